"""Instruction histogram of the large basic blocks of one kernel in a hipcc -save-temps .s file.

  python3 tools/isa_hist.py file.s kernel_symbol_substring [min_block_size]
"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
sub = sys.argv[2]
min_block = int(sys.argv[3]) if len(sys.argv) > 3 else 150
m = re.search(r'^(\S*%s\S*):.*?\n(.*?)\.end_amdhsa_kernel' % re.escape(sub), s, re.S | re.M)
name, body = m.group(1), m.group(2)
print(name)
for key in ('.amdhsa_next_free_vgpr', '.amdhsa_accum_offset', '.amdhsa_group_segment_fixed_size',
            '.amdhsa_private_segment_fixed_size', '; ScratchSize', '; Occupancy', '; NumVgprs', '; NumAgprs', '; LDSByteSize'):
    for l in re.findall(r'.*%s.*' % re.escape(key), s[m.start():m.end() + 4000])[:1]:
        print('  ', l.strip())
blocks, cur = [], None
for l in body.splitlines():
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t):
        cur = [t, []]
        blocks.append(cur)
    elif cur is not None and t and not t.startswith(('.', ';', '//')):
        cur[1].append(t.split()[0])
for b in blocks:
    if len(b[1]) >= min_block:
        c = collections.Counter(b[1])
        f64 = sum(v for k, v in c.items() if 'f64' in k)
        valu = sum(v for k, v in c.items() if k.startswith('v_'))
        lds = sum(v for k, v in c.items() if k.startswith('ds_'))
        print(f"{b[0]} total {len(b[1])}  valu {valu}  f64 {f64}  lds {lds}")
        print('   ', sorted(c.items(), key=lambda x: -x[1])[:45])
