#!/bin/bash
# Registers, scratch and LDS of the kernels in a built library:  tools/kernel_resources.sh lib.so [name substring]
set -e
LIB=$(readlink -f "$1"); PAT=${2:-}
TMP=$(mktemp -d); cd $TMP
cp "$LIB" lib.so   # the bundles are extracted next to the file
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null
for co in *gfx950*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$co" | python3 -c "
import sys, re
txt = sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
    blk = m.group(0)
    lds = re.search(r'\.group_segment_fixed_size:\s+(\d+)', blk)
    spill = re.search(r'\.vgpr_spill_count:\s+(\d+)', blk)
    if '$PAT' in m.group(1):
        print(f'{m.group(1)[:110]:110s} vgpr {m.group(3):>3s} scratch {m.group(2):>5s} spill {spill.group(1) if spill else \"?\":>3s} lds {lds.group(1) if lds else \"?\"}')
"
done
rm -rf $TMP
