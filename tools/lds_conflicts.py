"""LDS bank-conflict model for the wave-private transpositions of the NTT kernels (gfx950 rules).

  python3 tools/lds_conflicts.py

8-byte accesses: a load is served in two groups of 32 consecutive lanes over 64 banks (32 slots of a
double), a store in four groups of 16 consecutive lanes over 32 banks (16 slots).  Within a group
every extra distinct address on a busy bank costs one LDS cycle.  For every exchange between two
register layouts the script prints the extra cycles of one transposition (8 loads or 8 stores per
side) under the pad of one slot per 8 positions and under the per-exchange pad the kernels use
(a run of 2^(a-3) slots per 2^a positions, a = the larger of the two layout bases, at least 3).
"""


def position(base, t, k):  # position of register k of lane t in the layout with this base (ntt_wg.hpp)
    return (((t >> base) << (base + 3)) | (k << base) | (t & ((1 << base) - 1))) & 511


def extra_cycles(slots, group, banks):
    extra = 0
    for g0 in range(0, 64, group):
        per_bank = {}
        for lane in range(g0, g0 + group):
            per_bank.setdefault(slots[lane] % banks, set()).add(slots[lane])
        extra += max(len(v) for v in per_bank.values()) - 1
    return extra


def side(slot, base):  # (extra cycles of the 8 loads, of the 8 stores) of one layout
    loads = stores = 0
    for k in range(8):
        s = [slot(position(base, t, k)) for t in range(64)]
        loads += extra_cycles(s, 32, 32)
        stores += extra_cycles(s, 16, 16)
    return loads, stores


def main():
    for logn in (9, 10, 11):
        groups = (logn + 2) // 3
        bases = [0 if g == groups - 1 else logn - 3 * (g + 1) for g in range(groups)]
        for g in range(groups - 1):
            hi, lo = bases[g], bases[g + 1]
            if hi > 6:
                continue  # block-level exchange
            a = max(hi, 3)
            old = lambda j: j + (j >> 3)
            new = lambda j, a=a: j + ((j >> a) << (a - 3))
            assert max(new(j) for j in range(512)) < 576
            print(f"N = {1 << logn}: exchange of the layouts with bases {hi} and {lo}: "
                  f"pad per 8 positions (loads, stores) {side(old, hi)} / {side(old, lo)}, "
                  f"per-exchange pad {side(new, hi)} / {side(new, lo)}")


if __name__ == "__main__":
    main()
