// Prints the compile-time lazy-reduction plans of the inverse transforms (make_gs_plan, ntt512.hpp) as
// the kernels use them: one line per transform length,
//   <log2 N> <groups> <stages of the first group> <pre masks, groups-1 .. 0, three per group> <carry masks> <output bound>
// Host code only; tests/test_reduction_plan.py re-derives the bounds from these masks.
#include <cstdio>

#include "../tfhe_amd/csrc/ntt_wg.hpp"

using namespace tfhe_amd;

static void show(int log2n, int groups, int last, const GsPlan& pl) {
    std::printf("%d %d %d", log2n, groups, last);
    for (int g = groups - 1; g >= 0; g--)
        for (int s = 0; s < 3; s++) std::printf(" %u", pl.pre[g][s]);
    for (int g = groups - 1; g >= 0; g--) std::printf(" %u", pl.carry[g]);
    std::printf(" %.6f\n", pl.out);
}

int main() {
    show(9, 3, 3, kGsPlan512);  // wave-level transform of the default ring
    show(9, WgNtt<9>::GROUPS, WgNtt<9>::LAST, WgNtt<9>::kInversePlan);
    show(10, WgNtt<10>::GROUPS, WgNtt<10>::LAST, WgNtt<10>::kInversePlan);
    show(11, WgNtt<11>::GROUPS, WgNtt<11>::LAST, WgNtt<11>::kInversePlan);
    return 0;
}
