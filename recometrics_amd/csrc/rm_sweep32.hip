// rm_sweep32.hip -- dispatcher of the fp32 sweep's translation units, and the score-dump variant.
#include "rm_sweep32_launch.hpp"

namespace rm {

// nsub = 3 (three 32-item sub-tiles per step, 12 waves per block) exists for LDS lists up to 64 factors; sa.spec picks the
// specialisation of the epilogue's run-time switches (rm_sweep.hpp k_sweep SPEC; the kernels beyond 128 factors have none)
int launch_sweep32(bool auc, bool dump, int lmode, int nsub, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    const int spec = dump ? 0 : sa.spec;
    if (nsub == 3) {
        if (dump || lmode != LM_LDS || NG > 8) return -1;
        return spec == 1 ? launch_sweep32_n3_s1(auc, NG, grid, lds, stream, sa) : spec == 2 ? launch_sweep32_n3_s2(auc, NG, grid, lds, stream, sa)
                                                                                : launch_sweep32_n3_s0(auc, NG, grid, lds, stream, sa);
    }
    if (NG > 16) return launch_sweep32_large(auc, dump, lmode, NG, grid, lds, stream, sa);
    if (dump) return launch_small<false, true, LM_HBM, 2, true>(NG, grid, lds, stream, sa);
    if (lmode != LM_LDS)
        return spec == 1 ? launch_sweep32_hbm_s1(auc, lmode, NG, grid, lds, stream, sa) : spec == 2 ? launch_sweep32_hbm_s2(auc, lmode, NG, grid, lds, stream, sa)
                                                                                        : launch_sweep32_hbm_s0(auc, lmode, NG, grid, lds, stream, sa);
    return spec == 1 ? launch_sweep32_lds_s1(auc, NG, grid, lds, stream, sa) : spec == 2 ? launch_sweep32_lds_s2(auc, NG, grid, lds, stream, sa)
                                                                             : launch_sweep32_lds_s0(auc, NG, grid, lds, stream, sa);
}

} // namespace rm

#ifdef RM_STATS
extern "C" int rm_debug_stats(unsigned long long *out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(rm::g_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(rm::g_stats), z, sizeof(z)); }
    return 0;
}
#endif
