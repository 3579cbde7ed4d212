// rm_sweep32.hip -- fp32 sweep, up to 128 factors, two sub-tiles: LDS lists and the score-dump variant; dispatcher.
#include "rm_sweep32_launch.hpp"

namespace rm {

// nsub = 3 (three 32-item sub-tiles per step, 12 waves per block) exists for LDS lists up to 64 factors
int launch_sweep32(bool auc, bool dump, int lmode, int nsub, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    if (nsub == 3) return (dump || lmode != LM_LDS || NG > 8) ? -1 : launch_sweep32_n3(auc, NG, grid, lds, stream, sa);
    if (NG > 16) return launch_sweep32_large(auc, dump, lmode, NG, grid, lds, stream, sa);
    if (dump) return launch_small<false, true, LM_HBM, 2, true>(NG, grid, lds, stream, sa);
    if (lmode != LM_LDS) return launch_sweep32_hbm(auc, lmode, NG, grid, lds, stream, sa);
    return auc ? launch_small<true, false, LM_LDS, 2, true>(NG, grid, lds, stream, sa)
               : launch_small<false, false, LM_LDS, 2, true>(NG, grid, lds, stream, sa);
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
