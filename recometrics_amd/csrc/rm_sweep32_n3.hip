// rm_sweep32_n3.hip -- fp32 sweep, up to 64 factors, LDS lists, three item sub-tiles per step (12 waves per block).
#include "rm_sweep32_launch.hpp"

namespace rm {

int launch_sweep32_n3(bool auc, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    return auc ? launch_small<true, false, LM_LDS, 3, false>(NG, grid, lds, stream, sa)
               : launch_small<false, false, LM_LDS, 3, false>(NG, grid, lds, stream, sa);
}

} // namespace rm

#ifdef RM_STATS
extern "C" int rm_debug_stats_n3(unsigned long long *out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(rm::g_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(rm::g_stats), z, sizeof(z)); }
    return 0;
}
#endif
