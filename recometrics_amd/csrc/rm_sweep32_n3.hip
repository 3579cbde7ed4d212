// rm_sweep32_n3.hip -- fp32 sweep, up to 64 factors, LDS lists, three item sub-tiles per step (12 waves per block).
#include "rm_sweep32_launch.hpp"

namespace rm {

int launch_sweep32_n3(bool auc, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    return auc ? launch_small<true, false, LM_LDS, 3, false>(NG, grid, lds, stream, sa)
               : launch_small<false, false, LM_LDS, 3, false>(NG, grid, lds, stream, sa);
}

} // namespace rm
