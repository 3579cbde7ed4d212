// rm_sweep32_hbm.hip -- fp32 sweep, up to 128 factors, top-K lists in HBM (replace-the-minimum and append buffers).
#include "rm_sweep32_launch.hpp"

namespace rm {

int launch_sweep32_hbm(bool auc, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    if (lmode == LM_HBM)
        return auc ? launch_small<true, false, LM_HBM, 2, true>(NG, grid, lds, stream, sa)
                   : launch_small<false, false, LM_HBM, 2, true>(NG, grid, lds, stream, sa);
    return auc ? launch_small<true, false, LM_HBM_APPEND, 2, true>(NG, grid, lds, stream, sa)
               : launch_small<false, false, LM_HBM_APPEND, 2, true>(NG, grid, lds, stream, sa);
}

} // namespace rm
