// rm_sweep64_small.hip -- translation unit instantiating the fp64 (<= 64 factors) sweep kernels.
#include <hip/hip_runtime.h>
#include "rm_sweep64.hpp"

namespace rm {

template <bool AUC, bool DUMP, int LMODE>
static int launch_ng(int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa)
{
#define RM_LAUNCH(NGV)                                                                                               \
    case NGV: {                                                                                                      \
        auto kern = k_sweep64<NGV, AUC, DUMP, LMODE>;                                                                  \
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(SWEEP_THREADS), lds, stream, sa);                                        \
    } break;
    switch (NG) {
        RM_LAUNCH(2) RM_LAUNCH(4) RM_LAUNCH(8)
        default: return -1;
    }
#undef RM_LAUNCH
    return (int)hipGetLastError();
}

int launch_sweep64_small(bool auc, bool dump, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa)
{
    if (dump) return launch_ng<false, true, LM_HBM>(NG, grid, lds, stream, sa);
    if (auc) switch (lmode) {
        case LM_LDS: return launch_ng<true, false, LM_LDS>(NG, grid, lds, stream, sa);
        case LM_HBM: return launch_ng<true, false, LM_HBM>(NG, grid, lds, stream, sa);
        default: return launch_ng<true, false, LM_HBM_APPEND>(NG, grid, lds, stream, sa);
    }
    switch (lmode) {
        case LM_LDS: return launch_ng<false, false, LM_LDS>(NG, grid, lds, stream, sa);
        case LM_HBM: return launch_ng<false, false, LM_HBM>(NG, grid, lds, stream, sa);
        default: return launch_ng<false, false, LM_HBM_APPEND>(NG, grid, lds, stream, sa);
    }
}

} // namespace rm
