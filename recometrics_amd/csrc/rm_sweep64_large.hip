// rm_sweep64_large.hip -- translation unit instantiating the fp64 (65..512 factors) sweep kernels.
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
        RM_LAUNCH(16) RM_LAUNCH(32) RM_LAUNCH(64)
        default:
            if (NG <= 64 || NG % 8 != 0 || sa.ngt != NG) return -1;
            switch (0) { RM_LAUNCH(0) }                       // more than 512 factors: chunk count at run time
    }
#undef RM_LAUNCH
    return (int)hipGetLastError();
}

int launch_sweep64_large(bool auc, bool dump, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa)
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

int launch_sweep64(bool auc, bool dump, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa)
{
    return NG <= 8 ? launch_sweep64_small(auc, dump, lmode, NG, grid, lds, stream, sa)
                   : launch_sweep64_large(auc, dump, lmode, NG, grid, lds, stream, sa);
}

} // namespace rm
