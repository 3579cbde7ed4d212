// rm_sweep64_large.hip -- translation unit instantiating the fp64 (65..512 factors) sweep kernels.
#include <hip/hip_runtime.h>
#include "rm_sweep64.hpp"

namespace rm {

template <bool AUC, bool DUMP, bool LLDS>
static int launch_ng(int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa)
{
#define RM_LAUNCH(NGV)                                                                                               \
    case NGV: {                                                                                                      \
        auto kern = k_sweep64<NGV, AUC, DUMP, LLDS>;                                                                  \
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(SWEEP_THREADS), lds, stream, sa);                                        \
    } break;
    switch (NG) {
        RM_LAUNCH(16) RM_LAUNCH(32) RM_LAUNCH(64)
        default: return -1;
    }
#undef RM_LAUNCH
    return (int)hipGetLastError();
}

int launch_sweep64_large(bool auc, bool dump, bool llds, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa)
{
    if (dump) return launch_ng<false, true, false>(NG, grid, lds, stream, sa);
    if (auc) return llds ? launch_ng<true, false, true>(NG, grid, lds, stream, sa) : launch_ng<true, false, false>(NG, grid, lds, stream, sa);
    return llds ? launch_ng<false, false, true>(NG, grid, lds, stream, sa) : launch_ng<false, false, false>(NG, grid, lds, stream, sa);
}

int launch_sweep64(bool auc, bool dump, bool llds, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa)
{
    return NG <= 8 ? launch_sweep64_small(auc, dump, llds, NG, grid, lds, stream, sa)
                   : launch_sweep64_large(auc, dump, llds, NG, grid, lds, stream, sa);
}

} // namespace rm
