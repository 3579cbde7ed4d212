// rm_sweep32_large.hip -- translation unit instantiating the fp32 sweep kernels for 129..512 factors.
#include <hip/hip_runtime.h>
#include "rm_sweep.hpp"

namespace rm {

template <bool AUC, bool DUMP, bool LLDS>
static int launch_ng(int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
#define RM_LAUNCH(NGV)                                                                                               \
    case NGV: {                                                                                                      \
        auto kern = k_sweep<NGV, AUC, DUMP, LLDS>;                                                                  \
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(SWEEP_THREADS), lds, stream, sa);                                        \
    } break;
    switch (NG) {
        RM_LAUNCH(32) RM_LAUNCH(64)
        default: return -1;
    }
#undef RM_LAUNCH
    return (int)hipGetLastError();
}

int launch_sweep32_large(bool auc, bool dump, bool llds, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    if (dump) return launch_ng<false, true, false>(NG, grid, lds, stream, sa);
    if (auc) return llds ? launch_ng<true, false, true>(NG, grid, lds, stream, sa) : launch_ng<true, false, false>(NG, grid, lds, stream, sa);
    return llds ? launch_ng<false, false, true>(NG, grid, lds, stream, sa) : launch_ng<false, false, false>(NG, grid, lds, stream, sa);
}

} // namespace rm
