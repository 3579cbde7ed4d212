// rm_sweep32.hip -- translation unit instantiating the fp32 sweep kernels.
#include <hip/hip_runtime.h>
#include "rm_sweep.hpp"

namespace rm {

template <bool AUC, bool DUMP, int LMODE, int NSUB>
static int launch_ng(int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
#define RM_LAUNCH(NGV)                                                                                               \
    case NGV: {                                                                                                      \
        auto kern = k_sweep<NGV, AUC, DUMP, LMODE, NSUB>;                                                            \
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return (int)e;                                                                          \
        hipLaunchKernelGGL(kern, grid, dim3(256 * NSUB), lds, stream, sa);                                           \
    } break;
    switch (NG) {
        RM_LAUNCH(2) RM_LAUNCH(4) RM_LAUNCH(8)
        case 16: if (NSUB == 2) { constexpr int N16 = NSUB == 2 ? 16 : 8;       /* three sub-tiles exist up to 64 factors only */
            auto kern = k_sweep<N16, AUC, DUMP, LMODE, NSUB>;
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL(kern, grid, dim3(256 * NSUB), lds, stream, sa);
            break; }
            return -1;
        default: return -1;
    }
#undef RM_LAUNCH
    return (int)hipGetLastError();
}

// nsub = 3 (three 32-item sub-tiles per step, 12 waves per block) exists for LDS lists up to 64 factors
int launch_sweep32(bool auc, bool dump, int lmode, int nsub, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    if (nsub == 3) {
        if (dump || lmode != LM_LDS || NG > 8) return -1;
        return auc ? launch_ng<true, false, LM_LDS, 3>(NG, grid, lds, stream, sa) : launch_ng<false, false, LM_LDS, 3>(NG, grid, lds, stream, sa);
    }
    if (NG > 16) return launch_sweep32_large(auc, dump, lmode, NG, grid, lds, stream, sa);
    if (dump) return launch_ng<false, true, LM_HBM, 2>(NG, grid, lds, stream, sa);
    if (auc) switch (lmode) {
        case LM_LDS: return launch_ng<true, false, LM_LDS, 2>(NG, grid, lds, stream, sa);
        case LM_HBM: return launch_ng<true, false, LM_HBM, 2>(NG, grid, lds, stream, sa);
        default: return launch_ng<true, false, LM_HBM_APPEND, 2>(NG, grid, lds, stream, sa);
    }
    switch (lmode) {
        case LM_LDS: return launch_ng<false, false, LM_LDS, 2>(NG, grid, lds, stream, sa);
        case LM_HBM: return launch_ng<false, false, LM_HBM, 2>(NG, grid, lds, stream, sa);
        default: return launch_ng<false, false, LM_HBM_APPEND, 2>(NG, grid, lds, stream, sa);
    }
}

} // namespace rm

#ifdef RM_STATS
extern "C" int rm_debug_stats(unsigned long long *out, int reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(rm::g_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(rm::g_stats), z, sizeof(z)); }
    return 0;
}
#endif

