// rm_sweep32_launch.hpp -- shared by the translation units that instantiate the fp32 sweep (split so that they compile
// in parallel): one launcher per (AUC, DUMP, list mode, sub-tiles) over the factor-group counts a unit is built for.
#pragma once
#include <hip/hip_runtime.h>
#include "rm_sweep.hpp"

namespace rm {

template <int NGV, bool AUC, bool DUMP, int LMODE, int NSUB, int SPEC = 0>
static int launch_one(dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    auto kern = k_sweep<NGV, AUC, DUMP, LMODE, NSUB, SPEC>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, grid, dim3(256 * NSUB), lds, stream, sa);
    return (int)hipGetLastError();
}

// factor-group counts 2 ... 8 (and 16 when WITH16): the kernels for up to 64 (128) factors.  Every count up to 8 is instantiated
// (round 5): 50 factors -- the reference notebook's model -- run as 56, not as 64
template <bool AUC, bool DUMP, int LMODE, int NSUB, bool WITH16, int SPEC = 0>
static int launch_small(int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    switch (NG) {
        case 2: return launch_one<2, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa);
        case 3: return launch_one<3, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa);
        case 4: return launch_one<4, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa);
        case 5: return launch_one<5, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa);
        case 6: return launch_one<6, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa);
        case 7: return launch_one<7, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa);
        case 8: return launch_one<8, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa);
        // (between 64 and 128 factors: 80, 96 and 100 -- implicit-feedback libraries' usual defaults -- have kernels of their own)
        case 10: return WITH16 ? launch_one<WITH16 ? 10 : 8, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa) : -1;
        case 12: return WITH16 ? launch_one<WITH16 ? 12 : 8, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa) : -1;
        case 13: return WITH16 ? launch_one<WITH16 ? 13 : 8, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa) : -1;
        case 16: return WITH16 ? launch_one<WITH16 ? 16 : 8, AUC, DUMP, LMODE, NSUB, SPEC>(grid, lds, stream, sa) : -1;
        default: return -1;
    }
}

} // namespace rm
