// FETCH_SIZE calibration for THIS access pattern: a known number of bytes streamed once from HBM with the sweeps'
// LDS-DMA staging -- global_load_lds_dwordx4, one wave-instruction = two 512-byte runs 1 KiB apart, 32 KB units, eight waves per
// block -- and, for comparison, with plain 16-byte-per-lane loads to registers of 1 KiB contiguous per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 scratch/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- /tmp/fetch_calib     (bash scratch/fetch_calib.sh does both and the sum)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int UNIT = 32 * 1024;       // bytes per staged unit: fp32 128 factors x 64 items, fp64 64 factors x 64 items

// shape 0: the fp32 sweep's lane map ([g][h][64 items] x 16 B, piece j of wave (gi, sub)); shape 1: the fp64 sweep's ([g][q][row] x 16 B)
template <int SHAPE>
__global__ __launch_bounds__(512) void k_dma(const char *src, long long units, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave & 3, sub = wave >> 2, ul = lane & 31, h = lane >> 5;
    const unsigned lds_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)smem;
    unsigned voff[4];
    for (int j = 0; j < 4; j++) {
        const int pc = gi + 4 * j;
        voff[j] = SHAPE == 0 ? (unsigned)((pc * 2 * 64 + h * 64 + sub * 32 + ul) * 16)
                             : (unsigned)((((pc >> 1) * 4 + (pc & 1) * 2 + h) * 64 + sub * 32 + ul) * 16);
    }
    for (long long u = blockIdx.x; u < units; u += gridDim.x) {
        const char *base = src + u * UNIT;
        const int buf = (int)(u / gridDim.x) & 1;
        for (int j = 0; j < 4; j++) {
            const unsigned m0v = lds_base + (unsigned)(buf * UNIT + (sub * 16 + gi + 4 * j) * 1024);
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(voff[j]), "s"(base) : "memory", "m0");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (tid == 0) sink[blockIdx.x] = ((unsigned *)smem)[blockIdx.x & 1023];
}
__global__ __launch_bounds__(512) void k_plain(const uint4 *src, long long n16, unsigned *sink)
{
    unsigned acc = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
int main()
{
    const long long bytes = 2LL << 30, units = bytes / UNIT;
    char *buf; unsigned *sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4 * 4096)); CK(hipMemset(buf, 1, bytes));
    CK(hipFuncSetAttribute((const void *)k_dma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * UNIT));
    CK(hipFuncSetAttribute((const void *)k_dma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * UNIT));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_dma<0>, dim3(1024), dim3(512), 2 * UNIT, 0, buf, units, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("k_dma<0> (fp32 sweep's map)  %lld bytes  %.3f ms  %.2f TB/s\n", bytes, ms, bytes / (ms * 1e-3) / 1e12);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_dma<1>, dim3(1024), dim3(512), 2 * UNIT, 0, buf, units, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("k_dma<1> (fp64 sweep's map)  %lld bytes  %.3f ms  %.2f TB/s\n", bytes, ms, bytes / (ms * 1e-3) / 1e12);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_plain, dim3(2048), dim3(512), 0, 0, (const uint4 *)buf, bytes / 16, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); printf("k_plain (16 B per lane)       %lld bytes  %.3f ms  %.2f TB/s\n", bytes, ms, bytes / (ms * 1e-3) / 1e12);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
