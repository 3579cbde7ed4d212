// peak_f64.hip -- measured rate of v_mfma_f64_16x16x4_f64 (and, for reference, v_mfma_f32_32x32x2_f32) on this device: the fp64 roofline
// bench.py prices the fp64 sweep against (78.6 TFLOP/s) is AMD's MI355X figure, which is not in /opt/skills/guides/MI355X_MICROARCH.md;
// this probe shows what the matrix pipe sustains and how many cycles one instruction takes.
//   hipcc --offload-arch=gfx950 -O3 scratch/peak_f64.hip -o /tmp/peak_f64 && /tmp/peak_f64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int CHAINS> __global__ __launch_bounds__(256) void k64(int iters, double *out)
{
    d4 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = d4{0., 0., 0., 0.};
    const double a = threadIdx.x * 1e-9 + 1.0, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int u = 0; u < 8; u++)
            #pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678) out[0] = s;
}
template <int CHAINS> __global__ __launch_bounds__(256) void k32(int iters, float *out)
{
    f16v acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) acc[c][r] = 0.f;
    const float a = threadIdx.x * 1e-6f + 1.0f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int u = 0; u < 8; u++)
            #pragma unroll
            for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; c++) for (int r = 0; r < 16; r++) s += acc[c][r];
    if (s == 12345.678f) out[0] = s;
}
template <class F> float timed(F launch)
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(10); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(4000); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    double *o64; float *o32; (void)hipMalloc(&o64, 8); (void)hipMalloc(&o32, 4);
    int clk = 0; (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    printf("# %s, %d CUs, clock attribute %d kHz\n", pr.name, pr.multiProcessorCount, clk);
    const int blocks = pr.multiProcessorCount * 2;                    // 8 waves per CU = 2 per SIMD
    {
        const float ms = timed([&](int it) { hipLaunchKernelGGL((k64<2>), dim3(blocks), dim3(256), 0, 0, it, o64); });
        const double n_inst = (double)blocks * 4 * 4000.0 * 8 * 2, flop = n_inst * 2048.0;
        printf("v_mfma_f64_16x16x4_f64 : %.1f TFLOP/s  (%.1f %% of 78.6; %.1f cycles per instruction and SIMD at 2.4 GHz)\n", flop / ms / 1e9, 100 * flop / ms / 1e9 / 78.6,
               ms * 1e-3 * 2.4e9 / (n_inst / (pr.multiProcessorCount * 4)));
    }
    {
        const float ms = timed([&](int it) { hipLaunchKernelGGL((k32<2>), dim3(blocks), dim3(256), 0, 0, it, o32); });
        const double n_inst = (double)blocks * 4 * 4000.0 * 8 * 2, flop = n_inst * 4096.0;
        printf("v_mfma_f32_32x32x2_f32 : %.1f TFLOP/s  (%.1f %% of 157.3; %.1f cycles per instruction and SIMD at 2.4 GHz)\n", flop / ms / 1e9, 100 * flop / ms / 1e9 / 157.3,
               ms * 1e-3 * 2.4e9 / (n_inst / (pr.multiProcessorCount * 4)));
    }
    return 0;
}
