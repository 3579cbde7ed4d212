#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// mode bit0: waves 0-3 run an f32 MFMA dependent chain; bit1: waves 4-7 run a VALU fma chain (4 independent chains)
__global__ __launch_bounds__(512, 2) void k(int mode, int iters, float* out)
{
    const int wave = threadIdx.x >> 6;
    float r = 0;
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc = {0};
            float a = threadIdx.x * 1e-3f, b = 1.0001f;
            for (int i = 0; i < iters; i++) {
                #pragma unroll
                for (int j = 0; j < 16; j++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            r = acc[0] + acc[5];
        }
    } else {
        if (mode & 2) {
            float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, c = 1.0001f;
            for (int i = 0; i < iters; i++) {
                #pragma unroll
                for (int j = 0; j < 64; j++) { x0 = __builtin_fmaf(x0, c, x1); x1 = __builtin_fmaf(x1, c, x2); x2 = __builtin_fmaf(x2, c, x3); x3 = __builtin_fmaf(x3, c, x0); }
            }
            r = x0 + x1 + x2 + x3;
        }
    }
    if (r == 12345.678f) out[0] = r;
}
int main()
{
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 1; mode <= 3; mode++) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, 10, d);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, d);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per wave: mfma: iters*16 MFMAs of 64 cycles; valu: iters*256 fma instrs
        printf("mode %d: %.3f ms  (mfma chain %d x 64 cyc = %.0f kcyc; valu %d instr)\n", mode, ms, iters * 16, iters * 16 * 64 / 1e3, iters * 256);
    }
    return 0;
}
