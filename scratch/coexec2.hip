#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// waves 0-3: f32 MFMA chain (if mode&1); waves 4-7: side work of kind `kind` (if mode&2)
__global__ __launch_bounds__(512, 2) void k(int mode, int kind, int iters, float* out)
{
    __shared__ float lds[4096];
    const int wave = threadIdx.x >> 6;
    float r = 0;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
    __syncthreads();
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
    } else if (mode & 2) {
        if (kind == 0) {        // fp32 fma
            float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, c = 1.0001f;
            for (int i = 0; i < iters; i++) {
                #pragma unroll
                for (int j = 0; j < 64; j++) { x0 = __builtin_fmaf(x0, c, x1); x1 = __builtin_fmaf(x1, c, x2); x2 = __builtin_fmaf(x2, c, x3); x3 = __builtin_fmaf(x3, c, x0); }
            }
            r = x0 + x1 + x2 + x3;
        } else if (kind == 1) { // integer add/xor
            unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
            for (int i = 0; i < iters; i++) {
                #pragma unroll
                for (int j = 0; j < 64; j++) { x0 = (x0 + x1) ^ 0x5bd1e995u; x1 = (x1 + x2) ^ 0x1b873593u; x2 = (x2 + x3) ^ 0xcc9e2d51u; x3 = (x3 + x0) ^ 0x85ebca6bu; }
            }
            r = (float)(x0 + x1 + x2 + x3);
        } else if (kind == 2) { // compare + select
            float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f;
            unsigned b0 = 0, b1 = 0;
            for (int i = 0; i < iters; i++) {
                #pragma unroll
                for (int j = 0; j < 64; j++) { b0 = (x0 < x1) ? b0 + 128 : b0; b1 = (x2 < x3) ? b1 + 64 : b1; x0 = __uint_as_float(__float_as_uint(x0) ^ b1); x2 = __uint_as_float(__float_as_uint(x2) ^ b0); }
            }
            r = (float)(b0 + b1) + x0 + x2;
        } else {                // LDS reads (dependent address)
            unsigned a0 = threadIdx.x & 1023, a1 = (threadIdx.x * 7) & 1023;
            float s = 0;
            for (int i = 0; i < iters; i++) {
                #pragma unroll
                for (int j = 0; j < 64; j++) { float v0 = lds[a0], v1 = lds[a1 + 1024]; a0 = ((unsigned)v0 + 33) & 1023; a1 = ((unsigned)v1 + 17) & 1023; s += v0; }
            }
            r = s + a0 + a1;
        }
    }
    if (r == 12345.678f) out[0] = r;
}
int main()
{
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 1000;
    const char* names[4] = {"fma", "int add/xor", "cmp+cndmask", "lds read chain"};
    for (int kind = 0; kind < 4; kind++) {
        float t[4];
        for (int mode = 1; mode <= 3; mode++) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, kind, 10, d);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, kind, iters, d);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&t[mode], e0, e1);
        }
        printf("%-16s mfma %.3f  side %.3f  both %.3f  (sum %.3f, max %.3f)\n", names[kind], t[1], t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
    }
    return 0;
}
