// coexec3.hip -- does vector work hide behind an f32-input (or f64) MFMA chain on gfx950?
//   part 1 (same wave): a dependent MFMA chain with NF independent filler instructions after every MFMA, one wave per SIMD
//   part 2 (partner wave): waves 0-3 run the bare chain, waves 4-7 (their SIMD partners) run fillers only
// build: hipcc --offload-arch=gfx950 -O3 scratch/coexec3.hip -o scratch/coexec3 ; run: scratch/coexec3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// filler kinds: 0 v_add_f32, 1 v_or_b32, 2 v_cmp_lt_f32 + v_cndmask_b32 (pairs, counted as 2), 3 ds_read_b32, 4 s_nop-free SALU (s_add)
template <int KIND> __device__ __forceinline__ void filler(float (&x)[8], unsigned (&u)[8], int i, unsigned lds_addr)
{
    if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[i & 7]) : "v"(x[(i + 3) & 7]), "v"(x[(i + 5) & 7]));
    if (KIND == 1) asm volatile("v_or_b32 %0, %1, %2" : "=v"(u[i & 7]) : "v"(u[(i + 3) & 7]), "v"(u[(i + 5) & 7]));
    if (KIND == 2) {
        unsigned long long m;
        if (i & 1) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(u[i & 7]) : "v"(u[(i + 3) & 7]), "v"(u[(i + 5) & 7]) : "vcc");
        else asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(x[(i + 3) & 7]), "v"(x[(i + 5) & 7]) : "vcc");
        (void)m;
    }
    if (KIND == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(u[i & 7]) : "v"(lds_addr));
    if (KIND == 4) { unsigned s; asm volatile("s_add_u32 %0, %1, 1" : "=s"(s) : "s"((unsigned)i)); (void)s; }
}

// MF: 0 = v_mfma_f32_32x32x2_f32 (one chain), 1 = v_mfma_f32_16x16x4_f32 (one chain), 2 = the same, two chains in rotation,
//     3 = v_mfma_f64_16x16x4_f64 (one chain)
template <int MF, int KIND, int NF>
__global__ __launch_bounds__(256) void k_same(int iters, float *out, unsigned long long *cyc)
{
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = i;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned *)lds + (threadIdx.x & 63) * 4;
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x + i; }
    f32x16 a32 = {0}; f32x4 a16 = {0}, b16 = {0}; f64x4 a64 = {0};
    const float fa = threadIdx.x * 1e-3f, fb = 1.0001f;
    const double da = threadIdx.x * 1e-3, db = 1.0001;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        #pragma unroll
        for (int j = 0; j < 16; j++) {
            if (MF == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a32) : "v"(fa), "v"(fb));
            if (MF == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a16) : "v"(fa), "v"(fb));
            if (MF == 2) { if (j & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(b16) : "v"(fa), "v"(fb));
                           else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a16) : "v"(fa), "v"(fb)); }
            if (MF == 3) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a64) : "v"(da), "v"(db));
            #pragma unroll
            for (int f = 0; f < NF; f++) filler<KIND>(x, u, j * NF + f, lds_addr);
        }
        if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = a32[0] + a32[7] + a16[0] + b16[1] + (float)a64[0];
    for (int i = 0; i < 8; i++) r += x[i] + (float)u[i];
    if (r == 12345.678f) out[0] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// partner test: mode bit 0 = waves 0-3 run the chain, bit 1 = waves 4-7 run NFP fillers per "MFMA slot" (64 of them per iteration)
template <int MF, int KIND>
__global__ __launch_bounds__(512) void k_partner(int mode, int iters, float *out)
{
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 512) lds[i] = i;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned *)lds + (threadIdx.x & 63) * 4;
    const int wave = threadIdx.x >> 6;
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x + i; }
    f32x16 a32 = {0}; f32x4 a16 = {0}, b16 = {0}; f64x4 a64 = {0};
    const float fa = threadIdx.x * 1e-3f, fb = 1.0001f;
    const double da = threadIdx.x * 1e-3, db = 1.0001;
    if (wave < 4) {
        if (mode & 1) for (int it = 0; it < iters; it++) {
            #pragma unroll
            for (int j = 0; j < 16; j++) {
                if (MF == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a32) : "v"(fa), "v"(fb));
                if (MF == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a16) : "v"(fa), "v"(fb));
                if (MF == 2) { if (j & 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(b16) : "v"(fa), "v"(fb));
                               else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(a16) : "v"(fa), "v"(fb)); }
                if (MF == 3) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(a64) : "v"(da), "v"(db));
            }
        }
    } else if (mode & 2) {
        for (int it = 0; it < iters; it++) {
            #pragma unroll
            for (int j = 0; j < 128; j++) filler<KIND>(x, u, j, lds_addr);
            if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float r = a32[0] + a32[7] + a16[0] + b16[1] + (float)a64[0];
    for (int i = 0; i < 8; i++) r += x[i] + (float)u[i];
    if (r == 12345.678f) out[0] = r;
}

static float *d_out; static unsigned long long *d_cyc;
template <int MF, int KIND, int NF> void run_same(const char *mf, const char *kind)
{
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_same<MF, KIND, NF>), dim3(256), dim3(256), 0, 0, 10, d_out, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_same<MF, KIND, NF>), dim3(256), dim3(256), 0, 0, iters, d_out, d_cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("same-wave  %-22s filler %-12s x%2d per MFMA: %8.3f ms  %7.1f cycles per MFMA (s_memtime)\n", mf, kind, NF, ms, (double)c / (iters * 16.0));
}
template <int MF, int KIND> void run_partner(const char *mf, const char *kind)
{
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float t[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; mode++) {
        hipLaunchKernelGGL((k_partner<MF, KIND>), dim3(256), dim3(512), 0, 0, mode, 10, d_out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_partner<MF, KIND>), dim3(256), dim3(512), 0, 0, mode, iters, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&t[mode], e0, e1);
    }
    printf("partner    %-22s filler %-12s: chain %.3f ms, fillers %.3f ms, both %.3f ms (sum %.3f, max %.3f)\n", mf, kind, t[1], t[2], t[3], t[1] + t[2], t[1] > t[2] ? t[1] : t[2]);
}
#define SAME_ROW(MF, MFN, KIND, KN) run_same<MF, KIND, 0>(MFN, KN); run_same<MF, KIND, 2>(MFN, KN); run_same<MF, KIND, 4>(MFN, KN); \
    run_same<MF, KIND, 6>(MFN, KN); run_same<MF, KIND, 8>(MFN, KN); run_same<MF, KIND, 12>(MFN, KN); run_same<MF, KIND, 16>(MFN, KN); run_same<MF, KIND, 24>(MFN, KN);
int main()
{
    hipMalloc(&d_out, 4); hipMalloc(&d_cyc, 8);
    SAME_ROW(0, "f32_32x32x2 (1 chain)", 0, "v_add_f32")
    SAME_ROW(0, "f32_32x32x2 (1 chain)", 1, "v_or_b32")
    SAME_ROW(0, "f32_32x32x2 (1 chain)", 2, "cmp/cndmask")
    SAME_ROW(0, "f32_32x32x2 (1 chain)", 3, "ds_read_b32")
    SAME_ROW(0, "f32_32x32x2 (1 chain)", 4, "s_add_u32")
    SAME_ROW(1, "f32_16x16x4 (1 chain)", 0, "v_add_f32")
    SAME_ROW(2, "f32_16x16x4 (2 chains)", 0, "v_add_f32")
    SAME_ROW(2, "f32_16x16x4 (2 chains)", 2, "cmp/cndmask")
    SAME_ROW(3, "f64_16x16x4 (1 chain)", 0, "v_add_f32")
    SAME_ROW(3, "f64_16x16x4 (1 chain)", 2, "cmp/cndmask")
    SAME_ROW(3, "f64_16x16x4 (1 chain)", 3, "ds_read_b32")
    run_partner<0, 0>("f32_32x32x2", "v_add_f32"); run_partner<0, 1>("f32_32x32x2", "v_or_b32"); run_partner<0, 2>("f32_32x32x2", "cmp/cndmask");
    run_partner<0, 3>("f32_32x32x2", "ds_read_b32"); run_partner<0, 4>("f32_32x32x2", "s_add_u32");
    run_partner<2, 0>("f32_16x16x4 (2 chains)", "v_add_f32"); run_partner<2, 2>("f32_16x16x4 (2 chains)", "cmp/cndmask");
    run_partner<3, 0>("f64_16x16x4", "v_add_f32"); run_partner<3, 2>("f64_16x16x4", "cmp/cndmask"); run_partner<3, 3>("f64_16x16x4", "ds_read_b32");
    return 0;
}
