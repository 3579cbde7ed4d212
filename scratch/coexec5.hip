// coexec5.hip -- three waves per SIMD, each alternating a dependent f32 MFMA chain (32 x v_mfma_f32_32x32x2_f32 = one 32x32
// score tile at 64 factors) with an epilogue of NLEV search levels (16 x {ds_read_b32, v_or, v_cmp, v_cndmask}) -- the shape
// of the C2 sweep -- under different ways of arranging the waves of a SIMD in time.    gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define HC(x) (void)(x)
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NLEV, bool LDS>
__device__ __forceinline__ void epilogue(const f32x16 &acc, unsigned (&at)[16], unsigned base)
{
    #pragma unroll
    for (int lv = 0; lv < NLEV; lv++) {
        float pv[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            if (LDS) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(pv[r]) : "v"(at[r]), "n"(128 * 3));
            else pv[r] = __uint_as_float(at[r]);
        }
        if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            unsigned long long m; unsigned cand;
            asm volatile("v_or_b32 %0, %1, %2" : "=v"(cand) : "v"(at[r]), "v"(128u << (lv & 3)));
            asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(pv[r]), "v"(acc[r]));
            asm volatile("s_nop 1");
            asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(at[r]) : "v"(at[r]), "v"(cand), "s"(m));
        }
    }
    #pragma unroll
    for (int r = 0; r < 16; r++) at[r] = base + ((at[r] - base) & 0xfff);
}

// MODE 0 free running; 1 s_barrier per iteration (all waves in phase); 2 high priority while in the MFMA chain; 3 high priority
// in the epilogue; 4 free running, waves 4-7 and 8-11 start a third / two thirds of an iteration late; 5 one MFMA chain per SIMD
// at a time (LDS lock per SIMD position: waves w, w+4, w+8 share a SIMD)
template <int MODE, int NLEV, bool LDS>
__global__ __launch_bounds__(768) void k(int iters, float *out, unsigned long long *cyc)
{
    __shared__ float tab[32 * 64 * 4];
    __shared__ unsigned locks[4];
    for (int i = threadIdx.x; i < 32 * 64 * 4; i += blockDim.x) tab[i] = (float)(i % 977) * 0.01f;
    if (threadIdx.x < 4) locks[threadIdx.x] = 0;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) float *)tab + (wave & 3) * 8192 + (lane & 31) * 4;
    unsigned at[16];
    for (int r = 0; r < 16; r++) at[r] = base;
    f32x16 acc = {0};
    const float fa = threadIdx.x * 1e-3f, fb = 1.0001f;
    __attribute__((address_space(3))) unsigned *lock = (__attribute__((address_space(3))) unsigned *)&locks[wave & 3];
    if (MODE == 4) {          // initial offsets
        const int late = wave >> 2;
        for (int d = 0; d < late; d++) {
            #pragma unroll
            for (int j = 0; j < 11; j++) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
        }
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
        if (MODE == 1) __syncthreads();
        if (MODE == 2) __builtin_amdgcn_s_setprio(3);
        if (MODE == 3) __builtin_amdgcn_s_setprio(0);
        if (MODE == 5) {
            for (;;) {
                unsigned got = 1u;
                if (lane == 0) { unsigned expect = 0u; got = __hip_atomic_compare_exchange_strong(lock, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 0u : 1u; }
                if (__builtin_amdgcn_readfirstlane(got) == 0u) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        #pragma unroll
        for (int j = 0; j < 32; j++) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
        if (MODE == 5) { if (lane == 0) __hip_atomic_store(lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
        if (MODE == 2) __builtin_amdgcn_s_setprio(0);
        if (MODE == 3) __builtin_amdgcn_s_setprio(3);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // (MFMA results are read below)
        epilogue<NLEV, LDS>(acc, at, base);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = acc[0] + acc[9];
    for (int i = 0; i < 16; i++) r += (float)at[i];
    if (r == 12345.678f) out[0] = r;
    if (lane == 0) atomicMax(cyc, t1 - t0);
}
static float *d_out; static unsigned long long *d_cyc;
template <int MODE, int NLEV, bool LDS> void run(const char *name, int waves)
{
    const int iters = 400;
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE, NLEV, LDS>), dim3(256), dim3(256 * waves), 0, 0, 10, d_out, d_cyc);
    HC(hipDeviceSynchronize());
    HC(hipMemset(d_cyc, 0, 8));
    HC(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, NLEV, LDS>), dim3(256), dim3(256 * waves), 0, 0, iters, d_out, d_cyc);
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; HC(hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost));
    const double per = (double)c / iters;                    // cycles per iteration of the slowest wave = per SIMD step
    const double mfma = 2048.0 * waves, valu = 48.0 * NLEV * waves;
    printf("%-34s %d waves/SIMD, %d levels%s: %7.3f ms, %8.0f cycles per step; MFMA %5.0f (%.3f of the step), %4.0f vector instr -> %.2f cycles each beyond the MFMA\n",
           name, waves, NLEV, LDS ? " (LDS)" : "      ", ms, per, mfma, mfma / per, valu, (per - mfma) / valu);
}
#define ALL(NLEV, LDSF) \
    run<0, NLEV, LDSF>("free running", 1); run<0, NLEV, LDSF>("free running", 2); run<0, NLEV, LDSF>("free running", 3); \
    run<1, NLEV, LDSF>("s_barrier per step (in phase)", 3); run<2, NLEV, LDSF>("high priority in the MFMA chain", 3); \
    run<3, NLEV, LDSF>("high priority in the epilogue", 3); run<4, NLEV, LDSF>("staggered start", 3); run<5, NLEV, LDSF>("one MFMA chain per SIMD at a time", 3); \
    run<1, NLEV, LDSF>("s_barrier per step (in phase)", 2); run<5, NLEV, LDSF>("one MFMA chain per SIMD at a time", 2);
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    HC(hipMalloc(&d_out, 4)); HC(hipMalloc(&d_cyc, 8));
    ALL(6, true)
    ALL(6, false)
    ALL(3, true)
    return 0;
}
