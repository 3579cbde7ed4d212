// coexec4.hip -- vector issue rate of ONE SIMD with 1, 2, 3 waves (no MFMA anywhere), the price of some instructions, and
// whether scalar / LDS work of one wave hides behind vector work of its SIMD partner.   gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define HC(x) (void)(x)
// KIND: 0 v_add_f32 (independent), 1 v_pk_add_f32, 2 v_cvt_pk_u8_f32, 3 v_med3_f32, 4 v_lshl_add_u32, 5 v_cmp+v_cndmask via SGPR pair (2 instr),
//       6 s_add_u32 chain (scalar only), 7 ds_read_u8, 8 v_fma_f32, 9 v_pk_fma_f32, 10 v_max3_f32, 11 v_and_or_b32, 12 v_perm_b32
template <int KIND> __device__ __forceinline__ void work(float (&x)[8], unsigned (&u)[8], unsigned &s, unsigned lds_addr)
{
    #pragma unroll
    for (int i = 0; i < 64; i++) {
        const int a = i & 7, b = (i + 3) & 7, c = (i + 5) & 7;
        if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[a]) : "v"(x[b]), "v"(x[c]));
        if (KIND == 1) { typedef float f2 __attribute__((ext_vector_type(2))); f2 d, p = {x[b], x[c]}, q = {x[c], x[b]};
                         asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(p), "v"(q)); x[a] = d.x; }
        if (KIND == 2) asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(u[a]) : "v"(x[b]), "v"(u[c]));
        if (KIND == 3) asm volatile("v_med3_f32 %0, %1, %2, %3" : "=v"(x[a]) : "v"(x[b]), "v"(x[c]), "v"(x[(i + 1) & 7]));
        if (KIND == 4) asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]));
        if (KIND == 5) { unsigned long long m; asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(x[b]), "v"(x[c]));
                         asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]), "s"(m)); }
        if (KIND == 6) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s));
        if (KIND == 7) asm volatile("ds_read_u8 %0, %1" : "=v"(u[a]) : "v"(lds_addr));
        if (KIND == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[a]) : "v"(x[b]), "v"(x[c]), "v"(x[(i + 1) & 7]));
        if (KIND == 9) { typedef float f2 __attribute__((ext_vector_type(2))); f2 d, p = {x[b], x[c]}, q = {x[c], x[b]};
                         asm volatile("v_pk_fma_f32 %0, %1, %2, %1" : "=v"(d) : "v"(p), "v"(q)); x[a] = d.x; }
        if (KIND == 10) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(x[a]) : "v"(x[b]), "v"(x[c]), "v"(x[(i + 1) & 7]));
        if (KIND == 11) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]), "v"(u[(i + 1) & 7]));
        if (KIND == 12) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]), "v"(u[(i + 1) & 7]));
    }
    if (KIND == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// waves [0, 4) run KA, waves [4, 8) run KB, waves [8, 12) run KC; -1 = the wave exits at once
template <int KA, int KB, int KC>
__global__ __launch_bounds__(768) void k(int iters, float *out)
{
    __shared__ unsigned lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const unsigned lds_addr = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) unsigned *)lds + (threadIdx.x & 63) * 4;
    const int wave = threadIdx.x >> 6;
    float x[8]; unsigned u[8]; unsigned s = __builtin_amdgcn_readfirstlane(wave);
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x + i; }
    if (wave < 4) { if (KA >= 0) for (int it = 0; it < iters; it++) work<KA < 0 ? 0 : KA>(x, u, s, lds_addr); }
    else if (wave < 8) { if (KB >= 0) for (int it = 0; it < iters; it++) work<KB < 0 ? 0 : KB>(x, u, s, lds_addr); }
    else { if (KC >= 0) for (int it = 0; it < iters; it++) work<KC < 0 ? 0 : KC>(x, u, s, lds_addr); }
    float r = (float)s;
    for (int i = 0; i < 8; i++) r += x[i] + (float)u[i];
    if (r == 12345.678f) out[0] = r;
}
static float *d_out;
template <int KA, int KB, int KC> float run(int waves)
{
    const int iters = 1000;
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KA, KB, KC>), dim3(256), dim3(256 * waves), 0, 0, 10, d_out);
    HC(hipDeviceSynchronize());
    HC(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KA, KB, KC>), dim3(256), dim3(256 * waves), 0, 0, iters, d_out);
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}
int main()
{
    HC(hipMalloc(&d_out, 4));
    const double per = 1000.0 * 64.0; setvbuf(stdout, nullptr, _IONBF, 0);      // instruction slots per wave
    printf("# ms for 256k slots per wave; ns per slot in brackets (x clock = cycles; the clock under this load is ~2.1-2.4 GHz)\n");
    #define ROW(name, K) { float a = run<K, -1, -1>(1), b = run<K, K, -1>(2), c = run<K, K, K>(3); \
        printf("%-22s 1 wave/SIMD %.3f ms (%.2f ns)   2 waves %.3f ms (x%.2f)   3 waves %.3f ms (x%.2f)\n", name, a, a * 1e6 / per, b, b / a, c, c / a); }
    ROW("v_add_f32", 0) ROW("v_fma_f32", 8) ROW("v_pk_add_f32", 1) ROW("v_pk_fma_f32", 9) ROW("v_cvt_pk_u8_f32", 2) ROW("v_med3_f32", 3) ROW("v_max3_f32", 10)
    ROW("v_lshl_add_u32", 4) ROW("v_and_or_b32", 11) ROW("v_perm_b32", 12) ROW("v_cmp+v_cndmask (2)", 5) ROW("s_add_u32", 6) ROW("ds_read_u8", 7)
    { float a = run<0, -1, -1>(1), b = run<6, -1, -1>(1), c = run<0, 6, -1>(2); printf("v_add wave %.3f, s_add wave %.3f, side by side on one SIMD %.3f ms\n", a, b, c); }
    { float a = run<0, -1, -1>(1), b = run<7, -1, -1>(1), c = run<0, 7, -1>(2); printf("v_add wave %.3f, ds_read_u8 wave %.3f, side by side %.3f ms\n", a, b, c); }
    { float a = run<5, -1, -1>(1), b = run<6, -1, -1>(1), c = run<5, 6, -1>(2); printf("cmp/cndmask wave %.3f, s_add wave %.3f, side by side %.3f ms\n", a, b, c); }
    { float a = run<0, 0, -1>(2), b = run<6, -1, -1>(1), c = run<0, 0, 6>(3); printf("two v_add waves %.3f, s_add wave %.3f, all three %.3f ms\n", a, b, c); }
    return 0;
}
