// coexec6.hip -- issue cost of the forms a search level can be written in (gfx950): compare into an SGPR pair or into VCC,
// select from an SGPR pair or from VCC, 32-bit (e32) against 64-bit (e64 / VOP3) encodings; 1, 2, 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define HC(x) (void)(x)
template <int KIND> __device__ __forceinline__ void work(float (&x)[8], unsigned (&u)[8])
{
    unsigned long long m0 = 0, m1 = 0, m2 = 0;
    asm volatile("s_mov_b64 %0, 0x5555" : "=s"(m0)); asm volatile("s_mov_b64 %0, 0x3333" : "=s"(m1)); asm volatile("s_mov_b64 %0, 0x0f0f" : "=s"(m2));
    #pragma unroll
    for (int i = 0; i < 48; i++) {
        const int a = i & 7, b = (i + 3) & 7, c = (i + 5) & 7;
        if (KIND == 0) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(x[b]), "v"(x[c]));
        if (KIND == 1) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(x[b]), "v"(x[c]) : "vcc");
        if (KIND == 2) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]), "s"(m0));
        if (KIND == 3) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]) : "vcc");
        if (KIND == 4) asm volatile("v_or_b32_e32 %0, %1, %2" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]));
        if (KIND == 5) {        // the sweep's level today: or, cmp into one of three SGPR pairs, select from the pair written two scores ago
            asm volatile("v_or_b32_e32 %0, %1, %2" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]));
            if (i % 3 == 0) { asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(x[b]), "v"(x[c])); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[c]) : "v"(u[b]), "v"(u[a]), "s"(m1)); }
            if (i % 3 == 1) { asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(x[b]), "v"(x[c])); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[c]) : "v"(u[b]), "v"(u[a]), "s"(m2)); }
            if (i % 3 == 2) { asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(x[b]), "v"(x[c])); asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(u[c]) : "v"(u[b]), "v"(u[a]), "s"(m0)); }
        }
        if (KIND == 6) {        // through VCC, 32-bit encodings: cmp, or (of the next score), select
            asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(x[b]), "v"(x[c]) : "vcc");
            asm volatile("v_or_b32_e32 %0, %1, %2" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]));
            asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(u[c]) : "v"(u[b]), "v"(u[a]) : "vcc");
        }
        if (KIND == 7) {        // VCC, with the or folded away: select between two ready addresses is impossible, so: cmp, v_addc (2 idx + c)
            asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(x[b]), "v"(x[c]) : "vcc");
            asm volatile("v_or_b32_e32 %0, %1, %2" : "=v"(u[a]) : "v"(u[b]), "v"(u[c]));
            asm volatile("v_addc_co_u32_e32 %0, vcc, %1, %1, vcc" : "=v"(u[c]) : "v"(u[b]) : "vcc");
        }
        if (KIND == 8) asm volatile("v_sub_f32_e32 %0, %1, %2" : "=v"(x[a]) : "v"(x[b]), "v"(x[c]));
        if (KIND == 9) asm volatile("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(x[a]) : "v"(x[b]), "v"(x[c]), "v"(x[(i + 1) & 7]));
        if (KIND == 10) asm volatile("v_bfe_i32 %0, %1, 5, 1" : "=v"(u[a]) : "v"(u[b]));
        if (KIND == 11) asm volatile("v_max_f32_e32 %0, %1, %2" : "=v"(x[a]) : "v"(x[b]), "v"(x[c]));
    }
}
template <int KIND> __global__ __launch_bounds__(768) void k(int iters, float *out)
{
    float x[8]; unsigned u[8];
    for (int i = 0; i < 8; i++) { x[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x + i; }
    for (int it = 0; it < iters; it++) work<KIND>(x, u);
    float r = 0;
    for (int i = 0; i < 8; i++) r += x[i] + (float)u[i];
    if (r == 12345.678f) out[0] = r;
}
static float *d_out;
template <int KIND> float run(int waves)
{
    const int iters = 2000;
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256 * waves), 0, 0, 10, d_out);
    HC(hipDeviceSynchronize());
    HC(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256 * waves), 0, 0, iters, d_out);
    HC(hipEventRecord(e1)); HC(hipEventSynchronize(e1));
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}
int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    HC(hipMalloc(&d_out, 4));
    const double slots = 2000.0 * 48.0;
    printf("# ns per loop slot and wave (a slot = 1 instruction, or the 3 instructions of a search level for the last kinds), at 1 / 2 / 3 waves per SIMD; x = time relative to one wave\n");
    #define ROW(name, K, n) { float a = run<K>(1), b = run<K>(2), c = run<K>(3); \
        printf("%-46s 1 wave %.2f ns/instr   2 waves x%.2f   3 waves x%.2f  -> %.2f ns per instruction with 3 waves\n", name, a * 1e6 / slots / n, b / a, c / a, c * 1e6 / slots / n / 3); }
    ROW("v_cmp_lt_f32_e64 -> SGPR pair", 0, 1) ROW("v_cmp_lt_f32_e32 -> VCC", 1, 1) ROW("v_cndmask_b32_e64 <- SGPR pair", 2, 1) ROW("v_cndmask_b32_e32 <- VCC", 3, 1)
    ROW("v_or_b32_e32", 4, 1) ROW("v_sub_f32_e32", 8, 1) ROW("v_max_f32_e32", 11, 1) ROW("v_min3_f32 |a|,|b|,|c|", 9, 1) ROW("v_bfe_i32", 10, 1)
    ROW("level: or, cmp_e64 (3 pairs), cndmask_e64", 5, 3) ROW("level: cmp_e32 vcc, or, cndmask_e32 vcc", 6, 3) ROW("level: cmp_e32 vcc, or, v_addc_co_u32 vcc", 7, 3)
    return 0;
}
