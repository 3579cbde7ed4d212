// rm_noise.hpp -- the reference's tie-breaking noise, bit for bit (reference src/recometrics.hpp:528-534).
//
// With break_ties_with_noise the reference adds, to every CANDIDATE score of user u in ascending item order,
//     std::uniform_real_distribution<real_t>(-1e-12, 1e-12)(std::mt19937(seed + u)).
// libstdc++ (GCC 11, bits/random.tcc:3348-3380 generate_canonical, bits/random.h uniform_real_distribution::operator()):
//     float : one engine draw d;      r = float(d) / 2^32              (float(d) rounds to 24 bits; r >= 1 -> nextafter(1, 0))
//     double: two draws d1 then d2;   r = (double(d1) + double(d2) * 2^32) / 2^64   (the sum rounds to 53 bits; same clamp)
//     noise = r * (b - a) + a   with a = real_t(-1e-12), b = real_t(1e-12), one rounding per operation (the canonical
//     reference build has no fused multiply-add: oracle/Makefile)
// and the engine's seed is (seed + u) mod 2^32.  The stream cannot be entered in the middle (624-word state, no cheap
// jump), so a user whose ranking the noise can change gets its whole stream generated: one wavefront per user produces the
// tempered draws block by block (k_mt_draws), and k_noise_rows turns them into a row of per-ITEM noise values (0 for the
// train items, which draw nothing) that the sweep and the positives kernel add to their scores.
#pragma once
#include "rm_device.hpp"

namespace rm {

constexpr int MT_N = 624, MT_M = 397;
constexpr int MT_WAVES = 4;                       // generators (users) per block of k_mt_draws

// fp32: the noise can change a score only if |score| < 2^-15 (half an ulp of anything larger exceeds 1e-12); 2^-14 leaves
// a binade of margin.  Users with such a score among their test items or their top-K are the ones evaluated exactly.
constexpr float NOISE_ZONE_F32 = 6.103515625e-05f;

__device__ __forceinline__ unsigned mt_temper(unsigned y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// draws[row][0 .. need) of the users row_user[row]: need = (candidates of the user) * per_item draws, rounded up to blocks
__global__ __launch_bounds__(MT_WAVES * WAVE) void k_mt_draws(const int *row_user, int n_rows, unsigned long long seed, long long user0,
                                                                const int *train_p, int n, int per_item, unsigned *draws, long long ld)
{
    __shared__ unsigned state[MT_WAVES][MT_N];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = blockIdx.x * MT_WAVES + wv;
    if (row >= n_rows) return;
    const int u = row_user ? row_user[row] : row;
    unsigned *x = state[wv];
    if (lane == 0) {                                            // std::mersenne_twister_engine::seed(value)
        unsigned v = (unsigned)(seed + (unsigned long long)(user0 + u));
        x[0] = v;
        for (int i = 1; i < MT_N; i++) { v = 1812433253u * (v ^ (v >> 30)) + (unsigned)i; x[i] = v; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const long long need = (long long)(n - (train_p[u + 1] - train_p[u])) * per_item;
    unsigned *out = draws + (size_t)row * (size_t)ld;
    for (long long b0 = 0; b0 < need; b0 += MT_N) {
        // _M_gen_rand in place, 64 words per step in ascending order: a step reads x[i], x[i + 1] (not yet rewritten) and
        // x[(i + 397) mod 624] (old for i < 227, already new for i >= 227 -- exactly the sequential algorithm's view)
        for (int base = 0; base < MT_N; base += WAVE) {
            const int i = base + lane;
            unsigned nv = 0;
            if (i < MT_N) {
                const unsigned a = x[i], b = x[i + 1 == MT_N ? 0 : i + 1], c = x[i + MT_M >= MT_N ? i + MT_M - MT_N : i + MT_M];
                const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
                nv = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (i < MT_N) x[i] = nv;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        for (int i = lane; i < MT_N; i += WAVE) if (b0 + i < need) out[b0 + i] = mt_temper(x[i]);
    }
}

template <class T> __device__ __forceinline__ T noise_from_draws(const unsigned *d, long long c);
// NO fused multiply-add here: the reference rounds r * (b - a) before it adds a, and HIP's __fmul_rn / __fadd_rn are plain
// operators that the compiler's default -ffp-contract=fast fuses (found by scratch/fuzz.py: one noise value in ten was an ulp
// off, and an exact tie of two noisy scores in the reference was not a tie here).
template <> __device__ __forceinline__ float noise_from_draws<float>(const unsigned *d, long long c)
{
#pragma clang fp contract(off)
    float r = (float)d[c] * 2.3283064365386963e-10f;                            // float(d) / 2^32 (exact scaling)
    if (r >= 1.0f) r = 0.99999994f;                                             // nextafter(1.0f, 0.0f)
    const float a = (float)(-1e-12), b = (float)1e-12;
    const float scaled = r * (b - a);
    return scaled + a;
}
template <> __device__ __forceinline__ double noise_from_draws<double>(const unsigned *d, long long c)
{
#pragma clang fp contract(off)
    const double high = (double)d[2 * c + 1] * 4294967296.0;
    const double sum = (double)d[2 * c] + high;
    double r = sum * 5.421010862427522e-20;                                     // / 2^64 (exact scaling)
    if (r >= 1.0) r = 0.9999999999999999;                                       // nextafter(1.0, 0.0)
    const double a = -1e-12, b = 1e-12;
    const double scaled = r * (b - a);
    return scaled + a;
}

// E[row][item] = noise the reference adds to that item's score for the row's user (0 for train items and the padding):
// the item's candidate index = item - (train items below it) says which draw(s) it got
template <class T>
__global__ void k_noise_rows(const int *row_user, int n_rows, const int *train_p, const int *train_i, int n,
                             const unsigned *draws, long long d_ld, T *E, long long e_ld)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long row = t / e_ld, item = t % e_ld;
    if (row >= n_rows) return;
    const int u = row_user ? row_user[row] : (int)row;
    T e = 0;
    if (item < n) {
        const int *tr = train_i + train_p[u];
        const int len = train_p[u + 1] - train_p[u];
        int lo = 0, hi = len;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (tr[mid] < (int)item) lo = mid + 1; else hi = mid; }
        if (!(lo < len && tr[lo] == (int)item)) e = noise_from_draws<T>(draws + (size_t)row * (size_t)d_ld, item - lo);
    }
    E[(size_t)row * (size_t)e_ld + item] = e;
}

// The same from the DENSE train rows the sweep uses at small item counts (SweepArgs::train_bits: bit i of word i >> 5 = item i is
// a train item of the user, or lies beyond n): block per row, the number of train items in front of every 32-item word by a
// block-wide scan of the words' popcounts in LDS, then candidate index = item - (that count + popcount of the lower bits of the
// item's own word) -- no dependent walk through the CSR row per element (0.54 -> 0.1 ms for the 2,182 flagged users of
// BASELINE C2).  words <= TRAIN_BITS_MAX_WORDS (rm_prep.hpp).
constexpr int NOISE_ROWS_THREADS = 256;
// `masked` != 0: the rows mark the users' test items too (k_train_bits with test rows); the copy of the row in LDS gets the bits
// of the test items that are not train items cleared again -- the draws are indexed by the candidates, test items included.
template <class T>
__global__ __launch_bounds__(NOISE_ROWS_THREADS) void k_noise_rows_bits(const int *row_user, int n_rows, const unsigned *bits, int words, int n,
                                                                        int masked, const int *train_p, const int *train_i, const int *test_p, const int *test_i,
                                                                        const unsigned *draws, long long d_ld, T *E, long long e_ld)
{
    extern __shared__ int nr_lds[];                               // [words + 1] prefix counts, then [words] the row
    __shared__ int wave_total[NOISE_ROWS_THREADS / WAVE];
    int *nr_prefix = nr_lds;
    unsigned *lrow = (unsigned *)(nr_lds + words + 1);
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (row >= n_rows) return;
    const int u = row_user ? row_user[row] : row;
    const unsigned *brow = bits + (size_t)u * (size_t)words;
    for (int w = tid; w < words; w += NOISE_ROWS_THREADS) lrow[w] = brow[w];
    __syncthreads();
    if (masked) {
        const int *tr = train_i + train_p[u];
        const int ntr = train_p[u + 1] - train_p[u];
        for (int e = test_p[u] + tid; e < test_p[u + 1]; e += NOISE_ROWS_THREADS) {
            const int item = test_i[e];
            int lo = 0, hi = ntr;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (tr[mid] < item) lo = mid + 1; else hi = mid; }
            if (!(lo < ntr && tr[lo] == item)) atomicAnd(&lrow[item >> 5], ~(1u << (item & 31)));
        }
        __syncthreads();
    }
    // scan in chunks of the block size: every thread takes one word per chunk
    int carry = 0;
    for (int w0 = 0; w0 < words; w0 += NOISE_ROWS_THREADS) {
        const int w = w0 + tid;
        const int c = w < words ? __popc(lrow[w]) : 0;
        int x = c;
        #pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (lane == WAVE - 1) wave_total[wv] = x;
        __syncthreads();
        int before = carry;
        for (int i = 0; i < wv; i++) before += wave_total[i];
        if (w < words) nr_prefix[w] = before + x - c;             // exclusive
        int total = 0;
        for (int i = 0; i < NOISE_ROWS_THREADS / WAVE; i++) total += wave_total[i];
        carry += total;
        __syncthreads();
    }
    const unsigned *d = draws + (size_t)row * (size_t)d_ld;
    T *out = E + (size_t)row * (size_t)e_ld;
    for (long long item = tid; item < e_ld; item += NOISE_ROWS_THREADS) {
        T e = 0;
        if (item < n) {
            const int w = (int)(item >> 5), b = (int)(item & 31);
            const unsigned word = lrow[w];
            if (!((word >> b) & 1u)) e = noise_from_draws<T>(d, item - (nr_prefix[w] + __popc(word & ((1u << b) - 1u))));
        }
        out[item] = e;
    }
}

// tier 1 of the fp32 path: the users flagged in the first pass get rows (order irrelevant)
// (`skip`, optional: users that have been evaluated exactly already -- the exact pass beside the first sweep)
__global__ void k_noise_assign_rows(int m, const int *flag, const int *skip, int *noise_row, int *row_user, int *counter)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= m) return;
    int r = -1;
    if (flag[u] && !(skip && skip[u])) { r = atomicAdd(counter, 1); row_user[r] = u; }
    noise_row[u] = r;
}
// results of the exact pass that ran beside the first sweep (into buffers of its own) -> the caller's outputs, for its users
template <class T> struct ScatterArgs { const T *src[10]; T *dst[10]; int width[10]; };
template <class T>
__global__ void k_noise_scatter(int m, const int *picked, ScatterArgs<T> a)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= m || !picked[u]) return;
    #pragma unroll
    for (int i = 0; i < 10; i++) {
        if (!a.dst[i]) continue;
        const size_t o = (size_t)u * a.width[i];
        for (int j = 0; j < a.width[i]; j++) a.dst[i][o + j] = a.src[i][o + j];
    }
}
// the metric values of the users an exact pass evaluated, packed: out[f][...] for the f-th user of `row_user` (the host copies
// `count` short records instead of a metric block over every user of the range)
template <class T> struct GatherArgs { const T *src[10]; int width[10]; int off[10]; int out_w; };
template <class T>
__global__ void k_noise_gather(int count, const int *row_user, GatherArgs<T> a, T *out)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= count) return;
    const size_t u = (size_t)row_user[f];
    T *o = out + (size_t)f * a.out_w;
    #pragma unroll
    for (int i = 0; i < 10; i++) {
        if (!a.src[i]) continue;
        for (int j = 0; j < a.width[i]; j++) o[a.off[i] + j] = a.src[i][u * a.width[i] + j];
    }
}
__global__ void k_noise_select(int m, const int *noise_row, int r0, int r1, unsigned char *only)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < m) only[u] = noise_row[u] >= r0 && noise_row[u] < r1;
}

} // namespace rm
