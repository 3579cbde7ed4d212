// rm_finalize.hpp -- per-user metric finalisation from the sweep's partial results (one thread per user).
//
// Replaces reference src/recometrics.hpp:450-476 (NaN fill), :541-562 (validity checks), :589-788 (top-K walk and NaN
// overrides), :795-865 (ROC / PR AUC) and :868-961 (NDCG normalisation).  All metric arithmetic is fp64 with the
// reference's operation order, cast to real_t on store; log2(i+2) comes from a host table built with the same libm
// the CPU path uses.  ROC-AUC is formed in fp64 where the reference uses x87 long double (<= 1 ulp(fp64) apart).
#pragma once
#include "rm_device.hpp"
#include <type_traits>

namespace rm {

constexpr int MAX_PARTS = 128;
constexpr int FIN_THREADS = 128;      // block size of k_finalize
constexpr int FIN_TOPV = 64;          // sorted buffer of the largest test values (ideal DCG) per thread
template <class T> inline size_t finalize_lds_bytes(int K, int n_part) { return (sizeof(T) * (size_t)(K < FIN_TOPV ? K : FIN_TOPV) + 4 * (size_t)n_part) * FIN_THREADS; }

// per-slot result of the rank-histogram walk (k_auc_slots), combined per user by k_finalize
struct AucPart { unsigned long long sum_ranks; double s1, s2; int nvalid, pad; };

template <class T, class S> struct FinalArgs {     // T = real_t of inputs/outputs, S = score type of the sweep
    int m, n, K, n_part, req, cumulative, noise, gu, n_slots;
    const int *slot_user, *slot_chunk;
    const int *train_p, *test_p, *test_i; const T *test_v;
    const int *flags, *user_nslots, *uslot_base, *slot_index;
    const int *gj; const long long *grow;
    const Entry<S> *pl; const PartialStat<S> *pst; const unsigned *hist; const S *pos_score;
    const int *pos_item_tab;     // item ids of the tables of sorted positives (same shape as pos_score)
    AucPart *auc_part;           // [n_slots]
    T *heavy_topv;               // [m][heavy_ld] largest test values (descending) of users with more than
    unsigned char *heavy_nan;    // [m]    ... HEAVY_NPOS test items, and whether any of their values is NaN (k_top_values)
    const int *heavy_users; int n_heavy;     // those users (k_classify)
    int heavy_npos, heavy_ld;                // ... rows longer than heavy_npos; heavy_topv holds heavy_ld = min(K, longest test row) values per user
    const double *log2tab;
    T *p, *tp, *r, *ap, *tap, *ndcg, *hit, *rr, *roc, *pr;
    Entry<S> *merged;            // [m][K]   final ordered top-K (also the rm_rank_* output)
    long long *rank_sorted;      // [nnz_test] 1-based full-ranking position per SORTED positive (0 = masked), optional
    int *status;                 // [m] optional: 0 ranked, 1 skipped
    // streamed users (slots [stream_slot0, n_slots), rm_device.hpp STREAM_CLASS)
    int stream_slot0;
    int rank_generic;                                // A/B switch: k_rank_streamed without its fast routine
    int fin_slot0, fin_slot1;                        // slot range of a k_finalize launch
    int fused_auc;                                   // bit 0: k_rank_streamed also walks the counts of a row it holds whole (one block per row, table in LDS); bit 1: + the users' own test items
    int auc_defer_slot0;                             // k_finalize leaves ROC / PR-AUC of the slots from here on to k_finalize_auc (n_slots = none)
    const S *stream_scores; long long stream_ld;     // [n_stream][stream_ld] masked candidate scores written by the sweep
    const S *spos_score; const int *spos_item;       // [nnz_test] sorted positives of those users at test_p[u] + rank
    unsigned *shist;             // [nnz_test] at test_p[u] + j: candidates ranking above positive j but not above positive j + 1
    // top-K lists longer than the sweep keeps (k_metrics > 256): EVERY user is streamed and k_select_topk picks its top-K
    // from the stored row straight into `merged`
    int ext_topk; unsigned long long *sel_hi; unsigned *sel_lo; int sel_ld;      // scratch [n_slots][sel_ld] (sel_ld = power of two >= K)
    int collected;               // `merged` holds the ordered top-K already: k_collect_topk has written it (k_metrics beyond the LDS lists)
    int *noise_flag; Plan *plan; // first pass of an fp32 noise call: flag the users with a top-K score the noise can change
};

template <class T> __device__ __forceinline__ T qnan();
template <> __device__ __forceinline__ float qnan<float>() { return __int_as_float(0x7fc00000); }
template <> __device__ __forceinline__ double qnan<double>() { return __longlong_as_double(0x7ff8000000000000ll); }

template <class T, class S>
__device__ void fill_user_nan(const FinalArgs<T, S> &a, int u)
{
    T *top8[8] = {a.p, a.tp, a.r, a.ap, a.tap, a.ndcg, a.hit, a.rr};
    for (int q = 0; q < 8; q++) {
        T *arr = top8[q];
        if (!arr) continue;
        if (!a.cumulative) arr[u] = qnan<T>();
        else for (int i = 0; i < a.K; i++) arr[(size_t)u * a.K + i] = qnan<T>();
    }
    if (a.roc) a.roc[u] = qnan<T>();
    if (a.pr) a.pr[u] = qnan<T>();
}

template <class S> __device__ __forceinline__ bool ent_before(const Entry<S> &x, const Entry<S> &y)
{
    return x.s > y.s || (x.s == y.s && x.idx < y.idx);
}

// ROC / PR-AUC ingredients of one slot = one chunk of <= 63 sorted positives of a user (reference :795-865 walks the
// fully sorted candidate list instead).  Row b of the slot's histogram counts the candidates that rank between positive
// b - 1 and positive b, so a downward walk yields each positive's 1-based rank in the full ranking.  One thread per slot:
// the users of a group sit on adjacent threads and each row is one coalesced line; a heavy user's chunks run in parallel
// instead of back to back on that user's thread.
//   sum_ranks = sum of the ranks,  s1 = sum 1/rank,  s2 = sum i/rank  (i = 1.. in descending score order inside the chunk).
// With h0 positives in the higher-scored chunks the chunk adds  h0 * s1 + s2  to sum_i h_i/rank_i; a user with a single
// chunk gets exactly the reference's left-to-right accumulation.
template <class T, class S>
__global__ void k_auc_slots(FinalArgs<T, S> a)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.stream_slot0) return;                              // streamed users: k_rank_streamed / k_auc_streamed
    const int u = a.slot_user[slot], c = a.slot_chunk[slot];
    if (a.flags[u] & UF_ONLY_NDCG) return;
    const int te0 = a.test_p[u], npos = a.test_p[u + 1] - te0;
    const int GUr = a.gu;
    const int g = slot / GUr, ul = slot % GUr;
    const int PLg = (1 << a.gj[g]) - 1;
    const unsigned *H = a.hist + (a.grow[g] + g) * GUr + ul;
    const S *PS = a.pos_score + (a.grow[g] + g) * GUr + ul;
    const int pc = min(POS_CHUNK, npos - c * POS_CHUNK);
    unsigned long long above = 0, sum_ranks = 0;
    double s1 = 0, s2 = 0;
    int nvalid = 0;
    for (int b = PLg; b >= 1; b--) {
        above += H[(size_t)b * GUr];
        const int j = b - 1;
        if (j >= pc) continue;                                       // padding row above the chunk's positives
        const S ps = PS[(size_t)j * GUr];
        if (isinf(ps) && ps > 0) continue;                           // masked by the train row
        const unsigned long long rank = above + 1;
        sum_ranks += rank; nvalid++;
        s1 += 1. / (double)rank;
        s2 += (double)nvalid / (double)rank;
        if (a.rank_sorted) a.rank_sorted[te0 + c * POS_CHUNK + j] = (long long)rank;
    }
    AucPart r; r.sum_ranks = sum_ranks; r.s1 = s1; r.s2 = s2; r.nvalid = nvalid; r.pad = 0;
    a.auc_part[slot] = r;
}

// ---- the users' own test items, when the sweep did not see them ---------------------------------------------------------
// With dense train rows (small item counts) the rows mark the TEST items too (k_train_bits), so the sweep evaluates the
// candidates that are NOT test items: no candidate ever meets its own entry in the table of sorted positives -- the exact tie
// that three tiles out of four used to hold at 27k items, each one a trip through the tie rule (7 % of the sweep at BASELINE
// C2).  This kernel puts the test items back, one thread per user (slot), before anything reads the sweep's results:
//   * rank histogram: positive i (ascending (score, item desc) order = row i) outranks exactly the positives of the rows below
//     it, so it counts in bin i like every other candidate with i positives below it;
//   * validity statistics and top-K: the extra part `extra` of pst / pl = max / min / NaN of the test items' scores and their
//     K best in (score desc, item asc) order; k_finalize merges parts, whoever wrote them.
// Test items masked by the train row (+inf in the tables) are not candidates.  Only users with ONE slot reach this scheme
// (table users with <= 63 test items and streamed users); the host keeps the old scheme when a call has chunked users.
// Since round 4 the TABLE users are put back by the sweep itself (rm_sweep.hpp: the block of a user's first item range holds
// its table in LDS) and this kernel is launched for the streamed users only, `slot0` = their first slot.
constexpr int MERGE_WAVES = 4;                            // slots per block of k_merge_positives (one wavefront each)
template <class T, class S>
__global__ __launch_bounds__(MERGE_WAVES * WAVE) void k_merge_positives(FinalArgs<T, S> a, int extra, int slot0)
{
    const int slot = slot0 + __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6)), lane = threadIdx.x & 63;
    if (slot >= a.n_slots) return;
    const int u = a.slot_user[slot];
    const int K = a.K, NP = a.n_part;
    Entry<S> *L = (Entry<S> *)a.pl + ((size_t)slot * NP + extra) * K;
    S vmax = -(S)INFINITY, vmin = (S)INFINITY;
    int has_nan = 0, filled = 0;                                    // `filled` is wave-uniform
    if (!(a.flags[u] & UF_ONLY_NDCG)) {
        const int te0 = a.test_p[u], npos = a.test_p[u + 1] - te0;
        const S *PS = a.spos_score + te0;                           // the streamed user's positives, ascending (score, item desc)
        const int *PI = a.spos_item + te0;
        for (int top = npos - 1; top >= 0; top -= WAVE) {           // best rows first: lane 0 holds the best of the 64
            const int i = top - lane;
            const S x = i >= 0 ? PS[i] : (S)INFINITY;
            const bool masked = isinf(x) && x > 0;                  // masked by the train row (or beyond the row)
            const bool isn = x != x;
            const bool cand = !masked && !isn;
            has_nan |= isn ? 1 : 0;
            if (cand) {
                vmax = x > vmax ? x : vmax;
                vmin = x < vmin ? x : vmin;
                if (i >= 1) atomicAdd(&a.shist[te0 + i - 1], 1u);   // (k_rank_streamed may be counting beside this kernel)
            }
            if (filled < K) {                                       // (score desc, item asc) = descending row order
                const unsigned long long mk = __ballot(cand);
                const int pos = filled + __popcll(mk & ((1ull << lane) - 1ull));
                if (cand && pos < K) { L[pos].s = x; L[pos].idx = PI[i]; }
                filled += __popcll(mk);
            }
        }
    }
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const S om = __shfl_xor(vmax, d), on = __shfl_xor(vmin, d);
        vmax = om > vmax ? om : vmax; vmin = on < vmin ? on : vmin;
        has_nan |= __shfl_xor(has_nan, d);
    }
    for (int i = (filled < K ? filled : K) + lane; i < K; i += WAVE) { L[i].s = -(S)INFINITY; L[i].idx = IDX_EMPTY; }
    if (lane == 0) {
        PartialStat<S> ps;
        ps.vmax = vmax; ps.vmin = vmin; ps.rocsum = 0; ps.has_nan = has_nan; ps.pad = 0;
        ((PartialStat<S> *)a.pst)[(size_t)slot * NP + extra] = ps;
    }
}

// ---- streamed users: ranks of the positives from the score row the sweep stored --------------------------------------
// Block = (streamed user, range of its items).  The user's sorted positives sit in LDS (global memory when the row is
// longer than the LDS holds), every stored score is binary-searched among them with the sweep's tie rule (score desc,
// item asc: a candidate outranks the equal-scored positives with a LARGER item id), and bin b = "exactly b positives rank
// below the candidate" is counted at shist[test_p[u] + b - 1] (bin 0, below every positive, is never needed).
constexpr int STREAM_RANK_THREADS = 256;
constexpr int STREAM_RANK_ITEMS = 128;                   // items per thread and block (at most; rows are cut into equal pieces)
constexpr int STREAM_RANK_LDS = 20 * 1024;               // LDS per block: eight blocks per CU
// The LDS copy of the sorted positives is padded with +inf to a power of two (no bounds test in the search) and
// REPLICATED R times ([entry][R copies], lane l reads copy l mod R): with R = 32 lane l always hits bank l whatever it
// indexes, so the dependent reads of the search run at the conflict-free LDS rate (a single copy serves 64 random addresses
// per read at 3-5 bank cycles).  R = the largest power of two that fits next to the counters; rows too long even for one
// copy are searched in global memory.  ILP independent searches per thread advance level by level, all reads of a level
// in flight together.
// LV = log2(table entries) as a template constant for the usual depths (the search steps become immediates: offset of the
// LDS read, literal of the add), 0 = taken from `top` / `lgsb` at run time.  With LV known the copies fill 16 KiB.
template <class S, int ILP, int LV>
__device__ __forceinline__ void rank_streamed_lds(const S *row, long long i0, int ipt, int n, const int *pit, int P, int top_rt, int lgsb_rt,
                                                  unsigned tab_addr, unsigned hist_addr)
{
    const int top = LV ? (1 << LV) : top_rt;
    const int lgsb = LV ? 14 - LV : lgsb_rt;               // 16 KiB of copies: entry stride = 2^14 / entries
    typedef __attribute__((address_space(3))) const S *LdsS;
    typedef __attribute__((address_space(3))) unsigned *LdsU;
    const unsigned sb = 1u << lgsb;                               // bytes between consecutive entries of the lane's copy
    auto load_batch = [&](int it, S (&x)[ILP]) {
        #pragma unroll
        for (int q = 0; q < ILP; q++) {
            // coalesced; UNCONDITIONAL (index clamped, validity applied at use): a predicated load would hide the number
            // of loads in flight from the compiler and turn the wait for this batch into a wait for the prefetch too
            const long long item = i0 + (long long)(it + q) * STREAM_RANK_THREADS + threadIdx.x;
            x[q] = row[item < n ? item : n - 1];
        }
    };
    S nxt[ILP];
    load_batch(0, nxt);
    for (int it = 0; it < ipt; it += ILP) {
        S v[ILP]; unsigned o[ILP];                                // o = (number of positives below the candidate) * sb
        #pragma unroll
        for (int q = 0; q < ILP; q++) {                           // masked / NaN / beyond the row: bin 0
            const long long item = i0 + (long long)(it + q) * STREAM_RANK_THREADS + threadIdx.x;
            v[q] = (nxt[q] == nxt[q] && item < n) ? nxt[q] : -(S)INFINITY; o[q] = 0;
        }
        load_batch(it + ILP, nxt);                                // the next batch's HBM latency hides behind this batch's searches
        #pragma unroll
        for (unsigned st = ((unsigned)top >> 1) << lgsb; st >= sb; st >>= 1) {
            S pv[ILP];
            #pragma unroll
            for (int q = 0; q < ILP; q++) pv[q] = *(LdsS)(tab_addr + o[q] + st - sb);
            #pragma unroll
            for (int q = 0; q < ILP; q++) o[q] = (pv[q] < v[q]) ? o[q] + st : o[q];
        }
        S nx[ILP];
        #pragma unroll
        for (int q = 0; q < ILP; q++) nx[q] = *(LdsS)(tab_addr + o[q]);          // entry `top - 1` is +inf: always in range
        #pragma unroll
        for (int q = 0; q < ILP; q++) {
            int l = (int)(o[q] >> lgsb);
            if (nx[q] == v[q]) {                                  // exact tie with a positive (rare)
                const int item = (int)(i0 + (long long)(it + q) * STREAM_RANK_THREADS + threadIdx.x);
                while (l < P && *(LdsS)(tab_addr + ((unsigned)l << lgsb)) == v[q] && pit[l] > item) l++;
            }
            if (l >= 1) __hip_atomic_fetch_add((LdsU)(hist_addr + 4u * (unsigned)(l - 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// One wavefront per streamed user: ranks of its positives from the counts above (descending walk), the ROC ingredients,
// and the PR-AUC sum in the reference's own order -- left to right over the positives by descending score (:795-865) --
// so that PR-AUC is bit-identical for these users whatever the length of the row.  `count(j)` = scores counted in bin j
// ("exactly j + 1 positives rank below the candidate"), from the global counters (k_auc_streamed) or from the LDS of the block
// that has just counted them (k_rank_streamed, when one block holds a user's whole row).
template <class T, class S, class Count>
__device__ __forceinline__ void auc_walk_streamed(const FinalArgs<T, S> &a, int slot, int te0, int P, int lane, Count count)
{
    unsigned long long above = 0, sum_ranks = 0;
    double s2 = 0;
    int nvalid = 0;
    for (int base = P; base > 0; base -= WAVE) {
        const int j = base - 1 - lane;                            // lane 0 holds the best remaining positive
        const bool live = j >= 0;
        const unsigned long long h = live ? (unsigned long long)count(j) : 0ull;
        unsigned long long incl = h;
        #pragma unroll
        for (int dd = 1; dd < WAVE; dd <<= 1) {
            const unsigned lo32 = (unsigned)__shfl_up((int)(unsigned)incl, dd), hi32 = (unsigned)__shfl_up((int)(unsigned)(incl >> 32), dd);
            if (lane >= dd) incl += ((unsigned long long)hi32 << 32) | lo32;
        }
        const S ps = live ? a.spos_score[te0 + j] : (S)0;
        const bool valid = live && !(isinf(ps) && ps > 0);        // +inf = masked by the train row
        const unsigned long long vm = __ballot(valid);
        const unsigned long long rank = above + incl + 1;
        const int myidx = nvalid + __popcll(vm & ((1ull << lane) - 1ull)) + 1;
        const double term = valid ? (double)myidx / (double)rank : 0.;
        unsigned long long rs = valid ? rank : 0ull;
        #pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            const unsigned lo32 = (unsigned)__shfl_xor((int)(unsigned)rs, dd), hi32 = (unsigned)__shfl_xor((int)(unsigned)(rs >> 32), dd);
            rs += ((unsigned long long)hi32 << 32) | lo32;
        }
        sum_ranks += rs;
        for (unsigned long long mm = vm; mm; mm &= mm - 1) s2 += lane_bcast<double>(term, __ffsll((long long)mm) - 1);
        if (a.rank_sorted && valid) a.rank_sorted[te0 + j] = (long long)rank;
        const int last = (base < WAVE ? base : WAVE) - 1;         // lane of the lowest positive of this step
        above += ((unsigned long long)(unsigned)__shfl((int)(unsigned)(incl >> 32), last) << 32) | (unsigned)__shfl((int)(unsigned)incl, last);
        nvalid += __popcll(vm);
    }
    if (lane == 0) { AucPart r; r.sum_ranks = sum_ranks; r.s1 = 0; r.s2 = s2; r.nvalid = nvalid; r.pad = 0; a.auc_part[slot] = r; }
}

// does the block of k_rank_streamed that counts a streamed user's row also walk its counts?  (one block per row, table in LDS)
template <class S>
__device__ __forceinline__ bool rank_block_walks(int fused_auc, int P)
{
    if (!(fused_auc & 1)) return false;
    constexpr int per = (int)sizeof(S) / 4;
    int top = 1;
    while (top <= P) top <<= 1;
    return (long long)top * per + P + 1 <= STREAM_RANK_LDS / 4;
}

// fp32, the usual depths (64 ... 1023 positives: the copies fill the 16 KiB table), batches that lie inside the row: the same
// search with the sweep's instruction budget (rm_sweep.hpp auc_pass) -- 3 vector instructions and one LDS read per score and
// level instead of the compiler's 5 + 2 scalar ones, nothing per score for validity:
//   * the table is aligned to its size, so "entry e of the lane's copy" is tab_addr | (e << lgsb): a level is one OR (the
//     candidate address), one compare into one of three mask registers in rotation, one select two compares later;
//   * the two top levels compare against the three pivots every lane of the block shares (scalar registers): the quarter of
//     the table a score falls into is the number of sorted pivots below it -- three independent compare / select pairs, no read;
//   * a masked score is the NaN sentinel: every compare with it is false, it ends in bin 0 by itself; bin 0 is a word of
//     its own in front of the counters, so the histogram update needs no test either.
// Loads are unconditional 32-bit-indexed reads of whole batches; the ragged end of the row goes through the generic routine.
#define RM_FCMP_LT(m, p, x) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(p), "v"(x))
#define RM_FCMP_LT_S(m, p, x) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "s"(p), "v"(x))
#define RM_FSEL(d, a, b, m) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(m))
template <int LV>
__device__ __forceinline__ void rank_streamed_fast(const float *row, int i0, int iters, const int *pit, int P, unsigned tab_addr, unsigned hist0_addr,
                                                   float piv_lo, float piv_root, float piv_hi)
{
    constexpr int ILP = 8;
    constexpr int LGSB = 14 - LV;
    constexpr unsigned SB = 1u << LGSB;                           // bytes between consecutive entries of the lane's copy
    constexpr unsigned Q = SB << (LV - 2);                        // a quarter of the table
    typedef __attribute__((address_space(3))) const float *LdsS;
    const float *src = row + i0 + (int)threadIdx.x;
    float nxt[ILP];
    #pragma unroll
    for (int q = 0; q < ILP; q++) nxt[q] = src[q * STREAM_RANK_THREADS];
    const unsigned a1 = tab_addr | Q, a2 = tab_addr | (2 * Q), a3 = tab_addr | (3 * Q);
    for (int it = 0; it < iters; it += ILP) {
        float v[ILP]; unsigned at[ILP];
        #pragma unroll
        for (int q = 0; q < ILP; q++) { v[q] = nxt[q]; at[q] = tab_addr; }
        src += ILP * STREAM_RANK_THREADS;
        if (it + ILP < iters) {                                   // the next batch's HBM latency hides behind this batch's searches
            #pragma unroll
            for (int q = 0; q < ILP; q++) nxt[q] = src[q * STREAM_RANK_THREADS];
        }
        unsigned long long mk0, mk1, mk2;
        #pragma unroll
        for (int i = 0; i < 3 * ILP + 2; i++) {                   // i = pivot * ILP + score
            if (i < 3 * ILP) {
                const float pv = i < ILP ? piv_lo : (i < 2 * ILP ? piv_root : piv_hi);
                if (i % 3 == 0) RM_FCMP_LT_S(mk0, pv, v[i % ILP]); else if (i % 3 == 1) RM_FCMP_LT_S(mk1, pv, v[i % ILP]); else RM_FCMP_LT_S(mk2, pv, v[i % ILP]);
            }
            if (i >= 2) {
                const int j = i - 2;
                const unsigned tgt = j < ILP ? a1 : (j < 2 * ILP ? a2 : a3);
                if (j % 3 == 0) RM_FSEL(at[j % ILP], at[j % ILP], tgt, mk0); else if (j % 3 == 1) RM_FSEL(at[j % ILP], at[j % ILP], tgt, mk1); else RM_FSEL(at[j % ILP], at[j % ILP], tgt, mk2);
            }
        }
        #pragma unroll
        for (int st = 1 << (LV - 3); st >= 1; st >>= 1) {
            float pv[ILP];
            #pragma unroll
            for (int q = 0; q < ILP; q++) pv[q] = *(LdsS)(at[q] + (unsigned)(st - 1) * SB);
            #pragma unroll
            for (int i = 0; i < ILP + 2; i++) {
                // s_waitcnt lgkmcnt(4 / 0), the other counters left alone
                if (i == 0) __builtin_amdgcn_s_waitcnt(0xC47F); else if (i == 4) __builtin_amdgcn_s_waitcnt(0xC07F);
                if (i < ILP) { if (i % 3 == 0) RM_FCMP_LT(mk0, pv[i], v[i]); else if (i % 3 == 1) RM_FCMP_LT(mk1, pv[i], v[i]); else RM_FCMP_LT(mk2, pv[i], v[i]); }
                if (i >= 2) {
                    const int j = i - 2;
                    const unsigned cand = at[j] | ((unsigned)st * SB);
                    if (j % 3 == 0) RM_FSEL(at[j], at[j], cand, mk0); else if (j % 3 == 1) RM_FSEL(at[j], at[j], cand, mk1); else RM_FSEL(at[j], at[j], cand, mk2);
                }
            }
        }
        float nx[ILP];
        #pragma unroll
        for (int q = 0; q < ILP; q++) nx[q] = *(LdsS)(at[q]);             // entry `top - 1` is +inf: always in range
        bool tie = false;
        #pragma unroll
        for (int q = 0; q < ILP; q++) tie |= nx[q] == v[q];
        if (__any(tie)) {                                                 // exact tie with a positive (rare)
            #pragma unroll
            for (int q = 0; q < ILP; q++) {
                if (nx[q] == v[q]) {
                    const int item = i0 + (it + q) * STREAM_RANK_THREADS + (int)threadIdx.x;
                    int l = (int)((at[q] - tab_addr) >> LGSB);
                    while (l < P && *(LdsS)(tab_addr + ((unsigned)l << LGSB)) == v[q] && pit[l] > item) l++;
                    at[q] = tab_addr + ((unsigned)l << LGSB);
                }
            }
        }
        const unsigned one = 1u;
        #pragma unroll
        for (int q = 0; q < ILP; q++) {
            const unsigned ha = hist0_addr + (((at[q] - tab_addr) >> LGSB) << 2);
            asm volatile("ds_add_u32 %0, %1" :: "v"(ha), "v"(one) : "memory");
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);                                   // the asm atomics are invisible to the compiler
}

template <class T, class S>
__global__ __launch_bounds__(STREAM_RANK_THREADS) void k_rank_streamed(FinalArgs<T, S> a, int parts, int ipt, int row0)
{   // parts = blocks per row, ipt = items per thread (a multiple of 8): the row is cut into equal pieces
    __shared__ __attribute__((aligned(16384))) char rk_smem[STREAM_RANK_LDS];
    const int d = row0 + blockIdx.x / parts, part = blockIdx.x % parts;
    const int slot = a.stream_slot0 + d;
    const int u = a.slot_user[slot];
    if (a.flags[u] & UF_ONLY_NDCG) return;                            // (only when every user is streamed: no ranks wanted)
    const int te0 = a.test_p[u], P = a.test_p[u + 1] - te0;
    constexpr int WORDS = STREAM_RANK_LDS / 4;
    constexpr int per = (int)sizeof(S) / 4;                       // LDS words per score
    int top = 1;
    while (top <= P) top <<= 1;                                   // table = top entries: P positives, then +inf
    int R = 0;
    // (P + 1 counters: the word in front of them is bin 0, "below every positive", counted by the fast routine and never read)
    if ((long long)top * per + P + 1 <= WORDS) { R = 1; while (R < 32 && (long long)top * 2 * R * per + P + 1 <= WORDS) R *= 2; }
    const int *pit = a.spos_item + te0;
    const S *row = a.stream_scores + (size_t)d * (size_t)a.stream_ld;
    const long long i0 = (long long)part * STREAM_RANK_THREADS * ipt;
    if (R > 0) {
        S *lds_s = (S *)rk_smem;                                  // [top][R]
        unsigned *lds_h = (unsigned *)(rk_smem + sizeof(S) * (size_t)top * (size_t)R) + 1;
        for (int i = threadIdx.x; i < top * R; i += STREAM_RANK_THREADS) { const int e = i / R; lds_s[i] = e < P ? a.spos_score[te0 + e] : (S)INFINITY; }
        for (int i = threadIdx.x; i < P; i += STREAM_RANK_THREADS) lds_h[i] = 0u;
        __syncthreads();
        int lgsb = sizeof(S) == 4 ? 2 : 3;
        for (int r = R; r > 1; r >>= 1) lgsb++;
        const unsigned tab_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)rk_smem + (unsigned)((threadIdx.x & (R - 1)) * sizeof(S));
        const unsigned hist_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds_h;
        // (depth as a template constant when the copies fill the 4096-entry table: 64 ... 1023 positives)
        const bool full = (long long)top * R == 4096 / per;
        if (sizeof(S) == 4 && full && top >= 128 && top <= 1024 && !a.rank_generic) {
            // whole batches inside the row through the fast routine, the ragged end through the generic one
            const int inside = (int)((a.n - i0) / STREAM_RANK_THREADS);                      // iterations whose 256 items all lie in the row
            const int iters = inside <= 0 ? 0 : (inside < ipt ? inside : ipt) / 8 * 8;
            if (iters > 0) {
                const float *tabf = (const float *)lds_s;
                const float p_lo = tabf[(top / 4 - 1) * R], p_root = tabf[(top / 2 - 1) * R], p_hi = tabf[(3 * top / 4 - 1) * R];
                const float s_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p_lo)));
                const float s_root = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p_root)));
                const float s_hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p_hi)));
                const float *rowf = (const float *)row;
                const unsigned h0 = hist_addr - 4u;
                if (top == 128) rank_streamed_fast<7>(rowf, (int)i0, iters, pit, P, tab_addr, h0, s_lo, s_root, s_hi);
                else if (top == 256) rank_streamed_fast<8>(rowf, (int)i0, iters, pit, P, tab_addr, h0, s_lo, s_root, s_hi);
                else if (top == 512) rank_streamed_fast<9>(rowf, (int)i0, iters, pit, P, tab_addr, h0, s_lo, s_root, s_hi);
                else rank_streamed_fast<10>(rowf, (int)i0, iters, pit, P, tab_addr, h0, s_lo, s_root, s_hi);
            }
            if (iters < ipt) rank_streamed_lds<S, 8, 0>(row, i0 + (long long)iters * STREAM_RANK_THREADS, ipt - iters, a.n, pit, P, top, lgsb, tab_addr, hist_addr);
        } else
        if (full && top == 128) rank_streamed_lds<S, 8, 7>(row, i0, ipt, a.n, pit, P, top, lgsb, tab_addr, hist_addr);
        else if (full && top == 256) rank_streamed_lds<S, 8, 8>(row, i0, ipt, a.n, pit, P, top, lgsb, tab_addr, hist_addr);
        else if (full && top == 512) rank_streamed_lds<S, 8, 9>(row, i0, ipt, a.n, pit, P, top, lgsb, tab_addr, hist_addr);
        else if (full && top == 1024) rank_streamed_lds<S, 8, 10>(row, i0, ipt, a.n, pit, P, top, lgsb, tab_addr, hist_addr);
        else rank_streamed_lds<S, 8, 0>(row, i0, ipt, a.n, pit, P, top, lgsb, tab_addr, hist_addr);
        __syncthreads();
        if (parts == 1 && (a.fused_auc & 1)) {
            // this block has counted the user's whole row: its first wave walks the counts right here, out of LDS (k_auc_streamed's walk
            // without the trip through the global counters and without a launch behind this one).  When the dense rows mark the
            // users' own test items (fused_auc & 2) the sweep never saw those: positive i of the sorted table counts in bin i - 1,
            // exactly what k_merge_positives adds to the global counters
            if (threadIdx.x < WAVE) {
                const bool own = (a.fused_auc & 2) != 0;
                const S *sp = a.spos_score + te0;
                auc_walk_streamed<T, S>(a, slot, te0, P, (int)threadIdx.x, [&](int j) {
                    unsigned c = lds_h[j];
                    if (own && j + 1 < P) { const S x = sp[j + 1]; c += (!(isinf(x) && x > 0) && x == x) ? 1u : 0u; }
                    return c;
                });
            }
            return;
        }
        for (int i = threadIdx.x; i < P; i += STREAM_RANK_THREADS) { const unsigned c = lds_h[i]; if (c) atomicAdd(&a.shist[te0 + i], c); }
    } else {                                                      // very long row: search in global memory
        const S *tab = a.spos_score + te0;
        unsigned *hist = a.shist + te0;
        for (int it = 0; it < ipt; it++) {
            const long long item = i0 + (long long)it * STREAM_RANK_THREADS + threadIdx.x;
            if (item >= a.n) break;
            const S v = row[item];
            if (v != v) continue;
            int lo = 0;
            for (int st = top >> 1; st >= 1; st >>= 1) { const int idx = lo + st; if (idx <= P && tab[idx - 1] < v) lo = idx; }
            while (lo < P && tab[lo] == v && pit[lo] > (int)item) lo++;
            if (lo >= 1) atomicAdd(&hist[lo - 1], 1u);
        }
    }
}

// ---- top-K of a stored score row (k_metrics beyond what the sweep's lists hold) ---------------------------------------------
// Block per user.  Order everywhere: (score desc, item asc).  The K-th best score is found by a radix select over the
// order-preserving integer keys of the row (8 bits per pass, most significant first), the entries above it are gathered in any
// order, those EQUAL to it in item order until K are together, and the K entries are sorted by a block-wide bitonic network
// (in global scratch: K is unbounded; padded to a power of two with keys below every real one).
template <class S> struct SelKey;
template <> struct SelKey<float> { typedef unsigned T; static constexpr int BITS = 32; };
template <> struct SelKey<double> { typedef unsigned long long T; static constexpr int BITS = 64; };
constexpr int SELECT_THREADS = 256;

template <class T, class S>
__global__ __launch_bounds__(SELECT_THREADS) void k_select_topk(FinalArgs<T, S> a)
{
    typedef typename SelKey<S>::T KeyT;
    constexpr int BITS = SelKey<S>::BITS;
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sh_prefix;
    __shared__ int sh_remaining, sh_gt, sh_wsum[SELECT_THREADS / WAVE], sh_taken;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = a.stream_slot0 + row;
    const int u = a.slot_user[slot];
    const int n = a.n, K = a.K;
    const int C = n - (a.train_p[u + 1] - a.train_p[u]);
    const int W = K < C ? K : C;                                   // entries to produce (the row may hold NaN scores: then fewer exist)
    const S *sc = a.stream_scores + (size_t)row * (size_t)a.stream_ld;
    Entry<S> *M = a.merged + (size_t)u * K;
    unsigned long long *hi = a.sel_hi + (size_t)row * a.sel_ld;
    unsigned *lo = a.sel_lo + (size_t)row * a.sel_ld;
    if (tid == 0) { sh_prefix = 0ull; sh_remaining = W; }
    // ---- radix select: key of the W-th best score ----
    for (int pass = 0; pass < BITS / 8; pass++) {
        const int shift = BITS - 8 * (pass + 1);
        hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = sh_prefix;
        for (int i = tid; i < n; i += SELECT_THREADS) {
            const S x = sc[i];
            if (x != x) continue;                                    // masked by the train row / NaN score
            const KeyT key = ord_key(x);
            if (pass == 0 || (unsigned long long)(key >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int rem = sh_remaining, b = 255;
            unsigned long long total = 0;
            for (int q = 0; q < 256; q++) total += hist[q];
            if (total < (unsigned long long)rem) rem = (int)total;  // fewer real scores than W (only at the first pass)
            for (; b > 0; b--) { if ((int)hist[b] >= rem) break; rem -= (int)hist[b]; }
            sh_remaining = rem; sh_prefix = (prefix << 8) | (unsigned long long)b;
        }
        __syncthreads();
    }
    const KeyT T0 = (KeyT)sh_prefix;
    const int need_eq = sh_remaining;                                 // entries wanted among the scores equal to the W-th best
    // (how many exist in all: recount -- cheap, and it makes W_real exact when the row has fewer real scores than W)
    if (tid == 0) { sh_gt = 0; sh_taken = 0; }
    __syncthreads();
    // ---- gather: key > T0 anywhere, key == T0 in item order ----
    for (int base = 0; base < n; base += SELECT_THREADS) {
        const int i = base + tid;
        const S x = i < n ? sc[i] : (S)NAN;
        const bool real = x == x;
        const KeyT key = real ? ord_key(x) : (KeyT)0;
        const bool gt = real && key > T0, eq = real && key == T0;
        if (gt) { const int at = atomicAdd(&sh_gt, 1); hi[need_eq + at] = (unsigned long long)key; lo[need_eq + at] = ~(unsigned)i; }
        const unsigned long long em = __ballot(eq);
        if (lane == 0) sh_wsum[wave] = __popcll(em);
        __syncthreads();
        int before = sh_taken;
        for (int w = 0; w < wave; w++) before += sh_wsum[w];
        const int my = before + __popcll(em & ((1ull << lane) - 1ull));
        if (eq && my < need_eq) { hi[my] = (unsigned long long)key; lo[my] = ~(unsigned)i; }
        __syncthreads();
        if (tid == 0) { int t = sh_taken; for (int w = 0; w < SELECT_THREADS / WAVE; w++) t += sh_wsum[w]; sh_taken = t; }
        __syncthreads();
    }
    const int got = min(need_eq, sh_taken) + sh_gt;                   // == W unless the row has fewer real scores
    // the equal-keyed entries sit at [0, need_eq) and the larger ones behind them: close the gap when fewer equals exist
    // (cannot happen: need_eq <= number of equals by construction), pad to the power of two, sort
    for (int i = got + tid; i < a.sel_ld; i += SELECT_THREADS) { hi[i] = 0ull; lo[i] = 0u; }
    __syncthreads();
    int Kp = 2;
    while (Kp < got) Kp <<= 1;
    for (int k2 = 2; k2 <= Kp; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int idx = tid; idx < Kp; idx += SELECT_THREADS) {
                const int l = idx ^ j;
                if (l > idx) {
                    const unsigned long long ha = hi[idx], hb = hi[l];
                    const unsigned la = lo[idx], lb = lo[l];
                    const bool a_lt_b = ha < hb || (ha == hb && la < lb);
                    const bool desc = (idx & k2) == 0;
                    if (desc ? a_lt_b : !a_lt_b && !(ha == hb && la == lb)) { hi[idx] = hb; hi[l] = ha; lo[idx] = lb; lo[l] = la; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < K; i += SELECT_THREADS) {
        Entry<S> e;
        if (i < got) { e.s = ord_unkey((KeyT)hi[i]); e.idx = (int)~lo[i]; }
        else { e.s = (S)qnan<float>(); e.idx = -1; }
        M[i] = e;
    }
}

// ---- ordered top-K out of the sweep's lane buffers (k_metrics beyond the LDS lists; rm_list.hpp) -----------------------------
// Block per primary slot.  The user's candidates are spread over the lane buffers of every (item range, sub-tile wave) that swept
// it -- per wave the entries of its lanes, unsorted, every one at or above the bound the wave had when it appended it.  The block
// gathers them into LDS as (order-preserving score key, ~item) pairs -- only those that reach the user's final shared bound, a
// lower bound of its K-th best that at least K entries meet -- together with the entries of the "extra" part of `pl` (the user's own
// test items, when the sweep ran over rows that mask them: rm_sweep.hpp, k_merge_positives), sorts them by a bitonic network and
// writes the first K to `merged`, where k_finalize finds them.  When more entries come than LDS holds, it sorts and keeps K in
// between.  Order: (score desc, item asc) = (key desc, ~item desc).
// One WAVEFRONT per user, its entries in a private piece of LDS, no block barrier anywhere: a bitonic pass is `pairs / 64` compare-
// exchanges per lane between two wave-level fences.  (First version, round 6: a block of 256 threads per user with __syncthreads
// per pass -- 45 passes of a microsecond each with four blocks per CU: 4.5 ms for BASELINE C2's 138,493 users at K = 100.)
// CAPW = entries per wavefront: 1,024 (four users per block) up to K = 320, 4,096 (one user per block) beyond.
struct CollectGeom {
    int ublock0, n_ublocks, n_splits, tail_ublocks, tail_splits;   // the sweep's grid (rm_launch.hpp SweepArgs)
    int nsub, gu, lpu;                                              // sub-tile waves per group, users per group, lanes per user
    int lane_cap;
    int extra_part;                                                 // part of `pl` that holds the users' own test items, or -1
};
template <class S> struct CollectKey;
template <> struct CollectKey<float> { typedef unsigned T; };
template <> struct CollectKey<double> { typedef unsigned long long T; };
// (512 entries per wavefront up to K = 128: 16 KB of LDS per block of four users, and the CU holds its 32 wavefronts -- what a wavefront does per
// user is a chain of round trips to memory, more of them in flight is what helps: 1.43 -> 1.04 ms at BASELINE C2's shape with K = 21, 1.78 -> 1.43 with 100)
inline int collect_capw(int K, int lane_cap) { return K <= 128 ? 512 : (K + lane_cap <= 1024 && 2 * K <= 1024 ? 1024 : 4096); }
constexpr int COLLECT_MAX_ENTRIES = 4096;

#ifdef RM_CSTATS
__device__ unsigned long long g_cstats[16];                    // timing build: cycle counters of k_collect_topk's phases, summed over the wavefronts
#define RM_CSTAT(IDX, T0) do { const unsigned long long t_now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_cstats[IDX], t_now - (T0)); (T0) = t_now; } while (0)
#else
#define RM_CSTAT(IDX, T0) do { } while (0)
#endif
inline unsigned collect_grid(long long blocks) { return (unsigned)((blocks + 7) / 8 * 8); }
template <class T, class S, class ThrT, int CAPW>
__global__ __launch_bounds__(CAPW <= 1024 ? 256 : 64) void k_collect_topk(FinalArgs<T, S> a, CollectGeom g, const char *glists, const int *lane_cnt, const ThrT *thr_shared)
{
    typedef typename CollectKey<S>::T KeyT;
    constexpr int WPB = CAPW <= 1024 ? 4 : 1;                      // wavefronts (users) per block
    __shared__ KeyT kh_all[WPB * CAPW];
    __shared__ unsigned kl_all[WPB * CAPW];
    const int lane = threadIdx.x & 63, wv_in_blk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Sixteen neighbouring slots sit on neighbouring lanes of the same sweep waves: their entries share 128-byte lines.  Consecutive
    // block ids go round the eight XCDs, each with an L2 of its own -- in launch order a line was fetched over the fabric once per block
    // that touches it.  So the blocks of ONE XCD (ids b, b + 8, ...) take consecutive slots: a line comes in once and its other users
    // find it in that L2 (grid: a multiple of 8 blocks, collect_grid).
    const int per_xcd = gridDim.x >> 3;
    const int lblock = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int slot = lblock * WPB + wv_in_blk;
    if (slot >= a.n_slots) return;
#ifdef RM_CSTATS
    unsigned long long ct = __builtin_readcyclecounter();
#endif
    // (what a wavefront does per user is a chain of round trips to memory: the loads that depend on the slot alone go out together)
    const int chunk = a.slot_chunk[slot], u = a.slot_user[slot];
    const ThrT bound_raw = thr_shared[slot];
    // (no early exit for the few slots that are not a user's first: an exit here is a wait for `chunk` in front of every other load)
    const bool live = chunk == 0;
#ifdef RM_CSTATS
    asm volatile("" :: "s"(chunk), "s"(u));
    RM_CSTAT(0, ct);
#endif
    KeyT *kh = kh_all + wv_in_blk * CAPW;
    unsigned *kl = kl_all + wv_in_blk * CAPW;
    const int K = a.K;
    const int group = slot / g.gu, ul = slot % g.gu, gi = group % GROUPS_PER_BLOCK, blk_u = group / GROUPS_PER_BLOCK;
    const int nwaves = GROUPS_PER_BLOCK * g.nsub;
    // the sweep's two-level grid, inverted: the blocks that swept this user block (rm_sweep.hpp)
    const int rel = blk_u - g.ublock0, n_ub1 = g.n_ublocks - g.tail_ublocks;
    const bool in_tail = rel < g.tail_ublocks;
    const int nsplit = in_tail ? g.tail_splits : g.n_splits;
    const KeyT bound = (KeyT)bound_raw;                             // (0 = none: below every key)
    int cur = 0;                                                    // entries in LDS (wave-uniform)
    auto wave_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    // SELECT, then sort: the K best of the entries [0, cur) packed to the front in their old order.  The K-th best key by a radix
    // descent over the keys in LDS -- a pass per bit below the prefix all keys share (scores between a user's bound and its best
    // share a dozen leading bits), each a read, a compare and a ballot count per 64 entries -- then, only when more entries tie that
    // key than are wanted, the same descent over the item words of the ties.  (First version: a bitonic sort of EVERYTHING gathered,
    // 440 compare-exchanges per lane for 600 entries where this is ~25 passes of 10 reads and a 28-pass sort of the 128 kept.)
    // (the keys in REGISTERS for the passes: CAPW / 64 per lane, read from LDS once -- a pass per bit that re-read them was a chain of
    // LDS round trips, ~2,000 cycles per bit for 1,000 entries; the item words only when ties must be cut)
    // (always inline: out of line the closure -- `cur`, the LDS pointers -- lives in scratch memory, and so does every use of it elsewhere)
    auto select_best = [&]() __attribute__((always_inline)) {
        if (cur <= K) return;
        wave_sync();
        constexpr int ROWS = CAPW / WAVE;
        KeyT kr[ROWS];
        #pragma unroll
        for (int r = 0; r < ROWS; r++) { const int i = r * WAVE + lane; kr[r] = i < cur ? kh[i] : (KeyT)0; }
        KeyT all_and = ~(KeyT)0, all_or = 0;
        #pragma unroll
        for (int r = 0; r < ROWS; r++) if (r * WAVE + lane < cur) { all_and &= kr[r]; all_or |= kr[r]; }
        #pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { all_and &= __shfl_xor(all_and, d); all_or |= __shfl_xor(all_or, d); }
        const KeyT diff = all_and ^ all_or;
        constexpr int BITS = (int)sizeof(KeyT) * 8;
        int top = BITS - 1;                                         // highest bit in which two keys differ
        while (top >= 0 && !((diff >> top) & 1)) top--;
        KeyT Tk = top >= 0 ? (all_and & ~(((KeyT)2 << top) - 1)) : all_and;      // the shared prefix
        // (entries beyond `cur` hold key 0: below every candidate, which has a bit set)
        for (int bit = top; bit >= 0; bit--) {
            const KeyT cand = Tk | ((KeyT)1 << bit);
            int c = 0;
            #pragma unroll
            for (int r = 0; r < ROWS; r++) c += __popcll(__ballot(kr[r] >= cand));
            if (c >= K) Tk = cand;
        }
        int n_gt = 0, n_eq = 0;
        #pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const bool in = r * WAVE + lane < cur;
            n_gt += __popcll(__ballot(in && kr[r] > Tk)); n_eq += __popcll(__ballot(in && kr[r] == Tk));
        }
        const int need = K - n_gt;                                  // >= 1 of the n_eq ties: those with the largest item words
        unsigned Lk = 0u;
        if (n_eq > need) {
            const int rows = (cur + WAVE - 1) / WAVE;
            for (int bit = 31; bit >= 0; bit--) {
                const unsigned cand = Lk | (1u << bit);
                int c = 0;
                for (int r = 0; r < rows; r++) { const int i = r * WAVE + lane; c += __popcll(__ballot(i < cur && kh[i] == Tk && kl[i] >= cand)); }
                if (c >= need) Lk = cand;
            }
        }
        int out = 0;
        #pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if (r * WAVE >= cur) break;
            const int i = r * WAVE + lane;
            const KeyT x = kr[r]; const unsigned y = i < cur ? kl[i] : 0u;
            const bool keep = i < cur && (x > Tk || (x == Tk && y >= Lk));
            const unsigned long long m = __ballot(keep);
            if (keep) { const int at = out + __popcll(m & ((1ull << lane) - 1ull)); kh[at] = x; kl[at] = y; }      // (at <= i: a slot already read)
            out += __popcll(m);
        }
        cur = out;
        wave_sync();
    };
    // sort the entries [0, cur), cur <= K, descending: a bitonic network over PR * 64 >= cur elements IN REGISTERS (padded with keys below
    // every real one) -- element r * 64 + lane in register r: a compare-exchange at distance >= 64 is between two registers of a lane,
    // below 64 with the lane `lane ^ j` (a shuffle each for the key and the item word).  (First version: in LDS, four reads, four
    // writes and a fence per stage and pair: 36 stages of round trips for 256 entries.)
    auto sort_regs = [&](auto pr_tag) __attribute__((always_inline)) {
        constexpr int PR = decltype(pr_tag)::value, P = PR * WAVE;
        KeyT h[PR]; unsigned l[PR];
        #pragma unroll
        for (int r = 0; r < PR; r++) { const int i = r * WAVE + lane; const bool in = i < cur; h[r] = in ? kh[i] : (KeyT)0; l[r] = in ? kl[i] : 0u; }
        #pragma unroll
        for (int k2 = 2; k2 <= P; k2 <<= 1) {
            #pragma unroll
            for (int j = k2 >> 1; j > 0; j >>= 1) {
                if (j >= WAVE) {
                    const int jr = j / WAVE;
                    #pragma unroll
                    for (int r = 0; r < PR; r++) {
                        if (r & jr) continue;
                        const bool desc = (r & (k2 / WAVE)) == 0;
                        const bool lt = h[r] < h[r | jr] || (h[r] == h[r | jr] && l[r] < l[r | jr]);
                        if (desc ? lt : !lt) { const KeyT th = h[r]; h[r] = h[r | jr]; h[r | jr] = th; const unsigned tl = l[r]; l[r] = l[r | jr]; l[r | jr] = tl; }
                    }
                } else {
                    #pragma unroll
                    for (int r = 0; r < PR; r++) {
                        const KeyT oh = __shfl_xor(h[r], j); const unsigned ol = __shfl_xor(l[r], j);
                        const bool take_max = ((lane & j) == 0) == ((((r * WAVE) | lane) & k2) == 0);
                        const bool other_better = oh > h[r] || (oh == h[r] && ol > l[r]);
                        if (take_max ? other_better : !other_better) { h[r] = oh; l[r] = ol; }
                    }
                }
            }
        }
        #pragma unroll
        for (int r = 0; r < PR; r++) { kh[r * WAVE + lane] = h[r]; kl[r * WAVE + lane] = l[r]; }
        wave_sync();
    };
    auto sort_kept = [&]() __attribute__((always_inline)) {
        wave_sync();
        if (CAPW <= 1024) {                                    // (cur <= K <= 512 here: collect_capw)
            if (cur <= 64) sort_regs(std::integral_constant<int, 1>{});
            else if (cur <= 128) sort_regs(std::integral_constant<int, 2>{});
            else if (cur <= 256) sort_regs(std::integral_constant<int, 4>{});
            else sort_regs(std::integral_constant<int, 8>{});
        } else {
            if (cur <= 512) sort_regs(std::integral_constant<int, 8>{});
            else if (cur <= 1024) sort_regs(std::integral_constant<int, 16>{});
            else sort_regs(std::integral_constant<int, 32>{});
        }
    };
    auto reduce = [&]() __attribute__((always_inline)) { select_best(); };
    // one source: c entries, entry i fetched by `get(i, key, low)` (false: not a candidate)
    auto gather = [&](int c, auto get) __attribute__((always_inline)) {
        for (int i0 = 0; i0 < c; i0 += WAVE) {
            if (cur + WAVE > CAPW) reduce();
            KeyT key = 0; unsigned low = 0u;
            const bool take = i0 + lane < c && get(i0 + lane, key, low);
            const unsigned long long m = __ballot(take);
            if (take) { const int at = cur + __popcll(m & ((1ull << lane) - 1ull)); kh[at] = key; kl[at] = low; }
            cur += __popcll(m);
        }
    };
    // The sources: (item range, sub-tile wave, lane of the user) -- FOUR at a time, the same chunk of each in one batch of loads (a
    // user's entries sit 512 bytes apart: every load is a round trip to L2 or beyond, and source after source a wavefront had one
    // in flight).  fp32: score and item in one 8-byte load; fp64: the two arrays side by side.
    constexpr int G = 4;
    constexpr bool PAIRS = sizeof(S) == 4;                          // (rm_list.hpp LaneSel: fp32 (score, item) pairs; fp64 the wave's scores, then its item ids)
    const int per_range = g.nsub * g.lpu, nsrc = nsplit * per_range;
    const char *bs[G]; const int *bi[G]; int cn[G]; int cmax = 0;
    // (the four counts in ONE round trip: every address is computed first -- a source beyond the last reads entry 0 and is masked
    // afterwards --, then the four loads go out back to back.  The kernel runs beside k_rank_streamed, which streams 3 GB at BASELINE C2's
    // shape: a round trip to memory takes ~5 us there, and a wavefront's life is the number of them it makes one after the other;
    // timing build with cycle counters per phase, profiles/r6_ab_c2.txt r6w)
    auto describe = [&](int s0) __attribute__((always_inline)) {
        size_t cidx[G]; bool on[G];
        #pragma unroll
        for (int j = 0; j < G; j++) {
            const int src = s0 + j;
            on[j] = src < nsrc;
            const int srcc = on[j] ? src : 0;
            const int sp = srcc / per_range, rem = srcc % per_range, sub = rem / g.lpu, l = rem % g.lpu;
            const int bidx = in_tail ? n_ub1 * g.n_splits + sp * g.tail_ublocks + (g.tail_ublocks - 1 - rel)
                                     : sp * n_ub1 + (g.n_ublocks - 1 - rel);
            const size_t wv = (size_t)bidx * nwaves + (size_t)(sub * GROUPS_PER_BLOCK + gi);
            const int src_lane = ul + l * g.gu;
            const char *wbase = glists + wv * ((size_t)g.lane_cap * WAVE * (sizeof(S) + 4));
            cidx[j] = wv * WAVE + src_lane;
            bs[j] = wbase + (size_t)src_lane * (PAIRS ? 8 : sizeof(S));
            bi[j] = (const int *)(wbase + (size_t)g.lane_cap * WAVE * sizeof(S)) + src_lane;
        }
        int c[G];
        #pragma unroll
        for (int j = 0; j < G; j++) c[j] = lane_cnt[cidx[j]];
        cmax = 0;
        #pragma unroll
        for (int j = 0; j < G; j++) { cn[j] = (live && on[j]) ? c[j] : 0; cmax = cn[j] > cmax ? cn[j] : cmax; }
    };
#ifndef RM_ABL_COLLECT_NO_GATHER
    describe(0);                                                    // (the first sources' counts are on their way while the user's own test items come in)
#endif
#ifndef RM_ABL_COLLECT_NO_EXTRA
    if (g.extra_part >= 0) {
        const Entry<S> *px = a.pl + ((size_t)slot * a.n_part + g.extra_part) * K;
        gather(live ? K : 0, [&](int i, KeyT &key, unsigned &low) {
            Entry<S> e;                                             // (score and item in one load: the item alone first was a round trip more)
            __builtin_memcpy(&e, px + i, sizeof(e));
#ifdef RM_ABL_COLLECT_EXTRA_DISCARD
            if (e.idx != -77) return false;
#endif
            if (e.idx == IDX_EMPTY) return false;
            key = ord_key(e.s); low = ~(unsigned)e.idx;
            return true;
        });
    }
#endif
    RM_CSTAT(1, ct);
#ifdef RM_ABL_COLLECT_NO_GATHER
    for (int s0 = 0; s0 < 0; s0 += G) {
#else
    for (int s0 = 0; s0 < nsrc; s0 += G) {
#endif
        if (s0) describe(s0);
        // (measured and dropped: two chunks of each source per batch, eight loads in flight -- the room they need in LDS brings the
        // selection in the middle of the gather forward, 1.9 -> 2.6 ms at BASELINE C2's shape with k_metrics = 100)
        for (int i0 = 0; i0 < cmax; i0 += WAVE) {
            if (cur + G * WAVE > CAPW) reduce();
            const int i = i0 + lane;
            S sc[G]; int it[G];
            #pragma unroll
            for (int j = 0; j < G; j++) {
                sc[j] = (S)0; it[j] = 0;
                if (i < cn[j]) {
                    if (PAIRS) {
                        const uint2 e = *(const uint2 *)(bs[j] + (size_t)i * (WAVE * 8));
                        sc[j] = (S)__uint_as_float(e.x); it[j] = (int)e.y;
                    } else {
                        sc[j] = *(const S *)(bs[j] + (size_t)i * (WAVE * sizeof(S)));
                        it[j] = bi[j][(size_t)i * WAVE];
                    }
                }
            }
            #pragma unroll
            for (int j = 0; j < G; j++) {
                const KeyT key = ord_key(sc[j]);
                const bool take = i < cn[j] && key >= bound;
                const unsigned long long m = __ballot(take);
                if (take) { const int at = cur + __popcll(m & ((1ull << lane) - 1ull)); kh[at] = key; kl[at] = ~(unsigned)it[j]; }
                cur += __popcll(m);
            }
        }
    }
    wave_sync();
    RM_CSTAT(2, ct);
#ifndef RM_ABL_COLLECT_GATHER_ONLY                             // (timing only, scratch/build_abl.py: what the gather alone costs)
    select_best();
    sort_kept();
#endif
    RM_CSTAT(3, ct);
    if (!live) return;
#ifdef RM_ABL_COLLECT_NO_WRITE
    if (cur != 12345) return;
#endif
    Entry<S> *M = a.merged + (size_t)u * K;
    for (int i = lane; i < K; i += WAVE) {
        Entry<S> e;
        if (i < cur) { e.s = ord_unkey(kh[i]); e.idx = (int)~kl[i]; }
        else { e.s = (S)qnan<float>(); e.idx = -1; }
        M[i] = e;
    }
#ifdef RM_CSTATS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RM_CSTAT(4, ct);
    if (lane == 0) atomicAdd(&g_cstats[5], 1ull);
#endif
}

template <class T, class S>
__global__ void k_auc_streamed(FinalArgs<T, S> a, int row0, int row1)
{
    const int w = row0 + __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (w >= row1) return;
    const int slot = a.stream_slot0 + w;
    const int u = a.slot_user[slot];
    if (a.flags[u] & UF_ONLY_NDCG) return;
    const int te0 = a.test_p[u], P = a.test_p[u + 1] - te0;
    if (rank_block_walks<S>(a.fused_auc, P)) return;               // done by k_rank_streamed
    const unsigned *sh = a.shist + te0;
    auc_walk_streamed<T, S>(a, slot, te0, P, lane, [&](int j) { return sh[j]; });
}

// The L = min(K, npos) largest test VALUES of a user, descending (ideal DCG, reference :868-961), for users whose row is too
// long to walk on k_finalize's single thread: one wavefront per user, L rounds of "largest value after the previous pick in
// (value desc, position asc) order" with the row strided over the lanes.
template <class T, class S>
__global__ void k_top_values(FinalArgs<T, S> a)
{
    const int w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;     // (a wave's index: uniform, and known to be)
    if (w >= a.n_heavy) return;
    const int u = a.heavy_users[w];
    const int te0 = a.test_p[u], npos = a.test_p[u + 1] - te0;
    const int L = a.K < npos ? a.K : npos;
    if (L > a.heavy_ld) return;
    const T *tv = a.test_v + te0;
    T *out = a.heavy_topv + (size_t)u * a.heavy_ld;
    bool any_nan = false;
    T pv = 0; int pi = -1;
    // rows up to 64 * CACHE entries are read once into registers (all loads in flight together); longer ones are
    // re-read from L2 in every round
    constexpr int CACHE = 32;
    const bool cached = npos <= WAVE * CACHE;
    T xv[CACHE];
    if (cached) {
        #pragma unroll
        for (int i = 0; i < CACHE; i++) xv[i] = lane + WAVE * i < npos ? tv[lane + WAVE * i] : (T)0;
    }
    for (int r = 0; r < L; r++) {
        bool found = false; T bv = 0; int bi = 0x7fffffff;
        auto consider = [&](T x, int t) {
            if (r == 0) any_nan |= x != x;
            if (r > 0 && !(x < pv || (x == pv && t > pi))) return;
            if (!found || x > bv) { found = true; bv = x; bi = t; }          // ascending t: the first of equal values stays
        };
        if (cached) {
            #pragma unroll
            for (int i = 0; i < CACHE; i++) if (lane + WAVE * i < npos) consider(xv[i], lane + WAVE * i);
        } else {
            for (int t = lane; t < npos; t += WAVE) consider(tv[t], t);
        }
        #pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const T ov = __shfl_xor(bv, d); const int oi = __shfl_xor(bi, d); const bool of = __shfl_xor((int)found, d) != 0;
            if (of && (!found || ov > bv || (ov == bv && oi < bi))) { found = true; bv = ov; bi = oi; }
        }
        pv = bv; pi = bi;
        if (lane == 0) out[r] = bv;
    }
    const unsigned long long nm = __ballot(any_nan);
    if (lane == 0) a.heavy_nan[u] = nm ? 1 : 0;
}

// users that are not evaluated (reference :450-476): NaN in every requested output, empty ranking
template <class T, class S>
__global__ void k_finalize_skipped(FinalArgs<T, S> a)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= a.m || !(a.flags[u] & UF_NAN)) return;
    const int te0 = a.test_p[u], npos = a.test_p[u + 1] - te0;
    if (a.status) a.status[u] = 1;
    Entry<S> *M = a.merged + (size_t)u * a.K;
    for (int i = 0; i < a.K; i++) { M[i].s = (S)qnan<float>(); M[i].idx = -1; }
    fill_user_nan(a, u);                                            // rank_sorted was zeroed by the host
}

// ROC-AUC / PR-AUC of one user from the rank sums of its slots (reference :795-865)
template <class T, class S>
__device__ __forceinline__ void finalize_auc(const FinalArgs<T, S> &a, int u, int base, int npos, int C)
{
    const int nsl = a.user_nslots[u];
    unsigned long long sum_ranks = 0; int h = 0; double ap_full = 0;
    for (int c = nsl - 1; c >= 0; c--) {                           // chunks hold ascending scores: combine them downwards
        const AucPart r = a.auc_part[a.slot_index[base + c]];
        sum_ranks += r.sum_ranks;
        ap_full = nsl == 1 ? r.s2 : ap_full + ((double)h * r.s1 + r.s2);
        h += r.nvalid;
    }
    const unsigned long long P = (unsigned long long)npos, Nneg = (unsigned long long)C - P;
    if (a.roc) a.roc[u] = (T)(1. - (double)(sum_ranks - (P * (P + 1)) / 2) / (double)(P * Nneg));
    if (a.pr) a.pr[u] = (T)(ap_full / (double)npos);
}

// the two AUC outputs of the users k_finalize left open (slots [auc_defer_slot0, n_slots): the streamed users when their ranks are
// counted on the side stream)
template <class T, class S>
__global__ void k_finalize_auc(FinalArgs<T, S> a)
{
    const int slot = a.auc_defer_slot0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.n_slots || a.slot_chunk[slot] != 0) return;
    const int u = a.slot_user[slot];
    const T probe = a.roc ? a.roc[u] : a.pr[u];
    if (probe != probe) return;                                    // invalid user, or one without AUC (only_ndcg): NaN stays
    const int npos = a.test_p[u + 1] - a.test_p[u];
    const int C = a.n - (a.train_p[u + 1] - a.train_p[u]);
    finalize_auc(a, u, a.uslot_base[u], npos, C);
}

template <class T, class S>
__global__ void k_finalize(FinalArgs<T, S> a)
{
    // per-thread working arrays live in LDS as columns ([row][thread]): the partial-list cursors and the sorted buffer
    // of the largest test values are indexed dynamically, which in private memory means a scratch round trip per access
    extern __shared__ __attribute__((aligned(16))) char fin_smem[];
    T *topv_base = (T *)fin_smem;                                                     // [min(K, FIN_TOPV)][FIN_THREADS]
    unsigned short *head_base = (unsigned short *)(fin_smem + sizeof(T) * (a.K < FIN_TOPV ? a.K : FIN_TOPV) * FIN_THREADS);   // [n_part][FIN_THREADS]
    // One thread per SLOT, not per user: the 32 (16) users of a group sit on adjacent threads, so their reads of the
    // group's column-major tables (rank histogram, sorted positives: [row][user of the group]) coalesce into one line
    // per row, and the users of a group have similar row lengths.  Users without a slot are k_finalize_skipped's.
    const int slot = a.fin_slot0 + blockIdx.x * blockDim.x + threadIdx.x;          // this launch: slots [fin_slot0, fin_slot1)
    if (slot >= a.fin_slot1 || a.slot_chunk[slot] != 0) return;
    const int u = a.slot_user[slot];
    const int K = a.K, n = a.n;
    const int f = a.flags[u];
    const int te0 = a.test_p[u], npos = a.test_p[u + 1] - te0;
    if (a.status) a.status[u] = 1;
    Entry<S> *M = a.merged + (size_t)u * K;
    const bool merged_ready = a.ext_topk || a.collected;          // k_select_topk / k_collect_topk have written M already
    if (!merged_ready) for (int i = 0; i < K; i++) { M[i].s = (S)qnan<float>(); M[i].idx = -1; }

    const int ntr = a.train_p[u + 1] - a.train_p[u];
    const int C = n - ntr;
    const bool only_ndcg = f & UF_ONLY_NDCG, kleqn = f & UF_KLEQN;
    const int base = a.uslot_base[u];
    const int s0 = a.slot_index[base];
    const int NP = a.n_part;

    // ---- merge the partial top-K lists (each descending) and the validity stats ----
    unsigned short *head = head_base + threadIdx.x;               // head[q * FIN_THREADS]: entries of partial list q consumed
    unsigned short *alive = head_base + (size_t)NP * FIN_THREADS + threadIdx.x;     // alive[t * FIN_THREADS]: the parts that hold entries
    S vmax = -(S)INFINITY, vmin = (S)INFINITY; bool any_nan = false;
    const Entry<S> *PL = a.pl + (size_t)s0 * NP * K;
    int n_alive = 0;
    for (int q = 0; q < NP; q++) {
        // a part without entries (the sub-tile waves that share a group's LDS list write it once; item ranges this user's
        // block was not cut into) is left out of the merge: with many item ranges (few users: the exact pass of the tie noise)
        // two parts out of three are empty
        const bool empty = !merged_ready && PL[(size_t)q * K].idx == IDX_EMPTY;
        head[q * FIN_THREADS] = empty ? (unsigned short)K : (unsigned short)0;
        if (!empty) alive[(n_alive++) * FIN_THREADS] = (unsigned short)q;
        const PartialStat<S> ps = a.pst[(size_t)s0 * NP + q];
        vmax = ps.vmax > vmax ? ps.vmax : vmax;
        vmin = ps.vmin < vmin ? ps.vmin : vmin;
        any_nan |= ps.has_nan != 0;
    }
    for (int i = 0; i < (merged_ready ? 0 : K); i++) {
        int best = -1; Entry<S> be; be.s = 0; be.idx = 0;
        for (int t = 0; t < n_alive; t++) {
            const int q = alive[t * FIN_THREADS];
            const int hq = head[q * FIN_THREADS];
            if (hq >= K) continue;
            const Entry<S> e = PL[(size_t)q * K + hq];
            if (best < 0 || ent_before(e, be)) { best = q; be = e; }
        }
        if (best >= 0) head[best * FIN_THREADS]++;
        else be.idx = IDX_EMPTY;                                 // every part empty (e.g. all scores NaN)
        if (be.idx == IDX_EMPTY) { be.idx = -1; be.s = (S)qnan<float>(); }
        M[i] = be;
    }
    const int W = K < C ? K : C;

    // ---- validity (:517-526 with noise, :541-562 without; NaN anywhere => invalid, see DESIGN.md deviation D3) ----
    const bool ref_full = ((a.req & RQ_ROC) && !only_ndcg) || K >= C;
    bool invalid = any_nan;
    if (a.noise) invalid |= (vmax == vmin) || isinf(vmax) || isinf(vmin);
    else {
        const S hi = M[0].s, lo = ref_full ? vmin : M[W - 1].s;
        invalid |= isinf(hi) || isinf(lo) || hi == lo || M[W - 1].idx < 0;
    }
    if (invalid) {
        for (int i = 0; i < K; i++) { M[i].s = (S)qnan<float>(); M[i].idx = -1; }
        if (a.rank_sorted) for (int t = 0; t < npos; t++) a.rank_sorted[te0 + t] = 0;     // k_auc_slots ranked them already
        fill_user_nan(a, u); return;
    }
    if (a.status) a.status[u] = 0;
    if (a.noise_flag) {
        bool zone = false;
        for (int i = 0; i < W; i++) { const S x = M[i].s; zone |= (x < (S)0 ? -x : x) < (S)6.103515625e-05f; }
        if (zone && atomicExch(&a.noise_flag[u], 1) == 0) atomicAdd(&a.plan->n_noise_flagged, 1);
    }
#if defined(RM_ABL_FIN_STOP) && RM_ABL_FIN_STOP == 1
    return;
#endif

    const int *ti = a.test_i + te0;
    const T *tv = a.test_v ? a.test_v + te0 : nullptr;
    const size_t st = (size_t)u * K;
    T *cp = a.p ? a.p + st : nullptr, *ctp = a.tp ? a.tp + st : nullptr, *cr = a.r ? a.r + st : nullptr;
    T *cap = a.ap ? a.ap + st : nullptr, *ctap = a.tap ? a.tap + st : nullptr, *cndcg = a.ndcg ? a.ndcg + st : nullptr;
    T *chit = a.hit ? a.hit + st : nullptr, *crr = a.rr ? a.rr + st : nullptr;
    const bool cum = a.cumulative;

    // ---- top-K walk (:589-748) ----
    const bool top = a.req & (RQ_P | RQ_TP | RQ_R | RQ_AP | RQ_TAP | RQ_NDCG | RQ_HIT | RQ_RR);
    bool walked = false;
    int hits = 0, first = 0x7fffffff;
    double avg_p = 0, dcg = 0;
    if (top && (!kleqn || (a.req & (RQ_AP | RQ_TAP | RQ_RR | RQ_NDCG)))) {
        walked = true;
        for (int ix = 0; ix < W; ix++) {
            const int item = M[ix].idx;
            int lo = 0, hi = npos;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (ti[mid] < item) lo = mid + 1; else hi = mid; }
            if (lo < npos && ti[lo] == item) {
                hits++;
                avg_p += hits / (double)(ix + 1);
                dcg += tv ? ((double)tv[lo] / a.log2tab[ix]) : 0.;
                first = first < ix ? first : ix;
            }
            if (cum) {
                const int mn = (ix + 1) < npos ? (ix + 1) : npos;
                if (cp) cp[ix] = (T)(hits / (double)(ix + 1));
                if (ctp) ctp[ix] = (T)(hits / (double)mn);
                if (cr) cr[ix] = (T)(hits / (double)npos);
                if (cap) cap[ix] = (T)(avg_p / (double)npos);
                if (ctap) ctap[ix] = (T)(avg_p / (double)mn);
                if (cndcg) cndcg[ix] = (T)dcg;
                if (chit) chit[ix] = (T)(hits > 0);
                if (crr) crr[ix] = (T)(hits ? ((double)1 / (double)(first + 1)) : 0.);
            }
        }
        if (!cum) {
            const int mn = K < npos ? K : npos;
            if (a.p) a.p[u] = (T)((double)hits / (double)K);
            if (a.tp) a.tp[u] = (T)((double)hits / (double)mn);
            if (a.r) a.r[u] = (T)((double)hits / (double)npos);
            if (a.ap) a.ap[u] = (T)(avg_p / (double)npos);
            if (a.tap) a.tap[u] = (T)(avg_p / (double)mn);
            if (a.hit) a.hit[u] = (T)(hits > 0);
            if (a.rr) a.rr[u] = (T)(hits ? (1. / (double)(first + 1)) : 0.);
        } else if (K > C) {
            for (int i = C; i < K; i++) {
                if (cp) cp[i] = qnan<T>(); if (ctp) ctp[i] = qnan<T>(); if (cr) cr[i] = qnan<T>(); if (chit) chit[i] = qnan<T>();
                if (cap) cap[i] = cap[C - 1]; if (ctap) ctap[i] = ctap[C - 1]; if (crr) crr[i] = crr[C - 1]; if (cndcg) cndcg[i] = cndcg[C - 1];
            }
        }
    }

#if defined(RM_ABL_FIN_STOP) && RM_ABL_FIN_STOP == 2
    return;
#endif
    // ---- NaN overrides (:750-788) ----
    if (kleqn) {
        if (!cum) {
            if (a.p) a.p[u] = qnan<T>(); if (a.tp) a.tp[u] = qnan<T>(); if (a.r) a.r[u] = qnan<T>(); if (a.hit) a.hit[u] = qnan<T>();
        } else if (!walked) {
            for (int i = 0; i < K; i++) { if (cp) cp[i] = qnan<T>(); if (ctp) ctp[i] = qnan<T>(); if (cr) cr[i] = qnan<T>(); if (chit) chit[i] = qnan<T>(); }
        }
    } else if (only_ndcg) {
        if (!cum) {
            if (a.p) a.p[u] = qnan<T>(); if (a.tp) a.tp[u] = qnan<T>(); if (a.r) a.r[u] = qnan<T>(); if (a.ap) a.ap[u] = qnan<T>();
            if (a.tap) a.tap[u] = qnan<T>(); if (a.hit) a.hit[u] = qnan<T>(); if (a.rr) a.rr[u] = qnan<T>();
        } else {
            for (int i = 0; i < K; i++) {
                if (cp) cp[i] = qnan<T>(); if (ctp) ctp[i] = qnan<T>(); if (cr) cr[i] = qnan<T>(); if (cap) cap[i] = qnan<T>();
                if (ctap) ctap[i] = qnan<T>(); if (chit) chit[i] = qnan<T>(); if (crr) crr[i] = qnan<T>();
            }
        }
    }

    // ---- ROC-AUC / PR-AUC from the rank histograms (:795-865) ----
    if (only_ndcg) {
        if (a.roc) a.roc[u] = qnan<T>();
        if (a.pr) a.pr[u] = qnan<T>();
    } else if (a.req & (RQ_ROC | RQ_PR)) {
        if (slot >= a.auc_defer_slot0) {
            // the rank sums of this user are still being formed on the side stream (k_rank_streamed / k_auc_streamed): everything else
            // of the user is done here, beside them; k_finalize_auc fills these two in behind the join.  A zero says "valid, to be
            // filled in" (an invalid user's outputs are all NaN by now and stay so)
            if (a.roc) a.roc[u] = (T)0;
            if (a.pr) a.pr[u] = (T)0;
        } else finalize_auc(a, u, base, npos, C);
    }

#if defined(RM_ABL_FIN_STOP) && RM_ABL_FIN_STOP == 3
    return;
#endif
    // ---- NDCG normalisation (:868-961) ----
    if (a.ndcg) {
        const int L = K < npos ? K : npos;
        // the ideal DCG needs the L largest test values in descending order: ONE pass over the row keeping a small
        // sorted buffer (values only -- equal values are interchangeable in the sums); rows longer than the buffer's
        // K fall back to repeated selection
        constexpr int TOPV = FIN_TOPV;
        T *topv = topv_base + threadIdx.x;                         // topv[i * FIN_THREADS]
        bool has_nan_val = false;
        const bool buffered = L <= TOPV;
        auto pick_next = [&](bool have_prev, T pv, int pi, T &ov, int &oi) {
            bool found = false; T bv = 0; int bi = -1;
            for (int t = 0; t < npos; t++) {
                const T x = tv[t];
                if (have_prev && !(x < pv || (x == pv && t > pi))) continue;
                if (!found || x > bv) { found = true; bv = x; bi = t; }
            }
            ov = bv; oi = bi; return found;
        };
        // (k_top_values' list, one wave per such user: rows longer than a.heavy_npos -- HEAVY_NPOS, or FIN_TOPV when k_metrics is beyond
        // the buffer: the repeated selection below is L x npos DEPENDENT loads on one thread, 2.4 ms for ONE user with 70 test items
        // at K = 100 among BASELINE C4's 8,192 and seconds for a row of thousands)
        const bool listed = a.heavy_topv && npos > a.heavy_npos && L <= a.heavy_ld;
        const T *hv = listed ? a.heavy_topv + (size_t)u * a.heavy_ld : nullptr;
        if (listed) has_nan_val = a.heavy_nan[u] != 0;
        if (listed && buffered) {
            for (int i = 0; i < L; i++) topv[i * FIN_THREADS] = hv[i];
        } else if (listed) {
        } else if (buffered) {
            int cnt = 0;
            T kth = 0;                                               // topv[L - 1] once the buffer is full
            for (int t0 = 0; t0 < npos; t0 += 8) {                 // eight loads in flight: a heavy user's row is thousands long
                T xb[8];
                #pragma unroll
                for (int q = 0; q < 8; q++) xb[q] = t0 + q < npos ? tv[t0 + q] : (T)0;
                #pragma unroll
                for (int q = 0; q < 8; q++) {
                    if (t0 + q >= npos) break;
                    const T x = xb[q];
                    has_nan_val |= x != x;
                    if (cnt < L || x > kth) {
                        int j = cnt < L ? cnt : L - 1;
                        while (j > 0 && topv[(j - 1) * FIN_THREADS] < x) { topv[j * FIN_THREADS] = topv[(j - 1) * FIN_THREADS]; j--; }
                        topv[j * FIN_THREADS] = x;
                        cnt = cnt < L ? cnt + 1 : cnt;
                        if (cnt == L) kth = topv[(L - 1) * FIN_THREADS];
                    }
                }
            }
        } else {
            for (int t = 0; t < npos; t++) has_nan_val |= tv[t] != tv[t];
        }
        // value i of the descending order
        T pv = 0; int pi = -1; bool hp = false;
        auto next_value = [&](int i) -> T {
            if (buffered) return topv[i * FIN_THREADS];
            if (listed) return hv[i];
            T x; int xi; pick_next(hp, pv, pi, x, xi); hp = true; pv = x; pi = xi; return x;
        };
        T vmaxv = 0, vlast = 0;
        if (buffered) { vmaxv = topv[0]; vlast = topv[(L - 1) * FIN_THREADS]; }
        else if (listed) { vmaxv = hv[0]; vlast = hv[L - 1]; }
        else {
            for (int i = 0; i < L; i++) { const T x = next_value(i); if (i == 0) vmaxv = x; vlast = x; }
            hp = false; pv = 0; pi = -1;
        }
        if (has_nan_val || isinf(vmaxv) || isinf(vlast) || vmaxv <= 0) {
            if (!cum) a.ndcg[u] = qnan<T>(); else for (int i = 0; i < K; i++) cndcg[i] = qnan<T>();
            return;
        }
        double idcg = 0;
        if (!cum) {
            for (int ix = 0; ix < L; ix++) {
                const T x = next_value(ix);
                if (!(vlast >= 0) && x <= 0) break;
                idcg += (double)x / a.log2tab[ix];
            }
            a.ndcg[u] = (T)(dcg / idcg);
        } else {
            if (vlast >= 0) {
                for (int ix = 0; ix < L; ix++) {
                    const T x = next_value(ix);
                    idcg += (double)x / a.log2tab[ix];
                    cndcg[ix] = (T)((double)cndcg[ix] / idcg);                      // quirk Q5: divides the rounded DCG
                }
            } else {
                int ix = 0;
                for (; ix < L; ix++) {
                    const T x = next_value(ix);
                    if (x < 0) break;
                    idcg += (double)x / a.log2tab[ix];
                    cndcg[ix] = (T)((double)cndcg[ix] / idcg);
                }
                for (; ix < L; ix++) cndcg[ix] = (T)((double)cndcg[ix] / idcg);
            }
            if (npos < K) { const int e = K < C ? K : C; for (int i = npos; i < e; i++) cndcg[i] = cndcg[npos - 1]; }   // quirk Q6
        }
    }
}

} // namespace rm
