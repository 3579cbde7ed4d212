// rm_prep.hpp -- plan building, operand packing and positive-score kernels (everything before the sweep).
//
// Replaces, on device, the per-user bookkeeping of reference src/recometrics.hpp:439-497 (eligibility filter,
// candidate list) -- the candidate list itself is never materialised: train-item masking is a predicate in the
// sweep's epilogue.
#pragma once
#include "rm_device.hpp"

namespace rm {

struct ClassifyArgs {
    int m, n, K;
    const int *train_p, *test_p;
    int req;                 // RQ_* mask
    int cold, min_items_pool, min_pos_test;     // already clamped as reference recometrics.hpp:390-393
    int want_auc;            // ROC or PR requested
    int *flags;              // [m]
    int *user_nslots;        // [m]
    int *heavy_users;        // [m] list of the evaluated users with more than HEAVY_NPOS test items (plan->n_heavy of them)
    Plan *plan;
    const unsigned char *only;   // optional [m]: evaluate only the users with a non-zero entry (the others: UF_SKIP)
    int force_stream;            // 1 = every evaluated user is streamed (top-K picked from its stored scores: k_metrics > 256)
    int allow_stream;            // 1 = users with more than POS_CHUNK test items are streamed (the host has a budget for score rows; it
                                 //     looks at the count afterwards and plans again without when the rows do not fit)
    int check_ptr;               // 1 = validate the index pointers of every row here (first pass of a call over these users)
    int heavy_npos;              // test rows longer than this are listed in `heavy_users` (HEAVY_NPOS; FIN_TOPV when k_metrics is beyond k_finalize's buffer)
    long long nnz_train, nnz_test;
};

__device__ __forceinline__ int chunk_depth(int pc)      // smallest j with 2^j - 1 >= pc   (pc in 1..63)
{
    return 32 - __clz(pc);
}

// ---- validation of the CSR inputs (first kernels of the plan) ----------------------------------------------------------
// k_classify (ClassifyArgs::check_ptr): index pointers of the m rows start at >= 0, never decrease and end within the index arrays.
// k_check_csr_flat / k_check_csr_starts: every column index lies in [0, n) and every row ascends (equal neighbours allowed, as
// SciPy's has_sorted_indices).  They report through plan->csr_bad and the descent counters; every later kernel that
// would index by what these arrays hold (k_assign_slots, k_block_tables, k_train_bits) returns when INDPTR / INDEX is set.
// (the index pointers alone, for check_csr_now: the validation in front of the fp64 tie noise, which indexes by the rows before the plan runs)
__global__ void k_check_csr_ptr(int m, const int *train_p, long long nnz_train, const int *test_p, long long nnz_test, Plan *plan)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= m) return;
    const int a0 = train_p[u], a1 = train_p[u + 1], b0 = test_p[u], b1 = test_p[u + 1];
    if (a0 < 0 || a1 < a0 || (long long)a1 > nnz_train || b0 < 0 || b1 < b0 || (long long)b1 > nnz_test) { atomicOr(&plan->csr_bad, CSR_BAD_INDPTR); plan->csr_where = u; }
}
// The indices themselves, FLAT: one thread per four entries of the index array (16-byte loads, fully coalesced: the 80 MB of
// BASELINE C2 in ~25 us; a walk row by row -- 16 lanes per row, two dependent loads per step -- took 177 us).  Range: every
// entry in [0, n).  Order: every DESCENT idx[e] > idx[e + 1] of the flat array is counted; a descent is legitimate exactly when
// e + 1 starts a row, and k_check_csr_starts counts those per row boundary -- the rows are sorted iff the two counts agree
// (the host compares them in the plan read-back).  Entries [p[0], p[m]) of this call's rows only.
constexpr int CHECK_FLAT_THREADS = 256, CHECK_FLAT_BLOCKS = 1024;      // (a grid of at most that many blocks: one atomic per block at the end)
__global__ __launch_bounds__(CHECK_FLAT_THREADS) void k_check_csr_flat(int m, int n, const int *indptr, const int *idx, Plan *plan, int which)
{
    if (plan->csr_bad & CSR_BAD_INDPTR) return;
    const long long e0 = indptr[0], e1 = indptr[m];
    const bool aligned = (((size_t)idx) & 15) == 0;
    unsigned desc = 0; bool bad = false; long long bad_at = 0;
    // (aligned groups of four entries of the ARRAY, so that the 16-byte loads are aligned whatever e0 is)
    for (long long g = (e0 >> 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; (g << 2) < e1; g += (long long)gridDim.x * blockDim.x) {
        const long long e = g << 2;
        if (e >= e0 && e + 4 < e1 && aligned) {
            const int4 q = *(const int4 *)(idx + e);
            const int x4 = idx[e + 4];
            const int x[5] = {q.x, q.y, q.z, q.w, x4};
            #pragma unroll
            for (int i = 0; i < 4; i++) { if ((unsigned)x[i] >= (unsigned)n) { bad = true; bad_at = e + i; } desc += x[i] > x[i + 1]; }
        } else {
            for (int i = 0; i < 4; i++) {
                const long long f = e + i;
                if (f < e0 || f >= e1) continue;
                const int v = idx[f];
                if ((unsigned)v >= (unsigned)n) { bad = true; bad_at = f; }
                if (f + 1 < e1) desc += v > idx[f + 1];
            }
        }
    }
    if (bad) {                                                    // rare: name the row (the last row that starts at or before the entry)
        int lo = 0, hi = m;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((long long)indptr[mid] <= bad_at) lo = mid; else hi = mid - 1; }
        atomicOr(&plan->csr_bad, CSR_BAD_INDEX); plan->csr_where = lo;
    }
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) desc += __shfl_xor(desc, d);
    __shared__ unsigned blk;
    if (threadIdx.x == 0) blk = 0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && desc) atomicAdd(&blk, desc);
    __syncthreads();
    if (threadIdx.x == 0 && blk) atomicAdd(&plan->csr_desc_all[which], (unsigned long long)blk);
}
// the legitimate descents: one thread per row boundary (the smallest u with p[u] = s names each distinct start s once)
__global__ __launch_bounds__(1024) void k_check_csr_starts(int m, const int *train_p, const int *train_i, const int *test_p, const int *test_i, Plan *plan)
{
    if (plan->csr_bad & CSR_BAD_INDPTR) return;
    __shared__ unsigned blk[2];
    if (threadIdx.x < 2) blk[threadIdx.x] = 0;
    __syncthreads();
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned d0 = 0, d1 = 0;
    if (u >= 1 && u < m) {
        const int s = train_p[u], e1 = train_p[m], e0 = train_p[0];
        if (train_p[u - 1] < s && s > e0 && s < e1) d0 = train_i[s - 1] > train_i[s];
        const int t = test_p[u], f1 = test_p[m], f0 = test_p[0];
        if (test_p[u - 1] < t && t > f0 && t < f1) d1 = test_i[t - 1] > test_i[t];
    }
    const unsigned long long b0 = __ballot(d0), b1 = __ballot(d1);
    if ((threadIdx.x & 63) == 0) {
        if (b0) atomicAdd(&blk[0], (unsigned)__popcll(b0));
        if (b1) atomicAdd(&blk[1], (unsigned)__popcll(b1));
    }
    __syncthreads();
    if (threadIdx.x < 2 && blk[threadIdx.x]) atomicAdd(&plan->csr_desc_legit[threadIdx.x], (unsigned long long)blk[threadIdx.x]);
}

// reference recometrics.hpp:439-448, :479-486  (one thread per user)
// Also, in the same pass over the index pointers: their validation (check_ptr), the number of users with more than POS_CHUNK test
// items and the longest such row (plan->n_long, max_npos: eligibility is not looked at, as an upper bound of the streamed class).
__global__ void k_classify(ClassifyArgs a)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const bool skip = u < a.m && a.only && !a.only[u];
    bool live = u < a.m && !skip;
    if (skip) { a.flags[u] = UF_SKIP; a.user_nslots[u] = 0; }
    int ntr = 0, npos = 0;
    if (live) {
        const int a0 = a.train_p[u], a1 = a.train_p[u + 1], b0 = a.test_p[u], b1 = a.test_p[u + 1];
        if (a.check_ptr && (a0 < 0 || a1 < a0 || (long long)a1 > a.nnz_train || b0 < 0 || b1 < b0 || (long long)b1 > a.nnz_test)) {
            atomicOr(&a.plan->csr_bad, CSR_BAD_INDPTR); a.plan->csr_where = u;
        } else { ntr = a1 - a0; npos = b1 - b0; }
    }
    const int cand = a.n - ntr;
    bool isnan_user = !live || npos <= 0 || (ntr + npos >= a.n && !(a.req & RQ_NDCG)) || cand < a.min_items_pool ||
                      (!a.cold && ntr == 0) || npos < a.min_pos_test;
    const bool only_ndcg = (ntr + npos) >= a.n;
    const bool kleqn = cand <= a.K;
    if (!isnan_user && kleqn && !(a.req & (RQ_ROC | RQ_PR | RQ_AP | RQ_TAP | RQ_RR))) isnan_user = true;
    int f = 0, nsl = 0, myclass = -1, nfull = 0;
    if (isnan_user) f = UF_NAN;
    else {
        f = UF_ACTIVE | (only_ndcg ? UF_ONLY_NDCG : 0) | (kleqn ? UF_KLEQN : 0);
        if (a.force_stream || (a.want_auc && !only_ndcg && npos > POS_CHUNK && a.allow_stream)) {
            nsl = 1;                                            // streamed user: one slot, ranks from its stored score row
            myclass = STREAM_CLASS;
        } else if (a.want_auc && !only_ndcg) {
            nfull = npos / POS_CHUNK;
            const int rem = npos % POS_CHUNK;
            nsl = nfull + (rem ? 1 : 0);
            myclass = rem ? chunk_depth(rem) : -1;
        } else {
            nsl = 1;
            myclass = 0;
        }
    }
    // counts are aggregated per block in LDS, then one global atomic per block and class (the counters share a line)
    __shared__ int blk_count[N_CLASSES + 3];                      // [N_CLASSES] = evaluated users, [+1] long rows, [+2] the longest of them
    __shared__ int blk_heavy[2];                                  // heavy users of the block, their first place in the list
    if (threadIdx.x < N_CLASSES + 3) blk_count[threadIdx.x] = 0;
    if (threadIdx.x < 2) blk_heavy[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long act = __ballot(!isnan_user);
    if (act && lane == __ffsll((long long)act) - 1) atomicAdd(&blk_count[N_CLASSES], __popcll(act));
    for (int j = 0; j < N_CLASSES; j++) {
        const unsigned long long mk = __ballot(myclass == j);
        if (mk && lane == __ffsll((long long)mk) - 1) atomicAdd(&blk_count[j], __popcll(mk));
    }
    if (nfull) atomicAdd(&blk_count[MAX_J], nfull);
    const bool lng = live && npos > POS_CHUNK;
    const unsigned long long lm = __ballot(lng);
    if (lm) {
        int mx = lng ? npos : 0;
        #pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
        if (lane == __ffsll((long long)lm) - 1) { atomicAdd(&blk_count[N_CLASSES + 1], __popcll(lm)); atomicMax(&blk_count[N_CLASSES + 2], mx); }
    }
    if (live) { a.flags[u] = f; a.user_nslots[u] = nsl; }
    // (the list's places are taken per block: with k_metrics beyond k_finalize's buffer every row of more than 64 test items is
    // listed -- a fifth of the users at BASELINE C2's shape -- and one returning atomic per user on one counter is 10 ns each)
    const bool heavy = live && !isnan_user && npos > a.heavy_npos;
    const int heavy_at = heavy ? atomicAdd(&blk_heavy[0], 1) : 0;
    if (live && !isnan_user && only_ndcg) atomicAdd(&a.plan->n_only_ndcg, 1);                              // rare
    __syncthreads();
    if (threadIdx.x == 0 && blk_heavy[0]) blk_heavy[1] = atomicAdd(&a.plan->n_heavy, blk_heavy[0]);
    __syncthreads();
    if (heavy) a.heavy_users[blk_heavy[1] + heavy_at] = u;
    if (threadIdx.x < N_CLASSES && blk_count[threadIdx.x]) atomicAdd(&a.plan->class_count[threadIdx.x], blk_count[threadIdx.x]);
    if (threadIdx.x == N_CLASSES && blk_count[N_CLASSES]) atomicAdd(&a.plan->n_active, blk_count[N_CLASSES]);
    if (threadIdx.x == N_CLASSES + 1 && blk_count[N_CLASSES + 1]) { atomicAdd(&a.plan->n_long, blk_count[N_CLASSES + 1]); atomicMax(&a.plan->max_npos, blk_count[N_CLASSES + 2]); }
}

// Long arrays are scanned in two launches: k_scan_tiles (every block scans its own 1024 entries and reports its total) and
// k_scan_exclusive over the block totals; the consumer (k_assign_slots) adds its tile's offset itself.
__global__ __launch_bounds__(1024) void k_scan_tiles(const int *in, int *out, int count, int *tile_total)
{
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = blockIdx.x * 1024 + tid;
    const int v = i < count ? in[i] : 0;
    int x = v;
    #pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    if (i < count) out[i] = woff + x - v;
    if (tid == 1023) tile_total[blockIdx.x] = woff + x;
}
// exclusive scan of int array by ONE block of 1024 threads (m <= 2^31; a few hundred iterations at m = 1M)
__device__ inline void plan_classes(Plan *p, int gu, int n_slots)      // one thread; gu = users per group (32 fp32, 16 fp64)
{
    int off = 0, jmax = 0;
    for (int j = 0; j < N_CLASSES; j++) {
        p->class_offset[j] = off;
        off += p->class_count[j];
        if (p->class_count[j] && j <= MAX_J) jmax = j;         // deepest TABLE: streamed users have none
        p->class_cursor[j] = 0;
    }
    p->class_offset[N_CLASSES] = off;
    p->n_groups = (n_slots + gu - 1) / gu;
    p->jmax = jmax;
}

// (`classes`, optional: the thread that holds the total also lays out the depth classes -- what used to be a launch of one thread)
__global__ void k_scan_exclusive(const int *in, int *out, int count, int *total_out, Plan *classes, int gu)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < count; base += 1024) {
        const int i = base + tid;
        const int v = i < count ? in[i] : 0;
        int x = v;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const int carry = carry_s;
        if (i < count) out[i] = carry + woff + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0 && total_out) *total_out = carry_s;
    if (tid == 0 && classes) plan_classes(classes, gu, carry_s);
}

struct AssignArgs {
    int m;
    const int *test_p, *flags, *user_nslots, *uslot_base;
    int want_auc;
    Plan *plan;
    int *slot_user, *slot_chunk, *slot_index;
    unsigned char *slot_j;
    int *sc_user, *sc_chunk;     // work list of the streamed users' chunks of POS_CHUNK test entries (plan->n_stream_chunks of them)
    int force_stream;
    const int *tile_offset;      // optional: uslot_base holds scans of tiles of 1024 users (k_scan_tiles); tile t starts at tile_offset[t]
};

// scatter every (user, chunk) into its depth class; order inside a class is arbitrary (results do not depend on it).
// Cursor bumps are aggregated per block in LDS (ASSIGN_THREADS users): the 8 class cursors share one cache line, and one
// returning atomic per wave and class serialised the whole kernel behind that line.
constexpr int ASSIGN_THREADS = 1024;
__global__ __launch_bounds__(ASSIGN_THREADS) void k_assign_slots(AssignArgs a)
{
    __shared__ int blk_count[N_CLASSES], blk_base[N_CLASSES], blk_chunks, blk_chunk_base;
    if (a.plan->csr_bad & CSR_BAD_INDPTR) return;                 // (row lengths that cannot be trusted would index out of the slot arrays)
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < N_CLASSES) blk_count[threadIdx.x] = 0;
    if (threadIdx.x == N_CLASSES) blk_chunks = 0;
    __syncthreads();
    const int nsl = u < a.m ? a.user_nslots[u] : 0;
    const int npos = nsl ? a.test_p[u + 1] - a.test_p[u] : 0;
    int ubase = 0;
    if (u < a.m) {                                                // the scan's last step, in place: every later reader sees final values
        ubase = a.uslot_base[u] + (a.tile_offset ? a.tile_offset[u >> 10] : 0);
        if (a.tile_offset) ((int *)a.uslot_base)[u] = ubase;
    }
    const bool auc_user = nsl && a.want_auc && !(a.flags[u] & UF_ONLY_NDCG);
    // last (or only) chunk of every user: position inside the block's share of its class
    int jlast = -1, in_blk = 0;
    const bool streamed = nsl && (a.force_stream || (auc_user && nsl == 1 && npos > POS_CHUNK));
    if (nsl) jlast = streamed ? STREAM_CLASS : auc_user ? chunk_depth(min(POS_CHUNK, npos - (nsl - 1) * POS_CHUNK)) : 0;
    for (int j = 0; j < N_CLASSES; j++) {
        const unsigned long long mk = __ballot(jlast == j);
        if (!mk) continue;
        const int leader = __ffsll((long long)mk) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&blk_count[j], __popcll(mk));
        base = __shfl(base, leader);
        if (jlast == j) in_blk = base + __popcll(mk & ((1ull << lane) - 1));
    }
    // the full 63-positive chunks of heavy users (rare): nsl - 1 consecutive positions each in the deepest class
    int full_at = 0;
    if (nsl > 1) full_at = atomicAdd(&blk_count[MAX_J], nsl - 1);
    // the streamed users' work list of chunks: places inside the block's share by an LDS atomic, ONE global atomic per block (a
    // fifth of BASELINE C2's users are streamed: one returning global atomic each, all on one word, was most of this kernel's 30 us)
    const int nch = (streamed && auc_user) ? (npos + POS_CHUNK - 1) / POS_CHUNK : 0;
    int ch_at = 0;
    if (nch) ch_at = atomicAdd(&blk_chunks, nch);
    __syncthreads();
    if (threadIdx.x < N_CLASSES && blk_count[threadIdx.x]) blk_base[threadIdx.x] = atomicAdd(&a.plan->class_cursor[threadIdx.x], blk_count[threadIdx.x]);
    if (threadIdx.x == N_CLASSES && blk_chunks) blk_chunk_base = atomicAdd(&a.plan->n_stream_chunks, blk_chunks);
    __syncthreads();
    if (nch) {                                                   // the order of the work list does not matter
        const int at = blk_chunk_base + ch_at;
        for (int c = 0; c < nch; c++) { a.sc_user[at + c] = u; a.sc_chunk[at + c] = c; }
    }
    if (nsl) {
        const int pos = a.plan->class_offset[jlast] + blk_base[jlast] + in_blk;
        a.slot_user[pos] = u;
        a.slot_chunk[pos] = nsl - 1;
        a.slot_j[pos] = (unsigned char)jlast;
        a.slot_index[ubase + nsl - 1] = pos;
    }
    for (int c = 0; c < nsl - 1; c++) {
        const int pos = a.plan->class_offset[MAX_J] + blk_base[MAX_J] + full_at + c;
        a.slot_user[pos] = u;
        a.slot_chunk[pos] = c;
        a.slot_j[pos] = (unsigned char)MAX_J;
        a.slot_index[ubase + c] = pos;
    }
}

// per sweep block (4 groups): uniform tree depth jb = depth of its last slot (slots are sorted by depth); the block's positive
// tables take 2^jb - 1 rows per group.  ONE block of 1024 threads walks the sweep blocks 1024 at a time: depth and rows of each,
// their running sum, and from it gj / grow of the block's groups (what used to be k_block_rows -> a scan -> k_group_rows).
__global__ __launch_bounds__(1024) void k_block_tables(Plan *p, const unsigned char *slot_j, int *gj, long long *grow, int gu)
{
    __shared__ int wsum[16];
    __shared__ long long carry_s;
    if (p->csr_bad & CSR_BAD_INDPTR) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ng = p->n_groups, ns = p->n_slots;
    const int nb = (ng + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int b = base + tid;
        int jb = 0, ngb = 0;
        if (b < nb) {
            const int glast = min(ng, (b + 1) * GROUPS_PER_BLOCK) - 1;
            const int slast = min(ns, (glast + 1) * gu) - 1;
            jb = slot_j[slast];
            if (jb == STREAM_CLASS) {
                // streamed slots are the last class: a block of nothing else has no tables at all, the one block that straddles
                // the boundary sizes its tables for the deepest of its other slots
                const int first_stream = p->class_offset[STREAM_CLASS];
                jb = first_stream > b * GROUPS_PER_BLOCK * gu ? slot_j[first_stream - 1] : 0;
            }
            ngb = glast - b * GROUPS_PER_BLOCK + 1;
        }
        const int per_group = (1 << jb) - 1;
        const int v = ngb * per_group;
        int x = v;
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d); if (lane >= d) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; w++) woff += wsum[w];
        const long long carry = carry_s;
        const long long mine = carry + woff + x - v;               // rows in front of this block's tables
        for (int q = 0; q < ngb; q++) {
            const int g = b * GROUPS_PER_BLOCK + q;
            gj[g] = jb;
            grow[g] = mine + (long long)q * per_group;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + x;
        __syncthreads();
    }
    if (tid == 0) p->total_rows = carry_s;
}

// max |x| over a row-major matrix (rows x k, leading dimension ld) + a non-finite flag: lets the host prove that no
// score can overflow, in which case the sweep skips its NaN scan.
template <class T>
__global__ void k_absmax(const T *X, size_t ld, long long rows, int k, unsigned long long *amax_bits, int *nonfinite)
{
    double mx = 0.; bool bad = false;
    const long long total = rows * k;
    auto take = [&](T xv) {
        const double x = (double)xv;
        const double ax = x < 0 ? -x : x;
        bad |= !(ax <= 1.7976931348623157e308);          // NaN or Inf
        mx = ax > mx ? ax : mx;
    };
    if (ld == (size_t)k && (((size_t)X) & 15) == 0) {      // dense rows: one flat array, 16-byte loads, no division per element
        constexpr int V = 16 / (int)sizeof(T);
        typedef T VecT __attribute__((ext_vector_type(16 / sizeof(T))));
        const long long nv = total / V;
        // (four loads in flight per thread: with one, 35 MB of user factors took 34 us in front of the plan read-back)
        const long long stride = (long long)gridDim.x * blockDim.x;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += 4 * stride) {
            VecT q[4];
            #pragma unroll
            for (int u = 0; u < 4; u++) { const long long j = i + u * stride; q[u] = ((const VecT *)X)[j < nv ? j : i]; }
            #pragma unroll
            for (int u = 0; u < 4; u++)
                #pragma unroll
                for (int e = 0; e < V; e++) take(q[u][e]);
        }
        for (long long i = nv * V + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) take(X[i]);
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
            take(X[(size_t)(i / k) * ld + (i % k)]);
    }
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(mx, d); mx = o > mx ? o : mx; }
    // one atomic per BLOCK (the waves of a block meet in LDS first): thousands of atomics on one word serialise -- the kernel over
    // the 6.8 MB of BASELINE C2's item factors took longer (49 us) than the one over 35 MB of user factors with half the waves
    __shared__ unsigned long long blk_max;
    __shared__ int blk_bad;
    if (threadIdx.x == 0) { blk_max = 0ull; blk_bad = 0; }
    __syncthreads();
    const bool anybad = __any(bad);
    if ((threadIdx.x & 63) == 0) {
        if (!anybad) atomicMax(&blk_max, (unsigned long long)__double_as_longlong(mx));
        else atomicOr(&blk_bad, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (blk_bad) atomicOr(nonfinite, 1);
        else if (blk_max) atomicMax(amax_bits, blk_max);
    }
}

// A head start for the streaming top-K: the user's test items are candidates themselves, so the K-th best of THEIR scores (known
// before the sweep: k_pos_scores) is a valid lower bound of the K-th best score overall.  It seeds thr_shared, the bound every
// partial list of the user filters with from its first tile on (the sweep only ever raises it).  With a model that ranks the
// test items high this removes most of the warm-up inserts; with random factors about a fifth.  Users whose positives span
// several slots are left alone (their best positives are not in the primary slot's table).
template <class T, class KeyT>
__device__ __forceinline__ KeyT seed_of_slot(int slot, int stream_slot0, int K, int gu, const int *slot_user, const int *slot_chunk, const int *user_nslots,
                                             const int *flags, const int *test_p, const long long *grow, const T *pos_score, const T *spos_score)
{
    const int u = slot_user[slot];
    if (slot_chunk[slot] != 0 || user_nslots[u] != 1 || (flags[u] & UF_ONLY_NDCG)) return (KeyT)0;
    const int te0 = test_p[u], P = test_p[u + 1] - te0;
    if (P < K) return (KeyT)0;
    const T *tab; long long stride;
    if (slot >= stream_slot0) { tab = spos_score + te0; stride = 1; }
    else { const int g = slot / gu; tab = pos_score + (grow[g] + g) * gu + (slot % gu); stride = gu; }
    int nvalid = P;                                               // positives masked by the train row sort to the top as +inf
    while (nvalid > 0) { const T x = tab[(long long)(nvalid - 1) * stride]; if (isinf(x) && x > 0) nvalid--; else break; }
    if (nvalid < K) return (KeyT)0;
    // the K-th best DISTINCT score of the test row: a non-canonical CSR row may list an item twice (the reference only sorts
    // the rows, recometrics/__init__.py:478-486), and K entries are then fewer than K candidates -- entries with equal scores
    // count once, which can only lower the bound
    // (k_pos_place ranks the entries by (score, item): two entries of the same item share a rank and a row, and the row behind
    // them keeps its +inf filling -- such holes are not scores)
    int i = nvalid - 1, distinct = 1;
    T kth = tab[(long long)i * stride];
    while (distinct < K && i > 0) {
        const T x = tab[(long long)(--i) * stride];
        if (x != kth && !(isinf(x) && x > 0)) { kth = x; distinct++; }
    }
    if (distinct < K) return (KeyT)0;
    return kth == kth ? (KeyT)ord_key(kth) : (KeyT)0;
}
// (every slot's bound is WRITTEN -- 0 = "nothing yet" where there is no seed -- so the array needs no memset in front of the kernel)
template <class T, class KeyT>
__global__ void k_seed_thresholds(int n_slots, int stream_slot0, int K, int gu, const int *slot_user, const int *slot_chunk, const int *user_nslots,
                                  const int *flags, const int *test_p, const long long *grow, const T *pos_score, const T *spos_score, KeyT *thr_shared)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    thr_shared[slot] = seed_of_slot<T, KeyT>(slot, stream_slot0, K, gu, slot_user, slot_chunk, user_nslots, flags, test_p, grow, pos_score, spos_score);
}

// SAMPLE SEEDS (lane buffers, short item axes).  A list whose bound starts at -inf appends every score of its first tiles, and its
// first selections re-read all of that: at BASELINE C2's shape with k_metrics = 100 the selections were 29 % of the sweep's wave
// cycles (47 % at 256; profiles/r6_ab_c2.txt).  ANY K candidates of a user bound its K-th best score from below, so the host has the
// sweep's own kernel score the first `S` items of the catalogue for every slot (its DUMP variant: same packed operands, same
// matrix instructions -- the scores are the sweep's to the bit) and this kernel, a wavefront per primary slot, takes the K-th best
// of the row that is neither masked (the dense train rows leave the all-ones NaN in the accumulators) nor a train item (the sparse
// row's prefix below S) by a radix descent over the ordered keys -- 32 x NV ballots -- and raises the slot's shared bound to it.
// The sweep then starts with a pass rate of K / S instead of 1.  (k_seed_thresholds' bounds come from the positives alone: users
// with fewer than K of them get none.)
template <class S, class KeyT, int NV>
__global__ __launch_bounds__(256) void k_seed_from_sample(int n_slots, int K, const S *sample, const int *slot_user, const int *slot_chunk,
                                                          const int *train_p, const int *train_i, KeyT *thr_shared)
{
    constexpr int SN = NV * WAVE;
    __shared__ unsigned masked[4][SN / 32];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slot = blockIdx.x * 4 + wv;
    if (slot >= n_slots || slot_chunk[slot] != 0) return;
    const int u = slot_user[slot];
    if (u < 0) return;
    KeyT key[NV];
    const S *row = sample + (size_t)slot * SN;
    #pragma unroll
    for (int j = 0; j < NV; j++) { const S x = row[j * WAVE + lane]; key[j] = x == x ? (KeyT)ord_key(x) : (KeyT)0; }
    for (int w = lane; w < SN / 32; w += WAVE) masked[wv][w] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    const int tr0 = train_p[u], tr1 = train_p[u + 1];
    for (int i0 = tr0; i0 < tr1; i0 += WAVE) {                           // the sorted row's prefix of items below the sample's end
        const int t = i0 + lane < tr1 ? train_i[i0 + lane] : SN;
        if (t >= 0 && t < SN) atomicOr(&masked[wv][t >> 5], 1u << (t & 31));
        if (__ballot(t >= SN)) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    #pragma unroll
    for (int j = 0; j < NV; j++) { const int it = j * WAVE + lane; if ((masked[wv][it >> 5] >> (it & 31)) & 1u) key[j] = 0; }
    // the largest key with at least K keys at or above it (key 0: masked, never counted -- ord_key of a number is never 0)
    // (SEED_BITS of the key -- sign, exponent, 8 bits of the mantissa: the bound rounded down is a bound all the same, 0.4 % further down)
    constexpr int BITS = (int)sizeof(KeyT) * 8, SEED_BITS = sizeof(S) == 4 ? 17 : 20;
    KeyT best = 0;
    for (int bit = BITS - 1; bit > BITS - 1 - SEED_BITS; bit--) {
        const KeyT probe = best | ((KeyT)1 << bit);
        int cnt = 0;
        #pragma unroll
        for (int j = 0; j < NV; j++) cnt += __popcll(__ballot(key[j] >= probe));
        if (cnt >= K) best = probe;
    }
    // (keys at or below -inf's: no bound; below it they are NaNs)
    const KeyT ninf_key = sizeof(S) == 4 ? (KeyT)0x007fffffu : (KeyT)0x000fffffffffffffull;
    if (best > ninf_key && lane == 0) atomicMax(thr_shared + slot, best);
}

// the tables of sorted positives start as +inf everywhere, their rank histograms as zero: one launch for both arrays
template <class T> __global__ void k_init_tables(T *scores, unsigned *hist, long long count)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { scores[i] = (T)__int_as_float(0x7f800000); hist[i] = 0u; }
}

// dense train rows for the sweep (SweepArgs::train_bits): a wavefront builds a user's row in LDS (the padding beyond n set, LDS
// atomic OR per item) and writes it out in one coalesced sweep -- every word of the buffer is written, no memset.  The wavefronts
// are resident and walk the users with a stride; a user's index pointers and first items are loaded while the row of the user
// before is built and written (one user in flight per wavefront hides nothing: the round-4 kernel needed every wave slot of the
// device and 107 KB of LDS per CU for its 0.2 ms, and the plan's kernels, the read-back's copy and the positives' kernels beside
// it had to wait for a slot).  Four blocks per CU: half of the wave slots stay free.  words <= TRAIN_BITS_MAX_WORDS, a multiple of 6 (rows are 8-byte aligned).
// (Measured and dropped: no LDS, background by stores and the items by L2 atomics -- 5 ms for BASELINE C2's 20 M items.)
constexpr int TRAIN_BITS_MAX_WORDS = 4096;            // 131,072 items: 16 KiB of LDS per wavefront
constexpr int TRAIN_BITS_WAVES = 4;
// With `test_p` the TEST items are marked as well (`bits`), and the sweep then never sees a user's own test items: they come
// back in k_merge_positives (rm_finalize.hpp); the tie noise, which indexes its draws by the train items alone, clears the test
// items' bits in its own LDS copy of a row (rm_noise.hpp k_noise_rows_bits).
// `plan`: the CSR arrays are only walked when the plan's validation kernels found every index pointer and index in range (an item
// beyond n would be an LDS write out of the row).  `only` (optional): rows of the users with a non-zero entry only -- the others'
// are never read (a pass over a few flagged users of a large range).
struct TrainRowPtrs { int tr0, tr1, te0, te1; };
struct TrainRowHead { int ia, ib, it; };                          // a user's first 128 train items and first 64 test items (-1: none)
__global__ __launch_bounds__(TRAIN_BITS_WAVES * WAVE) void k_train_bits(int m, int n, int words, const int *train_p, const int *train_i,
                                                                        const int *test_p, const int *test_i, unsigned *bits,
                                                                        const Plan *plan, const unsigned char *only,
                                                                        unsigned char *ent_masked, int mark_test)
{
    extern __shared__ unsigned tb_lds[];                      // [TRAIN_BITS_WAVES][words]
    if (plan && (plan->csr_bad & (CSR_BAD_INDPTR | CSR_BAD_INDEX))) return;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    unsigned *row = tb_lds + (size_t)wv * words;
    const int stride = gridDim.x * TRAIN_BITS_WAVES;
    auto next_user = [&](int v) { while (v < m && only && !only[v]) v += stride; return v; };
    // (the index pointers are wave-uniform, but loaded as per-lane values: as scalars the compiler wants them at once -- a wait for
    // the load in front of everything else -- and the point is to have them arrive during the work on another row)
    int lane0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane0));
    auto ptrs = [&](int u) {
        TrainRowPtrs p;
        const int v = u + lane0;
        p.tr0 = train_p[v]; p.tr1 = train_p[v + 1];
        p.te0 = test_p ? test_p[v] : 0; p.te1 = test_p ? test_p[v + 1] : 0;
        return p;
    };
    auto head = [&](const TrainRowPtrs &p) {
        TrainRowHead h;
        h.ia = p.tr0 + lane < p.tr1 ? train_i[p.tr0 + lane] : -1;
        h.ib = p.tr0 + WAVE + lane < p.tr1 ? train_i[p.tr0 + WAVE + lane] : -1;
        h.it = p.te0 + lane < p.te1 ? test_i[p.te0 + lane] : -1;
        return h;
    };
    auto background = [n](int w) { return (w << 5) >= n ? 0xffffffffu : (((w << 5) + 32 > n) ? (0xffffffffu << (n & 31)) : 0u); };
    // two users ahead: the index pointers of the user after next and the first items of the next user are in flight while this
    // user's row is built and written.  Three sets of registers take turns (a loop over one set would copy the loaded values at
    // its back edge, i.e. wait for them there).
    struct Row { int u; TrainRowPtrs p; TrainRowHead h; };
    Row r0, r1, r2;
    r0.u = next_user(blockIdx.x * TRAIN_BITS_WAVES + wv);
    if (r0.u >= m) return;
    r0.p = ptrs(r0.u);
    r1.u = next_user(r0.u + stride);
    r1.p = r1.u < m ? ptrs(r1.u) : r0.p;
    r0.h = head(r0.p);
    r1.h = r0.h; r2 = r1;
    auto step = [&](Row &cur, Row &nxt, Row &nn) {
        nn.u = nxt.u < m ? next_user(nxt.u + stride) : m;
        if (nn.u < m) nn.p = ptrs(nn.u);
        if (nxt.u < m) nxt.h = head(nxt.p);
        for (int w = 2 * lane; w < words; w += 2 * WAVE) *(uint2 *)(row + w) = make_uint2(background(w), background(w + 1));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (cur.h.ia >= 0) atomicOr(&row[cur.h.ia >> 5], 1u << (cur.h.ia & 31));
        if (cur.h.ib >= 0) atomicOr(&row[cur.h.ib >> 5], 1u << (cur.h.ib & 31));
        for (int e = cur.p.tr0 + 2 * WAVE + lane; e < cur.p.tr1; e += WAVE) { const int item = train_i[e]; atomicOr(&row[item >> 5], 1u << (item & 31)); }
        if (ent_masked) {
            // with the train items in and before the test items are: is a test item a train item?  (what the positives' scores by
            // entry need to know, k_pos_apply_masked: one LDS read here instead of a binary search of the train row)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (cur.h.it >= 0) ent_masked[cur.p.te0 + lane] = (row[cur.h.it >> 5] >> (cur.h.it & 31)) & 1u;
            for (int e = cur.p.te0 + WAVE + lane; e < cur.p.te1; e += WAVE) { const int item = test_i[e]; ent_masked[e] = (row[item >> 5] >> (item & 31)) & 1u; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (mark_test) {
            if (cur.h.it >= 0) atomicOr(&row[cur.h.it >> 5], 1u << (cur.h.it & 31));
            for (int e = cur.p.te0 + WAVE + lane; e < cur.p.te1; e += WAVE) { const int item = test_i[e]; atomicOr(&row[item >> 5], 1u << (item & 31)); }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        unsigned *out = bits + (size_t)cur.u * words;
        // (streaming stores: 463 MB at BASELINE C2 that nothing reads before the sweep -- as ordinary stores they wash the item factors
        // out of the L2 under the positives' scores that run beside this kernel)
        for (int w = 2 * lane; w < words; w += 2 * WAVE) {
            const uint2 v = *(const uint2 *)(row + w);
            __builtin_nontemporal_store(v.x, out + w); __builtin_nontemporal_store(v.y, out + w + 1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        return nxt.u < m;
    };
    for (;;) {
        if (!step(r0, r1, r2)) break;
        if (!step(r1, r2, r0)) break;
        if (!step(r2, r0, r1)) break;
    }
}
// (`test_p` with `mark_test` = 0: the test rows are read for `ent_masked` only)
inline void launch_train_bits(hipStream_t stream, int m, int n, int words, const int *train_p, const int *train_i, const int *test_p, const int *test_i,
                              unsigned *bits, const Plan *plan, const unsigned char *only, unsigned char *ent_masked = nullptr, bool mark_test = true)
{
    const unsigned blocks = (unsigned)std::min<long long>(((long long)m + TRAIN_BITS_WAVES - 1) / TRAIN_BITS_WAVES, 256 * 4);
    hipLaunchKernelGGL(k_train_bits, dim3(blocks), dim3(TRAIN_BITS_WAVES * WAVE), sizeof(unsigned) * (size_t)TRAIN_BITS_WAVES * (size_t)words, stream,
                       m, n, words, train_p, train_i, test_p, test_i, bits, plan, only, ent_masked, (test_p && mark_test) ? 1 : 0);
}

template <class T> __global__ void k_fill(T *p, T v, long long count)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}

// ---- operand packing: the packed layouts ARE the LDS / register images the MFMA loop reads -----------------------
// fp32, v_mfma_f32_32x32x2_f32: lane (r = lane & 31, h = lane >> 5) supplies element [r][k = 2*step + h].  A b128
// read covers 4 steps, so a lane wants the 4 factors k = 8g + 2i + h (i = 0..3) contiguous:
//   packed[tile][g][h][row][i] = X[tile*ROWS + row][8g + 2i + h]        (zero for k >= k or row >= rows)
// The k order of the accumulate chain is unchanged: step 4g+i adds k = 8g+2i (h = 0) then k = 8g+2i+1 (h = 1).
template <class T>
__global__ void k_pack_items(const T *B, size_t ldb, int n, int k, int NG, int tile_items, float4 *Bp, long long total_f4)
{
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 of the image per thread
    if (o >= total_f4) return;
    const int row = (int)(o % tile_items);                                      // tile_items = 64 or 96 (rm_sweep.hpp NSUB)
    const int h = (int)((o / tile_items) & 1);
    const long long tg = o / (2 * tile_items);
    const int g = (int)(tg % NG);
    const long long tile = tg / NG;
    const long long item = tile * tile_items + row;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (item < n) {
        const T *src = B + (size_t)item * ldb;
        #pragma unroll
        for (int i = 0; i < 4; i++) { const int kk = 8 * g + 2 * i + h; if (kk < k) v[i] = (float)src[kk]; }
    }
    Bp[o] = make_float4(v[0], v[1], v[2], v[3]);
}

// The same image, a block per TILE: the tile's rows are read as rows (consecutive lanes along the factors: whole lines) into LDS
// and the image is written from there, 16 bytes per lane.  (k_pack_items reads four 4-byte pieces 8 bytes apart per lane, 64 rows per
// instruction: at 10 M items x 128 factors -- BASELINE C4 -- 10 GB moved in 3.8 ms.)  Row stride in LDS = 8 NG + 1 words: the
// lanes of the second phase walk down a column without bank conflicts.  Used while a tile fits (pack_items_tile_lds <= 96 KB).
inline size_t pack_items_tile_lds(int NG, int tile_items) { return sizeof(float) * (size_t)tile_items * (size_t)(8 * NG + 1); }
template <class T>
__global__ __launch_bounds__(256) void k_pack_items_tile(const T *B, size_t ldb, int n, int k, int NG, int tile_items, float4 *Bp)
{
    extern __shared__ float pk_tile[];                          // [tile_items][8 NG + 1]
    const int kw = 8 * NG, ld = kw + 1, tid = threadIdx.x;
    const long long item0 = (long long)blockIdx.x * tile_items;
    const int total = tile_items * kw;
    for (int base = 0; base < total; base += 8 * 256) {
        float v[8];
        #pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = base + q * 256 + tid;
            const int row = idx / kw, col = idx - row * kw;
            const long long item = item0 + row;
            v[q] = (idx < total && item < n && col < k) ? (float)B[(size_t)item * ldb + col] : 0.f;
        }
        #pragma unroll
        for (int q = 0; q < 8; q++) {
            const int idx = base + q * 256 + tid;
            const int row = idx / kw, col = idx - row * kw;
            if (idx < total) pk_tile[row * ld + col] = v[q];
        }
    }
    __syncthreads();
    const int units = NG * 2 * tile_items;
    float4 *out = Bp + (size_t)blockIdx.x * units;
    for (int o = tid; o < units; o += 256) {
        const int row = o % tile_items, h = (o / tile_items) & 1, g = o / (2 * tile_items);
        const float *src = pk_tile + row * ld + 8 * g + h;
        out[o] = make_float4(src[0], src[2], src[4], src[6]);
    }
}

template <class T>
__global__ void k_pack_users(const T *A, size_t lda, int k, int NG, const int *slot_user, int n_slots,
                             float4 *Ap, long long total_f4)
{
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total_f4) return;
    const int ul = (int)(o % GROUP_USERS);
    const int h = (int)((o / GROUP_USERS) & 1);
    const long long gg = o / (2 * GROUP_USERS);
    const int g = (int)(gg % NG);
    const long long group = gg / NG;
    const long long slot = group * GROUP_USERS + ul;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (slot < n_slots) {
        const T *src = A + (size_t)slot_user[slot] * lda;
        #pragma unroll
        for (int i = 0; i < 4; i++) { const int kk = 8 * g + 2 * i + h; if (kk < k) v[i] = (float)src[kk]; }
    }
    Ap[o] = make_float4(v[0], v[1], v[2], v[3]);
}

// fp64, v_mfma_f64_16x16x4_f64: lane (r = lane & 15, q = lane >> 4) supplies element [r][k = 4*step + q]; a b128 read
// covers 2 steps:   packed[tile][g][q][row][i] = X[tile*ROWS + row][8g + 4i + q]   (i = 0, 1)
template <class T>
__global__ void k_pack_items64(const T *B, size_t ldb, int n, int k, int NG, double2 *Bp, long long total_d2)
{
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total_d2) return;
    const int row = (int)(o % TILE_ITEMS);
    const int q = (int)((o / TILE_ITEMS) & 3);
    const long long tg = o / (4 * TILE_ITEMS);
    const int g = (int)(tg % NG);
    const long long tile = tg / NG;
    const long long item = tile * TILE_ITEMS + row;
    double v[2] = {0., 0.};
    if (item < n) {
        const T *src = B + (size_t)item * ldb;
        #pragma unroll
        for (int i = 0; i < 2; i++) { const int kk = 8 * g + 4 * i + q; if (kk < k) v[i] = (double)src[kk]; }
    }
    Bp[o] = make_double2(v[0], v[1]);
}

template <class T>
__global__ void k_pack_users64(const T *A, size_t lda, int k, int NG, const int *slot_user, int n_slots,
                               double2 *Ap, long long total_d2)
{
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= total_d2) return;
    const int ul = (int)(o % 16);
    const int q = (int)((o / 16) & 3);
    const long long gg = o / 64;
    const int g = (int)(gg % NG);
    const long long group = gg / NG;
    const long long slot = group * 16 + ul;
    double v[2] = {0., 0.};
    if (slot < n_slots) {
        const T *src = A + (size_t)slot_user[slot] * lda;
        #pragma unroll
        for (int i = 0; i < 2; i++) { const int kk = 8 * g + 4 * i + q; if (kk < k) v[i] = (double)src[kk]; }
    }
    Ap[o] = make_double2(v[0], v[1]);
}

// ---- positives: scores of the user's test items (same k-ordered fma chain as the sweep's MFMA), sorted --------------
// k-ordered fma chain; 16-byte loads when both rows allow it (the order of the fma's is what matters, not of the loads)
template <class T> __device__ __forceinline__ T chain_dot(const T *x, const T *y, int k);
template <> __device__ __forceinline__ float chain_dot<float>(const float *x, const float *y, int k)
{
    float s = 0.f;
    int t = 0;
    if ((((size_t)x | (size_t)y) & 15) == 0) {
        for (; t + 4 <= k; t += 4) {
            const float4 a = *(const float4 *)(x + t), b = *(const float4 *)(y + t);
            s = __builtin_fmaf(a.x, b.x, s); s = __builtin_fmaf(a.y, b.y, s);
            s = __builtin_fmaf(a.z, b.z, s); s = __builtin_fmaf(a.w, b.w, s);
        }
    }
    for (; t < k; t++) s = __builtin_fmaf(x[t], y[t], s);
    return s;
}
template <> __device__ __forceinline__ double chain_dot<double>(const double *x, const double *y, int k)
{
    double s = 0.;
    int t = 0;
    if ((((size_t)x | (size_t)y) & 15) == 0) {
        for (; t + 2 <= k; t += 2) {
            const double2 a = *(const double2 *)(x + t), b = *(const double2 *)(y + t);
            s = __builtin_fma(a.x, b.x, s); s = __builtin_fma(a.y, b.y, s);
        }
    }
    for (; t < k; t++) s = __builtin_fma(x[t], y[t], s);
    return s;
}

template <class T> struct PosArgs {
    int m, n, k;
    const T *A; size_t lda; const T *B; size_t ldb;
    const int *train_p, *train_i, *test_p, *test_i;
    const int *flags, *user_nslots, *uslot_base, *slot_index;
    const long long *grow;
    T *pos_tmp;          // [nnz_test] score of each test entry (+inf when the item is masked by the train row)
    int *pos_order;      // [nnz_test] ascending rank of the entry inside its row, order (score asc, item desc)
    T *pos_score;        // [(total_rows + n_groups)][32]: group g owns rows (grow[g] + g) .. + 2^j - 1, last row = +inf pad
    int *pos_item;       // same shape: item id of each sorted positive (tie resolution)
    int gu;              // users per group
    // streamed users (work list = their chunks): the sorted positives go to contiguous rows instead of group tables
    // tie noise (rm_noise.hpp): per-item noise rows added to the positives' scores; first pass of an fp32 noise call: flag
    // the users with a test item inside the zone the noise can change
    const int *noise_row; int noise_row0; const T *noise_E; long long noise_ld;
    int *noise_flag; Plan *plan;
    int stream;          // 1 = the work list is the streamed users' chunk list
    T *spos_score;       // [nnz_test] at test_p[u] + rank: scores ascending, order (score asc, item desc)
    int *spos_item;      // [nnz_test] their item ids
    unsigned long long *pos_key;   // fp32: [nnz_test + 8] the entry's packed key, (order-preserving score bits << 32) | ~item: an entry's rank in
                                   // (score asc, item desc) order is the number of smaller keys (k_pos_place reads them through the scalar cache)
};

__device__ __forceinline__ bool in_sorted_row(const int *row, int len, int item)
{
    int lo = 0, hi = len;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (row[mid] < item) lo = mid + 1; else hi = mid; }
    return lo < len && row[lo] == item;
}

// one wavefront per SLOT (= up to 63 test entries of one user, by row position; a heavy user's chunks run in parallel)
__device__ __forceinline__ float fma_chain_step(float a, float b, float s) { return __builtin_fmaf(a, b, s); }
__device__ __forceinline__ double fma_chain_step(double a, double b, double s) { return __builtin_fma(a, b, s); }

// The item-factor rows of the wave's test items are fetched 128 bytes at a time, COALESCED and WIDE: eight lanes cover that piece of
// a row with 16 bytes each, one load instruction covers it for 8 rows, eight instructions for all 63 -- a quarter of the load
// instructions of 4-byte lanes, which is what bound this kernel (an L2 gather of nnz_test rows: 1.5 GB at BASELINE C2).  The pieces
// go into a padded LDS image and each lane then walks its own row out of LDS; the user's factors ride in one register and are
// broadcast.  (A per-lane gather -- 16 bytes of 60-odd different rows per instruction -- thrashes the vector L1.)  The accumulate
// chain is chain_dot's: k-ordered fma from +0.  Rows that are not 16-byte aligned take element loads.
constexpr int POSS_WAVES = 4;                                   // wavefronts (slots) per block
template <class T>
__global__ __launch_bounds__(POSS_WAVES * WAVE) void k_pos_scores(PosArgs<T> a, const int *slot_user, const int *slot_chunk, int n_slots)
{
    constexpr int PB = 128;                                     // bytes of a row per staged piece
    constexpr int CH = PB / (int)sizeof(T);                     // factors per piece: 32 floats / 16 doubles
    constexpr int VE = 16 / (int)sizeof(T);                     // factors per 16-byte vector
    constexpr int LPR = PB / 16;                                // lanes per row piece
    constexpr int RPI = WAVE / LPR;                             // rows covered by one load instruction: 8
    constexpr int LD = CH + VE;                                 // padded row stride in LDS (144 B: conflict-free vector writes and row walks)
    typedef T VT __attribute__((ext_vector_type(16 / sizeof(T))));
    __shared__ __attribute__((aligned(16))) T rows[POSS_WAVES][WAVE][LD];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int w = blockIdx.x * POSS_WAVES + wv;
    if (w >= n_slots) return;
    const int u = slot_user[w], c0 = slot_chunk[w];
    if (a.flags[u] & UF_ONLY_NDCG) return;
    const int te0 = a.test_p[u], te1 = a.test_p[u + 1];
    const int tr0 = a.train_p[u], ntr = a.train_p[u + 1] - tr0;
    const int e = te0 + c0 * POS_CHUNK + lane;
    const int np = min(POS_CHUNK, te1 - (te0 + c0 * POS_CHUNK));         // entries of this slot: 1 .. 63
    const bool mine = lane < np;
    const int item = mine ? a.test_i[e] : 0;
    const bool masked = mine && ntr && in_sorted_row(a.train_i + tr0, ntr, item);
    const T *Au = a.A + (size_t)u * a.lda;
    const int f = lane % CH, rsub = lane / LPR, piece = lane % LPR;
    const bool vec_ok = ((((size_t)a.B) | (a.ldb * sizeof(T))) & 15) == 0;
    // the row this lane helps to fetch in each of the eight load instructions of a piece (same for every piece)
    const T *brow[WAVE / RPI];
    #pragma unroll
    for (int q = 0; q < WAVE / RPI; q++) {
        const int p = q * RPI + rsub;
        brow[q] = a.B + (size_t)__shfl(item, p < np ? p : 0) * a.ldb;
    }
    T s = 0;
    for (int k0 = 0; k0 < a.k; k0 += CH) {
        const T av = k0 + f < a.k ? Au[k0 + f] : (T)0;
        const int e0 = k0 + piece * VE;                             // first factor of this lane's 16 bytes
        VT bv[WAVE / RPI];
        #pragma unroll
        for (int q = 0; q < WAVE / RPI; q++) {
            const int p = q * RPI + rsub;
            VT x;
            #pragma unroll
            for (int i = 0; i < VE; i++) x[i] = (T)0;
            if (p < np && e0 < a.k) {
                if (vec_ok && e0 + VE <= a.k) x = *(const VT *)(brow[q] + e0);
                else {
                    #pragma unroll
                    for (int i = 0; i < VE; i++) if (e0 + i < a.k) x[i] = brow[q][e0 + i];
                }
            }
            bv[q] = x;
        }
        #pragma unroll
        for (int q = 0; q < WAVE / RPI; q++) *(VT *)&rows[wv][q * RPI + rsub][piece * VE] = bv[q];
        const int kc = min(CH, a.k - k0);
        for (int t = 0; t < kc; t++) s = fma_chain_step(lane_bcast<T>(av, t), rows[wv][mine ? lane : 0][t], s);
    }
    if (mine && !masked && a.noise_E) {
        const long long nr = a.noise_row ? a.noise_row[u] - a.noise_row0 : u;
        s += a.noise_E[(size_t)nr * (size_t)a.noise_ld + item];
    }
    if (mine && !masked && a.noise_flag && (s < (T)0 ? -s : s) < (T)6.103515625e-05f) {
        if (atomicExch(&a.noise_flag[u], 1) == 0) atomicAdd(&a.plan->n_noise_flagged, 1);
    }
    if (mine) {
        const T sv = masked ? (T)__int_as_float(0x7f800000) : s;
        a.pos_tmp[e] = sv;
        if (sizeof(T) == 4 && a.pos_key) a.pos_key[e] = ((unsigned long long)ord_key((float)sv) << 32) | (unsigned)~item;
    }
}

// ---- the same scores, by ENTRY instead of by slot ------------------------------------------------------------------------------
// k_pos_scores gives every slot a wavefront of its own: a user with a dozen test items keeps a dozen lanes busy, and every wavefront
// pays the whole chain of dependent loads (slot -> row bounds -> items -> train row search -> item factors) for at most 63 entries --
// at BASELINE C2 170,000 wavefronts of ~10 us each, 0.5 ms in front of the sweep, and NOT bound by the gather itself (test items
// squeezed into an eighth of the item factors: the same time; r5_ab_c2.txt).  By entry: a wavefront takes 64 CONSECUTIVE test entries
// whatever rows they belong to, every lane busy.  What an entry needs to know about its row depends on the CSR inputs alone: its
// user (k_entry_users) and whether the train row holds the item (k_test_masked).  All three kernels run BESIDE the plan chain, the
// plan read-back and the host's work behind it, on a stream of their own: nothing in them depends on the plan.  The user's factors are read per lane, 16 bytes at a time
// (the lanes of one user ask for the same addresses: one request per distinct user of the 64 entries).
// Same arithmetic: k-ordered fma chain from +0.  Users that are not evaluated (or need no ranks) are skipped per lane.
__global__ __launch_bounds__(256) void k_entry_users(int m, const int *test_p, int *ent_user, const Plan *plan)
{
    if (plan->csr_bad & CSR_BAD_INDPTR) return;
    const int lane = threadIdx.x & 63;
    const int ub = (int)(((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * WAVE;       // a wavefront per 64 users, one row at a time: coalesced writes
    if (ub >= m) return;
    const int u = ub + lane;
    const int s = u < m ? test_p[u] : 0, e = u < m ? test_p[u + 1] : 0;
    const int nu = min(WAVE, m - ub);
    for (int j = 0; j < nu; j++) {
        const int sj = __builtin_amdgcn_readlane(s, j), ej = __builtin_amdgcn_readlane(e, j);
        for (int x = sj + lane; x < ej; x += WAVE) ent_user[x] = ub + j;
    }
}
// k_test_masked: a block owns TM_USERS consecutive users.  Their train rows are CONSECUTIVE rows of the CSR, i.e. one contiguous
// piece of the index array: it is staged in LDS with coalesced loads and the binary searches run there (a search in memory is seven
// or eight DEPENDENT loads per entry at BASELINE C2: the kernel was 0.23 ms of latency).  The users are taken in runs whose rows fit
// TM_CAP entries together; a single row longer than that is searched in memory.  39.5 KB of LDS per block also means four blocks per
// CU: half of the wave slots stay free while this kernel runs beside the plan chain, whose blocks of 1,024 threads otherwise wait
// for sixteen slots of one CU to fall free at once (measured: the plan read-back 60 us later).
constexpr int TM_CAP = 9984, TM_USERS = 64;
__global__ __launch_bounds__(256) void k_test_masked(int m, const int *test_p, const int *test_i, const int *train_p, const int *train_i,
                                                     const int *ent_user, unsigned char *ent_masked, const Plan *plan)
{
    __shared__ int seg[TM_CAP];
    __shared__ int tp[TM_USERS + 1], trp[TM_USERS + 1];
    if (plan->csr_bad & CSR_BAD_INDPTR) return;
    const int u0 = blockIdx.x * TM_USERS, nu = min(TM_USERS, m - u0), tid = threadIdx.x;
    if (tid <= nu) { tp[tid] = test_p[u0 + tid]; trp[tid] = train_p[u0 + tid]; }
    __syncthreads();
    constexpr int SB = 4, EB = 4;                                 // loads in flight per thread: staging / entries
    for (int a = 0; a < nu; ) {
        int b = a + 1, longest = trp[a + 1] - trp[a];             // users [a, b): as many as fit, at least one
        while (b < nu && trp[b + 1] - trp[a] <= TM_CAP) { longest = max(longest, trp[b + 1] - trp[b]); b++; }
        const int s0 = trp[a], len = trp[b] - s0;
        const bool fits = len <= TM_CAP;
        if (fits) {
            for (int base = 0; base < len; base += SB * 256) {
                int v[SB];
                #pragma unroll
                for (int q = 0; q < SB; q++) { const int i = base + q * 256 + tid; v[q] = i < len ? train_i[s0 + i] : 0; }
                #pragma unroll
                for (int q = 0; q < SB; q++) { const int i = base + q * 256 + tid; if (i < len) seg[i] = v[q]; }
            }
            __syncthreads();
        }
        const int steps = 32 - __clz(longest);                    // binary search steps that settle the longest row of the run
        const int e_end = tp[b];
        for (int e0 = tp[a] + tid; e0 < e_end; e0 += EB * 256) {
            // EB entries per thread at a time: their loads, then their searches step by step (independent chains of LDS reads)
            int ul[EB], item[EB];
            #pragma unroll
            for (int q = 0; q < EB; q++) { const int e = e0 + q * 256; const bool in = e < e_end; ul[q] = in ? ent_user[e] - u0 : a; item[q] = in ? test_i[e] : 0; }
            if (fits) {
                int lo[EB], n[EB], end[EB];
                #pragma unroll
                for (int q = 0; q < EB; q++) { lo[q] = trp[ul[q]] - s0; end[q] = trp[ul[q] + 1] - s0; n[q] = end[q] - lo[q]; }
                for (int st = 0; st < steps; st++) {
                    #pragma unroll
                    for (int q = 0; q < EB; q++) {
                        const int half = n[q] >> 1, mid = lo[q] + half;
                        const bool lt = n[q] > 0 && seg[mid] < item[q];
                        lo[q] = lt ? mid + 1 : lo[q];
                        n[q] = lt ? n[q] - half - 1 : half;
                    }
                }
                #pragma unroll
                for (int q = 0; q < EB; q++) { const int e = e0 + q * 256; if (e < e_end) ent_masked[e] = lo[q] < end[q] && seg[lo[q]] == item[q]; }
            } else {
                #pragma unroll
                for (int q = 0; q < EB; q++) {
                    const int e = e0 + q * 256;
                    if (e >= e_end) continue;
                    const int tr0 = trp[ul[q]], ntr = trp[ul[q] + 1] - tr0;
                    ent_masked[e] = ntr && in_sorted_row(train_i + tr0, ntr, item[q]);
                }
            }
        }
        __syncthreads();
        a = b;
    }
}

constexpr int POSF_WAVES = 4;
template <class T>
__global__ __launch_bounds__(POSF_WAVES * WAVE) void k_pos_scores_flat(PosArgs<T> a, const int *ent_user)
{
    constexpr int PB = 64;                                      // bytes of a row per staged piece
    constexpr int CH = PB / (int)sizeof(T);                     // factors per piece: 32 floats / 16 doubles
    constexpr int VE = 16 / (int)sizeof(T);                     // factors per 16-byte vector
    constexpr int LPR = PB / 16;                                // lanes per row piece
    constexpr int RPI = WAVE / LPR;                             // rows covered by one load instruction: 8
    constexpr int LD = CH + VE;                                 // padded row stride in LDS (144 B)
    typedef T VT __attribute__((ext_vector_type(16 / sizeof(T))));
    __shared__ __attribute__((aligned(16))) T rows[POSF_WAVES][WAVE][LD];
    if (a.plan->csr_bad & (CSR_BAD_INDPTR | CSR_BAD_INDEX)) return;       // (the test items index the item factors)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long long e = (long long)a.test_p[0] + ((long long)blockIdx.x * POSF_WAVES + wv) * WAVE + lane;
    const bool in = e < (long long)a.test_p[a.m];
    const int u = in ? ent_user[e] : 0;
    const int item = in ? a.test_i[e] : 0;
    const int f = in ? a.flags[u] : 0;
    const bool mine = in && (f & UF_ACTIVE) && !(f & UF_ONLY_NDCG);
    const unsigned long long live = __ballot(mine);
    if (!live) return;
    const int rsub = lane / LPR, piece = lane % LPR;
    const bool vec_ok = ((((size_t)a.B) | (a.ldb * sizeof(T))) & 15) == 0, avec_ok = ((((size_t)a.A) | (a.lda * sizeof(T))) & 15) == 0;
    const T *brow[WAVE / RPI];
    #pragma unroll
    for (int q = 0; q < WAVE / RPI; q++) brow[q] = a.B + (size_t)__shfl(item, q * RPI + rsub) * a.ldb;
    const T *Au = a.A + (size_t)u * a.lda;
    T s = 0;
    for (int k0 = 0; k0 < a.k; k0 += CH) {
        const int e0 = k0 + piece * VE;                         // first factor of this lane's 16 bytes
        VT bv[WAVE / RPI];
        #pragma unroll
        for (int q = 0; q < WAVE / RPI; q++) {
            const int p = q * RPI + rsub;
            VT x;
            #pragma unroll
            for (int i = 0; i < VE; i++) x[i] = (T)0;
            if (((live >> p) & 1) && e0 < a.k) {
                if (vec_ok && e0 + VE <= a.k) x = *(const VT *)(brow[q] + e0);
                else {
                    #pragma unroll
                    for (int i = 0; i < VE; i++) if (e0 + i < a.k) x[i] = brow[q][e0 + i];
                }
            }
            bv[q] = x;
        }
        VT av[CH / VE];
        #pragma unroll
        for (int j = 0; j < CH / VE; j++) {
            VT x;
            #pragma unroll
            for (int i = 0; i < VE; i++) x[i] = (T)0;
            const int f0 = k0 + j * VE;
            if (mine && f0 < a.k) {
                if (avec_ok && f0 + VE <= a.k) x = *(const VT *)(Au + f0);
                else {
                    #pragma unroll
                    for (int i = 0; i < VE; i++) if (f0 + i < a.k) x[i] = Au[f0 + i];
                }
            }
            av[j] = x;
        }
        #pragma unroll
        for (int q = 0; q < WAVE / RPI; q++) *(VT *)&rows[wv][q * RPI + rsub][piece * VE] = bv[q];
        // (the pieces change lanes through LDS: fenced like k_train_bits / k_pos_place, not left to the compiler's alias analysis)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        const T *r = rows[wv][lane];
        if (k0 + CH <= a.k) {
            #pragma unroll
            for (int j = 0; j < CH / VE; j++) {
                const VT b = *(const VT *)(r + j * VE);
                #pragma unroll
                for (int i = 0; i < VE; i++) s = fma_chain_step(av[j][i], b[i], s);
            }
        } else {
            #pragma unroll
            for (int j = 0; j < CH / VE; j++) {
                const VT b = *(const VT *)(r + j * VE);
                #pragma unroll
                for (int i = 0; i < VE; i++) if (k0 + j * VE + i < a.k) s = fma_chain_step(av[j][i], b[i], s);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();      // the rows are rewritten by the next chunk
    }
    // (whether the train row holds the item is not known here: k_pos_apply_masked overwrites those entries afterwards.  A user
    // flagged for one of them goes through an exact pass it did not need -- same results.)
    if (mine && a.noise_E) {
        const long long nr = a.noise_row ? a.noise_row[u] - a.noise_row0 : u;
        s += a.noise_E[(size_t)nr * (size_t)a.noise_ld + item];
    }
    if (mine && a.noise_flag && (s < (T)0 ? -s : s) < (T)6.103515625e-05f) {
        if (atomicExch(&a.noise_flag[u], 1) == 0) atomicAdd(&a.plan->n_noise_flagged, 1);
    }
    if (mine) {
        a.pos_tmp[e] = s;
        if (sizeof(T) == 4 && a.pos_key) a.pos_key[e] = ((unsigned long long)ord_key((float)s) << 32) | (unsigned)~item;
    }
}
// test entries whose item is a train item of the user: score +inf (the reference never scores a train item, recometrics.hpp:491-497)
template <class T>
__global__ void k_pos_apply_masked(PosArgs<T> a, const unsigned char *ent_masked)
{
    if (a.plan->csr_bad & (CSR_BAD_INDPTR | CSR_BAD_INDEX)) return;
    const long long e = (long long)a.test_p[0] + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)a.test_p[a.m] || !ent_masked[e]) return;
    const T inf = (T)__int_as_float(0x7f800000);
    a.pos_tmp[e] = inf;
    if (sizeof(T) == 4 && a.pos_key) a.pos_key[e] = ((unsigned long long)ord_key((float)inf) << 32) | (unsigned)~a.test_i[e];
}

// One wavefront per SLOT (= up to 63 test entries of one user, by row position): rank of each of them among ALL the
// user's entries in (score asc, item desc) order, by all-pairs counting.  The row is walked 64 entries at a time out
// of registers (lane broadcasts), so the inner loop touches no memory, and a user with thousands of positives is
// spread over as many waves as it has slots instead of serialising the kernel behind one wave.
constexpr int PLACE_THREADS = 256;
template <class T>
__global__ __launch_bounds__(PLACE_THREADS) void k_pos_place(PosArgs<T> a, const int *slot_user, const int *slot_chunk, int n_slots)
{
    const int w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;     // (a wave's index: uniform, and known to be)
    if (w >= n_slots) return;
    const int u = slot_user[w], c0 = slot_chunk[w];
    if (a.flags[u] & UF_ONLY_NDCG) return;
    const int te0 = a.test_p[u], te1 = a.test_p[u + 1];
    const int e = te0 + c0 * POS_CHUNK + lane;
    const bool mine = lane < POS_CHUNK && e < te1;
    const T s = mine ? a.pos_tmp[e] : (T)0;
    const int item = mine ? a.test_i[e] : 0;
    int rank = 0;
    if (sizeof(T) == 4 && a.pos_key) {
        // fp32: (score asc, item desc) is the order of the packed key (ordered score bits << 32) | ~item, so an entry's rank is the
        // number of smaller keys (k_pos_scores wrote them).  The row's keys pass through a small LDS buffer of the wavefront, 64 at a
        // time: one coalesced load per piece (the next piece in flight while this one is compared), then every key is read back with
        // a wave-uniform address -- an LDS broadcast, two keys per instruction -- and costs one 64-bit compare and one add-with-carry.
        // (Round 4 read them through the scalar cache, eight keys per load: the loads come back one latency after the other, and the
        // all-pairs rank of the streamed users' long rows, 32 us of vector work at BASELINE C2, took 140 us.)
        // (Scores are never -0: the chain starts at +0.)
        __shared__ __attribute__((aligned(16))) unsigned long long kbuf[PLACE_THREADS / WAVE][2][WAVE];
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const unsigned long long key = ((unsigned long long)ord_key((float)s) << 32) | (unsigned)~item;
        const unsigned long long *keys = a.pos_key + te0;
        const int P = te1 - te0;
        // (beyond the row the buffer holds the largest key, which is smaller than nothing: whole groups of 16 are compared)
        unsigned long long nxt = lane < P ? keys[lane] : ~0ull;
        for (int j0 = 0; j0 < P; j0 += WAVE) {
            const unsigned long long cur = nxt;
            if (j0 + WAVE < P) nxt = j0 + WAVE + lane < P ? keys[j0 + WAVE + lane] : ~0ull;
            unsigned long long *kb = kbuf[wv][(j0 >> 6) & 1];
            kb[lane] = cur;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int cnt = min(WAVE, P - j0);
            typedef unsigned long long K2 __attribute__((ext_vector_type(2)));
            for (int j = 0; j < cnt; j += 16) {
                #pragma unroll
                for (int q = 0; q < 16; q += 2) { const K2 k2 = *(const K2 *)&kb[j + q]; rank += k2.x < key; rank += k2.y < key; }
            }
        }
    } else if (sizeof(T) == 4) {
        const unsigned long long key = ((unsigned long long)ord_key((float)s) << 32) | (unsigned)~item;
        for (int f0 = te0; f0 < te1; f0 += WAVE) {
            const int f = f0 + lane;
            const unsigned kh = f < te1 ? ord_key((float)a.pos_tmp[f]) : 0u;
            const unsigned kl = f < te1 ? (unsigned)~a.test_i[f] : 0u;
            const int cnt = min(WAVE, te1 - f0);
            for (int j = 0; j < cnt; j++) {
                const unsigned long long kj = ((unsigned long long)(unsigned)lane_bcast<int>((int)kh, j) << 32) | (unsigned)lane_bcast<int>((int)kl, j);
                rank += kj < key;
            }
        }
    } else {
        for (int f0 = te0; f0 < te1; f0 += WAVE) {
            const int f = f0 + lane;
            const T sf = f < te1 ? a.pos_tmp[f] : (T)0;
            const int itf = f < te1 ? a.test_i[f] : 0;
            const int cnt = min(WAVE, te1 - f0);
            for (int j = 0; j < cnt; j++) {
                const T sj = lane_bcast<T>(sf, j);
                const int ij = lane_bcast<int>(itf, j);
                rank += (sj < s) || (sj == s && ij > item);
            }
        }
    }
    if (mine && a.stream) {
        a.pos_order[e] = rank;
        a.spos_score[te0 + rank] = s;
        a.spos_item[te0 + rank] = item;
    } else if (mine) {
        a.pos_order[e] = rank;
        const int c = rank / POS_CHUNK, r = rank % POS_CHUNK;
        const int slot = a.slot_index[a.uslot_base[u] + c];
        const long long at = (a.grow[slot / a.gu] + slot / a.gu + r) * a.gu + (slot % a.gu);
        a.pos_score[at] = s;
        a.pos_item[at] = item;
    }
}

} // namespace rm
