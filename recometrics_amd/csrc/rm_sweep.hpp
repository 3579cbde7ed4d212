// rm_sweep.hpp -- THE hot kernel: score tile contraction (fp32 MFMA) fused with train masking, validity scan,
// streaming top-K select and AUC rank counting.
//
// Replaces reference src/recometrics.hpp:491-573 (candidate list, dot1 loop, validity scan, partial_sort / sort,
// test-item mask) for a block of 128 users at a time, without ever materialising scores:
//
//   D[item][user] = sum_t Bitem[t] * Auser[t]       v_mfma_f32_32x32x2_f32, accumulators start at +0 and k advances in
//                                                   index order => bit-identical to the reference's canonical fmaf chain
//   orientation    users on LANES (D column = lane & 31), items in the 16 accumulator registers.  Every per-user
//                  quantity of the epilogue (K-th best threshold, min, max, train cursor, AUC partial sum) is then ONE
//                  VGPR, and the per-user LDS tables (sorted positives, rank histogram, top-K list) are laid out
//                  [row][32 users] so that lane u always hits bank u: conflict-free whatever the data.
//   block          512 threads = 8 wavefronts = 4 user groups x 2 item sub-tiles.  Wavefronts w and w+4 share a SIMD
//                  and a user group; w runs {MFMA(tile t), epilogue(tile t-1)}, w+4 runs {epilogue(t-1), MFMA(t)}, so
//                  the matrix pipe and the VALU/LDS epilogue of the two overlap inside one barrier interval.
//   operands       user factors live in registers for the whole sweep (NG float4 per lane); packed item tiles
//                  (rm_prep.hpp k_pack_items) stream HBM -> registers -> LDS, double buffered, one barrier per tile.
#pragma once
#include "rm_device.hpp"

namespace rm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct SweepArgs {
    int n, K;
    int n_slots, n_groups, n_ublocks;     // n_ublocks = ceil(n_groups / 4)
    int n_splits, tiles_total;            // item splits (grid = n_ublocks * n_splits)
    int jmax;                             // LDS sizing (all blocks)
    int list_in_lds;
    const float4 *Ap, *Bp;
    const int *slot_user, *slot_chunk;
    const int *train_p, *train_i;
    const int *gj; const long long *grow;
    const float *pos_score;               // [(total_rows + n_groups)][32]  sorted positives, +inf padded (2^j rows per group)
    const int *pos_item;                  // same shape, item ids (read only when a candidate ties a positive's score)
    unsigned *hist;                       // [(total_rows + n_groups)][32]
    ListEntry *glists;                    // global list scratch when !list_in_lds: [block][wave][K][32]
    ListEntry *pl;                        // partial lists [slot][n_part][K]
    PartialStat<float> *pst;              // [slot][n_part]
    float *dump;                          // DUMP mode: dense [n_slots][n] scores
};

__device__ __forceinline__ bool entry_before(float s, int idx, float s2, int idx2)
{
    return s > s2 || (s == s2 && idx < idx2);
}

// per-lane insertion into the lane's user's descending list L[i*32] (i = 0..K-1); generic pointer (LDS or global)
__device__ __forceinline__ void list_insert(ListEntry *L, int K, float s, int item)
{
    const ListEntry last = L[(K - 1) * GROUP_USERS];
    if (!entry_before(s, item, last.s, last.idx)) return;
    int i = K - 1;
    while (i > 0) {
        const ListEntry e = L[(i - 1) * GROUP_USERS];
        if (entry_before(e.s, e.idx, s, item)) break;
        L[i * GROUP_USERS] = e;
        i--;
    }
    ListEntry ne; ne.s = s; ne.idx = item;
    L[i * GROUP_USERS] = ne;
}

// AUC rank counting for one tile: branchless lower_bound of every score in the lane's user's sorted positives
// (complete tree of 2^J - 1 rows, +inf padded), then one LDS atomic into the rank histogram.
template <int J>
__device__ __forceinline__ void auc_pass(const float (&v)[16], const char *posb, char *histb, unsigned &rocacc,
                                         const int *pos_item_g, int sb, int h)
{
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        const float s = v[r];
        unsigned base = 0;
        #pragma unroll
        for (int st = (J > 0 ? (1 << (J - 1)) : 0); st >= 1; st >>= 1) {
            const float pv = *(const float *)(posb + base + (st - 1) * 128);
            base = (pv < s) ? base + st * 128 : base;
        }
        // exact score tie with a positive (row `base` is the first positive not below s; the table has one +inf
        // pad row, so the read is always in range): the total order is (score desc, item asc), i.e. the candidate
        // also outranks the equal-scored positives with a LARGER item id.  Rare; positives' item ids stay in HBM.
        const float nx = *(const float *)(posb + base);
        if (__any(nx == s)) {
            if (nx == s) {
                const int item = sb + mfma32_row(r, h);
                unsigned t = base;
                while (t < (unsigned)(((1 << J) - 1) * 128) && *(const float *)(posb + t) == s && pos_item_g[t >> 2] > item) t += 128;
                base = t;
            }
        }
        rocacc += base;
        __hip_atomic_fetch_add((unsigned *)(histb + base), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

template <int NG, bool AUC, bool DUMP>
__global__ __launch_bounds__(SWEEP_THREADS, 2)
void k_sweep(SweepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF_F4 = NG * 2 * TILE_ITEMS;                 // float4 per packed tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gi = wave & 3, sub = wave >> 2;                   // group in block / 32-item sub-tile
    const int ul = lane & 31, h = lane >> 5;
    const int blk_u = blockIdx.x % a.n_ublocks, split = blockIdx.x / a.n_ublocks;
    const int group = blk_u * GROUPS_PER_BLOCK + gi;
    const bool group_ok = group < a.n_groups;
    const int slot = group * GROUP_USERS + ul;
    const bool slot_ok = group_ok && slot < a.n_slots;
    const int K = a.K, n = a.n;

    // ---- LDS carve: [B buf0 | B buf1 | lists (8 waves) | positives (4 groups) | histogram (4 groups)] ----
    float4 *ldsB = (float4 *)smem;
    char *p = smem + 2 * BUF_F4 * 16;
    ListEntry *lists_lds = (ListEntry *)p;
    if (a.list_in_lds) p += 8 * K * GROUP_USERS * (int)sizeof(ListEntry);
    const int PLmax = (1 << a.jmax) - 1;
    float *posL = (float *)p;  p += GROUPS_PER_BLOCK * (PLmax + 1) * GROUP_USERS * 4;
    unsigned *histL = (unsigned *)p;

    const int glast = min(a.n_groups, (blk_u + 1) * GROUPS_PER_BLOCK) - 1;
    const int jb = AUC ? a.gj[glast] : 0;                        // block-uniform tree depth
    const int PLb = (1 << jb) - 1;

    // ---- per-lane user state ----
    const int user = slot_ok ? a.slot_user[slot] : -1;
    const bool primary = slot_ok && a.slot_chunk[slot] == 0;
    float thr = primary ? neg_inf_f() : nan_sentinel_f();        // NaN threshold: "v >= thr" never true
    float vmax = neg_inf_f(), vmin = pos_inf_f();
    unsigned long long nanmask = 0, roc64 = 0;
    int ntc = 0, nte = 0, nt = IDX_EMPTY;

    // item range of this split
    const int tiles_per = (a.tiles_total + a.n_splits - 1) / a.n_splits;
    const int t0 = split * tiles_per, t1 = min(a.tiles_total, t0 + tiles_per);
    const int ntiles = max(0, t1 - t0);

    if (user >= 0) {
        ntc = a.train_p[user]; nte = a.train_p[user + 1];
        // first train item at or after this wave's first item (lower_bound)
        const int first_item = t0 * TILE_ITEMS;
        int lo = ntc, hi = nte;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.train_i[mid] < first_item) lo = mid + 1; else hi = mid; }
        ntc = lo;
        nt = ntc < nte ? a.train_i[ntc] : IDX_EMPTY;
    }

    // user factors -> registers (packed: [group][g][h][32][4 floats])
    float4 af[NG];
    #pragma unroll
    for (int g = 0; g < NG; g++)
        af[g] = group_ok ? a.Ap[((size_t)(group * NG + g) * 2 + h) * GROUP_USERS + ul] : make_float4(0.f, 0.f, 0.f, 0.f);

    // top-K list of this wave: [K][32 users]
    ListEntry *L = a.list_in_lds ? (lists_lds + wave * K * GROUP_USERS + ul)
                                 : (a.glists + ((size_t)blockIdx.x * 8 + wave) * K * GROUP_USERS + ul);
    if (h == 0) for (int i = 0; i < K; i++) { ListEntry e; e.s = neg_inf_f(); e.idx = IDX_EMPTY; L[i * GROUP_USERS] = e; }

    // positives -> LDS, histogram zeroed
    if (AUC) {
        for (int i = tid; i < GROUPS_PER_BLOCK * (PLb + 1) * GROUP_USERS; i += SWEEP_THREADS) {
            const int g4 = i / ((PLb + 1) * GROUP_USERS), rem = i % ((PLb + 1) * GROUP_USERS);
            const int gg = blk_u * GROUPS_PER_BLOCK + g4;
            posL[g4 * (PLmax + 1) * GROUP_USERS + rem] = gg < a.n_groups ? a.pos_score[(a.grow[gg] + gg) * GROUP_USERS + rem] : pos_inf_f();
        }
        for (int i = tid; i < GROUPS_PER_BLOCK * (PLmax + 1) * GROUP_USERS; i += SWEEP_THREADS) histL[i] = 0;
    }
    const char *posb = (const char *)(posL + gi * (PLmax + 1) * GROUP_USERS + ul);
    const int *pos_item_g = (AUC && group_ok) ? a.pos_item + (a.grow[group] + group) * GROUP_USERS + ul : nullptr;
    char *histb = (char *)(histL + gi * (PLmax + 1) * GROUP_USERS + ul);

    // ---- staging: packed tile (BUF_F4 float4, contiguous) HBM -> registers -> LDS ----
    constexpr int NST = (BUF_F4 + SWEEP_THREADS - 1) / SWEEP_THREADS;
    float4 st[NST];
    auto stage_load = [&](int tile) {
        const float4 *src = a.Bp + (size_t)tile * BUF_F4;
        #pragma unroll
        for (int i = 0; i < NST; i++) { const int idx = tid + i * SWEEP_THREADS; if (idx < BUF_F4) st[i] = src[idx]; }
    };
    auto stage_store = [&](int buf) {
        float4 *dst = ldsB + buf * BUF_F4;
        #pragma unroll
        for (int i = 0; i < NST; i++) { const int idx = tid + i * SWEEP_THREADS; if (idx < BUF_F4) dst[idx] = st[i]; }
    };

    // ---- MFMA: 32 items (registers) x 32 users (lanes), k in index order ----
    auto do_mfma = [&](f32x16 &acc, int buf) {
        const float4 *bb = ldsB + buf * BUF_F4 + h * TILE_ITEMS + sub * 32 + ul;
        #pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        #pragma unroll
        for (int g = 0; g < NG; g++) {
            const float4 b = bb[g * 2 * TILE_ITEMS];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, af[g].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, af[g].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, af[g].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, af[g].w, acc, 0, 0, 0);
        }
    };

    // ---- epilogue of one 32-item x 32-user tile ----
    auto do_epi = [&](const f32x16 &acc, int tile) {
        const int sb = tile * TILE_ITEMS + sub * 32;            // first item of this wave's sub-tile
        float v[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) v[r] = acc[r];
        if (DUMP) {
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int item = sb + mfma32_row(r, h);
                if (slot_ok && item < n) a.dump[(size_t)slot * n + item] = v[r];
            }
            return;
        }
        // (1) train-item / out-of-range masking (reference :491-497) + NaN detection (:517-518)
        const bool slow = __any(nt < sb + 32) || (sb + 32 > n);
        if (slow) {
            unsigned mbits = 0;
            while (nt < sb + 32) {
                if (nt >= sb) mbits |= 1u << (nt - sb);
                ntc++;
                nt = ntc < nte ? a.train_i[ntc] : IDX_EMPTY;
            }
            if (sb + 32 > n) mbits |= (n > sb) ? (0xffffffffu << (n - sb)) : 0xffffffffu;
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const bool mk = (mbits >> mfma32_row(r, h)) & 1u;
                nanmask |= __ballot(!mk && (v[r] != v[r]));
                v[r] = mk ? nan_sentinel_f() : v[r];
            }
        } else {
            #pragma unroll
            for (int r = 0; r < 16; r++) nanmask |= __ballot(v[r] != v[r]);
        }
        // (2) min / max over candidates (NaN-ignoring, so the sentinel is invisible) (:519-524)
        #pragma unroll
        for (int r = 0; r < 16; r++) { vmax = __builtin_fmaxf(vmax, v[r]); vmin = __builtin_fminf(vmin, v[r]); }
        // (3) streaming top-K: anything at or above the lane's K-th best goes through the insert path (:537-540)
        unsigned long long cm = 0;
        #pragma unroll
        for (int r = 0; r < 16; r++) cm |= __ballot(v[r] >= thr);
        if (cm) {
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const bool c = v[r] >= thr;
                if (__any(c)) {
                    const int item = sb + mfma32_row(r, h);
                    // lanes u and u+32 carry the same user (two item rows): one half of the wave at a time.  The wave
                    // barriers keep the compiler from fusing the two predicated inserts into one (legal per thread,
                    // wrong across lanes).
                    if (c && h == 0) list_insert(L, K, v[r], item);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (c && h == 1) list_insert(L, K, v[r], item);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (primary) thr = L[(K - 1) * GROUP_USERS].s;
                }
            }
        }
        // (4) AUC rank counting (replaces the full sort of :552 + the walk of :795-865)
        if (AUC) {
            unsigned rocacc = 0;
            switch (jb) {
                case 1: auc_pass<1>(v, posb, histb, rocacc, pos_item_g, sb, h); break;
                case 2: auc_pass<2>(v, posb, histb, rocacc, pos_item_g, sb, h); break;
                case 3: auc_pass<3>(v, posb, histb, rocacc, pos_item_g, sb, h); break;
                case 4: auc_pass<4>(v, posb, histb, rocacc, pos_item_g, sb, h); break;
                case 5: auc_pass<5>(v, posb, histb, rocacc, pos_item_g, sb, h); break;
                case 6: auc_pass<6>(v, posb, histb, rocacc, pos_item_g, sb, h); break;
                default: break;
            }
            roc64 += rocacc >> 7;
        }
    };

    // ---- main loop: one barrier per tile; acc ping-pong; role X = waves 0-3 (MFMA first), role Y = waves 4-7 ----
    f32x16 acc0, acc1;
    const bool roleX = sub == 0;
    if (ntiles > 0) { stage_load(t0); stage_store(0); }
    __syncthreads();
    auto step = [&](int i, f32x16 &cur, f32x16 &prev) {
        const bool has_cur = i < ntiles, has_next = i + 1 < ntiles;
        if (has_next) stage_load(t0 + i + 1);
        if (roleX && has_cur) do_mfma(cur, i & 1);
        if (i > 0) do_epi(prev, t0 + i - 1);
        if (!roleX && has_cur) do_mfma(cur, i & 1);
        if (has_next) stage_store((i + 1) & 1);
        __syncthreads();
    };
    for (int i = 0; i <= ntiles; i += 2) {
        step(i, acc0, acc1);
        if (i + 1 <= ntiles) step(i + 1, acc1, acc0);
    }
    if (DUMP) return;

    // ---- write this wave's partial: top-K list, validity stats, AUC sum; flush the LDS histogram ----
    const int n_part = a.n_splits * 2;
    const int part = split * 2 + sub;
    {   // lanes u and u+32 hold two halves of the same user's stats
        const float omax = __shfl_xor(vmax, 32), omin = __shfl_xor(vmin, 32);
        vmax = __builtin_fmaxf(vmax, omax); vmin = __builtin_fminf(vmin, omin);
        const unsigned long long oroc = __shfl_xor(roc64, 32);
        roc64 += oroc;
        const bool hn = ((nanmask >> ul) & 1ull) | ((nanmask >> (ul + 32)) & 1ull);
        if (slot_ok && h == 0) {
            PartialStat<float> ps;
            ps.vmax = vmax; ps.vmin = vmin; ps.rocsum = roc64; ps.has_nan = hn ? 1 : 0; ps.pad = 0;
            a.pst[(size_t)slot * n_part + part] = ps;
            ListEntry *dst = a.pl + ((size_t)slot * n_part + part) * K;
            for (int i = 0; i < K; i++) dst[i] = L[i * GROUP_USERS];
        }
    }
    if (AUC) {
        __syncthreads();
        for (int i = tid; i < GROUPS_PER_BLOCK * (PLb + 1) * GROUP_USERS; i += SWEEP_THREADS) {
            const int g4 = i / ((PLb + 1) * GROUP_USERS), rem = i % ((PLb + 1) * GROUP_USERS);
            const int gg = blk_u * GROUPS_PER_BLOCK + g4;
            const unsigned c = histL[g4 * (PLmax + 1) * GROUP_USERS + rem];
            if (gg < a.n_groups && c) atomicAdd(&a.hist[(a.grow[gg] + gg) * GROUP_USERS + rem], c);
        }
    }
}

} // namespace rm
