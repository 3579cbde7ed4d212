// rm_sweep64.hpp -- the fp64 sweep: same fused pipeline as rm_sweep.hpp (contraction + train masking + validity scan +
// streaming top-K + AUC rank counting) on v_mfma_f64_16x16x4_f64.
//
// Measured on gfx950 (scratch/probe_f64.hip): the f64 MFMA is bit-for-bit the k-ordered fma chain (accumulator in,
// k = 4*step + (lane >> 4) within an instruction), i.e. exactly reference src/recometrics.hpp:84-97 in its canonical
// build.  Geometry differences from the fp32 sweep:
//   * D is 16 items x 16 users: column (user) = lane & 15, row (item) = (lane >> 4) + 4*reg -- a user sits on FOUR lanes
//     (q = lane >> 4), a group is 16 users, a wave owns 16 users x 32 items per tile (two MFMA tiles, 8 values per lane);
//   * the packed operand image is [g][q][row][2 doubles]: a b128 read gives the lane its factors k = 8g + q and 8g + 4 + q;
//   * the factor axis is streamed in chunks of 64 factors (32 KiB LDS per buffer); up to 128 factors the user factors
//     stay in registers for the whole sweep (64 VGPRs), beyond that each chunk's are re-read from L2.
#pragma once
#include "rm_device.hpp"
#include "rm_list.hpp"
#include "rm_launch.hpp"

#ifndef RM_EARLY_ARRIVE
#define RM_EARLY_ARRIVE 4
#endif

namespace rm {

typedef double f64x4 __attribute__((ext_vector_type(4)));


typedef __attribute__((address_space(3))) u32x4 *LdsList64Ptr;
typedef u32x4 *GblList64Ptr;

__device__ __forceinline__ double pos_inf_d() { return __longlong_as_double(0x7ff0000000000000ll); }
__device__ __forceinline__ double neg_inf_d() { return __longlong_as_double(0xfff0000000000000ll); }
__device__ __forceinline__ double nan_sentinel_d() { return __longlong_as_double(-1ll); }

// Compare / select pairs as inline asm in a software-pipelined order, the LDS reads of a level waited for in two groups: see
// the fp32 sweep (rm_sweep.hpp, auc_pass) -- a wave pays for every issue slot, s_nop and s_waitcnt included.
#define RM_CMP_LT64(m, p, x) asm volatile("v_cmp_lt_f64 %0, %1, %2" : "=s"(m) : "v"(p), "v"(x))
#define RM_SEL32(d, a, b, m) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(m))

template <int J>
__device__ __forceinline__ void auc_pass64(const double (&v)[8], const char *posb, char *histb,
                                           const int *pos_item_g, int sb, int q)
{
    typedef __attribute__((address_space(3))) const double *LdsF64;
    const unsigned pos_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)posb;
    unsigned base[8];
    #pragma unroll
    for (int r = 0; r < 8; r++) base[r] = 0;
    unsigned long long mk0, mk1, mk2;
    #pragma unroll
    for (int st = (J > 0 ? (1 << (J - 1)) : 0); st >= 1; st >>= 1) {
        double pv[8];
        #pragma unroll
        for (int r = 0; r < 8; r++) pv[r] = *(LdsF64)(pos_addr + base[r] + (st - 1) * 128);
        #pragma unroll
        for (int i = 0; i < 8 + 2; i++) {
            if (i == 0) __builtin_amdgcn_s_waitcnt(0xC47F); else if (i == 4) __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(4), lgkmcnt(0)
            if (i < 8) { if (i % 3 == 0) RM_CMP_LT64(mk0, pv[i], v[i]); else if (i % 3 == 1) RM_CMP_LT64(mk1, pv[i], v[i]); else RM_CMP_LT64(mk2, pv[i], v[i]); }
            if (i >= 2) {
                const int j = i - 2;
                const unsigned cand = base[j] + (unsigned)(st * 128);
                if (j % 3 == 0) RM_SEL32(base[j], base[j], cand, mk0); else if (j % 3 == 1) RM_SEL32(base[j], base[j], cand, mk1); else RM_SEL32(base[j], base[j], cand, mk2);
            }
        }
    }
    double nx[8];
    #pragma unroll
    for (int r = 0; r < 8; r++) nx[r] = *(LdsF64)(pos_addr + base[r]);
    unsigned long long tie = 0;
    #pragma unroll
    for (int r = 0; r < 8; r++) tie |= __ballot(nx[r] == v[r]);
    if (tie) {
        // (the tile's first item made opaque inside the rare branch: visible, the eight item ids are loop-invariant code the compiler
        // hoists onto the common path in front of the switch over the table depths -- see the fp32 sweep, rm_sweep.hpp auc_pass)
        int sb_walk = sb;
        asm volatile("" : "+s"(sb_walk));
        #pragma unroll
        for (int r = 0; r < 8; r++) {
            if (nx[r] == v[r]) {
                const int item = sb_walk + (r >> 2) * 16 + q + 4 * (r & 3);
                unsigned t = base[r];
                while (t < (unsigned)(((1 << J) - 1) * 128) && *(const double *)(posb + t) == v[r] &&
                       pos_item_g[(t >> 7) * GROUP_USERS64] > item) t += 128;
                base[r] = t;
            }
        }
    }
    #pragma unroll
    for (int r = 0; r < 8; r++)
        __hip_atomic_fetch_add((unsigned *)(histb + (base[r] >> 1)), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// SPEC = 1: every score provably finite, no tie noise, top-K lists in reach -- the epilogue's run-time switches as constants (see
// the fp32 sweep, rm_sweep.hpp k_sweep SPEC); 0 = read from the argument block.  Picked by the host (Sweep64Args::spec).
template <int NGT, bool AUC, bool DUMP, int LMODE, int SPEC = 0>
__global__ __launch_bounds__(SWEEP_THREADS, 2)
void k_sweep64(Sweep64Args a)
{
    static_assert(SPEC >= 0 && SPEC <= 1 && !(SPEC && DUMP), "unknown specialisation");
    const bool f_nan = SPEC ? false : a.check_nan != 0, f_noise = SPEC ? false : a.noise_E != nullptr, f_ext = SPEC ? false : a.ext_topk != 0;
    constexpr bool LLDS = LMODE == LM_LDS;
    constexpr bool buffered = LMODE == LM_HBM_APPEND;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int GU = GROUP_USERS64;
    // NGT == 0: the group count is a run-time value (a.ngt, a multiple of 8): more than 512 factors
    constexpr bool NGT_RT = NGT == 0;
    constexpr int NGC = NGT_RT ? 8 : (NGT < 8 ? NGT : 8);       // factor groups (of 8) per LDS chunk
    constexpr int NC_CT = NGT_RT ? 0 : NGT / NGC;               // chunks per tile when known at compile time
    const int NGTV = NGT_RT ? a.ngt : NGT;
    const int NC = NGT_RT ? a.ngt / NGC : NC_CT;
    constexpr int BUF_D2 = NGC * 4 * TILE_ITEMS;                // double2 per chunk buffer
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave & 3, sub = wave >> 2;
    const int ul = lane & 15, q = lane >> 4;
    // user blocks are sorted by positive-tree depth (cheapest first): launch the deepest (slowest) ones first so that
    // the last round of the grid is made of the cheap ones -- and of SMALL ones: the tail_ublocks cheapest user blocks
    // come last, cut into tail_splits item ranges each (rm_launch.hpp)
    int blk_u, split, nsplit;
    {
        const int n_ub1 = a.n_ublocks - a.tail_ublocks, b1 = n_ub1 * a.n_splits;
        if ((int)blockIdx.x < b1) { blk_u = a.ublock0 + a.n_ublocks - 1 - (int)(blockIdx.x % n_ub1); split = blockIdx.x / n_ub1; nsplit = a.n_splits; }
        else { const int j = (int)blockIdx.x - b1; blk_u = a.ublock0 + a.tail_ublocks - 1 - j % a.tail_ublocks; split = j / a.tail_ublocks; nsplit = a.tail_splits; }
    }
    const int group = blk_u * GROUPS_PER_BLOCK + gi;
    const bool group_ok = group < a.n_groups;
    const int slot = group * GU + ul;
    const bool slot_ok = group_ok && slot < a.n_slots;
    const int K = a.K, n = a.n;

    const int glast = min(a.n_groups, (blk_u + 1) * GROUPS_PER_BLOCK) - 1;
    const int jb = AUC ? a.gj[glast] : 0;
    const int PLb = (1 << jb) - 1;

    // ---- LDS carve: [B buf0 | B buf1 | lists (8 waves) | positives (4 groups) | histogram (4 groups) | pending ... | sync] ----
    // tables are sized by the block's OWN depth; what the launch allocated beyond that (it is sized for the deepest
    // block) goes to this block's pending buffers
    f64x2 *ldsB = (f64x2 *)smem;
    char *p = smem + 2 * BUF_D2 * 16;
    u32x4 *lists_lds = (u32x4 *)p;
    if (LLDS) p += 8 * K * GU * 16;
    const int PLmax = PLb;
    double *posL = (double *)p;  p += GROUPS_PER_BLOCK * (PLmax + 1) * GU * 8;
    unsigned *histL = (unsigned *)p;  p += AUC ? GROUPS_PER_BLOCK * (PLmax + 1) * GU * 4 : 0;
    p = smem + (((int)(p - smem) + 15) & ~15);
    const int pend_room = (a.sync_off - (int)(p - smem)) / (8 * WAVE * 12);              // entries per lane that still fit
    char *pend_lds = p;

    const int user = slot_ok ? a.slot_user[slot] : -1;
    const bool primary = slot_ok && a.slot_chunk[slot] == 0;
    double thr = primary ? neg_inf_d() : nan_sentinel_d();
    double vmax = neg_inf_d(), vmin = pos_inf_d();
    unsigned long long nanmask = 0;
    int ntc = 0, nte = 0, nt = IDX_EMPTY, nt2 = IDX_EMPTY;     // train cursor: next item and the one after (prefetched)

    const int tiles_per = (a.tiles_total + nsplit - 1) / nsplit;
    const int t0 = split * tiles_per, t1 = min(a.tiles_total, t0 + tiles_per);
    const int ntiles = max(0, t1 - t0);

    if (user >= 0) {
        ntc = a.train_p[user]; nte = a.train_p[user + 1];
        const int first_item = t0 * TILE_ITEMS;
        int lo = ntc, hi = nte;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.train_i[mid] < first_item) lo = mid + 1; else hi = mid; }
        ntc = lo;
        nt = ntc < nte ? a.train_i[ntc] : IDX_EMPTY;
        nt2 = ntc + 1 < nte ? a.train_i[ntc + 1] : IDX_EMPTY;
    }

    // user factors -> registers: [group][g][q][16 users][2 doubles].  Up to 128 factors (64 VGPRs) they stay resident
    // for the whole sweep; beyond that each 64-factor chunk is re-read from L2 when its turn comes.
    constexpr bool AF_RESIDENT = !NGT_RT && NGT <= 16;
    // Streamed factor axis with an even, compile-time chunk count (256 and 512 factors): the user factors of a chunk arrive in two
    // halves -- the first one was loaded a whole chunk ahead (two register sets in rotation), the second one is requested when the
    // chunk begins and lands behind the first half's MFMAs (4 groups x 4 instructions x 64 cycles), so no matrix instruction waits
    // for L2 any more (profiles/r2_pmc_sq_C5.json: 0.4 loads per MFMA, each chunk opened with an L2 round trip).  16 VGPRs more.
    constexpr bool AF_HALF_AHEAD = !AF_RESIDENT && !NGT_RT && NGC == 8 && NC_CT % 2 == 0;
    constexpr int NAF = AF_RESIDENT ? NGT : NGC;
    constexpr int HALF = NGC / 2;
    f64x2 af[AF_HALF_AHEAD ? HALF : NAF];                        // half-ahead: the SECOND half of the current chunk
    f64x2 afA[AF_HALF_AHEAD ? HALF : 1], afB[AF_HALF_AHEAD ? HALF : 1];      // first halves: current / next chunk, in rotation
    const f64x2 *af_src = a.Ap + ((size_t)(group_ok ? group : 0) * NGTV * 4 + q) * GU + ul;    // + g * 4 * GU
    if (AF_RESIDENT) {
        #pragma unroll
        for (int g = 0; g < (AF_RESIDENT ? NGT : 0); g++) {
            f64x2 z; z.x = 0; z.y = 0;
            af[g] = group_ok ? af_src[(size_t)g * 4 * GU] : z;
        }
    }

    // top-K list owned by the q == 0 lane of the user: LDS [K][16 users] replace-the-minimum, the same scheme in HBM (K <= 32), or --
    // K > 32 -- per-LANE append buffers in HBM with lane-parallel selection (rm_list.hpp)
    const int CAP = 2 * K + 32;
    LdsList64Ptr Ll = (LdsList64Ptr)((GblList64Ptr)lists_lds + wave * K * GU + ul);
    GblList64Ptr Lr = (LLDS || buffered) ? nullptr : a.glists + ((size_t)blockIdx.x * 8 + wave) * GU * CAP + ul;
    // the wave's lane buffers, [entry][64 lanes]: lane_cap scores (8 B), then lane_cap item ids (4 B); the lane's next entry as a
    // 32-bit byte offset into the scores (the items' offset is half of it)
    const int lane_cap = a.lane_cap;
    const char *lb_scores = buffered ? (const char *)a.glists + ((size_t)blockIdx.x * 8 + wave) * ((size_t)lane_cap * (WAVE * 12)) : nullptr;
    const unsigned lb_items_disp = (unsigned)lane_cap * (WAVE * 8);      // from the wave's scores to its item ids: ONE scalar, not a second base pair (the kernel spills scalars)
    unsigned lb_off = (unsigned)lane * 8u;                        // (entries of the lane) * 512 + lane * 8
    const unsigned lb_trigger = (unsigned)(lane_cap - 7) << 9;    // a tile appends at most 8 per lane: select when cnt > lane_cap - 8
    double ws = neg_inf_d(); int widx = IDX_EMPTY, wpos = 0;
    if (q == 0 && (LLDS || !buffered)) for (int i = 0; i < K; i++) {
        if (LLDS) Ll[i * GU] = ListRaw<double>::pack(neg_inf_d(), IDX_EMPTY); else Lr[i * GU] = ListRaw<double>::pack(neg_inf_d(), IDX_EMPTY);
    }

    // Pending buffers (a.pend_cap > 0): every lane appends its own candidates to a small LDS buffer ([pend_cap][64 lanes],
    // scores and items in separate arrays), and the buffers of all 16 users of the wave are merged into their lists
    // together when one fills up -- instead of four 64-bit shuffles and up to four list updates by one owner lane per
    // score register that holds a candidate.  See the fp32 sweep.
    typedef __attribute__((address_space(3))) double *LdsF64Ptr;
    typedef __attribute__((address_space(3))) int *LdsI32Ptr;
    const int pend_want = pend_room < a.pend_cap ? pend_room : a.pend_cap;        // a.pend_cap = the most that is useful (0 = off)
    const int pend_cap = (buffered || pend_want < 2) ? 0 : pend_want;
    int pcnt = 0;
    LdsF64Ptr Ps = (LdsF64Ptr)pend_lds + wave * pend_cap * WAVE + lane;
    LdsI32Ptr Pi = (LdsI32Ptr)(pend_lds + 8 * pend_cap * WAVE * 8) + wave * pend_cap * WAVE + lane;
    auto offer_entry = [&](double s, int item) {                  // owner lanes only
        if (LLDS) { if (s >= ws) list_offer<double, GU>(Ll, K, s, item, ws, widx, wpos); }
        else if (s >= ws) list_offer<double, GU>(Lr, K, s, item, ws, widx, wpos);
    };
    bool merged = false;                                          // a merge since the bound was last published
    auto merge_pending = [&]() {
        merged = true;
        const int c1 = __shfl(pcnt, ul + 16), c2 = __shfl(pcnt, ul + 32), c3 = __shfl(pcnt, ul + 48);
        const int m01 = pcnt > c1 ? pcnt : c1, m23 = c2 > c3 ? c2 : c3;
        const int lim = q == 0 ? (m01 > m23 ? m01 : m23) : 0;
        for (int i = 0; wave_any(i < lim); i++) {
            if (q == 0) {
                if (i < pcnt) offer_entry(Ps[i * WAVE], Pi[i * WAVE]);
                if (i < c1) offer_entry(Ps[i * WAVE + 16], Pi[i * WAVE + 16]);
                if (i < c2) offer_entry(Ps[i * WAVE + 32], Pi[i * WAVE + 32]);
                if (i < c3) offer_entry(Ps[i * WAVE + 48], Pi[i * WAVE + 48]);
            }
        }
        pcnt = 0;
    };

    if (AUC) {
        for (int i = tid; i < GROUPS_PER_BLOCK * (PLb + 1) * GU; i += SWEEP_THREADS) {
            const int g4 = i / ((PLb + 1) * GU), rem = i % ((PLb + 1) * GU);
            const int gg = blk_u * GROUPS_PER_BLOCK + g4;
            posL[g4 * (PLmax + 1) * GU + rem] = gg < a.n_groups ? a.pos_score[(a.grow[gg] + gg) * GU + rem] : pos_inf_d();
        }
        for (int i = tid; i < GROUPS_PER_BLOCK * (PLmax + 1) * GU; i += SWEEP_THREADS) histL[i] = 0;
    }
    const char *posb = (const char *)(posL + gi * (PLmax + 1) * GU + ul);
    char *histb = (char *)(histL + gi * (PLmax + 1) * GU + ul);
    const int *pos_item_g = (AUC && group_ok) ? a.pos_item + (a.grow[group] + group) * GU + ul : nullptr;
    // streamed users (see the fp32 sweep): the lane writes its masked scores to the user's row in HBM
    const bool stream_lane = slot_ok && slot >= a.stream_slot0;
    const bool wave_streams = wave_any(stream_lane);
    double *stream_row = stream_lane ? a.stream_scores + (size_t)(slot - a.stream_slot0) * (size_t)a.stream_ld : nullptr;
    // tie noise (rm_noise.hpp): the lane's user's row of per-item noise values
    const double *noise_lane = (a.noise_E && user >= 0)
        ? a.noise_E + (size_t)(a.noise_row ? a.noise_row[user] - a.noise_row0 : user) * (size_t)a.noise_ld : nullptr;

    // unit u = (tile, chunk): contiguous BUF_D2 double2 of the packed image [tile][g][q][row][2].  The LDS image is
    // [sub][g][q][32 items]: a sub-tile is staged by the four waves that read it (see the fp32 sweep), 1 KiB pieces of
    // two 512-byte runs (q pair x 32 items)
    // (addressing as in the fp32 sweep: scalar tile base, one constant VGPR offset per piece, integer M0)
    const unsigned lds_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)smem;
    constexpr int STAGE_PIECES = (NGC * 2 + 3) / 4;
    unsigned stage_voff[STAGE_PIECES];
    #pragma unroll
    for (int j = 0; j < STAGE_PIECES; j++) {
        const int pc = gi + 4 * j;                           // piece = (g, q pair)
        stage_voff[j] = (unsigned)((((pc >> 1) * 4 + (pc & 1) * 2 + (lane >> 5)) * TILE_ITEMS + sub * 32 + (lane & 31)) * 16);
    }
    auto stage = [&](int tile, int chunk, int buf) {
        const char *src = (const char *)a.Bp + ((size_t)tile * NGTV + (size_t)chunk * NGC) * (4 * TILE_ITEMS * 16);
        // inline asm, not the builtin: see the note at the fp32 sweep's stage() (the compiler would otherwise wait for
        // the DMA in front of the next LDS read)
        #pragma unroll
        for (int j = 0; j < STAGE_PIECES; j++) {
            const int pc = gi + 4 * j;
            if ((NGC * 2) % 4 == 0 || pc < NGC * 2) {
                const unsigned m0v = lds_base + (unsigned)((buf * BUF_D2 + sub * NGC * 128 + pc * 64) * 16);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :: "s"(m0v), "v"(stage_voff[j]), "s"(src) : "memory", "m0");
            }
        }
    };

    unsigned long long thr_pub = 0;
    // select while waiting (see the fp32 sweep): the domain's selection epoch in LDS word 4 + sub
    typedef __attribute__((address_space(3))) unsigned *LdsEpochPtr;
    LdsEpochPtr sel_epoch = (LdsEpochPtr)(smem + a.sync_off) + 4 + sub;
    unsigned sel_seen = 0u;
    const unsigned lb_half = ((lb_trigger + (7u << 9)) >> 10) << 9;      // (lane_cap / 2) << 9
    unsigned lb_trig_now = sub == 0 ? lb_trigger : (unsigned)(lane_cap - lane_cap / 4) << 9;      // stagger of the two domains: see the fp32 sweep
    lb_trig_now = lb_trig_now < lb_trigger ? lb_trig_now : lb_trigger;                               // (never beyond the safe level)
    auto lane_bounds = [&]() {
        const double hi_hint = f_noise ? pos_inf_d() : LaneSel<double>::umax(vmax);
        // (out of line: inlined, the selection's working set is added to a register budget that is already full at 256 factors)
        const LaneSelResult<double> sr = lane_select_call<double>((double *)lb_scores + lane, (int *)(lb_scores + lb_items_disp) + lane, (int)(lb_off >> 9), K, primary, thr, hi_hint, n);
        const double t_new = sr.thr; const unsigned long long kk = sr.kth_key;
        lb_off = ((unsigned)sr.cnt << 9) | ((unsigned)lane * 8u);
        if (kk) {
            thr = t_new;
            if (q == 0 && kk > thr_pub) atomicMax(a.thr_shared + slot, kk);
            thr_pub = kk > thr_pub ? kk : thr_pub;
        }
    };
    auto do_epi = [&](const f64x4 &acc_lo, const f64x4 &acc_hi, int tile, unsigned long long thr_seen) {
        const int sb = tile * TILE_ITEMS + sub * 32;
        double v[8];
        #pragma unroll
        for (int r = 0; r < 4; r++) { v[r] = acc_lo[r]; v[4 + r] = acc_hi[r]; }
        if (DUMP) {
            #pragma unroll
            for (int r = 0; r < 8; r++) {
                const int item = sb + (r >> 2) * 16 + q + 4 * (r & 3);
                if (slot_ok && item < n) a.dump[(size_t)slot * n + item] = v[r];
            }
            return;
        }
        const bool slow = wave_any(nt < sb + 32) || (sb + 32 > n);
        if (slow) {
            unsigned mbits = 0;
            auto consume = [&]() {                                  // first consumption of a step peeled: see the fp32 sweep
                if (nt >= sb) mbits |= 1u << (nt - sb);
                ntc++;
                nt = nt2;
                nt2 = ntc + 1 < nte ? a.train_i[ntc + 1] : IDX_EMPTY;
            };
            if (nt < sb + 32) {
                consume();
                while (nt < sb + 32) consume();
            }
            if (sb + 32 > n) mbits |= (n > sb) ? (0xffffffffu << (n - sb)) : 0xffffffffu;
            #pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool mk = (mbits >> ((r >> 2) * 16 + q + 4 * (r & 3))) & 1u;
                if (f_nan) nanmask |= __ballot(!mk && (v[r] != v[r]));
                v[r] = mk ? nan_sentinel_d() : v[r];
            }
        } else if (f_nan) {
            #pragma unroll
            for (int r = 0; r < 8; r++) nanmask |= __ballot(v[r] != v[r]);
        }
        // (NaN-ignoring hardware max / min through asm: the generic fmax / fmin add a canonicalising v_max_f64 x, x, x per value,
        // and an f64 vector instruction occupies the pipe twice as long as an f32 one)
        {
            double m01, m23, m45, m67, n01, n23, n45, n67;
            asm("v_max_f64 %0, %1, %2" : "=v"(m01) : "v"(v[0]), "v"(v[1])); asm("v_max_f64 %0, %1, %2" : "=v"(m23) : "v"(v[2]), "v"(v[3]));
            asm("v_max_f64 %0, %1, %2" : "=v"(m45) : "v"(v[4]), "v"(v[5])); asm("v_max_f64 %0, %1, %2" : "=v"(m67) : "v"(v[6]), "v"(v[7]));
            asm("v_min_f64 %0, %1, %2" : "=v"(n01) : "v"(v[0]), "v"(v[1])); asm("v_min_f64 %0, %1, %2" : "=v"(n23) : "v"(v[2]), "v"(v[3]));
            asm("v_min_f64 %0, %1, %2" : "=v"(n45) : "v"(v[4]), "v"(v[5])); asm("v_min_f64 %0, %1, %2" : "=v"(n67) : "v"(v[6]), "v"(v[7]));
            asm("v_max_f64 %0, %1, %2" : "=v"(m01) : "v"(m01), "v"(m23)); asm("v_max_f64 %0, %1, %2" : "=v"(m45) : "v"(m45), "v"(m67));
            asm("v_min_f64 %0, %1, %2" : "=v"(n01) : "v"(n01), "v"(n23)); asm("v_min_f64 %0, %1, %2" : "=v"(n45) : "v"(n45), "v"(n67));
            asm("v_max_f64 %0, %1, %2" : "=v"(m01) : "v"(m01), "v"(m45)); asm("v_min_f64 %0, %1, %2" : "=v"(n01) : "v"(n01), "v"(n45));
            asm("v_max_f64 %0, %1, %2" : "=v"(vmax) : "v"(vmax), "v"(m01)); asm("v_min_f64 %0, %1, %2" : "=v"(vmin) : "v"(vmin), "v"(n01));
        }
        // tie noise (reference :531-534: added AFTER the validity scan, in real_t)
        if (f_noise && noise_lane) {
            #pragma unroll
            for (int r = 0; r < 8; r++) v[r] += noise_lane[sb + (r >> 2) * 16 + q + 4 * (r & 3)];
        }
        if (primary && thr_seen > thr_pub) { thr_pub = thr_seen; const double t = ord_unkey(thr_seen); thr = t > thr ? t : thr; }
        unsigned long long cm = 0;
        if (!f_ext) {                                           // (ext_topk: see the fp32 sweep)
            #pragma unroll
            for (int r = 0; r < 8; r++) cm |= __ballot(v[r] >= thr);
        }
        if (buffered) {
            // K > 32: every lane appends its own candidates to its own buffer (rm_list.hpp)
            if (cm) {
                const int sbq = sb + q;
                #pragma unroll
                for (int r = 0; r < 8; r++) {
                    if (v[r] >= thr) {
                        const int item = sbq + (r >> 2) * 16 + 4 * (r & 3);
                        const unsigned ioff = (lb_off >> 1) + lb_items_disp;
                        // (one statement, s_nop 4 in front: a base pair that was spilled is restored by v_readlane right before it, and a
                        // VALU write of an SGPR needs five wait states before a memory instruction reads it -- see the fp32 sweep)
                        asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2\n\tglobal_store_dword %3, %4, %5"
                                     :: "v"(lb_off), "v"(v[r]), "s"(lb_scores), "v"(ioff), "v"(item), "s"(lb_scores) : "memory");
                        lb_off += 512u;
                    }
                }
            }
        } else
        if (cm && pend_cap) {
            unsigned ov = 0;                                    // score registers that did not fit the lane's buffer
            #pragma unroll
            for (int r = 0; r < 8; r++) {
                const bool c = v[r] >= thr;
                if (__ballot(c)) {
                    if (c) {
                        if (pcnt < pend_cap) { Ps[pcnt * WAVE] = v[r]; Pi[pcnt * WAVE] = sb + (r >> 2) * 16 + 4 * (r & 3) + q; pcnt++; }
                        else ov |= 1u << r;
                    }
                }
            }
            bool more = wave_any(ov != 0);
            if (more || wave_any(pcnt >= pend_cap - 1)) merge_pending();
            while (more) {                                      // warm-up only
                unsigned ov2 = 0;
                #pragma unroll
                for (int r = 0; r < 8; r++) {
                    const bool c = ((ov >> r) & 1u) && v[r] >= thr;
                    if (__ballot(c)) {
                        if (c) {
                            if (pcnt < pend_cap) { Ps[pcnt * WAVE] = v[r]; Pi[pcnt * WAVE] = sb + (r >> 2) * 16 + 4 * (r & 3) + q; pcnt++; }
                            else ov2 |= 1u << r;
                        }
                    }
                }
                ov = ov2; more = wave_any(ov != 0);
                merge_pending();
            }
        } else if (cm) {
            #pragma unroll
            for (int r = 0; r < 8; r++) {
                if (wave_any(v[r] >= thr)) {
                    const double o1 = __shfl(v[r], ul + 16), o2 = __shfl(v[r], ul + 32), o3 = __shfl(v[r], ul + 48);
                    if (q == 0 && primary) {
                        const int ib = sb + (r >> 2) * 16 + 4 * (r & 3);
                        if (LLDS) {
                            if (v[r] >= ws) list_offer<double, GU>(Ll, K, v[r], ib, ws, widx, wpos);
                            if (o1 >= ws) list_offer<double, GU>(Ll, K, o1, ib + 1, ws, widx, wpos);
                            if (o2 >= ws) list_offer<double, GU>(Ll, K, o2, ib + 2, ws, widx, wpos);
                            if (o3 >= ws) list_offer<double, GU>(Ll, K, o3, ib + 3, ws, widx, wpos);
                        } else {
                            if (v[r] >= ws) list_offer<double, GU>(Lr, K, v[r], ib, ws, widx, wpos);
                            if (o1 >= ws) list_offer<double, GU>(Lr, K, o1, ib + 1, ws, widx, wpos);
                            if (o2 >= ws) list_offer<double, GU>(Lr, K, o2, ib + 2, ws, widx, wpos);
                            if (o3 >= ws) list_offer<double, GU>(Lr, K, o3, ib + 3, ws, widx, wpos);
                        }
                    }
                }
            }
        }
        if (!buffered && cm && (!pend_cap || merged)) {                      // with pending buffers the K-th best only moves in a merge
            merged = false;
            const double t2 = __shfl(ws, ul);
            if (primary) {
                thr = t2 > thr ? t2 : thr;
                const unsigned long long kk = ord_key(t2);
                if (q == 0 && kk > thr_pub) atomicMax(a.thr_shared + slot, kk);
                thr_pub = kk > thr_pub ? kk : thr_pub;
            }
        }
        if (wave_streams) {
            if (stream_lane) {
                #pragma unroll
                for (int r = 0; r < 8; r++) stream_row[sb + (r >> 2) * 16 + q + 4 * (r & 3)] = v[r];
            }
        }
        if (AUC) {
            switch (jb) {
                case 1: auc_pass64<1>(v, posb, histb, pos_item_g, sb, q); break;
                case 2: auc_pass64<2>(v, posb, histb, pos_item_g, sb, q); break;
                case 3: auc_pass64<3>(v, posb, histb, pos_item_g, sb, q); break;
                case 4: auc_pass64<4>(v, posb, histb, pos_item_g, sb, q); break;
                case 5: auc_pass64<5>(v, posb, histb, pos_item_g, sb, q); break;
                case 6: auc_pass64<6>(v, posb, histb, pos_item_g, sb, q); break;
                default: break;
            }
        }
        // K > 32: some lane is a tile away from a full buffer -> every user of the wave raises its bound (see the fp32 sweep)
        if (buffered) {
            const bool own = wave_any(lb_off >= lb_trig_now);
            unsigned ep = __builtin_amdgcn_readfirstlane(*sel_epoch);
            if (own || (ep != sel_seen && wave_any(lb_off >= lb_half))) {
                if (own && ep == sel_seen) {
                    if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"((unsigned)(__UINTPTR_TYPE__)sel_epoch), "v"(1u) : "memory");
                    ep++;
                }
                lane_bounds();
                lb_trig_now = lb_trigger;
            }
            sel_seen = ep;
        }
    };

    // ---- main loop over tiles, chunks of the factor axis statically unrolled inside; one barrier per chunk.  All waves
    // run in phase (MFMA chunks, then the tile's epilogue), as in the fp32 sweep. ----
    f64x4 clo, chi;
    const int nunits = ntiles * NC;
    // split barrier on an LDS arrival counter (see the fp32 sweep): arrive after the unit's last MFMA, wait before the
    // next unit touches the buffers; the epilogue in between absorbs the skew between waves
    typedef __attribute__((address_space(3))) unsigned *LdsSyncPtr;
    LdsSyncPtr arrive = (LdsSyncPtr)(smem + a.sync_off) + sub;       // one domain per sub-tile: its four waves
    if (tid < 8) ((LdsSyncPtr)(smem + a.sync_off))[tid] = 0u;       // arrival counters, and (words 4..7) the domains' selection epochs
    if (ntiles > 0) stage(t0, 0, 0);
    if (AF_HALF_AHEAD) {                          // first half of chunk 0 (in front of the drain below: nothing is pending at the loop's entry)
        #pragma unroll
        for (int g = 0; g < HALF; g++) afA[g] = af_src[(size_t)g * 4 * GU];
    }
    __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);      // builtin: the compiler's wait-count bookkeeping sees the drain
    __syncthreads();
    // shared K-th-best bound, read one tile ahead (drained by the closing wait); the seeded bound counts from the first tile on
    unsigned long long thr_seen = (primary && !DUMP) ? __hip_atomic_load(a.thr_shared + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    for (int i = 0; i < ntiles; i++) {
        const unsigned long long thr_next = (primary && !DUMP) ? __hip_atomic_load(a.thr_shared + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        // resident user factors are indexed af[c * NGC + gl]: the chunk loop must then be unrolled; when each chunk's
        // factors are re-read the loop stays rolled (4x less code and register pressure at 256 factors)
        if (AF_HALF_AHEAD) {
            auto unit_body = [&](int c, f64x2 (&cur)[AF_HALF_AHEAD ? HALF : 1], f64x2 (&nxt)[AF_HALF_AHEAD ? HALF : 1]) __attribute__((always_inline)) {
                const int unit = i * NC + c;
                const int buf = unit & 1;
                if (unit > 0) {
                    const unsigned target = 4u * (unsigned)unit;
#ifndef RM_ABL_NO_BARRIER                                                 // (timing only: what the coupling of the domain's four waves costs)
                    while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
                        __builtin_amdgcn_s_sleep(1);
                        if (buffered && !DUMP) {
                            const unsigned ep = __builtin_amdgcn_readfirstlane(*sel_epoch);
                            if (ep != sel_seen) { sel_seen = ep; if (wave_any(lb_off >= lb_half)) { lane_bounds(); lb_trig_now = lb_trigger; } }
                        }
                    }
#endif
                }
                // second half of this chunk first (it is needed 16 MFMAs from now), then the item tile of the next unit, then
                // the first half of the next chunk (needed a whole chunk from now; after the last chunk: chunk 0 of the next tile)
                #pragma unroll
                for (int g = 0; g < HALF; g++) af[g] = af_src[(size_t)(c * NGC + HALF + g) * 4 * GU];
                if (unit + 1 < nunits) {
                    const int nu = unit + 1;
                    stage(t0 + nu / NC, nu % NC, nu & 1);
                }
                const int cn = c + 1 == NC ? 0 : c + 1;
                #pragma unroll
                for (int g = 0; g < HALF; g++) nxt[g] = af_src[(size_t)(cn * NGC + g) * 4 * GU];
                __builtin_amdgcn_sched_barrier(0);               // (all requests in flight before the first matrix instruction)
                const f64x2 *bb = ldsB + buf * BUF_D2 + sub * NGC * 128 + q * 32 + ul;
                if (c == 0) {
                    #pragma unroll
                    for (int r = 0; r < 4; r++) { clo[r] = 0.; chi[r] = 0.; }
                }
                // EARLY ARRIVAL: the item operands of the unit's last EARLY factor groups are read together; once they are in registers this
                // wave is done with the buffer, so it arrives (its DMA share of the next unit has had most of the unit to land) BEFORE
                // it issues the matrix instructions of those groups -- the partners' wait at the next unit's start ends that much sooner
                // (the coupling of a sub-tile's four waves at four barriers per tile was 9 % of C5; r3_ab_c2.txt r3zp)
                constexpr int EARLY = RM_EARLY_ARRIVE < NGC ? RM_EARLY_ARRIVE : NGC;
                f64x2 tb0[EARLY], tb1[EARLY];
                #pragma unroll
                for (int gl = 0; gl < NGC; gl++) {
                    if (gl == NGC - EARLY) {
                        #pragma unroll
                        for (int e = 0; e < EARLY; e++) { tb0[e] = bb[(gl + e) * 128]; tb1[e] = bb[(gl + e) * 128 + 16]; }
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0) lgkmcnt(0) (expcnt left alone)
                        if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"((unsigned)(__UINTPTR_TYPE__)arrive), "v"(1u) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const f64x2 b0 = gl >= NGC - EARLY ? tb0[gl >= NGC - EARLY ? gl - (NGC - EARLY) : 0] : bb[gl * 128];
                    const f64x2 b1 = gl >= NGC - EARLY ? tb1[gl >= NGC - EARLY ? gl - (NGC - EARLY) : 0] : bb[gl * 128 + 16];
                    const f64x2 u = gl < HALF ? cur[gl < HALF ? gl : 0] : af[gl >= HALF ? gl - HALF : 0];
                    clo = __builtin_amdgcn_mfma_f64_16x16x4f64(b0.x, u.x, clo, 0, 0, 0);
                    chi = __builtin_amdgcn_mfma_f64_16x16x4f64(b1.x, u.x, chi, 0, 0, 0);
                    clo = __builtin_amdgcn_mfma_f64_16x16x4f64(b0.y, u.y, clo, 0, 0, 0);
                    chi = __builtin_amdgcn_mfma_f64_16x16x4f64(b1.y, u.y, chi, 0, 0, 0);
                }
                // the epilogue issues ahead of the SIMD partner's matrix instructions: it is the part of a unit whose length varies, and
                // the three waves that wait for this one at the next unit's barrier wait less (C5 0.608 -> 0.614, r3_ab_c2.txt r3zj)
                if (c == NC - 1) { __builtin_amdgcn_s_setprio(3); do_epi(clo, chi, t0 + i, thr_seen); __builtin_amdgcn_s_setprio(0); }
            };
            for (int c = 0; c < NC; c += 2) { unit_body(c, afA, afB); unit_body(c + 1, afB, afA); }
            thr_seen = thr_next;
            continue;
        }
        #pragma unroll(AF_RESIDENT ? NC_CT : 1)
        for (int c = 0; c < NC; c++) {
            const int unit = i * NC + c;
            const int buf = unit & 1;
            if (unit > 0) {                                                       // wait half of the split barrier
                const unsigned target = 4u * (unsigned)unit;
                while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (buffered && !DUMP) {
                        const unsigned ep = __builtin_amdgcn_readfirstlane(*sel_epoch);
                        if (ep != sel_seen) { sel_seen = ep; if (wave_any(lb_off >= lb_half)) { lane_bounds(); lb_trig_now = lb_trigger; } }
                    }
                }
            }
            // (the fp32 sweep's priority balancing of the two domains was measured here too: 1.2 % slower at C5, not adopted)
            if (unit + 1 < nunits) {
                const int nu = unit + 1;
                stage(t0 + nu / NC, nu % NC, nu & 1);
            }
            const f64x2 *bb = ldsB + buf * BUF_D2 + sub * NGC * 128 + q * 32 + ul;
            // streamed factor axis: this chunk's user factors come from L2 right before use.  (Prefetching them one chunk
            // ahead into the registers just consumed, as the fp32 sweep does, costs more in spills here than it hides.)
            if (!AF_RESIDENT) {
                #pragma unroll
                for (int gl = 0; gl < NGC; gl++) af[gl] = af_src[(size_t)(c * NGC + gl) * 4 * GU];
            }
            if (c == 0) {
                #pragma unroll
                for (int r = 0; r < 4; r++) { clo[r] = 0.; chi[r] = 0.; }
            }
            constexpr int EARLY = RM_EARLY_ARRIVE < NGC ? RM_EARLY_ARRIVE : NGC;           // early arrival: see the loop above
            f64x2 tb0[EARLY], tb1[EARLY];
            #pragma unroll
            for (int gl = 0; gl < NGC; gl++) {
                if (gl == NGC - EARLY) {
                    #pragma unroll
                    for (int e = 0; e < EARLY; e++) { tb0[e] = bb[(gl + e) * 128]; tb1[e] = bb[(gl + e) * 128 + 16]; }
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_waitcnt(0x0070);                                  // vmcnt(0) lgkmcnt(0): operands in registers, the next unit's DMA share landed
                    if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"((unsigned)(__UINTPTR_TYPE__)arrive), "v"(1u) : "memory");     // (see the fp32 sweep)
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f64x2 b0 = gl >= NGC - EARLY ? tb0[gl >= NGC - EARLY ? gl - (NGC - EARLY) : 0] : bb[gl * 128];
                const f64x2 b1 = gl >= NGC - EARLY ? tb1[gl >= NGC - EARLY ? gl - (NGC - EARLY) : 0] : bb[gl * 128 + 16];
                const f64x2 u = af[AF_RESIDENT ? c * NGC + gl : gl];
                clo = __builtin_amdgcn_mfma_f64_16x16x4f64(b0.x, u.x, clo, 0, 0, 0);
                chi = __builtin_amdgcn_mfma_f64_16x16x4f64(b1.x, u.x, chi, 0, 0, 0);
                clo = __builtin_amdgcn_mfma_f64_16x16x4f64(b0.y, u.y, clo, 0, 0, 0);
                chi = __builtin_amdgcn_mfma_f64_16x16x4f64(b1.y, u.y, chi, 0, 0, 0);
            }
            if (c == NC - 1) { __builtin_amdgcn_s_setprio(3); do_epi(clo, chi, t0 + i, thr_seen); __builtin_amdgcn_s_setprio(0); }
        }
        thr_seen = thr_next;
    }
    if (DUMP) return;
    if (pend_cap) merge_pending();

    // What follows reads its kernel arguments AGAIN, from the kernarg segment: loaded at the top they are eight scalar pairs alive across
    // the whole tile loop of a kernel that spills scalars (k_sweep64<32,true,false,2,1>: 43 spilled SGPRs with them, 22 without).
    {
    const Sweep64Args &a = *late_kernargs<Sweep64Args>();
    const int n_part = a.part_splits * 2;
    const int part = split * 2 + sub;
    {   // the four lanes of a user hold four quarters of its stats
        vmax = __builtin_fmax(vmax, __shfl_xor(vmax, 16)); vmax = __builtin_fmax(vmax, __shfl_xor(vmax, 32));
        vmin = __builtin_fmin(vmin, __shfl_xor(vmin, 16)); vmin = __builtin_fmin(vmin, __shfl_xor(vmin, 32));
        const bool hn = ((nanmask >> ul) | (nanmask >> (ul + 16)) | (nanmask >> (ul + 32)) | (nanmask >> (ul + 48))) & 1ull;
        if (slot_ok && q == 0) {
            PartialStat<double> ps;
            ps.vmax = vmax; ps.vmin = vmin; ps.rocsum = 0; ps.has_nan = hn ? 1 : 0; ps.pad = 0;
            a.pst[(size_t)slot * n_part + part] = ps;
            Entry<double> *dst = a.pl + ((size_t)slot * n_part + part) * K;
            if (LLDS) { list_sort_desc<double, GU>(Ll, K); for (int i = 0; i < K; i++) ListRaw<double>::unpack(Ll[i * GU], dst[i].s, dst[i].idx); }
            // a user block cut into fewer ranges than the arrays are laid out for: its first block fills in the missing parts
            if (split == 0) for (int sp = nsplit; sp < a.part_splits; sp++) {
                ps.vmax = neg_inf_d(); ps.vmin = pos_inf_d(); ps.has_nan = 0;
                a.pst[(size_t)slot * n_part + sp * 2 + sub] = ps;
                Entry<double> *de = a.pl + ((size_t)slot * n_part + sp * 2 + sub) * K;
                // (k_finalize looks at the first entry of an empty part only, and the fp32 sweep writes no more than that; here the
                // K-entry loop stays: with the single store this kernel's k_metrics > 32 instantiation came out 1.2 % slower at
                // BASELINE C5 -- register allocation, profiles/r4_ab_c2.txt r4s)
                if (a.pl) for (int i = 0; i < K; i++) { de[i].s = neg_inf_d(); de[i].idx = IDX_EMPTY; }
            }
        }
    }
    if (!LLDS && !buffered) {
        if (slot_ok && q == 0) {
            Entry<double> *dst = a.pl + ((size_t)slot * n_part + part) * K;
            list_sort_desc<double, GU>(Lr, K);
            for (int i = 0; i < K; i++) ListRaw<double>::unpack(Lr[i * GU], dst[i].s, dst[i].idx);
        }
    } else if (!LLDS && !a.ext_topk) {
        // the lanes' entries stay where they are: k_collect_topk (rm_finalize.hpp) writes the user's ordered top-K from them
        a.lane_cnt[((size_t)blockIdx.x * 8 + wave) * WAVE + lane] = (slot_ok && primary) ? (int)(lb_off >> 9) : 0;
    }
    if (AUC) {
        __syncthreads();
        for (int i = tid; i < GROUPS_PER_BLOCK * (PLb + 1) * GU; i += SWEEP_THREADS) {
            const int g4 = i / ((PLb + 1) * GU), rem = i % ((PLb + 1) * GU);
            const int gg = blk_u * GROUPS_PER_BLOCK + g4;
            const unsigned c = histL[g4 * (PLmax + 1) * GU + rem];
            if (gg < a.n_groups && c) atomicAdd(&a.hist[(a.grow[gg] + gg) * GU + rem], c);
        }
    }
    }
}

} // namespace rm
