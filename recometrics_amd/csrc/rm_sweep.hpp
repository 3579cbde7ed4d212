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
//   block          4 user groups x NSUB item sub-tiles of 32 = 4 NSUB wavefronts (NSUB = 2: 512 threads, 128 users x 64
//                  items per step; NSUB = 3 up to 64 factors).  Wavefronts w, w+4 (, w+8) share a SIMD and a user group;
//                  the top-K list of a group is one LDS list shared by them.  An f32-input MFMA chain does not co-execute
//                  with VALU / LDS work of the SIMD partner (scratch/coexec2.hip: time = sum), so the epilogue's
//                  instructions add to the MFMA time and the lever is the epilogue's instruction count.
//   operands       user factors live in registers for the whole sweep (NG float4 per lane, up to 128 factors; beyond
//                  that the factor axis is streamed in 128-factor chunks); packed item tiles (rm_prep.hpp
//                  k_pack_items) stream HBM -> LDS by LDS-DMA (inline asm), double buffered.
//   synchronisation  a split barrier on an LDS arrival counter instead of s_barrier per tile: arrive once the tile's last
//                  operands are in registers (before the last eight MFMAs are issued, when the user factors are resident),
//                  wait before the next tile touches the buffers, the whole epilogue in between -- per
//                  sub-tile (the four waves that stage and read the same 32 items), not per block; the sub-tile that is
//                  behind gets the higher issue priority (s_setprio), so that all reach the end of the range together.
//   grid           two levels (rm_launch.hpp): whole rounds of blocks with n_splits item ranges, then the cheapest user
//                  blocks cut into tail_splits smaller ranges that fill the last round.
//   diagnostics    the RM_ABL_* macros compile single stages out (wrong results, timing only): they are how the cost
//                  breakdown in DESIGN.md was measured and are never defined in a product build.
#pragma once
#include "rm_device.hpp"
#include "rm_list.hpp"
#include "rm_launch.hpp"

namespace rm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// NaN-ignoring hardware min / max (the masked-score sentinel is a NaN and must stay invisible).  Through inline asm: the
// generic fmaxf / fminf lower to the same instructions PLUS one canonicalising v_max_f32 x, x, x per MFMA result (the
// compiler cannot prove an accumulator is not a signalling NaN), which doubled the instruction count of the scan.
__device__ __forceinline__ float hw_max(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float hw_max3(float a, float b, float c) { float d; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float hw_min3(float a, float b, float c) { float d; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }


typedef __attribute__((address_space(3))) unsigned long long *LdsListPtr;
typedef u32x2 *GblListPtr;

// AUC rank counting for one tile: branchless lower_bound of every score in the lane's user's sorted positives
// (complete tree of 2^J - 1 rows, +inf padded), then one LDS atomic into the rank histogram.
typedef __attribute__((address_space(3))) const float *LdsF32Ptr;
typedef __attribute__((address_space(3))) unsigned *LdsU32Ptr;

// `pos_addr` = LDS byte address of row 0 of the lane's user in its group's table.  The table of a group is 2^J rows
// of 128 B and is ALIGNED to its own size, so "address of row r" = pos_addr | (r << 7): each level of the branchless
// lower_bound is one OR (candidate address), one compare, one select -- no add.
// One wavefront alone issues ONE instruction of any kind per ~4 cycles, and while its SIMD partners are inside their MFMA
// chains this pass runs alone: its cost is its instruction COUNT, s_nop and s_waitcnt included.  The compiler's version
// of a search level spends 6 issue slots per score -- v_or, s_waitcnt, v_cmp, s_nop (VALU-writes-SGPR hazard in front of
// the select), v_cndmask -- so the compare / select pairs are written as inline asm in a software-pipelined order: the
// select of score i issues after the compares of scores i+1 and i+2, which are the two wait states the hazard asks for
// (three mask registers in rotation), and the LDS reads of a level are waited for in four groups instead of one by one.
// (Measured and dropped: comparing into the EXEC mask -- v_cmpx, v_or of the step under the mask, s_mov exec -- is two vector
// instructions and a scalar one per score and level instead of three vector ones, and 1.3 % (C2) to 2.4 % (NS) SLOWER: a
// vector write of EXEC stalls the vector instruction behind it.  gpurun_out r4w.)
#define RM_CMP_LT(m, p, x) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(p), "v"(x))
#define RM_SEL(d, a, b, m) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(m))
__device__ __forceinline__ float hw_absmin3(float a, float b, float c) { float d; asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

template <int J>
__device__ __forceinline__ void auc_pass(const float (&v)[16], unsigned pos_addr,
                                         const int *pos_item_g, int sb, int h, float piv_root, float piv_lo, float piv_hi)
{
    // The 16 searches are independent: run them level by level (16 LDS reads in flight per level) instead of one
    // dependent 6-deep chain after another, and keep the histogram atomics out of the way until all reads are done.
    // The pivots of the two top levels of the lane's user (root, and the roots of its two subtrees) are per-sweep
    // constants held in registers: two of the dependent LDS round trips disappear.  They are sorted (lo <= root <= hi, the
    // table is padded with +inf at its end), so the subtree a score falls into is simply the NUMBER of them below it:
    // three independent compares, each selecting over the previous one, instead of a dependent compare-select-compare.
    // the block's histogram tables follow its four positives tables of 2^J rows each (k_sweep's carve): a compile-time
    // distance, so the LDS atomic below takes it as its immediate offset instead of an address add per score
    constexpr unsigned HIST_DELTA = (unsigned)GROUPS_PER_BLOCK * (1u << J) * GROUP_USERS * 4u;
    unsigned at[16];                                           // address of row `base`
#ifndef RM_TOP_LEVELS
#define RM_TOP_LEVELS 2
#endif
    // levels resolved from registers: 2 = root and the roots of its two subtrees (three independent compares per score), 1 = the
    // root alone, 0 = every level reads its pivot from LDS (A/B: profiles/r4_ab_c2.txt)
    constexpr int TOP = J >= 2 ? RM_TOP_LEVELS : 0;
    unsigned long long mk0, mk1, mk2;                          // lane masks in SGPR pairs, in rotation
#ifdef RM_ABL_LUT
    // timing model of a table-driven start (wrong results): 3 (4 at depth 6) vector instructions and one byte read per score
    // stand in for all but the last ABL_LEVELS levels
    constexpr int ABL_LEVELS = J >= 6 ? 3 : (J >= 4 ? 2 : J);
    if (J >= 3) {
        #pragma unroll
        for (int r = 0; r < 16; r++) {
            float x; unsigned qa, f;
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x) : "v"(v[r]), "v"(piv_lo), "v"(piv_hi));
            if (J >= 6) asm volatile("v_min_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(piv_root));
            asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %2" : "=v"(qa) : "v"(x), "v"(pos_addr));
            asm volatile("ds_read_u8 %0, %1" : "=v"(f) : "v"(qa));
            at[r] = f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        #pragma unroll
        for (int r = 0; r < 16; r++) asm volatile("v_lshl_add_u32 %0, %1, 7, %2" : "=v"(at[r]) : "v"(at[r] & 7u), "v"(pos_addr));
    } else
#endif
    if (TOP == 2) {
        constexpr unsigned Q = 128u << (J >= 2 ? J - 2 : 0);   // a quarter of the table
        const unsigned a1 = pos_addr | Q, a2 = pos_addr | (2 * Q), a3 = pos_addr | (3 * Q);
        #pragma unroll
        for (int r = 0; r < 16; r++) at[r] = pos_addr;
        #pragma unroll
        for (int i = 0; i < 48 + 2; i++) {                     // i = pivot * 16 + score
            if (i < 48) {
                const float pv = i < 16 ? piv_lo : (i < 32 ? piv_root : piv_hi);
                if (i % 3 == 0) RM_CMP_LT(mk0, pv, v[i % 16]); else if (i % 3 == 1) RM_CMP_LT(mk1, pv, v[i % 16]); else RM_CMP_LT(mk2, pv, v[i % 16]);
            }
            if (i >= 2) {
                const int j = i - 2;
                const unsigned tgt = j < 16 ? a1 : (j < 32 ? a2 : a3);
                if (j % 3 == 0) RM_SEL(at[j % 16], at[j % 16], tgt, mk0); else if (j % 3 == 1) RM_SEL(at[j % 16], at[j % 16], tgt, mk1); else RM_SEL(at[j % 16], at[j % 16], tgt, mk2);
            }
        }
    } else if (TOP == 1) {
        const unsigned a2 = pos_addr | (128u << (J >= 1 ? J - 1 : 0));         // the upper half of the table
        #pragma unroll
        for (int i = 0; i < 16 + 2; i++) {
            if (i < 16) { if (i % 3 == 0) RM_CMP_LT(mk0, piv_root, v[i]); else if (i % 3 == 1) RM_CMP_LT(mk1, piv_root, v[i]); else RM_CMP_LT(mk2, piv_root, v[i]); }
            if (i >= 2) {
                const int j = i - 2;
                if (j % 3 == 0) RM_SEL(at[j], pos_addr, a2, mk0); else if (j % 3 == 1) RM_SEL(at[j], pos_addr, a2, mk1); else RM_SEL(at[j], pos_addr, a2, mk2);
            }
        }
    } else {
        #pragma unroll
        for (int r = 0; r < 16; r++) at[r] = pos_addr;
    }
#ifdef RM_ABL_LUT
    constexpr int FIRST_ST = J >= 3 ? (1 << (ABL_LEVELS - 1)) : (J - TOP > 0 ? (1 << (J - TOP - 1)) : 0);
#else
    constexpr int FIRST_ST = J - TOP > 0 ? (1 << (J - TOP - 1)) : 0;
#endif
    #pragma unroll
    for (int st = FIRST_ST; st >= 1; st >>= 1) {
        float pv[16];
#ifdef RM_ABL_NOLDS_SEARCH
        // timing only (wrong results): the pivots come from a register move instead of LDS -- what the reads' latency costs
        #pragma unroll
        for (int r = 0; r < 16; r++) asm volatile("v_mov_b32 %0, %1" : "=v"(pv[r]) : "v"(at[r]));
#else
        #pragma unroll
        for (int r = 0; r < 16; r++) pv[r] = *(LdsF32Ptr)(at[r] + (st - 1) * 128);
        #pragma unroll
        for (int i = 0; i < 16 + 2; i++) {
            // s_waitcnt lgkmcnt(12 / 8 / 4 / 0) with the other counters left alone (gfx9 encoding, see WAIT_VMCNT0)
            if (i == 0) __builtin_amdgcn_s_waitcnt(0xCC7F); else if (i == 4) __builtin_amdgcn_s_waitcnt(0xC87F);
            else if (i == 8) __builtin_amdgcn_s_waitcnt(0xC47F); else if (i == 12) __builtin_amdgcn_s_waitcnt(0xC07F);
            if (i < 16) { if (i % 3 == 0) RM_CMP_LT(mk0, pv[i], v[i]); else if (i % 3 == 1) RM_CMP_LT(mk1, pv[i], v[i]); else RM_CMP_LT(mk2, pv[i], v[i]); }
            if (i >= 2) {
                const int j = i - 2;
                const unsigned cand = at[j] | (unsigned)(st * 128);
                if (j % 3 == 0) RM_SEL(at[j], at[j], cand, mk0); else if (j % 3 == 1) RM_SEL(at[j], at[j], cand, mk1); else RM_SEL(at[j], at[j], cand, mk2);
            }
        }
    }
#endif
    // exact score tie with a positive (row `base` is the first positive not below s; the table has one +inf pad
    // row, so the read is always in range): the total order is (score desc, item asc), i.e. the candidate also
    // outranks the equal-scored positives with a LARGER item id.  Rare; positives' item ids stay in HBM.  Detection: the
    // smallest |next positive - score| of the 16 (a masked score gives NaN, which v_min3 ignores) is zero.
    float nx[16];
    #pragma unroll
    for (int r = 0; r < 16; r++) nx[r] = *(LdsF32Ptr)(at[r]);
    // (plain subtractions: the sweep's translation units are compiled with -fno-slp-vectorize -- paired into v_pk_add_f32 they
    // issue at less than half the rate of two v_sub_f32, profiles/r3_coexec4_issue_rates.txt; written as asm statements
    // instead, the scheduler serialises the sixteen LDS reads behind two registers)
    float df[16];
    #pragma unroll
    for (int r = 0; r < 16; r++) {
        // the sixteen reads are waited for in four groups (lgkmcnt 12 / 8 / 4 / 0), not one by one: a wait is an issue slot too
        if (r == 0) __builtin_amdgcn_s_waitcnt(0xCC7F); else if (r == 4) __builtin_amdgcn_s_waitcnt(0xC87F);
        else if (r == 8) __builtin_amdgcn_s_waitcnt(0xC47F); else if (r == 12) __builtin_amdgcn_s_waitcnt(0xC07F);
        df[r] = nx[r] - v[r];
    }
    // Every positive meets ITSELF here (its own item is a candidate with exactly its score): at 27k items three tiles out of
    // four hold one, so the "rare" path is the common one at small item counts and its bookkeeping counts (10 % of the C2 sweep
    // before this form): the minima of the five register triples are kept, a triple is only looked into when one of its
    // lanes has a zero, and the walk starts from the entry already known to be equal.
    const float m0 = hw_absmin3(df[0], df[1], df[2]), m1 = hw_absmin3(df[3], df[4], df[5]), m2 = hw_absmin3(df[6], df[7], df[8]);
    const float m3 = hw_absmin3(df[9], df[10], df[11]), m4 = hw_absmin3(df[12], df[13], df[14]);
    const float dmin = hw_absmin3(hw_absmin3(m0, m1, m2), hw_absmin3(m3, m4, df[15]), df[15]);
#ifdef RM_ABL_NOTIE
    if (false) {
#else
    if (wave_any(dmin == 0.f)) {
#endif
        // (the tile's first item made opaque INSIDE the rare branch: left visible, the sixteen item ids of the walk are loop-invariant
        // code the compiler hoists in front of the switch over the table depths -- 24 v_or per tile on the common path, 6 % of the
        // sweep's vector instructions at BASELINE C2, for a branch one tile in forty takes)
        int sb_walk = sb;
        asm volatile("" : "+s"(sb_walk));
        auto walk = [&](int r) {
            if (df[r] == 0.f) {
                const int item = sb_walk + mfma32_row(r, h);
                unsigned t = at[r] - pos_addr;                       // row t holds a positive with exactly this score
                while (*(const int *)((const char *)pos_item_g + t) > item) {        // (the id table has the score table's [row][32 users] layout)
                    t += 128;
                    if (!(t < (unsigned)(((1 << J) - 1) * 128) && *(LdsF32Ptr)(pos_addr + t) == v[r])) break;
                }
                at[r] = pos_addr + t;
            }
        };
        if (wave_any(m0 == 0.f)) { walk(0); walk(1); walk(2); }
        if (wave_any(m1 == 0.f)) { walk(3); walk(4); walk(5); }
        if (wave_any(m2 == 0.f)) { walk(6); walk(7); walk(8); }
        if (wave_any(m3 == 0.f)) { walk(9); walk(10); walk(11); }
        if (wave_any(m4 == 0.f)) { walk(12); walk(13); walk(14); }
        if (wave_any(df[15] == 0.f)) walk(15);
    }
    // (inline asm: the six instantiations of this pass share one tail after the compiler's merge of the switch, which turned
    // the compile-time distance into a register and an address add per score)
    const unsigned one = 1u;
    #pragma unroll
    for (int r = 0; r < 16; r++)
        asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(at[r]), "v"(one), "n"(HIST_DELTA) : "memory");
}

#ifdef RM_STATS
#ifdef RM_STATS_TIME_ONLY
#define RM_STAT(i, x) do {} while (0)
#else
#define RM_STAT(i, x) do { if (lane == 0) atomicAdd(&g_stats[i], (unsigned long long)(x)); } while (0)
#endif
#else
#define RM_STAT(i, x) do {} while (0)
#endif

// LMODE (rm_device.hpp) is a template parameter, not a run-time switch: the three list schemes together do not fit the
// register budget of the 128-factor kernel without spilling.

// NSUB = 32-item sub-tiles per step (2 or 3): a block is 4 user groups x NSUB sub-tiles = 4 NSUB waves, NSUB per SIMD.
// Three need the LDS room and <= 168 VGPRs, i.e. k <= 64: there the epilogue dominates, two waves per SIMD leave the
// vector pipe idle a third of the time (profiles/r1_pmc_sq_C2.json), and the third wave fills it.
// SPEC: the run-time switches of the epilogue as compile-time constants of the two usual cases.  0 = every switch read from the
// argument block; 1 = dense train rows, every score provably finite, no tie noise, top-K lists in reach (small item counts:
// BASELINE C1 / C2); 2 = the same with the CSR cursor instead of dense rows (large item counts: the north-star shape, C3, C4).  The
// host picks the variant (SweepArgs::spec).  The switches cost a scalar branch each per tile and, held as lane masks across the tile
// loop, a dozen SGPR pairs of a kernel that spills SGPRs as it is: as constants 2.6 % at C2, 4.1 % at the north-star shape, 4.4 %
// at C3 (profiles/r4_ab_c2.txt, r4v / r4w).
template <int NGT, bool AUC, bool DUMP, int LMODE, int NSUB, int SPEC = 0>
__global__ __launch_bounds__(256 * NSUB)
void k_sweep(SweepArgs a)
{
    constexpr int TILE = 32 * NSUB;                             // items per step and per packed tile
    static_assert(SPEC >= 0 && SPEC <= 2 && !(SPEC && DUMP), "unknown specialisation");
    const bool f_bits = SPEC == 1 ? true : (SPEC == 2 ? false : a.train_bits != nullptr);
    const bool f_nan = SPEC ? false : a.check_nan != 0, f_noise = SPEC ? false : a.noise_E != nullptr, f_ext = SPEC ? false : a.ext_topk != 0;
#ifdef RM_STATS
    const unsigned long long prof_t0 = __builtin_readcyclecounter();
#endif
    constexpr int NWAVES = 4 * NSUB, THREADS = 64 * NWAVES;
    constexpr bool LLDS = LMODE == LM_LDS;
    constexpr bool buffered = LMODE == LM_HBM_APPEND;
    // factor axis: NGT groups of 8 factors; up to 128 factors (NGT <= 16) a tile is one LDS image and the user factors
    // stay in registers for the whole sweep; beyond that the axis is streamed in chunks of 128 factors (one barrier
    // per chunk) and each chunk of user factors is re-read from L2 when its turn comes.
    // NGT == 0: the group count is a run-time value (a.ngt, a multiple of 16): more than 512 factors, any number of chunks
    constexpr bool NGT_RT = NGT == 0;
    constexpr int NG = NGT_RT ? 16 : (NGT < 16 ? NGT : 16);     // groups per LDS chunk
    constexpr int NC_CT = NGT_RT ? 0 : NGT / NG;                // chunks per tile when known at compile time
    const int NGTV = NGT_RT ? a.ngt : NGT;
    const int NC = NGT_RT ? a.ngt / NG : NC_CT;
    constexpr bool AF_RESIDENT = NC_CT == 1;
    constexpr bool AF_PREFETCH = !AF_RESIDENT && LMODE != LM_HBM_APPEND;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BUF_F4 = NG * 2 * TILE;                 // float4 per packed tile
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar, and known to be)
    const int gi = wave & 3, sub = wave >> 2;                   // group in block / 32-item sub-tile
    const int ul = lane & 31, h = lane >> 5;
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
    const int slot = group * GROUP_USERS + ul;
    const bool slot_ok = group_ok && slot < a.n_slots;
    const int K = a.K, n = a.n;

    const int glast = min(a.n_groups, (blk_u + 1) * GROUPS_PER_BLOCK) - 1;
    const int jb = AUC ? a.gj[glast] : 0;                        // block-uniform tree depth
    const int PLb = (1 << jb) - 1;

    // ---- LDS carve: [B buf0 | B buf1 | lists (per wave) | positives (4 groups) | histogram (4 groups) | pending ... | sync] ----
    // The launch allocates for the deepest block (jmax); a block sizes its tables by its OWN depth jb and gives what is
    // left to the pending buffers (one block per CU either way, so the allocation is free).
    float4 *ldsB = (float4 *)smem;
    char *p = smem + 2 * BUF_F4 * 16;
    ListEntry *lists_lds = (ListEntry *)p;
    if (LLDS) p += GROUPS_PER_BLOCK * (K + 2) * GROUP_USERS * (int)sizeof(ListEntry);     // per group: K keys, worst key, its position
    const int PLmax = PLb;
    // each group's positives table (2^jb rows x 128 B) is aligned to its own size (see auc_pass)
    const unsigned tbytes = (unsigned)(PLmax + 1) * GROUP_USERS * 4;
    p = smem + (((unsigned)(p - smem) + tbytes - 1) / tbytes) * tbytes;
    float *posL = (float *)p;  p += GROUPS_PER_BLOCK * tbytes;
    unsigned *histL = (unsigned *)p;  p += AUC ? GROUPS_PER_BLOCK * tbytes : 0;
    const int pend_room = (a.sync_off - (int)(p - smem)) / (NWAVES * WAVE * 8);     // keys per lane that still fit
    char *pend_lds = p;

    // ---- per-lane user state ----
    const int user = slot_ok ? a.slot_user[slot] : -1;
    const bool primary = slot_ok && a.slot_chunk[slot] == 0;
#ifdef RM_ABL_TOPK_NOHIT
    float thr = pos_inf_f();
#else
    float thr = primary ? neg_inf_f() : nan_sentinel_f();        // NaN threshold: "v >= thr" never true
#endif
    float vmax = neg_inf_f(), vmin = pos_inf_f();
    bool track_min = true;                                      // wave-uniform (see the validity scan of the epilogue)
    unsigned long long nanmask = 0;
    int ntc = 0, nte = 0, nt = IDX_EMPTY, nt2 = IDX_EMPTY;     // train cursor: next item and the one after (prefetched)

    // item range of this split
    const int tiles_per = (a.tiles_total + nsplit - 1) / nsplit;
    const int t0 = split * tiles_per, t1 = min(a.tiles_total, t0 + tiles_per);
    const int ntiles = max(0, t1 - t0);

    // dense train rows (small item counts): one word per lane and tile instead of the cursor below
    const unsigned *tb_row = (a.train_bits && user >= 0) ? a.train_bits + (size_t)user * a.train_words : nullptr;
    // (the rows are <= 8 GiB in all, rm_lib.hip dense_rows_fit: a 32-bit word index, advanced by NSUB per tile, addresses them)
    unsigned tb_idx = (a.train_bits && user >= 0) ? (unsigned)user * (unsigned)a.train_words + (unsigned)(t0 * NSUB + sub) : 0u;
    if (user >= 0 && !f_bits) {
        ntc = a.train_p[user]; nte = a.train_p[user + 1];
        // first train item at or after this wave's first item (lower_bound)
        const int first_item = t0 * TILE;
        int lo = ntc, hi = nte;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.train_i[mid] < first_item) lo = mid + 1; else hi = mid; }
        ntc = lo;
        nt = ntc < nte ? a.train_i[ntc] : IDX_EMPTY;
        nt2 = ntc + 1 < nte ? a.train_i[ntc + 1] : IDX_EMPTY;
    }

    // user factors -> registers (packed: [group][g][h][32][4 floats])
    float4 af[NG];
    const float4 *af_src = a.Ap + ((size_t)(group_ok ? group : 0) * NGTV * 2 + h) * GROUP_USERS + ul;     // + g * 2 * GROUP_USERS
    // resident factors, or chunk 0 of a streamed axis (every later chunk is prefetched by do_mfma)
    #pragma unroll
    for (int g = 0; g < NG; g++)
        af[g] = (group_ok || !AF_RESIDENT) ? af_src[(size_t)g * 2 * GROUP_USERS] : make_float4(0.f, 0.f, 0.f, 0.f);

    // top-K list of this wave.  LDS: [K][32 users], unsorted, replace-the-minimum, owned by the lanes with h == 0.
    // HBM, K <= 32: the same scheme in memory.  HBM, K > 32: per-LANE append buffers + lane-parallel selection (rm_list.hpp).
    const int CAP = 2 * K + 32;
    // The LDS list is shared by the NSUB waves of a user group (same users, interleaved item sub-tiles): its K-th best is
    // the K-th best of everything the group has seen, not of a third of it, so fewer scores pass the bound (C2: 180
    // candidates per user and split instead of 410).  Rows K and K + 1 hold the worst key and its position; a wave takes
    // the group's lock (LDS word behind the arrival counters) around its updates -- pending-buffer merges, mostly.
    LdsListPtr Ll = (LdsListPtr)((unsigned long long *)lists_lds + gi * (K + 2) * GROUP_USERS + ul);
    LdsListPtr Lwk = Ll + K * GROUP_USERS, Lwp = Ll + (K + 1) * GROUP_USERS;
    LdsU32Ptr list_lock = (LdsU32Ptr)(smem + a.sync_off) + 4 + gi;
    unsigned long long wkey = 0;                                   // LDS list: key of its worst entry (0 = empty slot)
    // small K in HBM: [K][32 users] replace-the-minimum like the LDS list (cheaper than selections below K ~ 32)
    GblListPtr Lr = (LLDS || buffered) ? nullptr : a.glists + ((size_t)blockIdx.x * NWAVES + wave) * GROUP_USERS * CAP + ul;
    // K > 32: the wave's lane buffers, [entry][64 lanes] (score, item) pairs (a scalar base, the lane's next entry as a 32-bit byte
    // offset: an append is global_store_dwordx2 voff, pair, sbase -- no 64-bit vector address arithmetic)
    const int lane_cap = a.lane_cap;
    const char *lb_scores = buffered ? (const char *)a.glists + ((size_t)blockIdx.x * NWAVES + wave) * ((size_t)lane_cap * (WAVE * 8)) : nullptr;
    unsigned lb_off = (unsigned)lane * 8u;                        // (entries of the lane) * 512 + lane * 8
    const unsigned lb_trigger = (unsigned)(lane_cap - 15) << 9;   // a tile appends at most 16 per lane: select when cnt > lane_cap - 16
    float ws = neg_inf_f(); int widx = IDX_EMPTY, wpos = 0;
    if (LLDS) {
        if (sub == 0 && h == 0) for (int i = 0; i < K + 2; i++) Ll[i * GROUP_USERS] = 0ull;
        if (tid < 4) ((LdsU32Ptr)(smem + a.sync_off))[4 + tid] = 0u;
    } else if (h == 0 && !buffered) {
        for (int i = 0; i < K; i++) Lr[i * GROUP_USERS] = ListRaw<float>::pack(neg_inf_f(), IDX_EMPTY);
    }
    auto list_acquire = [&]() {                                   // wave-uniform; the holder never waits for another wave
        if (!LLDS) return;
        for (;;) {
            unsigned got = 1u;
            if (lane == 0) { unsigned expect = 0u; got = __hip_atomic_compare_exchange_strong(list_lock, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 0u : 1u; }
            if (__builtin_amdgcn_readfirstlane(got) == 0u) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (h == 0) { wkey = Lwk[0]; wpos = (int)Lwp[0]; }
    };
    auto list_release = [&]() {
        if (!LLDS) return;
        if (h == 0) { Lwk[0] = wkey; Lwp[0] = (unsigned long long)wpos; }
        if (lane == 0) __hip_atomic_store(list_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // Pending buffers (a.pend_cap > 0, whenever LDS has room): a candidate is first appended to its LANE's small buffer
    // ([pend_cap][64 lanes] packed keys per wave: one LDS write), and the buffers are merged into the lists for all 32
    // users of the wave at once when one of them fills up.  Offering candidates one score register at a time keeps 1-2
    // lanes busy per list update -- a replace-the-minimum scan in LDS, or a store plus K loads with their HBM round
    // trip when the lists live in HBM; the merge does the same work with every owner lane that has any.
    const int pend_want = pend_room < a.pend_cap ? pend_room : a.pend_cap;        // a.pend_cap = the most that is useful (0 = off)
    const int pend_cap = (buffered || pend_want < 2) ? 0 : pend_want;
    int pcnt = 0;
    LdsListPtr Pp = (LdsListPtr)((unsigned long long *)pend_lds + wave * pend_cap * WAVE + lane);
    auto offer_key = [&](unsigned long long key) {                // owner lanes only
        if (LLDS) { keylist_offer<GROUP_USERS>(Ll, K, key, wkey, wpos); return; }
        float s; int item;
        unpack_key(key, s, item);
        if (s >= ws) list_offer<float, GROUP_USERS>(Lr, K, s, item, ws, widx, wpos);
    };
    bool merged = false;                                          // a merge since the bound was last published
    auto merge_pending = [&]() {
        merged = true;
        const int pc = __shfl_xor(pcnt, 32);                      // the partner lane's count (same user, other item rows)
        const int lim = h == 0 ? (pcnt > pc ? pcnt : pc) : 0;
        RM_STAT(5, 1);
        list_acquire();
        for (int i = 0; wave_any(i < lim); i++) {
            RM_STAT(6, 1);
            if (h == 0) {
                if (i < pcnt) offer_key(Pp[i * WAVE]);
                if (i < pc) offer_key(Pp[i * WAVE + 32]);
            }
        }
        list_release();
        pcnt = 0;
        if (LLDS) ws = (wkey >> 32) ? ord_unkey((unsigned)(wkey >> 32)) : neg_inf_f();
    };

    // positives -> LDS, histogram zeroed
    if (AUC) {
        for (int i = tid; i < GROUPS_PER_BLOCK * (PLb + 1) * GROUP_USERS; i += THREADS) {
            const int g4 = i / ((PLb + 1) * GROUP_USERS), rem = i % ((PLb + 1) * GROUP_USERS);
            const int gg = blk_u * GROUPS_PER_BLOCK + g4;
            posL[g4 * (PLmax + 1) * GROUP_USERS + rem] = gg < a.n_groups ? a.pos_score[(a.grow[gg] + gg) * GROUP_USERS + rem] : pos_inf_f();
        }
        for (int i = tid; i < GROUPS_PER_BLOCK * (PLmax + 1) * GROUP_USERS; i += THREADS) histL[i] = 0;
    }
    const unsigned pos_addr = (unsigned)(size_t)(__attribute__((address_space(3))) const float *)(posL + gi * (PLmax + 1) * GROUP_USERS + ul);
    float piv_root = pos_inf_f(), piv_lo = pos_inf_f(), piv_hi = pos_inf_f();     // top two tree levels of the lane's user
    if (AUC && jb >= 2) {
        __syncthreads();                                                           // the tables above are complete
        const int r0 = (1 << (jb - 1)) - 1, d = 1 << (jb - 2);
        piv_root = *(LdsF32Ptr)(pos_addr + r0 * 128);
        piv_lo = *(LdsF32Ptr)(pos_addr + (r0 - d) * 128);
        piv_hi = *(LdsF32Ptr)(pos_addr + (r0 + d) * 128);
    }
    const int *pos_item_g = (AUC && group_ok) ? a.pos_item + (a.grow[group] + group) * GROUP_USERS + ul : nullptr;
    // streamed users (more than POS_CHUNK test items; the last slots): no table, no rank counting here -- the lane writes
    // its masked scores to the user's row in HBM and k_rank_streamed counts from there.  Wave-uniform flag: a wave with
    // none of them (all but the last few user blocks) pays one scalar branch per tile.
    const bool stream_lane = slot_ok && slot >= a.stream_slot0;
    const bool wave_streams = wave_any(stream_lane);
    float *stream_row = stream_lane ? a.stream_scores + (size_t)(slot - a.stream_slot0) * (size_t)a.stream_ld : nullptr;
    // tie noise (rm_noise.hpp), exact passes only: the lane's user's row of per-item noise values
    const float *noise_lane = (a.noise_E && user >= 0)
        ? a.noise_E + (size_t)(a.noise_row ? a.noise_row[user] - a.noise_row0 : user) * (size_t)a.noise_ld : nullptr;

    // ---- staging: packed tile (BUF_F4 float4, [g][h][TILE items]) HBM -> LDS by LDS-DMA (global_load_lds_dwordx4): no
    // staging registers, the wave never waits for the bytes before the end-of-step barrier.  One wave-instruction moves
    // 1 KiB (64 lanes x 16 B, two 512-byte runs of the packed tile) to a lane-linear piece of the LDS image, which is
    // laid out [sub][g][h][32 items]: a sub-tile's 32 items are staged by the four waves that read them (one per user
    // group) and by nobody else, so those four are a synchronisation domain of their own (main loop). ----
    // Addressing: the tile's base is scalar (64 bits: the packed image of 10M items is 5 GB), the lane's offset inside the
    // tile is a per-sweep constant in one VGPR per piece, the LDS destination is an integer in M0 -- no 64-bit vector adds,
    // no generic-to-LDS pointer casts (a null check each) per tile.
    const unsigned lds_base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char *)smem;
    constexpr int STAGE_PIECES = (NG + 3) / 4;
    unsigned stage_voff[STAGE_PIECES];
    #pragma unroll
    for (int j = 0; j < STAGE_PIECES; j++) stage_voff[j] = (unsigned)(((gi + 4 * j) * 2 * TILE + h * TILE + sub * 32 + ul) * 16);
    auto stage = [&](int unit, int buf) {                     // unit = tile * NC + chunk: contiguous in the packed image
#ifdef RM_ABL_SAME_TILE
        const char *src = (const char *)a.Bp + (size_t)(unit & 7) * (BUF_F4 * 16);
#else
        const char *src = (const char *)a.Bp + (size_t)unit * (BUF_F4 * 16);
#endif
        // Issued through inline asm on purpose: with the builtin the compiler assumes every later LDS read may alias
        // the DMA's LDS write and puts s_waitcnt vmcnt(0) in front of the MFMA operand reads, which serialises the
        // prefetch with the step it was meant to overlap.  The wait that matters is the explicit one before the
        // end-of-step barrier.
        #pragma unroll
        for (int j = 0; j < STAGE_PIECES; j++) {
            const int g = gi + 4 * j;
            if (NG % 4 == 0 || g < NG) {
                const unsigned m0v = lds_base + (unsigned)((buf * BUF_F4 + (sub * NG + g) * 64) * 16);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             :: "s"(m0v), "v"(stage_voff[j]), "s"(src) : "memory", "m0");
            }
        }
    };

    // ---- MFMA: 32 items (registers) x 32 users (lanes), k in index order ----
    // EARLY ARRIVAL (kernels with resident user factors: one unit per tile): see do_mfma
#if !defined(RM_FULL_BARRIER) && !defined(RM_ABL_NO_BARRIER)
    constexpr bool EARLY_ARRIVE = AF_RESIDENT && NG >= 4 && NG % 2 == 0;      // (an odd group count arrives behind its last matrix instruction)
#else
    constexpr bool EARLY_ARRIVE = false;
#endif
    // (the early arrival sits in front of factor groups NG - 2 and NG - 1 of do_mfma's loop, which advances by two: an odd group count
    // would never arrive and hang the sub-tile barrier)
    static_assert(!EARLY_ARRIVE || NG % 2 == 0, "early arrival needs an even number of factor groups");
    LdsU32Ptr arrive_p = (LdsU32Ptr)(smem + a.sync_off) + sub;
    auto do_mfma = [&](f32x16 &acc, int buf, int chunk, unsigned tile_bits) {
        const float4 *bb = ldsB + buf * BUF_F4 + sub * NG * 64 + h * 32 + ul;      // LDS image [sub][g][h][32 items]
        constexpr int G_STRIDE = 64;
        if (!AF_RESIDENT && !AF_PREFETCH) {
            #pragma unroll
            for (int g = 0; g < NG; g++) af[g] = af_src[(size_t)(chunk * NG + g) * 2 * GROUP_USERS];
        }
        if (chunk == 0) {
            // Dense train rows: a masked item's accumulator STARTS as the all-ones NaN instead of +0 and the matrix
            // instructions carry the NaN through -- masking costs the one v_bfe that replaces the zero, not a v_bfe and a
            // v_or per score after the fact (16 fewer vector instructions per tile; unmasked chains still start at +0).
            if (f_bits) {
                // (`tile_bits` arrives shifted by the lane half already -- see the loads: a shift here lands in the register
                // the allocator has just freed, accumulator 0, and costs sixteen moves to get out of the way again)
                const int mb = (int)tile_bits;
                #pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = __int_as_float(__builtin_amdgcn_sbfe(mb, (r & 3) + 8 * (r >> 2), 1));
            } else {
                #pragma unroll
                for (int r = 0; r < 16; r++) acc[r] = 0.f;
            }
        }
        // streamed factor axis: as soon as the four MFMAs of a factor group are issued its registers are free, and the
        // same group of the NEXT chunk is loaded into them -- a whole MFMA phase ahead of its use, drained by the
        // wait at the arrive point, so the L2 latency of the user factors is never in front of a matrix instruction.
        // (Not with the append-buffer lists: that variant has no registers to spare and the prefetch turns into spills.)
        const int next_chunk = chunk + 1 == NC ? 0 : chunk + 1;
        // The item operands of factor group g + 2 are read from LDS right after the four MFMAs of group g have been
        // issued (two register quads in rotation, the order pinned by scheduling barriers): a read is in flight for the
        // 256 cycles of the other quad's MFMAs and the chain never stalls for an LDS round trip.  (Left to itself the
        // compiler reads, waits, issues four MFMAs, reads again: ~100 idle cycles per 256 busy ones.)
        float4 b0 = bb[0], b1 = NG > 1 ? bb[G_STRIDE] : b0;
        #pragma unroll
        for (int g = 0; g < NG; g += 2) {
            // the operands of the last two factor groups are in registers: this wave is done with the buffer and arrives BEFORE their
            // eight matrix instructions are issued, not after -- the partners' wait ends that much sooner (a trick found in the fp64
            // sweep, where four barriers per tile made it worth 5 %; here +0.6 % at the north-star shape; r3_ab_c2.txt r3zs)
            if (EARLY_ARRIVE && g == NG - 2) {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0x0070);                  // vmcnt(0) lgkmcnt(0): operands landed, and the next tile's DMA share
                if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"((unsigned)(__UINTPTR_TYPE__)arrive_p), "v"(1u) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                const float4 u = af[g];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.x, u.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.y, u.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.z, u.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.w, u.w, acc, 0, 0, 0);
                if (AF_PREFETCH) af[g] = af_src[(size_t)(next_chunk * NG + g) * 2 * GROUP_USERS];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g + 2 < NG) b0 = bb[(g + 2) * G_STRIDE];
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) {
                const float4 u = af[g + 1];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.x, u.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.y, u.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.z, u.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.w, u.w, acc, 0, 0, 0);
                if (AF_PREFETCH) af[g + 1] = af_src[(size_t)(next_chunk * NG + g + 1) * 2 * GROUP_USERS];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g + 3 < NG) b1 = bb[(g + 3) * G_STRIDE];
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- epilogue of one 32-item x 32-user tile ----
    // last key this lane published / observed (lanes that own no list never follow the shared bound: all ones, so that the test
    // below is ONE compare whose lane mask the scalar unit can look at)
#ifdef RM_STATS
    unsigned long long st_sel = 0, st_bar = 0, st_app = 0; unsigned st_nsel = 0;
#endif
    unsigned thr_pub = primary ? 0u : 0xffffffffu;
    // SELECT WHILE WAITING: a selection is ~100 us of memory traffic during which the three other waves of the sub-tile's barrier
    // domain run into the next barrier and wait -- each wave's three selections per item range cost the domain twelve stalls
    // (profiles/r6_ab_c2.txt: 22 % of the wave cycles parked at the barrier, 16 % inside selections).  So a wave that starts a
    // selection says so in LDS (the domain's epoch word), and a wave that finds the epoch moved while it spins at the barrier -- or at
    // the end of its own epilogue -- runs ITS selection right then, provided its buffers are at least half full: the domain stalls
    // once for all four, and every one of them comes back with a tighter bound.
    LdsU32Ptr sel_epoch = (LdsU32Ptr)(smem + a.sync_off) + 4 + sub;      // (words 4..7: the list locks of the LDS lists, unused here)
    unsigned sel_seen = 0u;
    const unsigned lb_half = (unsigned)(lane_cap / 2) << 9;
    // STAGGER: the sub-tile domains fill their buffers at the same pace, so left alone they select at the same tiles and every wave of
    // the block sits in memory stalls together (a third of the wave cycles at BASELINE C2's shape with K = 100).  The FIRST selection
    // of the later sub-tiles comes early -- at 3/4 (and 1/2) of the buffer -- and since the stream position grows by a constant
    // factor from one selection to the next, the domains stay out of phase: while one waits for memory its SIMD partners compute.
    // (never later than the level that keeps a tile's sixteen appends inside the buffer: with fewer than 60 entries per lane 3/4 of the
    // buffer lies beyond it -- found by scratch/fuzz_r6.sh with every k_metrics forced through the lane buffers: a lane ran into the
    // next wave's rows.  Measured and dropped: the first selection as early as it can do anything, at (K + slack) / 2 + 4 entries per
    // lane instead of a full buffer -- a fifth fewer appends, one more selection: 11.1 against 10.9 ms at BASELINE C2's shape with
    // K = 100, 17.1 against 15.9 at K = 256; profiles/r6_ab_c2.txt)
    unsigned lb_trig_now = sub == 0 ? lb_trigger : (sub == 1 ? (unsigned)(lane_cap - lane_cap / 4) << 9 : lb_half);
    lb_trig_now = lb_trig_now < lb_trigger ? lb_trig_now : lb_trigger;
    auto lane_bounds = [&]() {
#ifdef RM_STATS
        const unsigned long long lb_t0 = __builtin_readcyclecounter();
#endif
        // (no entry of the user exceeds the larger of its two lanes' running maxima -- unless the tie noise moved it)
        const float hi_hint = f_noise ? pos_inf_f() : LaneSel<float>::umax(vmax);
        const LaneSelResult<float> sr = lane_select_call<float>((float *)lb_scores + 2 * lane, (int *)lb_scores + 2 * lane + 1, (int)(lb_off >> 9), K, primary, thr, hi_hint, n);
        const float t_new = sr.thr; const unsigned kk = sr.kth_key;
        lb_off = ((unsigned)sr.cnt << 9) | ((unsigned)lane * 8u);
#ifdef RM_STATS
        st_nsel++; st_sel += __builtin_readcyclecounter() - lb_t0;
#endif
        if (kk) {
            thr = t_new;
            if (h == 0 && kk > thr_pub) atomicMax(a.thr_shared + slot, kk);
            thr_pub = kk > thr_pub ? kk : thr_pub;
        }
    };
    auto do_epi = [&](const f32x16 &acc, int tile, unsigned thr_seen, unsigned tile_bits) {
        const int sb = tile * TILE + sub * 32;            // first item of this wave's sub-tile
        float v[16];
        #pragma unroll
        for (int r = 0; r < 16; r++) v[r] = acc[r];
        if (DUMP) {
            if ((n & 3) == 0) {                                 // a lane's four runs of four items, 16 bytes each (the sample seeds' launch)
                #pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int item = sb + mfma32_row(4 * q, h);
                    if (slot_ok && item < n) *(float4 *)(a.dump + (size_t)slot * n + item) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                }
                return;
            }
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int item = sb + mfma32_row(r, h);
                if (slot_ok && item < n) a.dump[(size_t)slot * n + item] = v[r];
            }
            return;
        }
        // (1) train-item / out-of-range masking (reference :491-497) + NaN detection (:517-518).  The NaN scan is
        // skipped when the host proved that no partial sum can overflow or be non-finite (k * max|A| * max|B| bound).
#ifdef RM_ABL_NO_MASK
        const bool slow = false;
#else
        const bool slow = !f_bits && (wave_any(nt < sb + 32) || (sb + 32 > n));
#endif
        if (f_bits) {                                             // masked in the accumulators already (do_mfma)
            if (f_nan) {
                const int mb = (int)tile_bits;
                #pragma unroll
                for (int r = 0; r < 16; r++) nanmask |= __ballot(!__builtin_amdgcn_sbfe(mb, (r & 3) + 8 * (r >> 2), 1) && (v[r] != v[r]));
            }
        } else if (slow) {
            unsigned mbits = 0u;
            // `nt2` was loaded when the previous item was consumed, in an earlier step, and is drained by that step's
            // closing wait: the first consumption of a step (peeled) needs no wait-count.  Only a lane that consumes a
            // second item in the same step waits for its own fresh load (and with it for the tile prefetch).
            auto consume = [&]() {
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
            // the sentinel is all ones: OR-ing the sign-extended mask bit into the score masks it (2 VALU per register)
            // (skipping the registers no lane masks, by a scalar union of the lanes' patterns, was measured 2-4 % SLOWER:
            // the scalar loop and 16 branches are more issue slots than the 25 vector instructions they save)
            const int mb = (int)(mbits >> (4 * h));
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const int mk = __builtin_amdgcn_sbfe(mb, (r & 3) + 8 * (r >> 2), 1);          // 0 or -1
                if (f_nan) nanmask |= __ballot(!mk && (v[r] != v[r]));
                v[r] = __int_as_float(__float_as_int(v[r]) | mk);
            }
        } else if (f_nan) {
            #pragma unroll
            for (int r = 0; r < 16; r++) nanmask |= __ballot(v[r] != v[r]);
        }
        // (2) min / max over candidates (NaN-ignoring, so the sentinel is invisible) (:519-524): v_max3 / v_min3 trees
        // (the maxima of the four register quads are kept: the top-K path below skips a whole quad with one test)
        // (ONE asm statement: the compiler puts an s_nop in front of every instruction that reads a register an asm statement has
        // just written -- it cannot see that the statement holds no instruction with a destination select -- which was one issue
        // slot per level of this tree)
        float qmax[4], tmax;
        asm("v_max_f32 %0, %8, %9\n\tv_max3_f32 %0, %6, %7, %0\n\t"
            "v_max_f32 %1, %12, %13\n\tv_max3_f32 %1, %10, %11, %1\n\t"
            "v_max_f32 %2, %16, %17\n\tv_max3_f32 %2, %14, %15, %2\n\t"
            "v_max_f32 %3, %20, %21\n\tv_max3_f32 %3, %18, %19, %3\n\t"
            "v_max_f32 %4, %2, %3\n\tv_max3_f32 %4, %0, %1, %4\n\t"
            "v_max_f32 %5, %5, %4"
            : "=&v"(qmax[0]), "=&v"(qmax[1]), "=&v"(qmax[2]), "=&v"(qmax[3]), "=&v"(tmax), "+v"(vmax)
            : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
              "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
        // The minimum is only ever compared with the maximum (all candidates equal, :524) and tested for infinity (:522).  When
        // the host has proved every score finite, it is tracked until every lane of the wave has seen two different scores
        // -- from then on "min < max" is settled for good and the eight instructions per tile are skipped (what is reported
        // is the minimum of the tiles seen until then: still below the maximum, which is all k_finalize asks of it).
        if (track_min) {
            const float tmin = hw_min3(hw_min3(v[0], v[1], v[2]), hw_min3(v[3], v[4], v[5]), hw_min3(v[6], v[7], v[8]));
            const float tmin2 = hw_min3(hw_min3(v[9], v[10], v[11]), hw_min3(v[12], v[13], v[14]), v[15]);
            vmin = hw_min3(vmin, tmin, tmin2);
            if (!f_nan) track_min = wave_any(slot_ok && !(vmax > vmin));
        }
        // tie noise (reference :531-534: added AFTER the validity scan, in real_t): wave-uniform branch, exact passes only
        if (f_noise) {
            if (noise_lane) {
                #pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const float4 e = *(const float4 *)(noise_lane + sb + 8 * q4 + 4 * h);
                    v[4 * q4] += e.x; v[4 * q4 + 1] += e.y; v[4 * q4 + 2] += e.z; v[4 * q4 + 3] += e.w;
                }
            }
            #pragma unroll
            for (int q4 = 0; q4 < 4; q4++) qmax[q4] = hw_max3(v[4 * q4], v[4 * q4 + 1], hw_max(v[4 * q4 + 2], v[4 * q4 + 3]));
            tmax = hw_max3(qmax[0], qmax[1], hw_max(qmax[2], qmax[3]));
        }
#ifndef RM_ABL_NO_TOPK
        // (3) streaming top-K: anything at or above the user's current K-th best is offered to the list (:537-540).
        // One compare per tile on the lane's tile maximum; the per-score work happens only in the rare hit path.
        // Lanes u (h = 0) and u + 32 (h = 1) carry two item rows of the same user: the h = 0 lane owns the list and
        // also takes its partner's candidate.
        // every partial list of the user (other sub-tile wave, other item splits) publishes its K-th best; the largest
        // of them is a valid lower bound of the final K-th best, so it filters for all of them
        // (behind a wave-level test: the shared bound moves a few dozen times per sweep, the seven vector instructions of the update
        // were paid on every tile)
        if (wave_any(thr_seen > thr_pub)) {
            asm volatile("" ::: "memory");                     // (a real branch: the compiler would turn the block into selects again)
            if (thr_seen > thr_pub) { thr_pub = thr_seen; const float t = ord_unkey(thr_seen); thr = t > thr ? t : thr; }
        }
        // (a.ext_topk: k_metrics beyond the lists' reach -- every lane streams its scores and k_select_topk picks the top-K)
        const unsigned long long cm = f_ext ? 0ull : __ballot(tmax >= thr);
        RM_STAT(0, 1); RM_STAT(1, cm != 0); RM_STAT(2, __popcll(cm));
#ifdef RM_STATS
        const unsigned long long ap_t0 = __builtin_readcyclecounter();
#endif
        if (buffered) {
            // K > 32: every lane appends its own candidates to its own buffer (rm_list.hpp): per score register a compare, one
            // 8-byte store under the lane mask, an add -- a register quad without a candidate in any lane is skipped with one test
            if (cm) {
                const int sbh = sb + 4 * h;
                #pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    if (!__ballot(qmax[qd] >= thr)) continue;
                    #pragma unroll
                    for (int r = 4 * qd; r < 4 * qd + 4; r++) {
                        if (v[r] >= thr) {
                            u32x2 e;
                            e.x = __float_as_uint(v[r]); e.y = (unsigned)(sbh + (r & 3) + 8 * (r >> 2));
                            // (s_nop 4: under register pressure the base pair is an SGPR spill, restored by v_readlane right in front of
                            // this statement -- a VALU write of an SGPR needs five wait states before a memory instruction reads it, and
                            // the compiler's hazard recognizer does not look inside inline asm: without them the store went to a stale
                            // base, a memory fault in exactly the kernels that spill the pair -- 24-40 factors, fp64 beyond 128)
                            asm volatile("s_nop 4\n\tglobal_store_dwordx2 %0, %1, %2" :: "v"(lb_off), "v"(e), "s"(lb_scores) : "memory");
                            lb_off += 512u;
                        }
                    }
                }
            }
#ifdef RM_STATS
            st_app += __builtin_readcyclecounter() - ap_t0;
#endif
        } else
        if (cm && pend_cap) {
            unsigned ov = 0;                                    // score registers that did not fit the lane's buffer
            #pragma unroll
            for (int qd = 0; qd < 4; qd++) {
                if (!__ballot(qmax[qd] >= thr)) continue;         // no lane has a candidate among these four registers
                #pragma unroll
                for (int r = 4 * qd; r < 4 * qd + 4; r++) {
                    // (the lane test alone: wrapped in a wave-level test it compiled into five scalar instructions in front of the
                    // same exec-mask branch)
                    const bool c = v[r] >= thr;
                    RM_STAT(3, __ballot(c) != 0); RM_STAT(4, __popcll(__ballot(c)));
                    if (c) {
                        if (pcnt < pend_cap) { Pp[pcnt * WAVE] = pack_key(v[r], sb + mfma32_row(r, h)); pcnt++; }
                        else ov |= 1u << r;
                    }
                }
            }
            bool more = wave_any(ov != 0);
            if (more || wave_any(pcnt >= pend_cap - 1)) merge_pending();
            while (more) {                                      // warm-up only: more candidates in one tile than a buffer holds
                unsigned ov2 = 0;
                #pragma unroll
                for (int r = 0; r < 16; r++) {
                    const bool c = ((ov >> r) & 1u) && v[r] >= thr;
                    if (__ballot(c)) {
                        if (c) {
                            if (pcnt < pend_cap) { Pp[pcnt * WAVE] = pack_key(v[r], sb + mfma32_row(r, h)); pcnt++; }
                            else ov2 |= 1u << r;
                        }
                    }
                }
                ov = ov2; more = wave_any(ov != 0);
                merge_pending();
            }
        } else if (cm) {
            list_acquire();
            if (LLDS) ws = (wkey >> 32) ? ord_unkey((unsigned)(wkey >> 32)) : neg_inf_f();
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned long long hitm = __ballot(v[r] >= thr);
                if (hitm) {
                    RM_STAT(3, 1); RM_STAT(4, __popcll(hitm));
                    // the partner's score is only fetched when a lane of the upper half has a candidate
                    const float other = (hitm >> 32) ? __shfl_xor(v[r], 32) : nan_sentinel_f();
                    if (h == 0 && primary) {
                        const int item0 = sb + mfma32_row(r, 0), item1 = sb + mfma32_row(r, 1);
                        if (LLDS) { if (v[r] >= ws) keylist_offer<GROUP_USERS>(Ll, K, pack_key(v[r], item0), wkey, wpos);
                                    if (other >= ws) keylist_offer<GROUP_USERS>(Ll, K, pack_key(other, item1), wkey, wpos);
                                    ws = (wkey >> 32) ? ord_unkey((unsigned)(wkey >> 32)) : neg_inf_f(); }
                        else      { if (v[r] >= ws) list_offer<float, GROUP_USERS>(Lr, K, v[r], item0, ws, widx, wpos);
                                    if (other >= ws) list_offer<float, GROUP_USERS>(Lr, K, other, item1, ws, widx, wpos); }
                    }
                }
            }
            list_release();
        }
        if (!buffered && cm && (!pend_cap || merged)) {         // with pending buffers the K-th best only moves in a merge
            merged = false;
            const float t2 = __shfl(ws, ul);
            if (primary) {
                thr = t2 > thr ? t2 : thr;
                const unsigned kk = ord_key(t2);
                if (h == 0 && kk > thr_pub) { atomicMax(a.thr_shared + slot, kk); }
                thr_pub = kk > thr_pub ? kk : thr_pub;
            }
        }
#endif
#ifndef RM_ABL_NO_AUC
        // (4) AUC rank counting (replaces the full sort of :552 + the walk of :795-865)
        if (wave_streams) {
            if (stream_lane) {                                  // registers 4q .. 4q+3 are four consecutive items
                #pragma unroll
                for (int q4 = 0; q4 < 4; q4++)
                    *(float4 *)(stream_row + sb + 8 * q4 + 4 * h) = make_float4(v[4 * q4], v[4 * q4 + 1], v[4 * q4 + 2], v[4 * q4 + 3]);
            }
        }
        if (AUC) {
            switch (jb) {
                case 1: auc_pass<1>(v, pos_addr, pos_item_g, sb, h, piv_root, piv_lo, piv_hi); break;
                case 2: auc_pass<2>(v, pos_addr, pos_item_g, sb, h, piv_root, piv_lo, piv_hi); break;
                case 3: auc_pass<3>(v, pos_addr, pos_item_g, sb, h, piv_root, piv_lo, piv_hi); break;
                case 4: auc_pass<4>(v, pos_addr, pos_item_g, sb, h, piv_root, piv_lo, piv_hi); break;
                case 5: auc_pass<5>(v, pos_addr, pos_item_g, sb, h, piv_root, piv_lo, piv_hi); break;
                case 6: auc_pass<6>(v, pos_addr, pos_item_g, sb, h, piv_root, piv_lo, piv_hi); break;
                default: break;
            }
        }
#endif
        // K > 32: some lane is a tile away from a full buffer -> every user of the wave raises its bound (at the END of the
        // epilogue: the tile's scores are dead, their registers hold the selection's loads in flight)
        if (buffered) {
            const bool own = wave_any(lb_off >= lb_trig_now);
            unsigned ep = __builtin_amdgcn_readfirstlane(*sel_epoch);
            if (own || (ep != sel_seen && wave_any(lb_off >= lb_half))) {
                if (own && ep == sel_seen) {                      // (nobody has asked yet: ask)
                    if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"((unsigned)(__UINTPTR_TYPE__)sel_epoch), "v"(1u) : "memory");
                    ep++;
                }
                lane_bounds();
                lb_trig_now = lb_trigger;
            }
            sel_seen = ep;
        }
    };

    // ---- main loop: one barrier per tile.  Measured on gfx950 (scratch/coexec2.hip): an f32-input MFMA chain and
    // VALU/LDS work of the SIMD partner wave do NOT overlap (time = sum, the f32 MFMA runs on the vector ALUs), so the
    // two waves of a SIMD run in phase: both chains back to back, then both epilogues sharing the VALU at full rate. ----
    f32x16 acc;
    const int nunits = ntiles * NC;
    // Synchronisation is a SPLIT barrier on an LDS counter, not s_barrier, and its domain is the FOUR waves that share a
    // 32-item sub-tile (one per user group / SIMD), not the block: a wave "arrives" for unit u once it has issued its last
    // MFMA of the unit (its LDS reads of that buffer are done) and its share of the next unit's DMA has landed; it only
    // "waits" -- for the 4 arrivals of unit u -- right before it touches the buffers again at the start of unit u + 1.
    // The whole epilogue sits between the two, so a wave whose epilogue runs long (train-mask walk, top-K merge, tie path)
    // delays its three partners only when it is a full epilogue behind, and the other sub-tiles' waves (its SIMD
    // neighbours) never: they drift apart freely and fill the vector pipe while it waits (12 -> 4 waves per domain:
    // -3.8 % at C2, gpurun_out r4a).  Arrivals of unit u + 1 cannot start before all of unit u are in, so one monotonic
    // counter per sub-tile is unambiguous: all arrived for unit u  <=>  counter >= 4 (u + 1).
    constexpr unsigned SYNC_WAVES = 4;
    LdsU32Ptr arrive = (LdsU32Ptr)(smem + a.sync_off) + sub;            // (words 4..7 of the area: the groups' list locks)
    if (tid < (buffered ? 8 : 4)) ((LdsU32Ptr)(smem + a.sync_off))[tid] = 0u;      // (+ the domains' selection epochs, words 4..7)
    if (ntiles > 0) stage(t0 * NC, 0);
    // first tile's word of the dense train row, shifted so that bit (r & 3) + 8 (r >> 2) is accumulator r's item for this lane half
    unsigned tile_bits = (tb_row && ntiles > 0) ? a.train_bits[tb_idx] >> (4 * h) : 0u;
    // vmcnt(0) through the builtin (not asm) so that the compiler's own wait-count bookkeeping sees the drain: every
    // load it issued before this point is known complete and needs no further wait inside the loop.
    __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);
    __syncthreads();
    // the shared K-th-best bound is read one tile ahead: its load is drained by the wait after the next MFMA phase,
    // never by a wait in the middle of an epilogue (which would also wait for the tile prefetch)
    auto load_thr = [&]() -> unsigned {
#ifdef RM_ABL_NO_THRSEEN
        return 0u;
#else
        return (primary && !DUMP) ? __hip_atomic_load(a.thr_shared + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#endif
    };
    unsigned thr_seen = load_thr();                             // (the seeded bound counts from the first tile on)
#ifdef RM_STATS
    const unsigned long long prof_t1 = __builtin_readcyclecounter();
#endif
    for (int i = 0; i < ntiles; i++) {
        unsigned thr_next = thr_seen;
        for (int c = 0; c < NC; c++) {
            const int unit = i * NC + c;
#if !defined(RM_FULL_BARRIER) && !defined(RM_ABL_NO_BARRIER)
            if (unit > 0) {                                                       // wait half of the split barrier
                const unsigned target = SYNC_WAVES * (unsigned)unit;
#ifdef RM_STATS
                const unsigned long long bw_t0 = __builtin_readcyclecounter();
#endif
                while (__hip_atomic_load(arrive, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (buffered && !DUMP) {                        // a partner is selecting: this wave's turn too (see lane_bounds)
                        const unsigned ep = __builtin_amdgcn_readfirstlane(*sel_epoch);
                        if (ep != sel_seen) { sel_seen = ep; if (wave_any(lb_off >= lb_half)) { lane_bounds(); lb_trig_now = lb_trigger; } }
                    }
                }
#ifdef RM_STATS
                st_bar += __builtin_readcyclecounter() - bw_t0;
#endif
            }
#ifndef RM_ABL_NO_PRIO
            // The sub-tile domains progress independently, and the instruction arbiter serves the oldest wave first: left
            // alone, sub-tile 0 finishes its range ~20 % ahead of sub-tile 2 and the block ends on one wave per SIMD.
            // Every fourth unit a wave compares its domain's arrival counter with the others' and sets its issue priority:
            // behind -> high, ahead -> low (more than half a unit apart), so that all domains reach the end together.
            if ((unit & 3) == 0) {
                const u32x4 cnt4 = *(__attribute__((address_space(3))) const u32x4 *)(smem + a.sync_off);
                const unsigned c0 = __builtin_amdgcn_readfirstlane(cnt4.x), c1 = __builtin_amdgcn_readfirstlane(cnt4.y), c2 = __builtin_amdgcn_readfirstlane(cnt4.z);
                const int sub_s = __builtin_amdgcn_readfirstlane(sub);                      // (wave-uniform, and known to be to the compiler)
                const unsigned mine = sub_s == 0 ? c0 : (sub_s == 1 ? c1 : c2);
                const unsigned o1 = sub_s == 0 ? c1 : c0, o2 = (NSUB == 3) ? (sub_s == 2 ? c1 : c2) : o1;
                const bool behind = mine + 2 < o1 || mine + 2 < o2, ahead = mine > o1 + 2 && mine > o2 + 2;
                if (behind) __builtin_amdgcn_s_setprio(2);
                else if (ahead) __builtin_amdgcn_s_setprio(0);
                else __builtin_amdgcn_s_setprio(1);
            }
#endif
#endif
#ifndef RM_ABL_NO_STAGE
            if (unit + 1 < nunits) stage(t0 * NC + unit + 1, (unit + 1) & 1);   // that buffer was last read one unit ago
#endif
            if (c == 0) thr_next = load_thr();
            // the NEXT tile's word of the dense train row (the accumulators start from it): in flight during the MFMA phase,
            // drained by the wait at the arrive point
            unsigned bits_next = 0u;
            if (c == NC - 1 && tb_row && i + 1 < ntiles) { tb_idx += NSUB; bits_next = a.train_bits[tb_idx]; }      // (shifted once it has landed, below)
#ifndef RM_ABL_NO_MFMA
            do_mfma(acc, unit & 1, c, tile_bits);
#endif
#if !defined(RM_FULL_BARRIER) && !defined(RM_ABL_NO_BARRIER)
#ifndef RM_ABL_NO_ARRIVE_WAIT
            __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);                              // arrive half: the next unit's DMA share has landed
#endif
            // (a bare ds_add: the builtin goes through the compiler's wave-aggregation of atomics, a dozen instructions per tile;
            // the wave's LDS operations are issued in order, so the arrival cannot overtake its operand reads)
            if (!EARLY_ARRIVE && lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"((unsigned)(__UINTPTR_TYPE__)arrive), "v"(1u) : "memory");
#endif
#ifndef RM_ABL_NO_EPI
            if (c == NC - 1) do_epi(acc, t0 + i, thr_seen, tile_bits);
#endif
            if (c == NC - 1) tile_bits = bits_next >> (4 * h);
#if defined(RM_FULL_BARRIER)
            __builtin_amdgcn_s_waitcnt(WAIT_VMCNT0);                              // the DMA of the next unit has landed
            __syncthreads();
#endif
        }
        thr_seen = thr_next;
    }
    if (DUMP) return;
#ifdef RM_STATS
    const unsigned long long prof_t2 = __builtin_readcyclecounter();
#endif
    if (pend_cap) merge_pending();
#ifdef RM_STATS
    if (lane == 0 && gi == 0 && sub == 1) atomicAdd(&g_stats[14], prof_t2 - prof_t1);
    if (lane == 0 && gi == 0 && sub == 2) atomicAdd(&g_stats[15], prof_t2 - prof_t1);
    if (lane == 0 && gi == 3 && sub == 0) atomicAdd(&g_stats[8], prof_t2 - prof_t1);
#endif
    if (LLDS) __syncthreads();                                  // every wave of the group has merged into the shared list
#ifdef RM_STATS
    if (threadIdx.x == 0) atomicAdd(&g_stats[9], __builtin_readcyclecounter() - prof_t2);
#endif

    // ---- write this wave's partial: top-K list, validity stats, AUC sum; flush the LDS histogram ----
    // (the kernel arguments of this part -- six pointers and a handful of counts -- are read AGAIN, from the kernarg segment: loaded at
    // the top they stay in SGPRs across the whole tile loop of a kernel that spills scalars; rm_device.hpp late_kernargs)
    {
    const SweepArgs &a = *late_kernargs<SweepArgs>();
    const int n_part = a.part_splits * NSUB + a.part_extra;
    const int part = split * NSUB + sub;
    {   // lanes u and u+32 hold two halves of the same user's stats
        const float omax = __shfl_xor(vmax, 32), omin = __shfl_xor(vmin, 32);
        vmax = __builtin_fmaxf(vmax, omax); vmin = __builtin_fminf(vmin, omin);
        const bool hn = ((nanmask >> ul) & 1ull) | ((nanmask >> (ul + 32)) & 1ull);
        if (slot_ok && h == 0) {
            PartialStat<float> ps;
            ps.vmax = vmax; ps.vmin = vmin; ps.rocsum = 0; ps.has_nan = hn ? 1 : 0; ps.pad = 0;
            a.pst[(size_t)slot * n_part + part] = ps;
            ListEntry *dst = a.pl + ((size_t)slot * n_part + part) * K;
            if (LLDS && sub == 0) { keylist_sort_desc<GROUP_USERS>(Ll, K); for (int i = 0; i < K; i++) unpack_key(Ll[i * GROUP_USERS], dst[i].s, dst[i].idx); }
            // (an EMPTY part is marked by its first entry alone -- k_finalize looks at nothing else of it, rm_finalize.hpp -- so the
            // parts that hold nothing cost one scattered 8-byte store per lane instead of K)
            if (LLDS && sub != 0) { dst[0].s = neg_inf_f(); dst[0].idx = IDX_EMPTY; }      // the group's list is written once
            // a user block cut into fewer ranges than the arrays are laid out for: its first block fills in the missing parts
            if (split == 0) for (int sp = nsplit; sp < a.part_splits; sp++) {
                ps.vmax = neg_inf_f(); ps.vmin = pos_inf_f(); ps.has_nan = 0;
                a.pst[(size_t)slot * n_part + sp * NSUB + sub] = ps;
                ListEntry *de = a.pl + ((size_t)slot * n_part + sp * NSUB + sub) * K;
                if (a.pl) { de[0].s = neg_inf_f(); de[0].idx = IDX_EMPTY; }
            }
        }
    }
    // The users' own TEST items, which a sweep over dense rows that mark them never saw (a.part_extra; rm_lib.hip `mask_test`), come
    // back here for the table users -- once per user, in the block of its first item range, by the lane that owns the user: the
    // group's table of sorted positives is in LDS, row i (ascending (score, item desc) order) outranks exactly the positives of the
    // rows below it and so counts in bin i like any candidate with i positives below it; the rows walked downwards are the test items
    // in (score desc, item asc) order, of which the first K are the extra part of the user's top-K lists and the first / last
    // the maximum / minimum of the part's validity statistics.  +inf rows (padding, or a test item that is also a train item) are no
    // candidates.  (Streamed users: k_merge_positives, rm_finalize.hpp.  This was that kernel's job for everybody: 0.25 ms per
    // step at BASELINE C2, one wavefront per user re-reading from HBM what the block holds in LDS.)
    if (AUC && a.part_extra && split == 0 && sub == 0 && h == 0 && slot_ok && slot < a.stream_slot0) {
        const int pe = n_part - 1;
        ListEntry *dx = a.pl + ((size_t)slot * n_part + pe) * K;
        const unsigned hcol = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)(histL + gi * (PLmax + 1) * GROUP_USERS + ul);
        float bmax = neg_inf_f(), bmin = pos_inf_f();
        int filled = 0;
        for (int i = PLb - 1; i >= 0; i--) {
            const float x = *(LdsF32Ptr)(pos_addr + (unsigned)i * 128u);
            if (x < pos_inf_f()) {
                if (filled == 0) bmax = x;
                bmin = x;
                asm volatile("ds_add_u32 %0, %1" :: "v"(hcol + (unsigned)i * 128u), "v"(1u) : "memory");
                if (filled < K) { dx[filled].s = x; dx[filled].idx = pos_item_g[i * GROUP_USERS]; }
                filled++;
            }
        }
        for (int i = filled; i < K; i++) { dx[i].s = neg_inf_f(); dx[i].idx = IDX_EMPTY; }
        PartialStat<float> ps;
        ps.vmax = bmax; ps.vmin = bmin; ps.rocsum = 0; ps.has_nan = 0; ps.pad = 0;
        a.pst[(size_t)slot * n_part + pe] = ps;
    }
    if (!LLDS && !buffered) {
        if (slot_ok && h == 0) {
            ListEntry *dst = a.pl + ((size_t)slot * n_part + part) * K;
            list_sort_desc<float, GROUP_USERS>(Lr, K);
            for (int i = 0; i < K; i++) ListRaw<float>::unpack(Lr[i * GROUP_USERS], dst[i].s, dst[i].idx);
        }
    } else if (!LLDS && !a.ext_topk) {
        // the lanes' entries stay where they are: k_collect_topk (rm_finalize.hpp) reads the buffers of all item ranges and sub-tile
        // waves of a user and writes its ordered top-K
        a.lane_cnt[((size_t)blockIdx.x * NWAVES + wave) * WAVE + lane] = (slot_ok && primary) ? (int)(lb_off >> 9) : 0;
    }
    if (AUC) {
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the asm histogram atomics are invisible to the compiler
        __syncthreads();
        for (int i = tid; i < GROUPS_PER_BLOCK * (PLb + 1) * GROUP_USERS; i += THREADS) {
            const int g4 = i / ((PLb + 1) * GROUP_USERS), rem = i % ((PLb + 1) * GROUP_USERS);
            const int gg = blk_u * GROUPS_PER_BLOCK + g4;
            const unsigned c = histL[g4 * (PLmax + 1) * GROUP_USERS + rem];
            if (gg < a.n_groups && c) atomicAdd(&a.hist[(a.grow[gg] + gg) * GROUP_USERS + rem], c);
        }
    }
    }
#ifdef RM_STATS
    if (lane == 0) { atomicAdd(&g_stats[5], (unsigned long long)st_nsel); atomicAdd(&g_stats[6], st_sel); atomicAdd(&g_stats[7], st_bar); atomicAdd(&g_stats[4], st_app); }
    if (threadIdx.x == 0) {
        const unsigned long long prof_t3 = __builtin_readcyclecounter();
        atomicAdd(&g_stats[10], prof_t1 - prof_t0); atomicAdd(&g_stats[11], prof_t2 - prof_t1); atomicAdd(&g_stats[12], prof_t3 - prof_t2);
        atomicAdd(&g_stats[13], (unsigned long long)ntiles);
    }
#endif
}

} // namespace rm
