// rm_launch.hpp -- argument blocks of the two sweep kernels and the launch entry points of their translation units
// (the sweeps are compiled in separate .hip files so that the build parallelises).
#pragma once
#include "rm_device.hpp"
#include "rm_list.hpp"

namespace rm {

typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int GROUP_USERS64 = 16;      // users on the lanes of one wavefront in the fp64 sweep (MFMA 16x16x4: D column = lane & 15)

struct SweepArgs {
    int n, K;
    int ngt;                              // factor groups of 8 when the kernel takes them at run time (more than 512 factors)
    int ext_topk;                         // 1 = no top-K lists here: every lane streams its scores, k_select_topk picks the top-K
    int spec;                             // specialisation of the epilogue's switches (k_sweep SPEC): 0 read them here, 1 / 2 the usual cases
    int n_slots, n_groups, n_ublocks;     // n_ublocks = user blocks of THIS launch (ceil(n_groups / 4) when there is one)
    int ublock0;                          // first user block of this launch (depth-split calls launch twice)
    int n_splits, tiles_total;            // item splits of the main part of the grid
    // Two-level grid: the LAST `tail_ublocks` user blocks in launch order (the cheapest) are cut into `tail_splits` item
    // ranges instead of n_splits -- small blocks that fill the last, partial round of the grid.  part_splits =
    // max(n_splits, tail_splits) is the split dimension of pl / pst (grid = (n_ublocks - tail_ublocks) * n_splits +
    // tail_ublocks * tail_splits)
    int tail_ublocks, tail_splits, part_splits;
    int part_extra;                       // parts of pl / pst behind the sweep's own (1 = the user's test items, k_merge_positives)
    int jmax;                             // LDS sizing (all blocks)
    int list_in_lds;
    int check_nan;                        // 0 when the host proved all scores finite (skips the NaN scan)
    int buffered_lists;                   // HBM lists: 1 = append buffer + compaction (large K), 0 = replace-the-minimum
    int pend_cap, pend_off;               // LDS lists: per-lane pending buffers of pend_cap keys at byte offset pend_off (0 = none)
    int sync_off;                         // byte offset of the split-barrier counter in LDS
    const float4 *Ap, *Bp;
    const int *slot_user, *slot_chunk;
    const int *train_p, *train_i;
    // optional dense form of the train rows (small item counts): bit i of train_bits[user * train_words + (item >> 5)] = item is
    // a train item of the user OR lies beyond n; one load per lane and tile replaces the cursor walk over the CSR row
    const unsigned *train_bits; int train_words;
    const int *gj; const long long *grow;
    const float *pos_score;               // [(total_rows + n_groups)][32]  sorted positives, +inf padded (2^j rows per group)
    const int *pos_item;                  // same shape, item ids (read only when a candidate ties a positive's score)
    unsigned *hist;                       // [(total_rows + n_groups)][32]
    unsigned *thr_shared;                 // [n_slots] order-preserving key of the best K-th-best any partial of the user has seen
    u32x2 *glists;                        // list scratch in HBM when the lists do not fit LDS: K <= 32 [block][wave][K][32]; K > 32 the waves' lane
                                          // buffers [block][wave]{[lane_cap][64] scores, [lane_cap][64] item ids} (rm_list.hpp)
    int lane_cap; int *lane_cnt;          // K > 32: entries per lane buffer (a multiple of 16); [block][wave][64] entries left at the end of the sweep
    ListEntry *pl;                        // partial lists [slot][n_part][K]
    PartialStat<float> *pst;              // [slot][n_part]
    float *dump;                          // DUMP mode: dense [n_slots][n] scores
    // streamed users = slots [stream_slot0, n_slots): their masked candidate scores go to stream_scores[slot - stream_slot0][item]
    int stream_slot0;
    long long stream_ld;                  // row stride in elements (= tiles_total * items per tile)
    float *stream_scores;
    // tie noise (rm_noise.hpp): noise_E[row of the lane's user][item] is added to every score after the validity scan
    const int *noise_row; int noise_row0; const float *noise_E; long long noise_ld;
};

struct Sweep64Args {
    int n, K;
    int ngt;                              // factor groups of 8 when the kernel takes them at run time (more than 512 factors)
    int ext_topk;                         // 1 = no top-K lists here: every lane streams its scores, k_select_topk picks the top-K
    int spec;                             // specialisation of the epilogue's switches (k_sweep64 SPEC): 0 read them here, 1 the usual case
    int n_slots, n_groups, n_ublocks;     // n_ublocks = user blocks of THIS launch
    int ublock0;                          // first user block of this launch (depth-split calls launch twice)
    int n_splits, tiles_total;
    int tail_ublocks, tail_splits, part_splits;       // as in SweepArgs
    int jmax;
    int check_nan;
    int buffered_lists;
    int sync_off;                         // byte offset of the split-barrier counter in LDS
    int pend_cap, pend_off;               // per-lane pending buffers (scores [cap][8 waves][64] then items), 0 = none
    const f64x2 *Ap, *Bp;
    const int *slot_user, *slot_chunk;
    const int *train_p, *train_i;
    const int *gj; const long long *grow;
    const double *pos_score;              // [(total_rows + n_groups)][16]
    const int *pos_item;
    unsigned *hist;                       // [(total_rows + n_groups)][16]
    unsigned long long *thr_shared;       // [n_slots] (64-bit keys)
    u32x4 *glists;                        // K <= 32: [block][wave][K][16]; K > 32: lane buffers [block][wave]{[lane_cap][64] scores, [lane_cap][64] item ids}
    int lane_cap; int *lane_cnt;          // as in SweepArgs
    Entry<double> *pl;                    // [slot][n_part][K]
    PartialStat<double> *pst;             // [slot][n_part]
    double *dump;
    int stream_slot0;                     // streamed users: see SweepArgs
    long long stream_ld;
    double *stream_scores;
    const int *noise_row; int noise_row0; const double *noise_E; long long noise_ld;     // tie noise: see SweepArgs
};

// return 0 = launched, -1 = unsupported factor-group count, otherwise a hipError_t
int launch_sweep32(bool auc, bool dump, int lmode, int nsub, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_hbm_s0(bool auc, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_hbm_s1(bool auc, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_hbm_s2(bool auc, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_lds_s0(bool auc, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_lds_s1(bool auc, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_lds_s2(bool auc, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_n3_s0(bool auc, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_n3_s1(bool auc, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_n3_s2(bool auc, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep32_large(bool auc, bool dump, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa);
int launch_sweep64(bool auc, bool dump, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa);
int launch_sweep64_small_s0(bool auc, bool dump, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa);
int launch_sweep64_small_s1(bool auc, bool dump, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa);
int launch_sweep64_large_s0(bool auc, bool dump, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa);
int launch_sweep64_large_s1(bool auc, bool dump, int lmode, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa);

} // namespace rm
