// rm_device.hpp -- shared device-side definitions for the gfx950 (MI355X / CDNA4) hot path.
//
// Path: per-user ranking metrics of recometrics' calc_metrics<real_t>
// (reference src/recometrics.hpp:359-965).  Terminology follows the reference's domain:
// users, items, factors, train/test CSR rows, candidates, top-K, positives.
//
// Device data model (see DESIGN.md for the full picture)
//   slot      one lane-owner in the sweep: (user, chunk c of <= 63 of its sorted positives).  A user with P
//             positives owns max(1, ceil(P/63)) slots; slot 0 ("primary") also owns the user's top-K/validity state.
//   group     32 consecutive slots = the users one wavefront carries on its lanes (users-on-lanes MFMA orientation).
//   tile      64 consecutive items; a packed item tile is the exact LDS image the MFMA loop reads.
//   partial   one wavefront's private result for a slot (its 32-item half of every tile of its item split).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rm {

constexpr int WAVE = 64;
constexpr int GROUP_USERS = 32;        // users on the lanes of one wavefront (fp32 MFMA 32x32x2: D column = lane & 31)
constexpr int TILE_ITEMS = 64;         // items per packed tile (two 32-item MFMA sub-tiles)
constexpr int SWEEP_THREADS = 512;     // 8 wavefronts: 4 groups x 2 sub-tiles
constexpr int GROUPS_PER_BLOCK = 4;
constexpr int POS_CHUNK = 63;          // positives per slot = rows of a complete binary search tree of depth 6
constexpr int MAX_J = 6;
// Depth class MAX_J + 1 = "streamed" users: users with more than POS_CHUNK test items whose candidate scores the sweep
// writes to HBM instead of counting ranks in LDS; their ranks come from k_rank_streamed (rm_finalize.hpp).  One sweep
// slot per such user whatever the length of its test row -- no lane recomputes a contraction for an extra chunk.
constexpr int STREAM_CLASS = MAX_J + 1;
constexpr int N_CLASSES = MAX_J + 2;
constexpr int IDX_EMPTY = 0x7fffffff;
constexpr int HEAVY_NPOS = 256;        // test rows longer than this get wave-per-user treatment where one thread would crawl
// where a sweep wave's top-K lists live: LM_LDS replace-the-minimum lists in LDS; LM_HBM the same scheme in HBM (lists that
// do not fit LDS, K <= 32); LM_HBM_APPEND per-user append buffers in HBM with wave-cooperative compaction (K > 32)
enum : int { LM_LDS = 0, LM_HBM = 1, LM_HBM_APPEND = 2 };
// s_waitcnt immediate on gfx9/CDNA: vmcnt = bits [3:0] and [15:14], expcnt = [6:4], lgkmcnt = [11:8]; this one waits for
// vmcnt == 0 and leaves the other two counters alone
constexpr int WAIT_VMCNT0 = 0x0F70;

// per-user flags written by k_classify
// UF_SKIP: the user is not part of this pass at all (second pass of a noise call: only the flagged users are re-evaluated);
// nothing of it is read or written
enum : int { UF_NAN = 1, UF_ONLY_NDCG = 2, UF_KLEQN = 4, UF_ACTIVE = 8, UF_SKIP = 16 };

// which outputs the caller asked for (NULL pointer == not requested, reference recometrics.hpp:370-379)
enum : int { RQ_P = 1, RQ_TP = 2, RQ_R = 4, RQ_AP = 8, RQ_TAP = 16, RQ_NDCG = 32, RQ_HIT = 64, RQ_RR = 128, RQ_ROC = 256, RQ_PR = 512 };

template <class T> struct Entry { T s; int idx; };     // (score, item) -- top-K list element
typedef Entry<float> ListEntry;

template <class T> struct PartialStat {          // one wavefront's validity / AUC state for one slot
    T vmax, vmin;
    unsigned long long rocsum;                    // unused (the rank sums come from the histograms, k_auc_slots); keeps the layout
    int has_nan; int pad;
};

struct Plan {                                     // produced on device, read back once by the host
    int n_active;      // users that need scoring
    int n_slots;
    int n_groups;
    int jmax;          // deepest positive tree over all groups (0 when no AUC is requested)
    long long total_rows;   // sum over groups of (2^j - 1)
    int class_count[N_CLASSES];
    int class_offset[N_CLASSES + 1];
    int class_cursor[N_CLASSES];
    int n_long;             // users with more than POS_CHUNK test items (k_count_long: an upper bound of the streamed class)
    int stream_enable;      // 1 = those users are streamed (their score rows fit the HBM budget), 0 = they get one slot per chunk
    int max_npos;           // longest test row (k_count_long)
    int n_stream_chunks;    // chunks of POS_CHUNK test entries over all streamed users (work items of the positives kernels)
    int nonfinite;                      // some factor of A is NaN / Inf
    int nonfinite_b;                    // some factor of B is NaN / Inf
    int n_noise_flagged;                // fp32 + noise, first pass: users whose ranking the noise can change (rm_noise.hpp)
    int n_heavy;                        // evaluated users with more than HEAVY_NPOS test items (listed by k_classify)
    int n_only_ndcg;                    // evaluated users whose train and test rows cover every item (no tables of positives)
    unsigned long long amax_a, amax_b;  // bit patterns of max|A|, max|B| as doubles (non-negative doubles order like u64)
    int csr_bad;                        // CSR_* bits: what k_check_csr_ptr / k_check_csr_rows found wrong with the caller's CSR arrays
    int csr_where;                      // a user (row) that shows the defect, for the message
    // sortedness, counted: descents (idx[e] > idx[e + 1]) over the whole index array, and those that sit on a row boundary --
    // the rows are sorted exactly when the two agree ([0] train, [1] test; k_check_csr_flat / k_check_csr_starts)
    // (a cache line of their own: atomics on one line serialise in L2, and the validation kernels run BESIDE the plan chain, whose
    // kernels bump the counters above)
    alignas(128) unsigned long long csr_desc_all[2];
    unsigned long long csr_desc_legit[2];
    unsigned long long csr_pad[12];
};
// Validation of the caller's CSR arrays on the device (the reference's callers guarantee sorted rows, recometrics/__init__.py:35-41,
// :553-558, and nobody range-checks: on the CPU a bad index is a segfault).  INDPTR / INDEX defects end the call with RM_ERR_INVALID;
// rows that are merely unsorted are sorted by the library (host side, csrc/rm_csr.cpp) and the call runs again.
enum : int { CSR_BAD_INDPTR = 1, CSR_BAD_INDEX = 2, CSR_UNSORTED_TRAIN = 4, CSR_UNSORTED_TEST = 8 };

__device__ __forceinline__ float nan_sentinel_f() { return __int_as_float(0xffffffff); }
__device__ __forceinline__ float pos_inf_f() { return __int_as_float(0x7f800000); }
__device__ __forceinline__ float neg_inf_f() { return __int_as_float(0xff800000); }

// order-preserving integer keys: key(a) < key(b) <=> a < b for non-NaN floats; key(-inf) > 0, so 0 = "nothing yet"
__device__ __forceinline__ unsigned ord_key(float x) { const unsigned b = __float_as_uint(x); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }
__device__ __forceinline__ float ord_unkey(unsigned k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu)); }
__device__ __forceinline__ unsigned long long ord_key(double x)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return b ^ ((b >> 63) ? 0xffffffffffffffffull : 0x8000000000000000ull);
}
__device__ __forceinline__ double ord_unkey(unsigned long long k)
{
    return __longlong_as_double((long long)(k ^ ((k >> 63) ? 0x8000000000000000ull : 0xffffffffffffffffull)));
}

// broadcast of lane `src` (wave-uniform index): v_readlane_b32, no LDS crossbar round trip
template <class T> __device__ __forceinline__ T lane_bcast(T v, int src);
template <> __device__ __forceinline__ int lane_bcast<int>(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
template <> __device__ __forceinline__ float lane_bcast<float>(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
template <> __device__ __forceinline__ double lane_bcast<double>(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// "some lane": the lane mask of the condition itself tested by the scalar unit.  (HIP's __any() and __ballot() take an int: a
// condition that is already a lane mask is turned into a 0 / 1 vector and compared again -- two vector instructions more per
// use, and the sweeps ask several times per tile.)
__device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }

// row of accumulator register r in a 32x32 MFMA result for the lane half h (cdna_hip_programming.md section 3)
// the kernel's argument struct (its first and only explicit argument) read from the kernarg segment at the point of use: the opaque
// asm keeps the compiler from merging these loads with the ones at the top of the kernel, whose results would stay in SGPRs until here
template <class A> __device__ __forceinline__ const A *late_kernargs()
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const A *)p;
}
__device__ __forceinline__ constexpr int mfma32_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

} // namespace rm
