// rm_lib.hip -- host orchestration + the C-ABI of include/recometrics_hip.h  (gfx950 only).
//
// Host counterpart of reference src/recometrics.hpp:359-436 (prologue: clamps, scratch) and of the instantiation
// shims src/recometrics_instantiated.cpp:30-143.  The reference's OpenMP loop over users (:428-437) becomes:
//   plan (classify users, slots, groups)  ->  pack operands  ->  positives  ->  k_sweep  ->  k_finalize
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <type_traits>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <csignal>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/recometrics_hip.h"
#include "rm_device.hpp"
#include "rm_prep.hpp"
#include "rm_launch.hpp"
#include "rm_finalize.hpp"
#include "rm_noise.hpp"

namespace {

using namespace rm;

thread_local std::string g_err;
struct RmError { int code; std::string msg; };

#define HIP_CHECK(expr)                                                                                   \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess)                                                                             \
            throw RmError{e_ == hipErrorOutOfMemory ? RM_ERR_NOMEM : RM_ERR_HIP,                          \
                          std::string(#expr) + ": " + hipGetErrorString(e_)};                             \
    } while (0)

// ---- switches of tests and A/B timing ----------------------------------------------------------------------------------
// The environment is read ONCE, when the library is loaded (and again only when a test asks: rm_debug_reload_switches), never on
// the path of a call.  Every switch chooses among code paths that produce the same results (DESIGN.md section 7 lists them;
// tests/test_switches_cpu.py compares this table with that list); nothing here turns a feature of the call off -- the timing
// builds that drop the tie noise are compiled with -DRM_ABL_NOISE_OFF=1 (scratch/build_abl.py), not switched at run time.
#ifndef RM_ABL_NOISE_OFF
#define RM_ABL_NOISE_OFF 0
#endif
struct Switches {
    bool no_train_bits, no_spec, no_side, ext_topk, no_early_bits, no_test_mask, hbm_lists, nsub2, no_pending, no_pos_keys, no_pos_beside,
         no_seed, no_depth_split, rank_generic, no_fused_auc, no_defer_auc, noise_sequential, no_ext_bits, one_context, noise_per_batch,
         host_trace, no_noise_beside_last, no_pack_beside, no_pos_flat;
    long long free_mb, stream_budget_mb, dense_always_mb, noise_budget_mb, lane_cap_min, lane_min_k, lane_cap_set, sample_seed, split_slack;      // -1 = not set
    double batch_users;                                                        // 0 = not set
    int ramp;                                                                  // 0 = not set
    std::string splits;
    static bool on(const char *name) { return getenv(name) != nullptr; }
    static long long num(const char *name) { const char *e = getenv(name); return e ? atoll(e) : -1; }
    void load()
    {
        no_train_bits = on("RM_DEBUG_NO_TRAIN_BITS"); no_spec = on("RM_DEBUG_NO_SPEC"); no_side = on("RM_DEBUG_NO_SIDE");
        ext_topk = on("RM_DEBUG_EXT_TOPK"); no_early_bits = on("RM_DEBUG_NO_EARLY_BITS"); no_test_mask = on("RM_DEBUG_NO_TEST_MASK");
        hbm_lists = on("RM_DEBUG_HBM_LISTS"); nsub2 = on("RM_DEBUG_NSUB2"); no_pending = on("RM_DEBUG_NO_PENDING");
        no_pos_keys = on("RM_DEBUG_NO_POS_KEYS"); no_pos_beside = on("RM_DEBUG_NO_POS_BESIDE"); no_seed = on("RM_DEBUG_NO_SEED");
        no_depth_split = on("RM_DEBUG_NO_DEPTH_SPLIT"); rank_generic = on("RM_DEBUG_RANK_GENERIC"); no_fused_auc = on("RM_DEBUG_NO_FUSED_AUC");
        no_defer_auc = on("RM_DEBUG_NO_DEFER_AUC"); noise_sequential = on("RM_DEBUG_NOISE_SEQUENTIAL"); no_ext_bits = on("RM_DEBUG_NO_EXT_BITS");
        one_context = on("RM_DEBUG_ONE_CONTEXT"); noise_per_batch = on("RM_DEBUG_NOISE_PER_BATCH"); host_trace = on("RM_HOST_TRACE");
        no_noise_beside_last = on("RM_DEBUG_NO_NOISE_BESIDE_LAST"); no_pack_beside = on("RM_DEBUG_NO_PACK_BESIDE");
        no_pos_flat = on("RM_DEBUG_NO_POS_FLAT");
        free_mb = num("RM_DEBUG_FREE_MB"); stream_budget_mb = num("RM_STREAM_BUDGET_MB"); dense_always_mb = num("RM_DEBUG_DENSE_ALWAYS_MB");
        noise_budget_mb = num("RM_NOISE_BUDGET_MB");
        lane_min_k = num("RM_DEBUG_LANE_MIN_K");                // the smallest k_metrics that takes the lane buffers instead of LDS / HBM lists (A/B timing)
        lane_cap_set = num("RM_DEBUG_LANE_CAP");                // entries per lane buffer (A/B timing; rounded to 16, never below what a selection needs)
        sample_seed = num("RM_DEBUG_SAMPLE_SEED");            // items of the sample that seeds the lane buffers' bounds: 0 = none, else forced to 64 / 256 / 1024 / 2048 / 4096 (A/B timing, tests)
        split_slack = num("RM_DEBUG_SPLIT_SLACK");            // percent of a round charged to the part that ends the sweep's grid (A/B timing)
        lane_cap_min = num("RM_DEBUG_LANE_CAP_MIN");          // the smallest lane buffers that work: a selection every few tiles (tests)
        const char *b = getenv("RM_BATCH_USERS"); batch_users = b ? atof(b) : 0.0;
        const char *r = getenv("RM_DEBUG_RAMP"); ramp = r ? atoi(r) : 0;
        const char *s = getenv("RM_DEBUG_SPLITS"); splits = s ? s : "";
    }
    Switches() { load(); }
};
Switches g_sw;

// bytes the cached workspaces of ALL contexts hold (RM_DEBUG_FREE_MB: tests simulate a device with that much memory, so that
// the memory-bound regimes -- user batches sized by the score rows of k_metrics > 256 -- are reached with small inputs)
std::atomic<long long> g_ws_bytes{0};
inline long long debug_free_cap()
{
    return g_sw.free_mb >= 0 ? (g_sw.free_mb << 20) : -1;
}

// RM_HOST_TRACE: host-side time stamps from inside run() (the phases of one batch), picked up by the host entry's trace line
struct RunTrace {
    std::mutex mu; std::vector<std::pair<const char *, double>> pts; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void point(const char *what) { std::lock_guard<std::mutex> lk(mu); if (pts.size() > 4096) pts.clear(); pts.emplace_back(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};
RunTrace g_run_trace;
#define RM_TRACE_POINT(what) do { if (g_sw.host_trace) g_run_trace.point(what); } while (0)

// ---- cached device workspace (per device), so that repeated calls do not pay hipMalloc ----
struct Workspace {
    std::map<std::string, std::pair<void *, size_t>> bufs;
    void *get(const std::string &name, size_t bytes)
    {
        bytes = std::max<size_t>(bytes, 256);
        auto &b = bufs[name];
        if (b.second < bytes) {
            if (b.first) { (void)hipFree(b.first); g_ws_bytes -= (long long)b.second; b.first = nullptr; b.second = 0; }
            const size_t cap = bytes + bytes / 8;
            const long long sim = debug_free_cap();
            if (sim >= 0 && g_ws_bytes.load() + (long long)cap > sim)
                throw RmError{RM_ERR_NOMEM, "hipMalloc(" + name + ", " + std::to_string(cap) + " B): beyond RM_DEBUG_FREE_MB"};
            hipError_t e = hipMalloc(&b.first, cap);
            if (e != hipSuccess) { b.first = nullptr; throw RmError{RM_ERR_NOMEM, "hipMalloc(" + name + ", " + std::to_string(cap) + " B): " + hipGetErrorString(e)}; }
            b.second = cap;
            g_ws_bytes += (long long)cap;
        }
        return b.first;
    }
    void release()
    {
        for (auto &kv : bufs) if (kv.second.first) { (void)hipFree(kv.second.first); g_ws_bytes -= (long long)kv.second.second; }
        bufs.clear();
    }
};
// One context per (device, shard slot): the cached workspace plus the events / side stream that belong to that device.
// Slot 0 serves ordinary calls; a call sharded inside the library (rm_set_devices) uses slot i for the i-th entry of the device
// list, so that the same physical device can appear more than once ("virtual shards").  A context is used by ONE call at a
// time (`mu` is held while that call enqueues its work), and a call whose stream differs from the previous call's waits for
// that call's `done` event before it touches any workspace buffer: calls on different streams or threads serialise on the
// device instead of corrupting each other's buffers.
struct Ctx {
    int device = 0, slot = 0;
    std::mutex mu;
    Workspace ws;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false, ev_recorded = false;
    hipStream_t side_stream = nullptr;      // second sweep launch of a depth-split call runs beside the first
    hipStream_t pos_stream = nullptr; hipEvent_t pos_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // the streamed users' positives beside the table users' (run())
    hipEvent_t side_ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done = nullptr;               // end of the device work of the most recent call on this context
    hipStream_t own_stream = nullptr;        // stream of a shard worker (multi-device calls)
    hipStream_t up_stream = nullptr;         // host-pointer calls: uploads of the NEXT user batch run beside the current batch's kernels
    hipEvent_t up_ev[2] = {nullptr, nullptr};
    int *pinned_small = nullptr; hipEvent_t flags_ev = nullptr, pass_ev = nullptr;   // tie noise: flag count read-back, flags ready, exact pass done
    void *pinned = nullptr; size_t pinned_bytes = 0;   // page-locked staging for the metric block of a batch (one D2H copy instead of ten)
    void *pinned_get(size_t bytes)
    {
        if (pinned_bytes < bytes) {
            if (pinned) { (void)hipHostFree(pinned); pinned = nullptr; pinned_bytes = 0; }
            if (hipHostMalloc(&pinned, bytes + bytes / 8, hipHostMallocDefault) != hipSuccess) { pinned = nullptr; throw RmError{RM_ERR_NOMEM, "hipHostMalloc of the output staging buffer failed"}; }
            pinned_bytes = bytes + bytes / 8;
        }
        return pinned;
    }
    double timings[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int timed_slots = 0, total_slots = 0;     // slots (user lanes) of the sweep launch the "sweep" timing brackets / of the call
    double acc[4] = {0, 0, 0, 0};            // prep / sweep / finalize / total ms of the batches already read back (host entry)
    // what the packed item image (workspace buffer "Bp") currently holds, for batches of one host call that share B
    unsigned long long packed_tag = 0; int packed_tile = 0, packed_ng = 0; const void *packed_ptr = nullptr;
    const void *bits_ptr = nullptr; long long bits_words = 0; int bits_m = 0;      // dense train rows as last built (set_train_bits)
    unsigned long long bits_tag = 0; const int *bits_train_p = nullptr;             // ... by which call (Call::items_tag) and for which rows
    bool bits_masked = false;                                                       // ... with the test items marked too
    bool bits_partial = false;                                                      // ... for a subset of the users only (Call::only_users)
    bool high_priority = false;          // streams of this context are created with the highest priority (the exact passes of the tie noise)
    Plan *pinned_plan = nullptr;                                                    // page-locked landing place of the plan read-back
    const void *log2_ptr = nullptr; int log2_K = 0;                                 // the DCG discount table the workspace holds (run())
    unsigned long long packed_amax_b = 0; int packed_nonfinite_b = 0;
};
// A stream of a context.  The NOISE_SLOT contexts ask for the highest priority the device offers: their work -- the exact pass of the
// tie noise over a few thousand users -- is enqueued BESIDE a sweep that fills every compute unit, and at normal priority its kernels
// are only served when that sweep has no more blocks to place (measured: the pass started when the sweep ended); at high priority its
// workgroups go first whenever a compute unit frees up.
inline hipError_t create_stream(hipStream_t *st, bool high_priority)
{
    if (high_priority) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            return hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest);
        (void)hipGetLastError();
    }
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}
std::mutex g_ctx_mu;
std::map<std::pair<int, int>, std::unique_ptr<Ctx>> g_ctx;
thread_local Ctx *g_last_ctx = nullptr;      // context of the most recent call on this thread (rm_get_timings)

Ctx &context(int slot)
{
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto &p = g_ctx[std::make_pair(dev, slot)];
    if (!p) { p.reset(new Ctx()); p->device = dev; p->slot = slot; }
    return *p;
}
// the second context of a host-pointer call: its odd batches run there (own workspace, events, stream), so that a batch's
// prep -- two dozen short launches and a plan read-back -- overlaps the sweep of the batch before it
// (a third kind of context, NOISE_SLOT apart from its owner: the exact pass of the tie noise beside a first sweep -- also of a
// batch that itself runs on a peer context)
constexpr int PEER_SLOT = 1 << 12, NOISE_SLOT = 1 << 13;
Ctx &peer_context(const Ctx &cx, int offset = PEER_SLOT)
{
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    auto &p = g_ctx[std::make_pair(cx.device, cx.slot + offset)];
    if (!p) { p.reset(new Ctx()); p->device = cx.device; p->slot = cx.slot + offset; p->high_priority = offset == NOISE_SLOT; }
    return *p;
}

template <class T> struct Call {          // one calc_metrics call; every pointer is a DEVICE pointer
    const T *A; size_t lda; const T *B; size_t ldb;
    int m, n, k;
    const int *train_p, *train_i; long long nnz_train;
    const int *test_p, *test_i; const T *test_v; long long nnz_test;
    int K; bool cumulative, noise;
    T *out[10];                            // p, tp, r, ap, tap, ndcg, hit, rr, roc, pr
    bool cold; int min_items_pool, min_pos_test;
    // optional ranking outputs (device)
    int *topk_idx; T *topk_score; long long *pos_rank; int *status;
    // batches of one host call share the item factors: a non-zero tag says "the packed image and the |B| bound made for this
    // tag are still valid" (set by run_host_range; 0 = always repack)
    unsigned long long items_tag;
    // tie noise (rm_noise.hpp)
    unsigned long long seed; long long user0;      // the engine of local user u is seeded with seed + user0 + u
    const unsigned char *only_users;               // optional [m]: evaluate only these users, leave the others' outputs alone
    const int *noise_row; int noise_row0;          // row of each user in noise_E (null: row = user)
    const T *noise_E; long long noise_ld;          // per-item noise rows; null = scores as they are
    int *noise_flag;                               // optional [m] out (fp32 first pass): users the noise can change
    bool same_train_rows = false;                  // a later pass of the same call over the same users' rows: dense train rows may be reused
    long long eval_users = -1;                     // users this pass evaluates when fewer than m (only_users given); -1 = all m
    // fp32 tie noise, first pass: right before the sweep is launched the flags set so far (users with a TEST item in the noise
    // zone: all of them are known by then) are copied to `flag_snapshot`, their number to `*flag_count_host` (page-locked), and
    // `flags_event` is recorded -- the exact pass of those users can start beside the sweep (run_call)
    int *flag_snapshot = nullptr; int *flag_count_host = nullptr; hipEvent_t flags_event = nullptr;
    // dense train rows built by another pass of the same call over the same users (the exact noise pass beside the first sweep
    // reads the first pass's rows instead of building 463 MB of its own at BASELINE C2)
    const unsigned *ext_bits = nullptr; long long ext_words = 0; bool ext_masked = false;
    // fp32 tie noise of a host-pointer call in user batches: the batch only runs its FIRST pass and flags the users the noise can
    // touch in this array (its slice of a range-wide one); the exact pass over all flagged users of the range follows the
    // last batch (noise_exact_pass) -- per batch it costs two host-side waits that keep the next batch from being enqueued
    int *first_pass_flags = nullptr;
    // the CSR arrays of these users have been validated by an earlier pass of the same call (plan kernels k_check_csr_*)
    bool csr_checked = false;
    // dense train rows: decided once per call (-1 = not yet: run() asks dense_rows_fit itself) -- every pass of a call gets the
    // answer the first one got, whatever the passes in between have allocated
    int dense_fit = -1;
};
// internal status: some CSR row is not sorted (the entry points sort a copy of the rows and run again; never returned to a caller)
constexpr int RM_INTERNAL_UNSORTED = 1000;

inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

template <class T> __global__ void k_iota(T *p, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = (T)i;
}

template <class S>
__global__ void k_export_rank(int m, int K, const Entry<S> *merged, int *topk_idx, S *topk_score, const int *flags)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)m * K || (flags[i / K] & UF_SKIP)) return;
    topk_idx[i] = merged[i].idx;
    topk_score[i] = merged[i].s;
}

__global__ void k_export_pos_rank(long long nnz, int m, const int *test_p, const int *flags, const int *pos_order,
                                  const long long *rank_sorted, long long *pos_rank, int have_ranks)
{
    // one thread per user row (rows are short).  Users that were never ranked (skipped, or NDCG-only) have no entry in
    // pos_order -- whatever the workspace held before -- and report rank 0.
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= m || (flags[u] & UF_SKIP)) return;
    const bool ranked = have_ranks && (flags[u] & UF_ACTIVE) && !(flags[u] & UF_ONLY_NDCG);
    for (int e = test_p[u]; e < test_p[u + 1]; e++) pos_rank[e] = ranked ? rank_sorted[test_p[u] + pos_order[e]] : 0;
}

// ---- per-precision traits: which sweep kernel, which operand image, how many users ride on a wavefront ----
template <class T> struct Prec;
template <> struct Prec<float> {
    static constexpr int GU = GROUP_USERS;               // users per group
    typedef float4 PackT;  typedef u32x2 ListT;  typedef SweepArgs Args;
    // factor groups of 8: the instantiated counts up to 512 factors, beyond that whole 128-factor chunks (run-time count)
    static int supported_ng(int k) { const int ng = (k + 7) / 8; for (int o : {2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 16, 32, 64}) if (ng <= o) return o; return (ng + 15) / 16 * 16; }
    static const char *limit() { return "unsupported factor count"; }
    static size_t lds_b(int NG, int tile = TILE_ITEMS) { return 2ull * std::min(NG, 16) * 2 * tile * 16; }
    static long long items_units(int tiles, int NG, int tile = TILE_ITEMS) { return (long long)tiles * NG * 2 * tile; }
    static long long users_units(int groups, int NG) { return (long long)groups * NG * 2 * GU; }
    static constexpr bool has_pending = true, pending_for_append = false;   // appends of K > 32 are single stores already
    static constexpr size_t pend_key_bytes = 8;
    static constexpr int pend_cap_max = 8;                // keys per lane: measured flat from 3 to 8 at C2, best at 6-8
    static constexpr int max_nsub = 3;
    static size_t lists_b(int, int K) { return (size_t)GROUPS_PER_BLOCK * (K + 2) * GU * 8; }      // one list per group, shared by its waves
    static constexpr bool block_carve = true;
    static void set_pending(SweepArgs &sa, int cap, int off) { sa.pend_cap = cap; sa.pend_off = off; }
    static void set_sync(SweepArgs &sa, int off) { sa.sync_off = off; }
    static void set_ublocks(SweepArgs &sa, int first, int count) { sa.ublock0 = first; sa.n_ublocks = count; }
    // K > 32: entries per lane buffer of the sweep (rm_list.hpp; a multiple of 16, and a selection's K + slack survivors plus a tile's
    // sixteen appends must fit one lane: 2K + 16), lanes per user
    static int lane_cap(int K) { return (2 * K + 16 + 15) / 16 * 16; }
    static constexpr int lanes_per_user = 2, lane_tile = 16;
};
template <> struct Prec<double> {
    static constexpr int GU = GROUP_USERS64;
    typedef double2 PackT;  typedef u32x4 ListT;  typedef Sweep64Args Args;
    static int supported_ng(int k) { const int ng = (k + 7) / 8; for (int o : {2, 3, 4, 5, 6, 7, 8, 16, 32, 64}) if (ng <= o) return o; return (ng + 7) / 8 * 8; }
    static const char *limit() { return "unsupported factor count"; }
    static size_t lds_b(int NG, int = TILE_ITEMS) { return 2ull * std::min(NG, 8) * 4 * TILE_ITEMS * 16; }
    static long long items_units(int tiles, int NG, int = TILE_ITEMS) { return (long long)tiles * NG * 4 * TILE_ITEMS; }
    static long long users_units(int groups, int NG) { return (long long)groups * NG * 4 * GU; }
    static constexpr bool has_pending = true, pending_for_append = true;    // saves four 64-bit shuffles per candidate register
    static constexpr size_t pend_key_bytes = 12;
    static constexpr int pend_cap_max = 3;                // a user spans four lanes here: larger buffers only delay the bound (measured)
    static constexpr int max_nsub = 2;
    static size_t lists_b(int ns, int K) { return 4ull * ns * K * GU * sizeof(ListT); }               // one list per wave
    static constexpr bool block_carve = true;
    static void set_pending(Sweep64Args &sa, int cap, int off) { sa.pend_cap = cap; sa.pend_off = off; }
    static void set_sync(Sweep64Args &sa, int off) { sa.sync_off = off; }
    static void set_ublocks(Sweep64Args &sa, int first, int count) { sa.ublock0 = first; sa.n_ublocks = count; }
    // (four lanes per user: K + slack + a tile's eight appends must fit ONE lane, the user's four hold ~5K together)
    static int lane_cap(int K) { return (K + lane_sel_slack(K) + 8 + 16 + 15) / 16 * 16; }
    static constexpr int lanes_per_user = 4, lane_tile = 8;
};

inline void check_launch(int rc)
{
    if (rc == -1) throw RmError{RM_ERR_UNSUPPORTED, "unsupported factor count"};
    if (rc != 0) throw RmError{RM_ERR_HIP, std::string("sweep launch: ") + hipGetErrorString((hipError_t)rc)};
}
inline void dispatch_sweep(bool auc, bool dump, bool llds, int nsub, int NG, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs &sa)
{
    // list mode of the fp32 sweep (rm_sweep.hpp): 0 = LDS, 1 = HBM replace-the-minimum (K <= 32), 2 = HBM append buffers
    check_launch(launch_sweep32(auc, dump, llds ? 0 : (sa.buffered_lists ? 2 : 1), nsub, NG, grid, lds, stream, sa));
}
inline void dispatch_sweep(bool auc, bool dump, bool llds, int, int NG, dim3 grid, size_t lds, hipStream_t stream, const Sweep64Args &sa)
{
    check_launch(launch_sweep64(auc, dump, llds ? 0 : (sa.buffered_lists ? 2 : 1), NG, grid, lds, stream, sa));
}

inline void pack_operands(const float *A, size_t lda, const float *B, size_t ldb, int n, int k, int NG, int tile_items, const int *slot_user,
                          int n_slots, float4 *Ap, long long ap, float4 *Bp, long long bp, hipStream_t stream, bool items = true)
{
    if (items) {
        const size_t lds = pack_items_tile_lds(NG, tile_items);
        const long long tiles = bp / ((long long)NG * 2 * tile_items);
        // (the attribute is per DEVICE -- rm_set_devices drives several -- so it is set at every launch, like every other kernel of
        // the library that takes more than 64 KB; a refusal falls back to the kernel that needs no LDS)
        bool tiled = lds <= 96 * 1024 && tiles * NG * 2 * tile_items == bp;
        if (tiled && lds > 48 * 1024)
            tiled = hipFuncSetAttribute((const void *)k_pack_items_tile<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        if (tiled) {
            hipLaunchKernelGGL(k_pack_items_tile<float>, dim3((unsigned)tiles), dim3(256), lds, stream, B, ldb, n, k, NG, tile_items, Bp);
            if (hipGetLastError() != hipSuccess) tiled = false;
        }
        if (!tiled) hipLaunchKernelGGL(k_pack_items<float>, dim3(cdiv(bp, 256)), dim3(256), 0, stream, B, ldb, n, k, NG, tile_items, Bp, bp);
    }
    if (ap > 0) hipLaunchKernelGGL(k_pack_users<float>, dim3(cdiv(ap, 256)), dim3(256), 0, stream, A, lda, k, NG, slot_user, n_slots, Ap, ap);
}
inline void pack_operands(const double *A, size_t lda, const double *B, size_t ldb, int n, int k, int NG, int, const int *slot_user,
                          int n_slots, double2 *Ap, long long ap, double2 *Bp, long long bp, hipStream_t stream, bool items = true)
{
    if (items) hipLaunchKernelGGL(k_pack_items64<double>, dim3(cdiv(bp, 256)), dim3(256), 0, stream, B, ldb, n, k, NG, Bp, bp);
    if (ap > 0) hipLaunchKernelGGL(k_pack_users64<double>, dim3(cdiv(ap, 256)), dim3(256), 0, stream, A, lda, k, NG, slot_user, n_slots, Ap, ap);
}

constexpr size_t LDS_LIMIT = 160 * 1024;
constexpr size_t SYNC_BYTES = 32;          // end of the sweep's LDS: arrival counters of the sub-tiles (4 words), list locks of the groups (4 words)

// HBM the score rows of streamed users (and of every user when k_metrics > 256) may take: a third of what is free now, or
// RM_STREAM_BUDGET_MB (tests)
// (what the workspace already holds of exactly these buffers counts as free: a later batch of the same size must get the
// same answer as the first one, whose rows are still cached -- otherwise its budget shrinks and an equal-sized batch fails)
inline long long free_plus_owned(const Workspace &ws, std::initializer_list<const char *> names)
{
    size_t fr = 0, tot = 0;
    HIP_CHECK(hipMemGetInfo(&fr, &tot));
    const long long sim = debug_free_cap();
    if (sim >= 0) fr = (size_t)std::max<long long>(0, std::min<long long>((long long)fr, sim - g_ws_bytes.load()));
    long long owned = 0;
    for (const char *nm : names) { auto it = ws.bufs.find(nm); if (it != ws.bufs.end()) owned += (long long)it->second.second; }
    return (long long)fr + owned;
}
inline long long stream_budget_bytes(const Workspace &ws)
{
    if (g_sw.stream_budget_mb >= 0) return g_sw.stream_budget_mb << 20;
    return free_plus_owned(ws, {"stream_scores", "sel_hi", "sel_lo"}) / 3;
}

// Which top-K scheme a pass takes (rm_sweep.hpp LMODE):
//   lane lists   per-lane append buffers + lane-parallel selection + k_collect_topk (rm_list.hpp): k_metrics from `lane_min_k` (20 / 14; the
//                replace-the-minimum lists rescan K entries per insert: BASELINE C2's shape took 8.1 / 15.0 / 26.8 ms at K = 20 /
//                32 / 50 in LDS or HBM lists against 8.1 / 8.7 / 10.0 here, profiles/r6_ksweep_C2.txt) up to what k_collect_topk sorts in LDS (K + one lane buffer <= 4,096 entries: K <= 1,354), while the
//                buffers -- 8 waves x 64 lanes x lane_cap entries per block of the sweep's grid -- fit a third of the free memory;
//   lists        below that: in LDS while they fit, else replace-the-minimum in HBM;
//   ext_topk     beyond: one score row per user + k_select_topk.
// (Larger buffers -- sized so that one selection per item range is the rule -- were measured in round 6 and are SLOWER: what a range
// costs is the entries its selections scan in all, which hardly depends on the size, and larger buffers lose the L2: BASELINE C2's shape
// at K = 100: 12.9 ms with 304 entries per lane, 12.7 with 448, against 12.4 with the base size and 12.2 with 160; profiles/r6_ab_c2.txt)
template <class T> inline long long lane_list_bytes(int K, long long n_blocks) { return n_blocks * 8 * WAVE * (long long)Prec<T>::lane_cap(K) * (long long)(sizeof(T) + 4); }
// (`three_subtiles`: the kernel the replace-the-minimum lists would run on has three sub-tiles per step -- fp32 up to 64 factors -- which
// the lane buffers do not (a user's candidates over three waves: slower, r6h): there the lists hold out until K = 20.  Where both run
// two sub-tiles the lane buffers' lighter epilogue wins from K = 14: north-star shape K = 16 77.4 against 78.9 ms, K = 20 77.6 / 79.9,
// K = 12 77.45 / 77.41; C3 (cumulative K = 20) 110.9 / 115.2, K = 10 equal, K = 5 111.0 / 108.4; profiles/r6_ab_c2.txt r6m)
template <class T> inline bool lane_lists_possible(int K, bool three_subtiles = false)
{
    // (fp64: every k_metrics -- a user sits on four lanes there and the lists' owner lane collects from all of them: C5's shape K = 5
    // 73.4 against 74.8 ms, K = 10 73.6 / 76.4, K = 20 75.1 / 84.2; C2's shape in fp64 K = 10 4.50 / 5.22, K = 14 4.50 / 6.28)
    const long long min_k = g_sw.lane_min_k >= 0 ? g_sw.lane_min_k : (three_subtiles ? 20 : (sizeof(T) == 8 ? 1 : 14));
    return !g_sw.ext_topk && K >= min_k && K + Prec<T>::lane_cap(K) <= COLLECT_MAX_ENTRIES;
}
// (the grid of the sweep is only known behind the plan: user blocks of the call, or a few rounds of 256 blocks when there are few)
template <class T> inline long long lane_blocks_bound(long long m) { return (m + 4 * Prec<T>::GU - 1) / (4 * Prec<T>::GU) + 1024; }
inline long long lane_budget_bytes(const Workspace &ws) { return free_plus_owned(ws, {"glists", "stream_scores", "sel_hi", "sel_lo"}) / 3; }
template <class T> inline bool lane_lists_fit(const Workspace &ws, int K, long long m, bool three_subtiles)
{
    return lane_lists_possible<T>(K, three_subtiles) && lane_list_bytes<T>(K, lane_blocks_bound<T>(m)) <= lane_budget_bytes(ws);
}

// Dense train rows for the fp32 sweep when they are small (m * n / 8 bytes <= 1 GiB, e.g. 463 MB at BASELINE C2): with ~100
// train items per user among 27k items, some lane of a wave has one in nearly every 32-item sub-tile, and the per-item walk of
// the CSR cursor (compare, consume, reload, loop) was 9 % of the C2 sweep; one word per lane and tile replaces it.
// (rows are padded to 192 items = a whole tile of either size, 64 or 96: the row stride does not depend on the sweep's geometry,
// which is only known behind the plan read-back -- the rows are built before it, beside the plan kernels)
inline long long dense_row_words(long long n) { return (n + 191) / 192 * 6; }
// Up to 1 GiB of rows always; beyond that (many users at a small item count: 1M users x 27k items = 3.4 GB) when they take no more
// than a quarter of the HBM that is free -- what the workspace already holds of the rows and of the streamed users' score rows counts
// as free, so that equal calls get equal answers -- and no more than 8 GiB: the sweep indexes the rows by 32-bit word offsets.
// (One answer per call: the first pass asks, Call::dense_fit carries the answer to the others.)
inline bool dense_rows_fit(const Workspace &ws, int m, long long n)
{
    const long long words = dense_row_words(n);
    if (words > TRAIN_BITS_MAX_WORDS || g_sw.no_train_bits) return false;
    const unsigned long long bytes = (unsigned long long)m * (unsigned long long)words * 4ull;
    unsigned long long always = 1ull << 30;
    if (g_sw.dense_always_mb >= 0) always = (unsigned long long)g_sw.dense_always_mb << 20;      // (tests: the branch below at small sizes)
    if (bytes <= always) return true;
    if (bytes > (8ull << 30)) return false;
    return (long long)bytes <= free_plus_owned(ws, {"train_bits", "stream_scores", "sel_hi", "sel_lo"}) / 4;
}
// `mask_test`: the rows mark the users' TEST items as well (k_merge_positives puts them back after the sweep; rm_noise.hpp clears
// the test items' bits in its own copy of a row)
// `early`: the rows were launched before the plan was known (run(): on the side stream, beside the plan kernels) with the guess
// `early_masked`; when the guess holds nothing is launched here
template <class C> inline void set_train_bits(SweepArgs &sa, Ctx &cx, const C &c, int m, int n, bool fit, hipStream_t stream, bool mask_test, bool early = false, bool early_masked = false,
                                              unsigned char *ent_masked = nullptr)
{
    Workspace &ws = cx.ws;
    const long long words = dense_row_words(n);
    const size_t bytes = (size_t)m * (size_t)words * 4;
    sa.train_bits = nullptr; sa.train_words = 0;
    if (!fit) { cx.bits_tag = 0; return; }
    unsigned *bits = (unsigned *)ws.get("train_bits", bytes);
    // (the exact second pass of the tie noise evaluates a subset of the same users: the rows of the first pass are still there)
    const bool ready = (early && early_masked == mask_test) ||
                       (c.same_train_rows && !cx.bits_partial && cx.bits_ptr == (const void *)bits && cx.bits_words == words && cx.bits_m == m && cx.bits_masked == mask_test);
    if (!ready) {
        launch_train_bits(stream, m, n, (int)words, c.train_p, c.train_i, (mask_test || ent_masked) ? c.test_p : nullptr, c.test_i, bits,
                          (const Plan *)ws.get("plan", sizeof(Plan)), c.only_users, ent_masked, mask_test);
        cx.bits_partial = c.only_users != nullptr;
    }
    cx.bits_ptr = (const void *)bits; cx.bits_words = words; cx.bits_m = m; cx.bits_masked = mask_test;
    cx.bits_tag = c.items_tag; cx.bits_train_p = c.train_p;
    sa.train_bits = bits; sa.train_words = (int)words;
}
template <class C> inline void set_train_bits(Sweep64Args &, Ctx &, const C &, int, int, bool, hipStream_t, bool, bool = false, bool = false, unsigned char * = nullptr) {}
inline void set_part_extra(SweepArgs &sa, int extra) { sa.part_extra = extra; }
inline void set_part_extra(Sweep64Args &, int) {}
// the sweep variant with the epilogue's switches as constants (rm_sweep.hpp k_sweep SPEC): 1 = dense train rows, 2 = CSR cursor,
// both with every score finite, no tie noise and the top-K lists in reach; anything else reads the switches at run time
inline void set_spec(SweepArgs &sa)
{
    sa.spec = (sa.check_nan || sa.noise_E || sa.ext_topk || g_sw.no_spec) ? 0 : (sa.train_bits ? 1 : 2);
}
inline void set_spec(Sweep64Args &sa) { sa.spec = (sa.check_nan || sa.noise_E || sa.ext_topk || g_sw.no_spec) ? 0 : 1; }
inline void set_ext_bits(SweepArgs &sa, const unsigned *bits, int words) { sa.train_bits = bits; sa.train_words = words; }
inline void set_ext_bits(Sweep64Args &, const unsigned *, int) {}

// Sample seeds for the lane buffers (k_seed_from_sample): how many items, and the two launches.  Not when a
// score may be non-finite (the NaN check wants every score looked at) and not under the tie noise (the sample is scored without it).
template <class T> inline int sample_seed_items(const Workspace &ws, int K, int n, long long n_slots, bool off, bool lists = false)
{
    if (off || g_sw.sample_seed == 0) return 0;
    int S;
    if (g_sw.sample_seed > 0) {          // forced (tests, A/B timing): the largest size at or below the value; 64 and 256 for small catalogues
        S = 64;
        for (int c : {256, 1024, 2048, 4096}) if (g_sw.sample_seed >= c) S = c;
    } else {
        // A sample of S items costs S / n of the sweep's matrix work and gives the K-th best of S as the bound.  Measured at BASELINE
        // C2's shape (26,744 items; profiles/r6_ab_c2.txt): 1,024 items beat 2,048 up to k_metrics ~ 128, 2,048 beyond, 4,096 from ~ 400.
        // (k_metrics = 500: sweep 22.5 ms without, 13.9 with 2,048, 10.9 with 4,096 in front of which the sample costs 1.9 ms more; 1,000: 36.8 / 22.8 / 16.0)
        S = K <= 128 ? 1024 : (K <= 384 ? 2048 : 4096);
        // The replace-the-minimum lists (fp32, k_metrics below the lane buffers') rescan their K entries per insert, and most inserts come
        // while the bound is still low: a sample of 256 pays from K = 12 -- C2's shape, M users/s without / with: K = 11 17.3 / 17.0, 12 16.6 / 16.8,
        // 14 15.8 / 16.5, 16 15.0 / 15.7, 18 14.7 / 15.2 (1,024 items: the same within 1 %); at K = 10 the sweep goes 6.34 -> 6.24 ms and the
        // sample costs 0.14
        if (lists) { if (K < 12) return 0; S = 256; }
        if ((long long)S * 6 > n || K * 2 > S) return 0;
        // (fp64 takes the lane buffers from k_metrics = 1: at C2's shape in fp64 the sample pays from ~ 16 -- step 19.6 -> 19.0 ms at 20,
        // 25.9 -> 22.2 at 100, 34.8 -> 30.7 at 256; even at 10)
        if (sizeof(T) == 8 && K < 16) return 0;
    }
    if (S > n || K > S) return 0;
    if ((long long)sizeof(T) * n_slots * S > free_plus_owned(ws, {"sample_scores"}) / 8) return 0;
    return S;
}
template <class T, class Args>
inline void seed_from_sample(const Args &sa, Workspace &ws, int S, int NG, int n_slots, int n_ublocks, hipStream_t stream, const void *items64 = nullptr, void *lists = nullptr)
{
    typedef Prec<T> P;
    typedef typename std::remove_pointer<decltype(Args{}.thr_shared)>::type ThrT;
    T *sample = (T *)ws.get("sample_scores", sizeof(T) * (size_t)n_slots * (size_t)S);
    Args sd = sa;
    sd.n = S; sd.tiles_total = S / TILE_ITEMS; sd.K = 1; sd.n_splits = 1; sd.tail_ublocks = 0; sd.tail_splits = 1; sd.part_splits = 1;
    sd.buffered_lists = 0; sd.ext_topk = 0; sd.lane_cap = 0; sd.lane_cnt = nullptr; sd.spec = 0; sd.dump = sample;
    // (a sweep with three sub-tiles per step has its item image in 96-item tiles and its dense train rows in words of them: the DUMP
    // variant, two sub-tiles, gets an image of the sample of its own and no dense rows -- the seed kernel walks the sparse rows)
    if (items64) { sd.Bp = (decltype(sd.Bp))items64; sd.glists = (decltype(sd.glists))lists; set_ext_bits(sd, nullptr, 0); }
    P::set_pending(sd, 0, 0);
    P::set_sync(sd, (int)P::lds_b(NG));
    dispatch_sweep(false, true, false, 2, NG, dim3((unsigned)n_ublocks), P::lds_b(NG) + SYNC_BYTES, stream, sd);
    const dim3 grid((unsigned)cdiv(n_slots, 4)), block(256);
#define RM_SEED_LAUNCH(NV) hipLaunchKernelGGL((k_seed_from_sample<T, ThrT, NV>), grid, block, 0, stream, n_slots, sa.K, (const T *)sample, sa.slot_user, sa.slot_chunk, sa.train_p, sa.train_i, sa.thr_shared)
    switch (S) {
    case 4096: RM_SEED_LAUNCH(64); break;
    case 2048: RM_SEED_LAUNCH(32); break;
    case 1024: RM_SEED_LAUNCH(16); break;
    case 256: RM_SEED_LAUNCH(4); break;
    default: RM_SEED_LAUNCH(1); break;
    }
#undef RM_SEED_LAUNCH
    check_launch(hipGetLastError());
}

// what the plan's validation kernels found wrong with the caller's CSR arrays -> the error the entry points see
template <class T>
void throw_csr_defects(const Plan &hp, const Call<T> &c, Ctx &cx)
{
    const bool unsorted = hp.csr_desc_all[0] != hp.csr_desc_legit[0] || hp.csr_desc_all[1] != hp.csr_desc_legit[1];
    if (!hp.csr_bad && !unsorted) return;
    cx.bits_tag = 0; cx.bits_ptr = nullptr;                        // (dense train rows launched beside the plan were not built)
    if (hp.csr_bad & CSR_BAD_INDPTR)
        throw RmError{RM_ERR_INVALID, "CSR index pointers of row " + std::to_string((long long)c.user0 + hp.csr_where) + " are negative, decreasing or beyond the index array"};
    if (hp.csr_bad & CSR_BAD_INDEX)
        throw RmError{RM_ERR_INVALID, "CSR column index out of range [0, " + std::to_string(c.n) + ") in row " + std::to_string((long long)c.user0 + hp.csr_where)};
    throw RmError{RM_INTERNAL_UNSORTED, "CSR rows are not sorted"};
}
// the validation of the index ARRAYS (the index pointers are k_classify's / k_check_csr_ptr's): `nnz_*` bound the grids -- the
// kernels look up the entry range of this call's rows themselves
inline void launch_csr_index_checks(int m, int n, const int *train_p, const int *train_i, long long nnz_train, const int *test_p, const int *test_i, long long nnz_test,
                                    Plan *plan, hipStream_t stream)
{
    hipLaunchKernelGGL(k_check_csr_starts, dim3(cdiv(m, 1024)), dim3(1024), 0, stream, m, train_p, train_i, test_p, test_i, plan);
    auto blocks = [](long long nnz) { return dim3((unsigned)std::min<long long>((nnz / 4 + 2 + CHECK_FLAT_THREADS - 1) / CHECK_FLAT_THREADS, CHECK_FLAT_BLOCKS)); };
    if (nnz_train > 0) hipLaunchKernelGGL(k_check_csr_flat, blocks(nnz_train), dim3(CHECK_FLAT_THREADS), 0, stream, m, n, train_p, train_i, plan, 0);
    if (nnz_test > 0) hipLaunchKernelGGL(k_check_csr_flat, blocks(nnz_test), dim3(CHECK_FLAT_THREADS), 0, stream, m, n, test_p, test_i, plan, 1);
}
inline void launch_csr_checks(int m, int n, const int *train_p, const int *train_i, long long nnz_train, const int *test_p, const int *test_i, long long nnz_test,
                              Plan *plan, const unsigned char *, hipStream_t stream)
{
    hipLaunchKernelGGL(k_check_csr_ptr, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, train_p, nnz_train, test_p, nnz_test, plan);
    launch_csr_index_checks(m, n, train_p, train_i, nnz_train, test_p, test_i, nnz_test, plan, stream);
}
// the validation alone, with its own wait: for the callers that index by the CSR arrays BEFORE the pipeline runs (the fp64 tie noise
// builds its noise rows -- candidate index = item - train items below it -- in front of run())
template <class T>
void check_csr_now(const Call<T> &c, hipStream_t stream, Ctx &cx)
{
    Plan *plan = (Plan *)cx.ws.get("plan", sizeof(Plan));
    if (cx.ev_valid) HIP_CHECK(hipStreamWaitEvent(stream, cx.done, 0));
    HIP_CHECK(hipMemsetAsync(plan, 0, sizeof(Plan), stream));
    launch_csr_checks(c.m, c.n, c.train_p, c.train_i, c.nnz_train, c.test_p, c.test_i, c.nnz_test, plan, c.only_users, stream);
    Plan hp;
    HIP_CHECK(hipMemcpyAsync(&hp, plan, sizeof(Plan), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    throw_csr_defects(hp, c, cx);
}

// ---------------------------------------------------------------------------------------------------------------------
// device pipeline (T = float: v_mfma_f32_32x32x2_f32 sweep; T = double: v_mfma_f64_16x16x4_f64 sweep)
// ---------------------------------------------------------------------------------------------------------------------
template <class T>
void run(const Call<T> &c, hipStream_t stream, Ctx &cx)
{
    typedef Prec<T> P;
    constexpr int GU = P::GU;
    Workspace &ws = cx.ws;
    double *g_timings = cx.timings;
    hipEvent_t *g_ev = cx.ev;
    const int m = c.m, n = c.n, k = c.k, K = c.K;
    // reference recometrics.hpp:390-393
    const int min_items_pool = std::max(std::max(c.min_items_pool, K), 2);
    const int min_pos_test = std::min(c.min_pos_test, 1);
    int req = 0;
    for (int i = 0; i < 10; i++) if (c.out[i]) req |= (1 << i);
    const bool want_auc = req & (RQ_ROC | RQ_PR);

    const int NG = P::supported_ng(k);
    if (NG < 0) throw RmError{RM_ERR_UNSUPPORTED, std::string(P::limit()) + " (got " + std::to_string(k) + ")"};

    if (!cx.ev_valid) {
        for (int i = 0; i < 5; i++) HIP_CHECK(hipEventCreate(&cx.ev[i]));
        HIP_CHECK(hipEventCreateWithFlags(&cx.done, hipEventDisableTiming));
        cx.ev_valid = true;
    } else {
        HIP_CHECK(hipStreamWaitEvent(stream, cx.done, 0));          // the previous call on this context may still be running on another stream
    }
    g_last_ctx = &cx;
    HIP_CHECK(hipEventRecord(g_ev[0], stream));
    // side stream of the context: kernels that do not depend on one another run beside the main stream's (the dense train rows beside
    // the positives, the streamed users' ranks beside the rest of the finalisation; a depth-split call's second sweep launch)
    auto side_stream = [&]() -> hipStream_t {
        if (!cx.side_stream) {
            HIP_CHECK(create_stream(&cx.side_stream, cx.high_priority));
            for (int i = 0; i < 7; i++) HIP_CHECK(hipEventCreateWithFlags(&cx.side_ev[i], hipEventDisableTiming));
        }
        return cx.side_stream;
    };
    // (an error between a fork and its join must not leave the side stream reading the call's buffers behind the caller's back)
    struct SideGuard {
        Ctx &cx; int pending = 0;                               // pieces of side-stream work the main stream has not been told to wait for yet
        ~SideGuard() { if (pending > 0 && cx.side_stream) (void)hipStreamSynchronize(cx.side_stream); }
    } side_guard{cx};
    auto fork_side = [&]() { hipStream_t sd = side_stream(); HIP_CHECK(hipEventRecord(cx.side_ev[0], stream)); HIP_CHECK(hipStreamWaitEvent(sd, cx.side_ev[0], 0)); side_guard.pending++; return sd; };
    auto join_side = [&]() { HIP_CHECK(hipEventRecord(cx.side_ev[1], cx.side_stream)); HIP_CHECK(hipStreamWaitEvent(stream, cx.side_ev[1], 0)); side_guard.pending = 0; };
    const bool use_side = !g_sw.no_side;

    RM_TRACE_POINT("run: start");
    // ---- plan ----
    int *flags = (int *)ws.get("flags", sizeof(int) * (size_t)m);
    int *user_nslots = (int *)ws.get("user_nslots", sizeof(int) * (size_t)m);
    int *uslot_base = (int *)ws.get("uslot_base", sizeof(int) * (size_t)m);
    Plan *plan = (Plan *)ws.get("plan", sizeof(Plan));
    int *heavy_users = (int *)ws.get("heavy_users", sizeof(int) * (size_t)m);
    ClassifyArgs ca{m, n, K, c.train_p, c.test_p, req, c.cold ? 1 : 0, min_items_pool, min_pos_test, want_auc ? 1 : 0,
                    flags, user_nslots, heavy_users, plan};
    ca.only = c.only_users;
    ca.heavy_npos = K > FIN_TOPV ? FIN_TOPV : HEAVY_NPOS;
    // the caller's CSR arrays are validated in front of everything that indexes by them (an out-of-range column index in
    // k_train_bits would be a memory fault; on the CPU reference it is a segfault): the index pointers by k_classify itself, the
    // 80 MB of indices of BASELINE C2 by k_check_csr_rows beside the plan chain (~40 us)
    ca.check_ptr = c.csr_checked ? 0 : 1; ca.nnz_train = c.nnz_train; ca.nnz_test = c.nnz_test;
    // Users with more than POS_CHUNK test items are "streamed" (rm_device.hpp STREAM_CLASS) when a score row for each of
    // them fits the HBM budget: a third of the free memory unless RM_STREAM_BUDGET_MB says otherwise (0 = never; such
    // users then take one sweep slot per chunk of their test row -- same results, the contraction repeated per chunk).
    // The plan assumes they fit; the host looks at their number in the read-back and, should the rows not fit, plans once more
    // without streaming (what used to be two launches in front of k_classify -- count, decide -- on every call).
    const long long stream_ld_max = ((long long)n + 191) / 192 * 192;             // row stride for either tile size (64 / 96 items)
    // k_metrics beyond the sweep's lists (append buffers + wave compaction reach 256): every user is streamed and
    // k_select_topk picks its top-K from the stored row -- any k_metrics <= n, at one score row of HBM per user
    const bool want_lane = lane_lists_fit<T>(ws, K, c.eval_users >= 0 ? std::min<long long>(c.eval_users, m) : m, P::max_nsub >= 3 && NG <= 8);
    const bool ext_topk = (!want_lane && K > 256) || g_sw.ext_topk;
    long long stream_cap = 0;
    if (want_auc || ext_topk) {
        stream_cap = stream_budget_bytes(ws) / (stream_ld_max * (long long)sizeof(T));
        // (a pass over a subset of the users -- the exact second pass of the fp32 tie noise -- stores rows for that subset only)
        const long long m_rows = c.eval_users >= 0 ? std::min<long long>(c.eval_users, m) : m;
        if (ext_topk) {
            if (m_rows > stream_cap) throw RmError{RM_ERR_NOMEM, "k_metrics > 256 keeps one score row (" + std::to_string(stream_ld_max * (long long)sizeof(T)) +
                                       " B) per user in device memory: " + std::to_string(m_rows) + " users do not fit, at most " + std::to_string(stream_cap) + " per call"};
            ca.force_stream = 1;
        } else if (stream_cap > 0) ca.allow_stream = 1;
    }
    const long long slot_bound = (long long)m + c.nnz_test / POS_CHUNK + 1;
    const long long group_bound = slot_bound / GU + 2;
    int *slot_user = (int *)ws.get("slot_user", sizeof(int) * (size_t)slot_bound);
    int *slot_chunk = (int *)ws.get("slot_chunk", sizeof(int) * (size_t)slot_bound);
    int *slot_index = (int *)ws.get("slot_index", sizeof(int) * (size_t)slot_bound);
    unsigned char *slot_j = (unsigned char *)ws.get("slot_j", (size_t)slot_bound);
    int *gj = (int *)ws.get("gj", sizeof(int) * (size_t)group_bound);
    long long *grow = (long long *)ws.get("grow", sizeof(long long) * (size_t)group_bound);
    int *sc_user = (int *)ws.get("sc_user", sizeof(int) * (size_t)slot_bound);
    int *sc_chunk = (int *)ws.get("sc_chunk", sizeof(int) * (size_t)slot_bound);
    int *tile_total = nullptr, *tile_offset = nullptr;
    const int n_tiles = (int)cdiv(m, 1024);
    if (m > 8192) {                                          // one block walking the whole array costs ~0.5 us per 1024 entries
        tile_total = (int *)ws.get("scan_tile_total", sizeof(int) * (size_t)n_tiles);
        tile_offset = (int *)ws.get("scan_tile_offset", sizeof(int) * (size_t)n_tiles);
    }
    const bool items_known = c.items_tag != 0 && c.items_tag == cx.packed_tag;       // a later batch of the same host call
    // log2(i + 2) for the DCG discounts, from the host's libm like the reference's (:620,:902, int -> double log2).  The table
    // depends on K alone: it stays in the workspace, and only a longer one (or a moved buffer) is uploaded again -- the copy
    // comes from pageable memory, which blocks the host and waits for the stream
    double *log2tab = (double *)ws.get("log2tab", sizeof(double) * (size_t)K);
    if (cx.log2_ptr != (const void *)log2tab || cx.log2_K < K) {
        std::vector<double> lt((size_t)K);
        for (int i = 0; i < K; i++) lt[i] = std::log2(i + 2);
        HIP_CHECK(hipMemcpy(log2tab, lt.data(), sizeof(double) * (size_t)K, hipMemcpyHostToDevice));       // (synchronous: `lt` is on the stack)
        cx.log2_ptr = (const void *)log2tab; cx.log2_K = K;
    }
    if (!cx.pinned_plan) HIP_CHECK(hipHostMalloc((void **)&cx.pinned_plan, sizeof(Plan), hipHostMallocDefault));
    // (one answer per call: a later pass -- the exact passes of the tie noise, on this or on a peer context with less free memory --
    // carries the first pass's answer; rows handed over by another pass are proof that they fit)
    const bool dense_ok = std::is_same<T, float>::value && (c.ext_bits ? true : c.dense_fit >= 0 ? c.dense_fit != 0 : dense_rows_fit(ws, m, n));
    bool bits_early = false, bits_early_masked = false, masked_from_bits = false;
    // the positives' stream: the streamed users' chain beside the table users', and what is made per test entry beside the plan chain
    struct PosGuard { hipStream_t st = nullptr; ~PosGuard() { if (st) (void)hipStreamSynchronize(st); } } pos_guard;     // (an error between fork and join)
    auto pos_stream = [&]() {
        if (!cx.pos_stream) {
            HIP_CHECK(create_stream(&cx.pos_stream, cx.high_priority));
            for (int i = 0; i < 5; i++) HIP_CHECK(hipEventCreateWithFlags(&cx.pos_ev[i], hipEventDisableTiming));
        }
        return cx.pos_stream;
    };
    const bool flat_early = want_auc && !c.only_users && c.nnz_test > 0 && !g_sw.no_pos_flat;
    int *ent_user = nullptr; unsigned char *ent_masked = nullptr;
    PosArgs<T> pf{};
    if (flat_early) {
        ent_user = (int *)ws.get("ent_user", sizeof(int) * (size_t)c.nnz_test);
        ent_masked = (unsigned char *)ws.get("ent_masked", (size_t)c.nnz_test);
        pf.m = m; pf.n = n; pf.k = k; pf.A = c.A; pf.lda = c.lda; pf.B = c.B; pf.ldb = c.ldb;
        pf.train_p = c.train_p; pf.train_i = c.train_i; pf.test_p = c.test_p; pf.test_i = c.test_i; pf.flags = flags;
        pf.pos_tmp = (T *)ws.get("pos_tmp", sizeof(T) * (size_t)c.nnz_test);
        if (sizeof(T) == 4 && !g_sw.no_pos_keys) pf.pos_key = (unsigned long long *)ws.get("pos_key", 8 * ((size_t)c.nnz_test + 8));
        pf.noise_row = c.noise_row; pf.noise_row0 = c.noise_row0; pf.noise_E = c.noise_E; pf.noise_ld = c.noise_ld;
        pf.noise_flag = c.noise_flag; pf.plan = plan;
    }
    auto launch_flat = [&](hipStream_t es) {
        hipLaunchKernelGGL(k_pos_scores_flat<T>, dim3(cdiv(c.nnz_test, POSF_WAVES * WAVE)), dim3(POSF_WAVES * WAVE), 0, es, pf, ent_user);
        // test items that are train items: +inf, once the answer (the dense train rows' kernel, or k_test_masked) is there
        if (masked_from_bits) HIP_CHECK(hipStreamWaitEvent(es, cx.side_ev[5], 0));
        else hipLaunchKernelGGL(k_test_masked, dim3(cdiv(m, TM_USERS)), dim3(256), 0, es, m, c.test_p, c.test_i, c.train_p, c.train_i, ent_user, ent_masked, plan);
        hipLaunchKernelGGL(k_pos_apply_masked<T>, dim3(cdiv(c.nnz_test, 256)), dim3(256), 0, es, pf, ent_masked);
    };
    Plan hp;
    for (int attempt = 0; ; attempt++) {
        // ---- the plan chain: five launches that depend on one another, on the call's stream (index pointers only) ----
        if (attempt == 0) HIP_CHECK(hipMemsetAsync(plan, 0, sizeof(Plan), stream));
        else {
            // (a second plan keeps the count of the users the tie noise's first pass has flagged so far: the positives' scores of the
            // first attempt, which count them, are not made again)
            const size_t at = offsetof(Plan, n_noise_flagged);
            HIP_CHECK(hipMemsetAsync(plan, 0, at, stream));
            HIP_CHECK(hipMemsetAsync((char *)plan + at + sizeof(int), 0, sizeof(Plan) - at - sizeof(int), stream));
        }
        hipLaunchKernelGGL(k_classify, dim3(cdiv(m, 1024)), dim3(1024), 0, stream, ca);
        // (the two forks behind k_classify, recorded BEFORE the rest of the chain is enqueued: the host needs ~4 us per launch, and the
        // chain's kernels used to reach the device 60 us late, behind everything that was enqueued for the other streams)
        if (use_side) {
            side_stream(); pos_stream();
            HIP_CHECK(hipEventRecord(cx.side_ev[0], stream));
            HIP_CHECK(hipEventRecord(cx.pos_ev[0], stream));
        }
        if (tile_total) {
            hipLaunchKernelGGL(k_scan_tiles, dim3(n_tiles), dim3(1024), 0, stream, user_nslots, uslot_base, m, tile_total);
            hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, stream, tile_total, tile_offset, n_tiles, &plan->n_slots, plan, GU);
        } else {
            hipLaunchKernelGGL(k_scan_exclusive, dim3(1), dim3(1024), 0, stream, user_nslots, uslot_base, m, &plan->n_slots, plan, GU);
        }
        AssignArgs aa{m, c.test_p, flags, user_nslots, uslot_base, want_auc ? 1 : 0, plan, slot_user, slot_chunk, slot_index, slot_j, sc_user, sc_chunk,
                      ext_topk ? 1 : 0, tile_offset};
        hipLaunchKernelGGL(k_assign_slots, dim3(cdiv(m, ASSIGN_THREADS)), dim3(ASSIGN_THREADS), 0, stream, aa);
        hipLaunchKernelGGL(k_block_tables, dim3(1), dim3(1024), 0, stream, plan, slot_j, gj, grow, GU);
        // ---- beside it: what reads the index arrays and the factors ----
        // On the side stream the CSR rows' validation (gated on the index pointers k_classify has just checked), then the dense train
        // rows (fp32, small item counts; set_train_bits), which depend on the CSR inputs alone: they run during the rest of the plan
        // chain, the read-back -- the host's one wait of the call -- and the positives' scores.  Whether the rows also mark the test
        // items (`mask_test`) is only decided behind the read-back; the guess here is the usual answer, and a wrong guess costs one
        // more launch of the kernel behind it.  On the positives' stream max |A| and max |B| and the user of every test entry.  The
        // plan carries the checks' verdicts and the maxima: its read-back waits for both streams.
        hipStream_t aux = stream, aux2 = stream;
        if (use_side) {
            aux = cx.side_stream; aux2 = cx.pos_stream;
            HIP_CHECK(hipStreamWaitEvent(aux, cx.side_ev[0], 0)); side_guard.pending++;
            HIP_CHECK(hipStreamWaitEvent(aux2, cx.pos_ev[0], 0)); pos_guard.st = aux2;
        }
        if (!c.csr_checked) launch_csr_index_checks(m, n, c.train_p, c.train_i, c.nnz_train, c.test_p, c.test_i, c.nnz_test, plan, aux);
        if (use_side) HIP_CHECK(hipEventRecord(cx.side_ev[2], aux));
        hipLaunchKernelGGL(k_absmax<T>, dim3(512), dim3(256), 0, aux2, c.A, c.lda, (long long)m, k, &plan->amax_a, &plan->nonfinite);
        if (!items_known) hipLaunchKernelGGL(k_absmax<T>, dim3(1024), dim3(256), 0, aux2, c.B, c.ldb, (long long)n, k, &plan->amax_b, &plan->nonfinite_b);
        if (use_side) HIP_CHECK(hipEventRecord(cx.pos_ev[2], aux2));
        // (the user of every test entry, for the positives' scores by entry: index pointers only)
        if (attempt == 0 && flat_early) hipLaunchKernelGGL(k_entry_users, dim3(cdiv(cdiv(m, WAVE) * WAVE, 256)), dim3(256), 0, aux2, m, c.test_p, ent_user, plan);
        // (463 MB of writes at BASELINE C2; four resident blocks per CU leave half of the wave slots to the plan's kernels and the
        // read-back's copy.  With the positives' scores by entry the kernel also says which test items are train items, `ent_masked`:
        // a bit of the row it has just built.)
        if (attempt == 0 && std::is_same<T, float>::value && use_side && !c.ext_bits && dense_ok && !g_sw.no_early_bits) {
            SweepArgs probe{};
            const bool guess = want_auc && !ext_topk && !g_sw.no_test_mask;
            const unsigned *had = (const unsigned *)cx.bits_ptr;
            const bool reuse = c.same_train_rows && had && !cx.bits_partial && cx.bits_words == dense_row_words(n) && cx.bits_m == m && cx.bits_masked == guess &&
                               had == (const unsigned *)ws.get("train_bits", (size_t)m * (size_t)dense_row_words(n) * 4);
            if (!reuse) {
                set_train_bits(probe, cx, c, m, n, dense_ok, aux, guess, false, false, flat_early ? ent_masked : nullptr);
                HIP_CHECK(hipEventRecord(cx.side_ev[4], aux));
                HIP_CHECK(hipEventRecord(cx.side_ev[5], aux));
                bits_early = true; bits_early_masked = guess; masked_from_bits = flat_early;
            }
        }
        if (use_side) { HIP_CHECK(hipStreamWaitEvent(stream, cx.side_ev[2], 0)); HIP_CHECK(hipStreamWaitEvent(stream, cx.pos_ev[2], 0)); }
        HIP_CHECK(hipMemcpyAsync(cx.pinned_plan, plan, sizeof(Plan), hipMemcpyDeviceToHost, stream));
        if (attempt == 0 && flat_early) {
            // The scores of the test entries (k_pos_scores_flat) depend on the inputs and on the users' flags alone: they run on the
            // positives' stream beside the read-back and the host's work behind it.  They index the item factors by the test items, so
            // they follow the index checks (and return when those found a defect) -- and they follow the plan's last kernel and the
            // copy: their blocks take every wave slot they find, and a plan kernel's block of 1,024 threads then waits for sixteen
            // slots of one CU to fall free at once (measured: the read-back 0.2 ms late).  Without dense train rows k_test_masked says
            // which test items are train items, behind the scores.
            if (use_side) {
                HIP_CHECK(hipEventRecord(cx.pos_ev[3], stream));
                HIP_CHECK(hipStreamWaitEvent(cx.pos_stream, cx.pos_ev[3], 0));
            }
            launch_flat(aux2);
            if (use_side) HIP_CHECK(hipEventRecord(cx.pos_ev[1], cx.pos_stream));
        }
        RM_TRACE_POINT("run: plan chain + side kernels enqueued");
        HIP_CHECK(hipStreamSynchronize(stream));
        RM_TRACE_POINT("run: plan read back");
        // (the stream has waited for the checks and the maxima: only the dense train rows of the first attempt may still be running on
        // the side stream, only the positives' scores on theirs)
        if (use_side && !(attempt == 0 && bits_early)) side_guard.pending--;
        if (use_side && !flat_early) pos_guard.st = nullptr;
        hp = *cx.pinned_plan;
        throw_csr_defects(hp, c, cx);
        // the streamed users' score rows must fit the budget; if not (memory pressure), plan again with those users in chunks
        if (ca.allow_stream && !ca.force_stream && hp.class_count[STREAM_CLASS] > stream_cap && attempt == 0) {
            ca.allow_stream = 0; ca.check_ptr = 0;
            // (attempt 0's kernels on BOTH side streams read `flags` and `plan`, which the second plan rewrites: wait for them)
            if (use_side) {
                HIP_CHECK(hipStreamSynchronize(cx.side_stream)); side_guard.pending = bits_early ? 1 : 0;
                if (cx.pos_stream) HIP_CHECK(hipStreamSynchronize(cx.pos_stream));
            }
            continue;
        }
        break;
    }
    throw_csr_defects(hp, c, cx);

    const int n_slots = hp.n_slots, n_groups = hp.n_groups;
    const int jmax = want_auc ? hp.jmax : 0;
    // streamed users own the last slots; the tables and their kernels cover slots [0, stream_slot0)
    const int n_stream = (want_auc || ext_topk) ? hp.class_count[STREAM_CLASS] : 0;
    const int stream_slot0 = n_stream > 0 ? hp.class_offset[STREAM_CLASS] : n_slots;
    // |any partial sum| <= k * max|A| * max|B|: if that is comfortably finite in T, no score is NaN / Inf
    if (items_known) { hp.amax_b = cx.packed_amax_b; hp.nonfinite_b = cx.packed_nonfinite_b; }
    else { cx.packed_amax_b = hp.amax_b; cx.packed_nonfinite_b = hp.nonfinite_b; }
    double amax_a, amax_b;
    std::memcpy(&amax_a, &hp.amax_a, 8); std::memcpy(&amax_b, &hp.amax_b, 8);
    const double tmax = std::is_same<T, float>::value ? 3.0e38 : 1.0e308;
    const bool check_nan = hp.nonfinite || hp.nonfinite_b || !((double)k * amax_a * 1.001 < tmax / std::max(amax_b, 1e-300));
    const int n_ublocks = (n_groups + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK;

    // ---- sweep geometry ----
    // (each group's positives table is aligned to its own size, 2^j rows of GU scores: worst-case padding = one table)
    // sub-tiles per step (waves per block = 4 nsub): three when the fp32 kernel for <= 64 factors keeps its lists in LDS
    // next to the larger item tile -- the third wave per SIMD fills the vector pipe the epilogue leaves idle
    int nsub = 2;
    auto lds_need_j = [&](bool with_lists, int ns, int j) {                      // LDS of a block of depth j
        const size_t head = P::lds_b(NG, 32 * ns) + (with_lists ? P::lists_b(ns, K) : 0);
        const size_t tb = ((size_t)1 << j) * GU * sizeof(T);
        return want_auc ? (head + tb - 1) / tb * tb + (size_t)GROUPS_PER_BLOCK * (1 << j) * GU * (sizeof(T) + 4) : head;
    };
    auto lds_need_n = [&](bool with_lists, int ns) { return lds_need_j(with_lists, ns, jmax); };
    // (three sub-tiles with the lane buffers of larger k_metrics were built and measured in round 6 -- 162-168 VGPRs, no spills -- and are
    // SLOWER: a user's candidates are then spread over three waves whose bounds each see a third of the items; BASELINE C2's shape at
    // K = 100: 13.4 against 12.4 ms, profiles/r6_ab_c2.txt)
    if (!ext_topk && !want_lane && P::max_nsub >= 3 && NG <= 8 && lds_need_n(true, 3) + SYNC_BYTES <= LDS_LIMIT && !g_sw.hbm_lists && !g_sw.nsub2)
        nsub = 3;
    // (four sub-tiles -- sixteen waves, four per SIMD, 128 registers each -- were built in round 5, passed the parity tests and were
    // 1.9 % SLOWER at BASELINE C2: profiles/r5_ab_c2.txt r5a; the patches are scratch/dropped/r5_nsub4*)
    const int tile_items = 32 * nsub, n_waves = 4 * nsub;
    const int tiles_total = (n + tile_items - 1) / tile_items;
    // LDS admits one block per CU, so the grid runs in rounds of 256 blocks, and a block costs its tiles plus a fixed part:
    //   block(s) = tiles_total / s + 13 (tables into LDS, the top-K lists filling up, list sort, histogram flush)
    //              + 0.4 % of the sweep per 10 of k_metrics (every item range restarts the streaming top-K lists)
    // (coefficients measured on MI355X, profiles/r2_splits*: C2, NS, C3- and C4-shaped runs).  One split count for the whole
    // grid leaves the last round partly empty -- 8.45 rounds cost 9 at C2 -- so the grid has two levels: whole rounds of
    // blocks with n_splits item ranges, then the `tail_ublocks` cheapest user blocks cut into `tail_splits` (more, smaller)
    // ranges that fill the last round.  Blocks differ in duration by ~ +-20 % (depth of the positive trees, streamed
    // users), so a part that ends the grid is charged max(whole rounds, rounds + 0.2).
    int n_splits = 1, tail_ublocks = 0, tail_splits = 0;
    if (n_ublocks > 0) {
        const int n_cu = 256;
        // (ranges of at least 16 tiles: a call with few users -- the second, exact pass of the tie noise over the flagged
        // users, a small batch -- is cut into many ranges so that its few user blocks still fill the chip)
        const int max_splits = std::max(1, std::min(MAX_PARTS / nsub, tiles_total / 16));
        const int max_tail = max_splits;
        const double fixed = 13.0 + 0.004 * std::max(0.25, K / 10.0) * tiles_total;
        auto block = [&](int sct) { return (double)tiles_total / sct + fixed; };
        // (0.2 of a round with the lane buffers, whose blocks differ by their selections; 0.1 with the lists: round 6, scratch/r6_slack.sh --
        // tutorial shape 2.57 -> 2.54 ms, 20,000 users of C2 1.14 -> 1.11, north-star shape 77.3 -> 77.2; C4 / C5 lose 5 % / 1 % below 0.2)
        const double slack = g_sw.split_slack >= 0 ? g_sw.split_slack / 100.0 : (want_lane ? 0.2 : 0.1);
        auto last = [&](double rounds) { return std::max(std::ceil(rounds - 1e-9), rounds + slack); };
        double best = 1e300;
        for (int sct = 1; sct <= max_splits; sct++) {
            const double t1 = last((double)n_ublocks * sct / n_cu) * block(sct);
            if (t1 < best - 1e-9) { best = t1; n_splits = sct; tail_ublocks = 0; tail_splits = 0; }
            const long long full = (long long)n_ublocks * sct / n_cu;               // whole rounds of the main part
            const int main_ub = (int)(full * n_cu / sct), tail_ub = n_ublocks - main_ub;
            if (full == 0 || tail_ub == 0) continue;
            for (int ts = sct + 1; ts <= max_tail; ts++) {
                const double t2 = (double)main_ub * sct / n_cu * block(sct) + last((double)tail_ub * ts / n_cu) * block(ts);
                if (t2 < best * 0.99) { best = t2; n_splits = sct; tail_ublocks = tail_ub; tail_splits = ts; }
            }
        }
    }
    if (const char *e = g_sw.splits.empty() ? nullptr : g_sw.splits.c_str()) {                                // A/B timing and tests: "S" or "S,tail_ublocks,tail_splits"
        int v[3] = {1, 0, 0};
        sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]);
        n_splits = std::max(1, std::min(v[0], std::max(1, MAX_PARTS / nsub)));
        tail_ublocks = std::max(0, std::min(v[1], n_ublocks)); tail_splits = std::max(1, std::min(v[2], std::max(1, MAX_PARTS / nsub)));
        if (tail_ublocks == 0) tail_splits = 0;
    }
    const int part_splits = std::max(n_splits, tail_splits);
    // fp32, small item counts (dense train rows), ROC / PR-AUC wanted: the rows mark the users' test items too, the sweep never
    // meets a candidate that IS one of the user's positives (an exact tie, resolved by the tie rule three tiles out of four at
    // 27k items), and k_merge_positives puts the test items back.  Not when a score can be non-finite (a test item masked by the
    // train row is marked +inf in the tables), not with chunked long rows (more slots than users: their best positives are not in the primary slot),
    // not beyond the lists (k_select_topk works on the stored rows).
    bool mask_test = std::is_same<T, float>::value && want_auc && !ext_topk && !check_nan && n_slots > 0 &&
                     dense_ok && hp.n_slots == hp.n_active && hp.n_only_ndcg == 0 &&
                     nsub * part_splits + 1 <= MAX_PARTS && !g_sw.no_test_mask;
    // rows handed over by another pass: usable when they were built the way this pass would build them (unmasked rows are
    // always valid: the old scheme)
    const bool use_ext_bits = c.ext_bits != nullptr && c.ext_words == dense_row_words(n) && (!c.ext_masked || mask_test);
    if (use_ext_bits) mask_test = c.ext_masked;
    const int n_part = nsub * part_splits + (mask_test ? 1 : 0);
    const int part_extra = mask_test ? 1 : 0;
    auto lds_need = [&](bool with_lists) { return lds_need_n(with_lists, nsub); };
    const bool list_in_lds = !ext_topk && !want_lane && lds_need(true) + SYNC_BYTES <= LDS_LIMIT && !g_sw.hbm_lists;
    size_t lds_total = lds_need(list_in_lds);
    // per-lane pending buffers for top-K candidates behind everything else when 2..8 keys per lane still fit
    // (fp32: not for the append-buffer lists of K > 32, whose appends are already single stores)
    const bool want_pending = !ext_topk && !want_lane && P::has_pending && !g_sw.no_pending;
    const size_t per_key = (size_t)n_waves * WAVE * P::pend_key_bytes;            // one key per lane and wave
    int pend_cap = 0; size_t pend_off = 0, sync_off = 0;
    if (P::block_carve) {
        // the kernel sizes its tables per block (by the block's own depth) and computes the pending capacity from what
        // is left below the counter: allocate for the deepest block plus, if it still fits, 8 keys per lane
        pend_cap = want_pending ? P::pend_cap_max : 0;
        lds_total = std::min<size_t>(LDS_LIMIT, lds_total + SYNC_BYTES + pend_cap * per_key);
        sync_off = lds_total - SYNC_BYTES;                         // split-barrier counter of the sweep, last 16 bytes
    } else {
        sync_off = lds_total; lds_total += SYNC_BYTES;
        pend_off = lds_total;
        if (want_pending) {
            pend_cap = (int)std::min<size_t>(8, (LDS_LIMIT - lds_total) / per_key);
            if (pend_cap < 2) pend_cap = 0;
            lds_total += pend_cap * per_key;
        }
    }

    Entry<T> *merged = (Entry<T> *)ws.get("merged", sizeof(Entry<T>) * (size_t)m * K);
    // the argument block of the finalisation kernels is filled in as the pieces come into being
    long long *rank_sorted = nullptr;
    if (c.pos_rank) rank_sorted = (long long *)ws.get("rank_sorted", sizeof(long long) * (size_t)std::max<long long>(c.nnz_test, 1));
    if (rank_sorted) HIP_CHECK(hipMemsetAsync(rank_sorted, 0, sizeof(long long) * (size_t)std::max<long long>(c.nnz_test, 1), stream));
    FinalArgs<T, T> fa{};
    fa.m = m; fa.n = n; fa.K = K; fa.req = req; fa.cumulative = c.cumulative ? 1 : 0; fa.noise = c.noise ? 1 : 0; fa.gu = GU;
    fa.train_p = c.train_p; fa.test_p = c.test_p; fa.test_i = c.test_i; fa.test_v = c.test_v;
    fa.flags = flags; fa.user_nslots = user_nslots; fa.uslot_base = uslot_base; fa.slot_index = slot_index;
    fa.gj = gj; fa.grow = grow; fa.log2tab = log2tab;
    fa.p = c.out[0]; fa.tp = c.out[1]; fa.r = c.out[2]; fa.ap = c.out[3]; fa.tap = c.out[4];
    fa.ndcg = c.out[5]; fa.hit = c.out[6]; fa.rr = c.out[7]; fa.roc = c.out[8]; fa.pr = c.out[9];
    fa.merged = merged; fa.rank_sorted = rank_sorted; fa.status = c.status;
    fa.noise_flag = c.noise_flag; fa.plan = plan;
    fa.n_slots = n_slots; fa.slot_user = slot_user; fa.slot_chunk = slot_chunk;
    fa.stream_slot0 = stream_slot0;
    if (want_auc && n_slots > 0) fa.auc_part = (AucPart *)ws.get("auc_part", sizeof(AucPart) * (size_t)n_slots);
    // ideal-DCG values of the users with very long test rows (k_top_values: a wavefront per such user, a chain of K dependent
    // rounds -- latency, 0.16 ms at BASELINE C2): they depend on the test rows alone, so they are computed on the side stream
    // beside the preparation kernels instead of between the sweep and k_finalize, which waits for them
    // (launched BEHIND the operand packing on the side stream, which the sweep waits for: with k_metrics > 64 every row of more than
    // 64 test items is listed and the kernel is half a millisecond at BASELINE C2's shape)
    bool topv_pending = false;
    fa.heavy_npos = ca.heavy_npos;
    auto launch_top_values = [&]() {
        if (!(fa.ndcg && hp.n_heavy > 0)) return;
        // (min(K, longest test row) values per user: with a cap of 256 the rows beyond it fell back to k_finalize's repeated selection on
        // one thread -- L x positives dependent loads: 160 of the 190 ms of a K = 300 step at BASELINE C2's shape)
        fa.heavy_ld = std::max(1, std::min(K, hp.max_npos));
        fa.heavy_topv = (T *)ws.get("heavy_topv", sizeof(T) * (size_t)m * (size_t)fa.heavy_ld);
        fa.heavy_nan = (unsigned char *)ws.get("heavy_nan", (size_t)m);
        fa.heavy_users = heavy_users; fa.n_heavy = hp.n_heavy;
        hipStream_t tv_stream = stream;
        if (use_side) tv_stream = fork_side();
        hipLaunchKernelGGL((k_top_values<T, T>), dim3(cdiv((long long)hp.n_heavy * WAVE, 256)), dim3(256), 0, tv_stream, fa);
        if (use_side) { HIP_CHECK(hipEventRecord(cx.side_ev[3], tv_stream)); topv_pending = true; }
    };
    const int stream_parts = (int)cdiv(n, STREAM_RANK_THREADS * STREAM_RANK_ITEMS);
    const int stream_ipt = ((int)cdiv(n, (long long)stream_parts * STREAM_RANK_THREADS) + 7) / 8 * 8;     // equal pieces of the row
    auto rank_streamed_rows = [&](int r0, int r1, hipStream_t st) {
        if (r1 <= r0) return;
        hipLaunchKernelGGL((k_rank_streamed<T, T>), dim3((unsigned)((long long)(r1 - r0) * stream_parts)), dim3(STREAM_RANK_THREADS), 0, st, fa, stream_parts, stream_ipt, r0);
    };
    auto auc_streamed_rows = [&](int r0, int r1, hipStream_t st) {
        if (r1 <= r0) return;
        hipLaunchKernelGGL((k_auc_streamed<T, T>), dim3(cdiv((long long)(r1 - r0) * WAVE, 256)), dim3(256), 0, st, fa, r0, r1);
    };
    T *pos_score = nullptr; unsigned *hist = nullptr; int *pos_order = nullptr, *pos_item = nullptr;
    Entry<T> *pl = nullptr; PartialStat<T> *pst = nullptr;
    T *stream_scores = nullptr, *spos_score = nullptr; int *spos_item = nullptr; unsigned *shist = nullptr;
    const long long stream_ld = (long long)tiles_total * tile_items;
    struct { bool on = false; const char *glists = nullptr; const int *lane_cnt = nullptr; const void *thr = nullptr; CollectGeom g{}; } collect;

    if (n_slots > 0) {
        // ---- pack operands into the MFMA images ----
        const long long bp_units = P::items_units(tiles_total, NG, tile_items), ap_units = P::users_units(n_groups, NG);
        typename P::PackT *Bp = (typename P::PackT *)ws.get("Bp", 16 * (size_t)bp_units);
        typename P::PackT *Ap = (typename P::PackT *)ws.get("Ap", 16 * (size_t)ap_units);
        // the packed item image survives between the batches of one host call (same B, same geometry)
        const bool items_packed = items_known && cx.packed_tile == tile_items && cx.packed_ng == NG && cx.packed_ptr == (const void *)Bp;
        // Only the sweep reads the packed images: with a side stream they are made there, behind the dense train rows and beside the
        // positives' kernels (120 us of BASELINE C2's preparation that sat in front of k_pos_scores); the sweep's launch waits for both.
        const bool packs_side = use_side && want_auc && cx.side_stream != nullptr && !g_sw.no_pack_beside;
        pack_operands(c.A, c.lda, c.B, c.ldb, n, k, NG, tile_items, slot_user, n_slots, Ap, ap_units, Bp, bp_units, packs_side ? cx.side_stream : stream, !items_packed);
        if (packs_side) {
            HIP_CHECK(hipEventRecord(cx.side_ev[4], cx.side_stream));   // (behind the rows, when they were launched: one event for both)
            if (!bits_early) side_guard.pending++;
        }
        cx.packed_tag = c.items_tag; cx.packed_tile = tile_items; cx.packed_ng = NG; cx.packed_ptr = (const void *)Bp;
        launch_top_values();

        // ---- dense train rows (fp32, small item counts) ----
        // (measured: on the side stream beside the positives' kernels they gain nothing -- both are bound by memory; r3_ab_c2.txt)
        typename P::Args sa{};
        // (rows launched beside the plan read-back that turn out not to be the ones wanted -- a wrong guess of `mask_test`, rows handed over
        // by another pass -- are waited for here, before anything is launched over them; the usual case waits in front of the sweep)
        bool bits_wait = bits_early || packs_side;
        if (bits_early && (bits_early_masked != mask_test || (use_ext_bits && dense_ok))) {
            HIP_CHECK(hipStreamWaitEvent(stream, cx.side_ev[4], 0));
            side_guard.pending--; bits_wait = false;
        }
        if (use_ext_bits && dense_ok) set_ext_bits(sa, c.ext_bits, (int)c.ext_words);
        else set_train_bits(sa, cx, c, m, n, dense_ok, stream, mask_test, bits_early, bits_early_masked);

        // ---- positives ----
        if (want_auc) {
            const long long rows = hp.total_rows + n_groups;       // 2^j rows per group (one +inf pad row each)
            pos_score = (T *)ws.get("pos_score", sizeof(T) * (size_t)(rows + 1) * GU);
            pos_item = (int *)ws.get("pos_item", sizeof(int) * (size_t)(rows + 1) * GU);
            hist = (unsigned *)ws.get("hist", sizeof(unsigned) * (size_t)(rows + 1) * GU);
            T *pos_tmp = (T *)ws.get("pos_tmp", sizeof(T) * (size_t)std::max<long long>(c.nnz_test, 1));
            pos_order = (int *)ws.get("pos_order", sizeof(int) * (size_t)std::max<long long>(c.nnz_test, 1));
            PosArgs<T> pa{m, n, k, c.A, c.lda, c.B, c.ldb, c.train_p, c.train_i, c.test_p, c.test_i,
                          flags, user_nslots, uslot_base, slot_index, grow, pos_tmp, pos_order, pos_score, pos_item, GU};
            pa.noise_row = c.noise_row; pa.noise_row0 = c.noise_row0; pa.noise_E = c.noise_E; pa.noise_ld = c.noise_ld;
            pa.noise_flag = c.noise_flag; pa.plan = plan;
            if (sizeof(T) == 4 && !g_sw.no_pos_keys)
                pa.pos_key = (unsigned long long *)ws.get("pos_key", 8 * ((size_t)std::max<long long>(c.nnz_test, 1) + 8));
            // Scores of the test entries: by entry (k_pos_scores_flat, launched beside the plan chain: every lane busy) unless the call
            // evaluates a few of its users only -- then a wavefront per slot is less work.
            const bool flat = flat_early;
            // The streamed users' positives (the all-pairs rank of long test rows: vector work) run on a stream of their own beside
            // the table users'
            hipStream_t ps = stream;
            const bool pos_beside = use_side && n_stream > 0 && stream_slot0 > 0 && !g_sw.no_pos_beside;
            if (pos_beside) {
                // (not the side stream: that one carries the dense train rows and the operand packing, 0.3 ms the streamed users' chain
                // used to queue behind.  The host has waited for the call's stream since: nothing to wait for over there.)
                ps = pos_stream();
                pos_guard.st = ps;
            }
            PosArgs<T> pb = pa;
            // (with the scores by entry the positives' stream is busy with them: the tables are filled on the call's stream meanwhile)
            const hipStream_t fill = (flat && pos_beside) ? stream : ps;
            if (n_stream > 0) {
                const size_t nz = (size_t)std::max<long long>(c.nnz_test, 1);
                spos_score = (T *)ws.get("spos_score", sizeof(T) * nz);
                spos_item = (int *)ws.get("spos_item", sizeof(int) * nz);
                shist = (unsigned *)ws.get("shist", sizeof(unsigned) * nz);
                pb.stream = 1; pb.spos_score = spos_score; pb.spos_item = spos_item;
                // (+inf in every rank: entries that repeat an item -- a non-canonical CSR row -- share a rank and leave one unused)
                hipLaunchKernelGGL(k_init_tables<T>, dim3(cdiv((long long)nz, 256)), dim3(256), 0, fill, spos_score, shist, (long long)nz);
            }
            const int nsc = hp.n_stream_chunks;
            hipLaunchKernelGGL(k_init_tables<T>, dim3(cdiv(rows * GU, 256)), dim3(256), 0, stream, pos_score, hist, rows * GU);
            if (flat) {
                if (pos_beside && n_stream > 0) { HIP_CHECK(hipEventRecord(cx.pos_ev[2], stream)); HIP_CHECK(hipStreamWaitEvent(ps, cx.pos_ev[2], 0)); }
                // (the scores are the last thing on the positives' stream: whoever is not on it waits for them)
                if (use_side) { HIP_CHECK(hipStreamWaitEvent(stream, cx.pos_ev[1], 0)); if (!pos_beside) pos_guard.st = nullptr; }
            } else {
                if (n_stream > 0) hipLaunchKernelGGL(k_pos_scores<T>, dim3(cdiv(nsc, POSS_WAVES)), dim3(POSS_WAVES * WAVE), 0, ps, pb, sc_user, sc_chunk, nsc);
                if (stream_slot0 > 0) hipLaunchKernelGGL(k_pos_scores<T>, dim3(cdiv(stream_slot0, POSS_WAVES)), dim3(POSS_WAVES * WAVE), 0, stream, pa, slot_user, slot_chunk, stream_slot0);
            }
            if (n_stream > 0) hipLaunchKernelGGL(k_pos_place<T>, dim3(cdiv((long long)nsc * WAVE, 256)), dim3(256), 0, ps, pb, sc_user, sc_chunk, nsc);
            if (stream_slot0 > 0) hipLaunchKernelGGL(k_pos_place<T>, dim3(cdiv((long long)stream_slot0 * WAVE, 256)), dim3(256), 0, stream, pa, slot_user, slot_chunk, stream_slot0);
            if (pos_beside) { HIP_CHECK(hipEventRecord(cx.pos_ev[4], ps)); HIP_CHECK(hipStreamWaitEvent(stream, cx.pos_ev[4], 0)); pos_guard.st = nullptr; }
        }

        if (n_stream > 0) stream_scores = (T *)ws.get("stream_scores", sizeof(T) * (size_t)n_stream * (size_t)stream_ld);
        if (!ext_topk) pl = (Entry<T> *)ws.get("pl", sizeof(Entry<T>) * (size_t)n_slots * n_part * K);
        pst = (PartialStat<T> *)ws.get("pst", sizeof(PartialStat<T>) * (size_t)n_slots * n_part);
        typename P::ListT *glists = nullptr;
        const unsigned n_blocks = (unsigned)((n_ublocks - tail_ublocks) * n_splits + tail_ublocks * tail_splits);
        const bool lane_lists = want_lane && !ext_topk;                     // per-lane append buffers + k_collect_topk (rm_list.hpp)
        int lane_cap = lane_lists ? P::lane_cap(K) : 0;
        if (lane_lists && g_sw.lane_cap_set > 0) lane_cap = (int)std::min<long long>(std::max<long long>(P::lane_cap(K), (g_sw.lane_cap_set + 15) / 16 * 16), (COLLECT_MAX_ENTRIES - K) / 16 * 16);
        if (lane_lists && g_sw.lane_cap_min >= 0)        // K + slack survivors and one tile's appends (16: both precisions' bound) in one lane
            lane_cap = std::min(lane_cap, (K + lane_sel_slack(K) + 16 + 1 + 15) / 16 * 16);
        int *lane_cnt = nullptr;
        if (lane_lists) {
            glists = (typename P::ListT *)ws.get("glists", (size_t)n_blocks * n_waves * WAVE * (size_t)lane_cap * (sizeof(T) + 4));
            lane_cnt = (int *)ws.get("lane_cnt", sizeof(int) * (size_t)n_blocks * n_waves * WAVE);
        } else if (!list_in_lds && !ext_topk) {
            glists = (typename P::ListT *)ws.get("glists", sizeof(typename P::ListT) * (size_t)n_blocks * n_waves * GU * (2 * K + 32));
        }

        typedef typename std::remove_pointer<decltype(typename P::Args{}.thr_shared)>::type ThrT;
        ThrT *thr_shared = (ThrT *)ws.get("thr_shared", sizeof(ThrT) * (size_t)n_slots);
        if (want_auc && !ext_topk && !g_sw.no_seed)
            hipLaunchKernelGGL((k_seed_thresholds<T, ThrT>), dim3(cdiv(n_slots, 256)), dim3(256), 0, stream, n_slots, stream_slot0, K, GU, slot_user, slot_chunk,
                               user_nslots, flags, c.test_p, grow, pos_score, spos_score, thr_shared);
        else HIP_CHECK(hipMemsetAsync(thr_shared, 0, sizeof(ThrT) * (size_t)n_slots, stream));
        sa.thr_shared = thr_shared;
        sa.n = n; sa.K = K; sa.ngt = NG; sa.n_slots = n_slots; sa.n_groups = n_groups; sa.n_ublocks = n_ublocks;
        sa.n_splits = n_splits; sa.tail_ublocks = tail_ublocks; sa.tail_splits = tail_splits; sa.part_splits = part_splits; sa.tiles_total = tiles_total; sa.jmax = jmax; sa.check_nan = check_nan ? 1 : 0; sa.buffered_lists = (want_lane || ext_topk) ? 1 : 0; sa.ext_topk = ext_topk ? 1 : 0;
        sa.Ap = (decltype(sa.Ap))Ap; sa.Bp = (decltype(sa.Bp))Bp; sa.slot_user = slot_user; sa.slot_chunk = slot_chunk;
        sa.train_p = c.train_p; sa.train_i = c.train_i; sa.gj = gj; sa.grow = grow;
        sa.pos_score = pos_score; sa.pos_item = pos_item; sa.hist = hist; sa.glists = glists; sa.lane_cap = lane_cap; sa.lane_cnt = lane_cnt; sa.pl = pl; sa.pst = pst; sa.dump = nullptr;
        P::set_pending(sa, pend_cap, (int)pend_off);
        P::set_sync(sa, (int)sync_off);
        sa.stream_slot0 = stream_slot0; sa.stream_ld = stream_ld; sa.stream_scores = stream_scores;
        set_part_extra(sa, part_extra);
        sa.noise_row = c.noise_row; sa.noise_row0 = c.noise_row0; sa.noise_E = c.noise_E; sa.noise_ld = c.noise_ld;
        set_spec(sa);

        // the dense train rows, launched beside the plan read-back: the sweep reads them, and so do the passes that start at
        // `flags_event` (their noise rows, their own sweeps) -- nothing in front of this point does
        if (bits_wait) {
            HIP_CHECK(hipStreamWaitEvent(stream, cx.side_ev[4], 0));
            side_guard.pending--;
        }
        // Sample seeds (k_seed_from_sample, rm_prep.hpp): the sweep's DUMP variant scores the first S items for every slot, a
        // wavefront per slot takes the K-th best candidate of them, and the lane buffers start with a pass rate of K / S.
        const bool sample_lists = !lane_lists && !ext_topk && sizeof(T) == 4;      // (the LDS / HBM lists: seeded as well, from a smaller sample)
        const int sample_S = (lane_lists || sample_lists) ? sample_seed_items<T>(ws, K, n, n_slots, check_nan || c.noise_E != nullptr, sample_lists) : 0;
        if (sample_S > 0 && !sample_lists) seed_from_sample<T>(sa, ws, sample_S, NG, n_slots, n_ublocks, stream);
        else if (sample_S > 0) {
            const long long units = P::items_units(sample_S / TILE_ITEMS, NG);
            typename P::PackT *Bs = (typename P::PackT *)ws.get("sample_items", 16 * (size_t)units);
            void *Ls = ws.get("sample_lists", sizeof(typename P::ListT) * (size_t)n_ublocks * 8 * GU * (2 + 32));
            pack_operands(c.A, c.lda, c.B, c.ldb, sample_S, k, NG, TILE_ITEMS, slot_user, 0, Ap, 0, Bs, units, stream, true);
            seed_from_sample<T>(sa, ws, sample_S, NG, n_slots, n_ublocks, stream, Bs, Ls);
        }
        if (c.flag_snapshot) {
            HIP_CHECK(hipMemcpyAsync(c.flag_snapshot, c.noise_flag, sizeof(int) * (size_t)m, hipMemcpyDeviceToDevice, stream));
            HIP_CHECK(hipMemcpyAsync(c.flag_count_host, &plan->n_noise_flagged, sizeof(int), hipMemcpyDeviceToHost, stream));
            HIP_CHECK(hipEventRecord(c.flags_event, stream));
        }
        RM_TRACE_POINT("run: preparation enqueued");
        HIP_CHECK(hipEventRecord(g_ev[1], stream));
        // Depth split: when only the deepest user blocks force the lists out of LDS (the allocation is sized per launch,
        // the tables per block), the shallow blocks [0, u_split) get their own launch with LDS lists.  The two launches
        // run side by side on two streams so that neither pays a partially filled last round of its own.
        int u_split = 0, j_shallow = -1;
        if (P::block_carve && !list_in_lds && !want_lane && K <= 32 && want_auc && !g_sw.hbm_lists && !g_sw.no_depth_split) {
            for (int j = jmax - 1; j >= 0 && j_shallow < 0; j--)
                if (lds_need_j(true, nsub, j) + SYNC_BYTES <= LDS_LIMIT) j_shallow = j;
            if (j_shallow >= 0) u_split = hp.class_offset[j_shallow + 1] / (GROUPS_PER_BLOCK * GU);
        }
        if (u_split > 0) {
            hipStream_t g_side_stream = side_stream();
            hipEvent_t *g_side_ev = cx.side_ev;
            typename P::Args sb = sa;                              // the deep blocks: lists in HBM, as computed above
            P::set_ublocks(sb, u_split, n_ublocks - u_split);
            typename P::Args sl = sa;                              // the shallow blocks: lists in LDS
            P::set_ublocks(sl, 0, u_split);
            // (the tail of the two-level grid = the cheapest user blocks = the first ones: the shallow launch's, then the deep one's)
            sl.tail_ublocks = std::min(tail_ublocks, u_split);
            sb.tail_ublocks = tail_ublocks - sl.tail_ublocks;
            const size_t lds_l = std::min<size_t>(LDS_LIMIT, lds_need_j(true, nsub, j_shallow) + SYNC_BYTES + (want_pending ? P::pend_cap_max : 0) * per_key);
            P::set_pending(sl, want_pending ? P::pend_cap_max : 0, 0);
            P::set_sync(sl, (int)(lds_l - SYNC_BYTES));
            HIP_CHECK(hipEventRecord(g_side_ev[0], stream));
            HIP_CHECK(hipStreamWaitEvent(g_side_stream, g_side_ev[0], 0));
            dispatch_sweep(want_auc, false, false, nsub, NG, dim3((unsigned)((n_ublocks - u_split - sb.tail_ublocks) * n_splits + sb.tail_ublocks * tail_splits)), lds_total, g_side_stream, sb);
            HIP_CHECK(hipEventRecord(g_side_ev[1], g_side_stream));
            dispatch_sweep(want_auc, false, true, nsub, NG, dim3((unsigned)((u_split - sl.tail_ublocks) * n_splits + sl.tail_ublocks * tail_splits)), lds_l, stream, sl);
            HIP_CHECK(hipStreamWaitEvent(stream, g_side_ev[1], 0));
        } else {
            // (Measured and dropped: the blocks made of streamed users only as a second launch of the sweep variant without rank
            // counting on a side stream, followed there by k_rank_streamed, beside the main launch.  At C2 the step time did
            // not move -- 11.61 vs 11.59 ms -- and at the north-star shape, where one launch is exactly one round of 256
            // blocks, the side launch ran AFTER the main one instead of beside it: 61 ms of tail.  gpurun_out r2q.)
            dispatch_sweep(want_auc, false, list_in_lds, nsub, NG, dim3(n_blocks), lds_total, stream, sa);
        }
        RM_TRACE_POINT("run: sweep enqueued");
        HIP_CHECK(hipEventRecord(g_ev[2], stream));
        if (lane_lists) {
            collect.on = true; collect.glists = (const char *)glists; collect.lane_cnt = lane_cnt; collect.thr = (const void *)thr_shared;
            collect.g = CollectGeom{sa.ublock0, sa.n_ublocks, n_splits, tail_ublocks, tail_splits, nsub, GU, P::lanes_per_user, lane_cap, part_extra ? n_part - 1 : -1};
        }
        g_timings[4] = u_split > 0 ? 2 : 1; g_timings[5] = n_splits; g_timings[6] = n_blocks; g_timings[7] = (double)lds_total;
        cx.timed_slots = n_slots;
        cx.total_slots = n_slots;
    } else {
        if (c.flag_snapshot) {                                       // (no user to evaluate: nothing is flagged)
            HIP_CHECK(hipMemsetAsync(c.flag_snapshot, 0, sizeof(int) * (size_t)m, stream));
            *c.flag_count_host = 0;
            HIP_CHECK(hipEventRecord(c.flags_event, stream));
        }
        HIP_CHECK(hipEventRecord(g_ev[1], stream));
        HIP_CHECK(hipEventRecord(g_ev[2], stream));
        g_timings[4] = 0; g_timings[5] = 0; g_timings[6] = 0; g_timings[7] = 0;
    }

    // ---- finalize ----
    fa.n_part = n_part; fa.pl = pl; fa.pst = pst; fa.hist = hist; fa.pos_score = pos_score; fa.pos_item_tab = pos_item;
    // the streamed users' ranks (HBM-bound: the stored score rows) run on the side stream beside the short kernels of the others
    fa.stream_slot0 = stream_slot0; fa.stream_scores = stream_scores; fa.stream_ld = stream_ld;
    fa.spos_score = spos_score; fa.spos_item = spos_item; fa.shist = shist;
    fa.rank_generic = (g_sw.rank_generic ? 1 : 0);
    const bool ranks_beside = use_side && n_slots > 0 && want_auc && n_stream > 0;
    // one block of k_rank_streamed per row (rows up to 32,768 items) with the user's table in LDS: the block also walks its counts
    // (k_auc_streamed's job); k_auc_streamed is then only launched when some row is too long for that
    bool auc_launch = true;
    if (stream_parts == 1 && !g_sw.no_fused_auc) {
        fa.fused_auc = 1 | (mask_test ? 2 : 0);
        long long top = 1;
        while (top <= hp.max_npos) top <<= 1;
        // (k_metrics > 256 streams everybody without counting the long rows first: the longest one is not known then)
        auc_launch = ext_topk || !(top * (long long)(sizeof(T) / 4) + hp.max_npos + 1 <= STREAM_RANK_LDS / 4);
    }
    hipStream_t rank_stream = stream;
    if (ranks_beside) { rank_stream = fork_side(); rank_streamed_rows(0, n_stream, rank_stream); }
    if (mask_test) {
        // (the table users' test items were put back by the sweep, rm_sweep.hpp; the streamed users' are this kernel's)
        if (n_slots > stream_slot0)
            hipLaunchKernelGGL((k_merge_positives<T, T>), dim3(cdiv(n_slots - stream_slot0, MERGE_WAVES)), dim3(MERGE_WAVES * WAVE), 0, stream, fa, n_part - 1, stream_slot0);
        if (ranks_beside && auc_launch) {                          // (it counts the streamed users' own test items: before k_auc_streamed)
            HIP_CHECK(hipEventRecord(cx.side_ev[2], stream));
            HIP_CHECK(hipStreamWaitEvent(rank_stream, cx.side_ev[2], 0));
        }
    }
    if (ranks_beside && auc_launch) auc_streamed_rows(0, n_stream, rank_stream);
    hipLaunchKernelGGL((k_finalize_skipped<T, T>), dim3(cdiv(m, 256)), dim3(256), 0, stream, fa);
    if (topv_pending) {                                           // (k_top_values, launched beside the preparation)
        HIP_CHECK(hipStreamWaitEvent(stream, cx.side_ev[3], 0));
        side_guard.pending--;
    }
    if (collect.on) {
        // k_metrics beyond the LDS lists: the users' ordered top-K out of the sweep's lane buffers, straight into `merged` (behind
        // k_merge_positives, whose part of `pl` -- the streamed users' own test items -- is one of the inputs)
        typedef typename std::remove_pointer<decltype(typename P::Args{}.thr_shared)>::type ThrT;
        fa.collected = 1;
        if (collect_capw(K, collect.g.lane_cap) == 512)
            hipLaunchKernelGGL((k_collect_topk<T, T, ThrT, 512>), dim3(collect_grid(cdiv(n_slots, 4))), dim3(256), 0, stream, fa, collect.g, collect.glists, collect.lane_cnt, (const ThrT *)collect.thr);
        else if (collect_capw(K, collect.g.lane_cap) == 1024)
            hipLaunchKernelGGL((k_collect_topk<T, T, ThrT, 1024>), dim3(collect_grid(cdiv(n_slots, 4))), dim3(256), 0, stream, fa, collect.g, collect.glists, collect.lane_cnt, (const ThrT *)collect.thr);
        else
            hipLaunchKernelGGL((k_collect_topk<T, T, ThrT, 4096>), dim3(collect_grid(n_slots)), dim3(64), 0, stream, fa, collect.g, collect.glists, collect.lane_cnt, (const ThrT *)collect.thr);
    }
    if (n_slots > 0 && ext_topk) {
        int sel_ld = 2;
        while (sel_ld < K) sel_ld <<= 1;
        fa.ext_topk = 1; fa.sel_ld = sel_ld;
        fa.sel_hi = (unsigned long long *)ws.get("sel_hi", sizeof(unsigned long long) * (size_t)n_slots * sel_ld);
        fa.sel_lo = (unsigned *)ws.get("sel_lo", sizeof(unsigned) * (size_t)n_slots * sel_ld);
        hipLaunchKernelGGL((k_select_topk<T, T>), dim3(n_slots), dim3(SELECT_THREADS), 0, stream, fa);
    }
    if (n_slots > 0) {
        if (want_auc) {
            if (n_stream > 0 && !ranks_beside) { rank_streamed_rows(0, n_stream, stream); if (auc_launch) auc_streamed_rows(0, n_stream, stream); }
            if (stream_slot0 > 0) hipLaunchKernelGGL((k_auc_slots<T, T>), dim3(cdiv(stream_slot0, 256)), dim3(256), 0, stream, fa);
        }
        const size_t fin_lds = finalize_lds_bytes<T>(K, n_part);
        HIP_CHECK(hipFuncSetAttribute((const void *)k_finalize<T, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds));
        auto finalize_slots = [&](int s0, int s1) {
            if (s1 <= s0) return;
            fa.fin_slot0 = s0; fa.fin_slot1 = s1;
            hipLaunchKernelGGL((k_finalize<T, T>), dim3(cdiv(s1 - s0, FIN_THREADS)), dim3(FIN_THREADS), fin_lds, stream, fa);
        };
        fa.auc_defer_slot0 = n_slots;
        if (ranks_beside && (req & (RQ_ROC | RQ_PR)) && !g_sw.no_defer_auc) {
            // nothing of k_finalize but the two AUC values of the streamed users needs the side stream: it runs for everybody while
            // their ranks are still being counted there, and a short kernel fills those two in behind the join
            fa.auc_defer_slot0 = stream_slot0;
            finalize_slots(0, n_slots);
            join_side();
            hipLaunchKernelGGL((k_finalize_auc<T, T>), dim3(cdiv(n_slots - stream_slot0, 256)), dim3(256), 0, stream, fa);
        } else if (ranks_beside) {
            finalize_slots(0, stream_slot0);
            join_side();
            finalize_slots(stream_slot0, n_slots);
        } else finalize_slots(0, n_slots);
    }
    HIP_CHECK(hipGetLastError());
    if (c.topk_idx)
        hipLaunchKernelGGL(k_export_rank<T>, dim3(cdiv((long long)m * K, 256)), dim3(256), 0, stream, m, K, merged, c.topk_idx, c.topk_score, flags);
    if (c.pos_rank)           // (per user, never a memset of the whole array: a batch must not clear what other batches wrote)
        hipLaunchKernelGGL(k_export_pos_rank, dim3(cdiv(m, 128)), dim3(128), 0, stream, c.nnz_test, m, c.test_p, flags, pos_order, rank_sorted, c.pos_rank,
                           (want_auc && n_slots > 0) ? 1 : 0);
    RM_TRACE_POINT("run: finalisation enqueued");
    HIP_CHECK(hipEventRecord(g_ev[3], stream));
    HIP_CHECK(hipEventRecord(cx.done, stream));
    cx.ev_recorded = true;
    HIP_CHECK(hipGetLastError());
}


// ---- tie noise on top of run() (rm_noise.hpp) ---------------------------------------------------------------------------
// fp64: the noise (|e| <= 1e-12) is far above an ulp of any ordinary score, every score changes: every user is evaluated
// with its noise row, in user batches whose rows (draws + noise values, 2 x 8 n bytes per user) fit the memory budget.
// fp32: the noise only changes scores below 2^-15 in magnitude; a first pass evaluates everybody on the plain scores and
// flags the users who have such a score among their test items or their top-K -- the only ones whose ranking the noise
// can touch -- and only those are evaluated again, exactly.
std::atomic<unsigned long long> g_call_counter{1};       // tags of calls: the packed item image survives between the passes / batches of one

template <class T> struct NoiseGeom {
    int per; long long e_ld, d_ld, cap;
    NoiseGeom(const Workspace &ws, int n)
    {
        per = sizeof(T) == 4 ? 1 : 2;
        e_ld = ((long long)n + 191) / 192 * 192;                                  // covers either tile size of the sweep
        d_ld = ((long long)n * per + MT_N - 1) / MT_N * MT_N;
        const long long row_bytes = e_ld * (long long)sizeof(T) + d_ld * 4;
        long long budget;
        if (g_sw.noise_budget_mb >= 0) budget = g_sw.noise_budget_mb << 20;
        else budget = free_plus_owned(ws, {"noise_draws", "noise_rows"}) / 3;
        cap = std::max<long long>(1, std::min<long long>(budget / row_bytes, 1 << 20));
    }
};
// per-user mt19937(seed + user) draws -> per-item noise rows for `rows` users (row_user: their indices, null = users 0 .. rows - 1)
template <class T>
void noise_make_rows(Ctx &cx, const Call<T> &c0, const NoiseGeom<T> &g, const int *row_user, int rows, const int *train_p, long long user0,
                     unsigned *D, T *E, hipStream_t stream, const unsigned *own_bits = nullptr, long long own_words = 0)
{
    const int n = c0.n, m = c0.m;
    hipLaunchKernelGGL(k_mt_draws, dim3(cdiv(rows, MT_WAVES)), dim3(MT_WAVES * WAVE), 0, stream, row_user, rows, c0.seed, user0,
                       train_p, n, g.per, D, g.d_ld);
    // the dense train rows of this call's own first pass (fp32, small item counts), when they cover exactly these users
    const bool dense = cx.bits_tag != 0 && cx.bits_tag == c0.items_tag && cx.bits_train_p == train_p && cx.bits_m == m && row_user != nullptr && !cx.bits_partial;
    if (own_bits)                                                  // unmasked dense rows built for exactly these users (run_host_range)
        hipLaunchKernelGGL(k_noise_rows_bits<T>, dim3(rows), dim3(NOISE_ROWS_THREADS), sizeof(int) * (size_t)(2 * own_words + 1), stream, row_user, rows,
                           own_bits, (int)own_words, n, 0, train_p, c0.train_i, c0.test_p, c0.test_i, D, g.d_ld, E, g.e_ld);
    else if (dense)
        hipLaunchKernelGGL(k_noise_rows_bits<T>, dim3(rows), dim3(NOISE_ROWS_THREADS), sizeof(int) * (size_t)(2 * cx.bits_words + 1), stream, row_user, rows,
                           (const unsigned *)cx.bits_ptr, (int)cx.bits_words, n, cx.bits_masked ? 1 : 0, train_p, c0.train_i, c0.test_p, c0.test_i,
                           D, g.d_ld, E, g.e_ld);
    else
        hipLaunchKernelGGL(k_noise_rows<T>, dim3(cdiv((long long)rows * g.e_ld, 256)), dim3(256), 0, stream, row_user, rows, train_p, c0.train_i, n,
                           D, g.d_ld, E, g.e_ld);
}

// The exact pass of the fp32 tie noise over the users flagged in `flag` [m]: rows of noise for them (in batches of what the memory
// budget allows), the whole pipeline again for exactly those users.  `n_known` < 0: the number of flagged users is counted here
// (one wait for the stream).  Returns that number; `row_user_out` (optional) receives the workspace array of their indices.
template <class T>
int noise_exact_pass(const Call<T> &c0, const int *flag, const int *skip, int n_known, hipStream_t stream, Ctx &cx, const int **row_user_out = nullptr)
{
    Workspace &ws = cx.ws;
    const int m = c0.m;
    const NoiseGeom<T> g(ws, c0.n);
    int *noise_row = (int *)ws.get("noise_row", sizeof(int) * (size_t)m);
    int *row_user = (int *)ws.get("noise_row_user", sizeof(int) * (size_t)(n_known >= 0 ? std::max(n_known, 1) : m));
    int *counter = (int *)ws.get("noise_counter", sizeof(int));
    unsigned char *only = (unsigned char *)ws.get("noise_only", (size_t)m);
    HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(int), stream));
    hipLaunchKernelGGL(k_noise_assign_rows, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, flag, skip, noise_row, row_user, counter);
    int n_flagged = n_known;
    if (n_known < 0) {
        HIP_CHECK(hipMemcpyAsync(cx.pinned_small + 2, counter, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        n_flagged = cx.pinned_small[2];
    }
    if (row_user_out) *row_user_out = row_user;
    for (long long r0 = 0; r0 < n_flagged; r0 += g.cap) {
        const int rows = (int)std::min<long long>(g.cap, n_flagged - r0);
        hipLaunchKernelGGL(k_noise_select, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, noise_row, (int)r0, (int)r0 + rows, only);
        unsigned *D = (unsigned *)ws.get("noise_draws", sizeof(unsigned) * (size_t)rows * (size_t)g.d_ld);
        T *E = (T *)ws.get("noise_rows", sizeof(T) * (size_t)rows * (size_t)g.e_ld);
        noise_make_rows<T>(cx, c0, g, row_user + r0, rows, c0.train_p, c0.user0, D, E, stream);
        Call<T> c = c0;
        c.only_users = only; c.noise_row = noise_row; c.noise_row0 = (int)r0; c.noise_E = E; c.noise_ld = g.e_ld; c.noise_flag = nullptr;
        c.first_pass_flags = nullptr; c.flag_snapshot = nullptr;
        c.eval_users = rows;
        c.csr_checked = true;                                        // (the first pass looked at every row)
        run<T>(c, stream, cx);
    }
    return n_flagged;
}

// `defer` (host-pointer calls in user batches): the fp32 noise path ends with a look at what the first pass flagged on top of the
// users evaluated beside it -- a wait for the whole batch.  With `defer` that tail is handed back instead of run: the caller
// enqueues its next batch first and runs the tail when it needs this batch's results (true = a sequential exact pass rewrote
// outputs behind whatever the caller had enqueued after this call).
template <class T>
void run_call(const Call<T> &c_in, hipStream_t stream, Ctx &cx, std::function<bool()> *defer = nullptr)
{
    if (defer) *defer = nullptr;
    if (!c_in.noise || RM_ABL_NOISE_OFF) { run<T>(c_in, stream, cx); return; }
    Call<T> c0 = c_in;
    if (c0.items_tag == 0) c0.items_tag = g_call_counter.fetch_add(1);                   // (a device-pointer call: its passes share B)
    Workspace &ws = cx.ws;
    const int m = c0.m;
    if (!cx.pinned_small) {
        HIP_CHECK(hipHostMalloc((void **)&cx.pinned_small, 64, hipHostMallocDefault));
        HIP_CHECK(hipEventCreateWithFlags(&cx.flags_ev, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&cx.pass_ev, hipEventDisableTiming));
    }
    if (sizeof(T) == 4 && c0.first_pass_flags) {                    // a batch of a host-pointer call: see Call::first_pass_flags
        Call<T> c1 = c0;
        c1.noise_flag = c0.first_pass_flags;
        run<T>(c1, stream, cx);
        return;
    }
    if (sizeof(T) == 4 && c0.dense_fit < 0) c0.dense_fit = dense_rows_fit(ws, m, c0.n) ? 1 : 0;
    const NoiseGeom<T> g(ws, c0.n);
    const long long e_ld = g.e_ld, d_ld = g.d_ld, cap = g.cap;
    auto make_rows_into = [=, &cx](const int *row_user, int rows, const int *train_p, long long user0, unsigned *D, T *E, hipStream_t st) {
        noise_make_rows<T>(cx, c0, g, row_user, rows, train_p, user0, D, E, st);
    };
    auto make_rows = [=, &ws](const int *row_user, int rows, const int *train_p, long long user0, unsigned *&D, T *&E) {
        D = (unsigned *)ws.get("noise_draws", sizeof(unsigned) * (size_t)rows * (size_t)d_ld);
        E = (T *)ws.get("noise_rows", sizeof(T) * (size_t)rows * (size_t)e_ld);
        make_rows_into(row_user, rows, train_p, user0, D, E, stream);
    };
    if (sizeof(T) == 8) {
        if (!c0.csr_checked) { check_csr_now<T>(c0, stream, cx); c0.csr_checked = true; }     // (the noise rows below index by the CSR arrays)
        const long long width = c0.cumulative ? c0.K : 1;
        for (long long b0 = 0; b0 < m; b0 += cap) {
            const int mb = (int)std::min<long long>(cap, m - b0);
            Call<T> c = c0;
            c.A = c0.A + (size_t)b0 * c0.lda; c.m = mb; c.train_p = c0.train_p + b0; c.test_p = c0.test_p + b0; c.user0 = c0.user0 + b0;
            for (int i = 0; i < 10; i++) if (c0.out[i]) c.out[i] = c0.out[i] + (size_t)b0 * (i >= 8 ? 1 : width);
            if (c0.topk_idx) { c.topk_idx = c0.topk_idx + (size_t)b0 * c0.K; c.topk_score = c0.topk_score + (size_t)b0 * c0.K; c.status = c0.status + b0; }
            if (c0.only_users) c.only_users = c0.only_users + b0;
            unsigned *D; T *E;
            make_rows(nullptr, mb, c.train_p, c.user0, D, E);
            c.noise_E = E; c.noise_ld = e_ld; c.noise_row = nullptr; c.noise_row0 = 0;
            run<T>(c, stream, cx);
        }
        return;
    }
    int *flag = (int *)ws.get("noise_flag", sizeof(int) * (size_t)m);
    HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int) * (size_t)m, stream));
    Call<T> c1 = c0;
    c1.noise_flag = flag;
    // Every user the noise can touch through a TEST item in the zone -- in practice all the flagged users -- is known once the
    // positives are scored, before the first sweep starts: their exact pass (noise rows, a small sweep, two dozen short
    // launches: 1.4 ms at BASELINE C2) runs on the PEER context and its stream BESIDE the first sweep, writes into buffers of its
    // own, and a scatter behind both puts its results over the first pass's.  Users flagged only by the first pass's
    // k_finalize (a top-K score in the zone) get the sequential exact pass below, as does everybody when the ranking outputs
    // of rm_rank_* are wanted.
    const bool beside = !c0.topk_idx && !c0.only_users && !g_sw.noise_sequential;
    int *snap = nullptr;
    if (beside) {
        snap = (int *)ws.get("noise_flag_snap", sizeof(int) * (size_t)m);
        *cx.pinned_small = 0;
        c1.flag_snapshot = snap; c1.flag_count_host = cx.pinned_small; c1.flags_event = cx.flags_ev;
    }
    // The exact pass beside the first sweep: enqueued right behind the first pass, on the NOISE_SLOT context's HIGH-PRIORITY streams
    // (create_stream) -- at normal priority its kernels would only be served when the sweep, which fills every compute unit, drains.
    // (Measured and dropped: enqueueing it from inside the first pass's run(), in front of the sweep's dispatch:
    // the host-side chain of the pass, two waits, keeps the sweep off the device for 0.8 ms.)
    Plan *plan = (Plan *)ws.get("plan", sizeof(Plan));
    int n_beside = 0;
    std::unique_lock<std::mutex> peer_lock;
    ScatterArgs<T> sc_keep{};
    auto beside_pass = [&]() {
        HIP_CHECK(hipEventSynchronize(cx.flags_ev));                 // (early: the sweep has only just been launched)
        const int n_early = *cx.pinned_small;
        if (n_early > 0 && n_early <= cap) {
            Ctx &pc = peer_context(cx, NOISE_SLOT);
            peer_lock = std::unique_lock<std::mutex>(pc.mu);
            if (!pc.own_stream) HIP_CHECK(create_stream(&pc.own_stream, pc.high_priority));
            hipStream_t ps = pc.own_stream;
            Workspace &pw = pc.ws;
            HIP_CHECK(hipStreamWaitEvent(ps, cx.flags_ev, 0));       // flags, their snapshot and the dense train rows are in place
            int *noise_row = (int *)pw.get("noise_row", sizeof(int) * (size_t)m);
            int *row_user = (int *)pw.get("noise_row_user", sizeof(int) * (size_t)n_early);
            int *counter = (int *)pw.get("noise_counter", sizeof(int));
            unsigned char *only = (unsigned char *)pw.get("noise_only", (size_t)m);
            HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(int), ps));
            hipLaunchKernelGGL(k_noise_assign_rows, dim3(cdiv(m, 256)), dim3(256), 0, ps, m, snap, (const int *)nullptr, noise_row, row_user, counter);
            hipLaunchKernelGGL(k_noise_select, dim3(cdiv(m, 256)), dim3(256), 0, ps, m, noise_row, 0, n_early, only);
            unsigned *D = (unsigned *)pw.get("noise_draws", sizeof(unsigned) * (size_t)n_early * (size_t)d_ld);
            T *E = (T *)pw.get("noise_rows", sizeof(T) * (size_t)n_early * (size_t)e_ld);
            make_rows_into(row_user, n_early, c0.train_p, c0.user0, D, E, ps);
            Call<T> c = c0;
            const long long width = c0.cumulative ? c0.K : 1;
            static const char *tnames[10] = {"t_p", "t_tp", "t_r", "t_ap", "t_tap", "t_ndcg", "t_hit", "t_rr", "t_roc", "t_pr"};
            ScatterArgs<T> sc{};
            for (int i = 0; i < 10; i++) {
                sc.width[i] = i >= 8 ? 1 : (int)width;
                sc.dst[i] = c0.out[i];
                c.out[i] = c0.out[i] ? (T *)pw.get(tnames[i], sizeof(T) * (size_t)m * (size_t)sc.width[i]) : nullptr;
                sc.src[i] = c.out[i];
            }
            c.only_users = only; c.noise_row = noise_row; c.noise_row0 = 0; c.noise_E = E; c.noise_ld = e_ld; c.noise_flag = nullptr;
            c.eval_users = n_early;
            c.csr_checked = true;
            if (cx.bits_tag != 0 && cx.bits_tag == c0.items_tag && cx.bits_train_p == c0.train_p && cx.bits_m == m && !cx.bits_partial && !g_sw.no_ext_bits) {
                c.ext_bits = (const unsigned *)cx.bits_ptr; c.ext_words = cx.bits_words; c.ext_masked = cx.bits_masked;      // the first pass's rows
            }
            // (whatever fails in there -- the peer context duplicates workspace under memory pressure --, nothing of it may still be
            // reading this context's buffers (flag snapshot, dense train rows, A / B) when the error reaches the caller)
            try { run<T>(c, ps, pc); }
            catch (...) {
                (void)hipStreamSynchronize(ps);
                if (pc.side_stream) (void)hipStreamSynchronize(pc.side_stream);
                g_last_ctx = &cx;
                throw;
            }
            g_last_ctx = &cx;                                        // rm_get_timings reports the main pass, not the small exact one
            HIP_CHECK(hipEventRecord(cx.pass_ev, ps));
            sc_keep = sc;
            n_beside = n_early;
        }
    };
    run<T>(c1, stream, cx);
    if (beside) beside_pass();
    if (n_beside > 0) {
        HIP_CHECK(hipStreamWaitEvent(stream, cx.pass_ev, 0));        // behind the first pass (stream order) AND the exact one
        hipLaunchKernelGGL(k_noise_scatter<T>, dim3(cdiv(m, 256)), dim3(256), 0, stream, m, snap, sc_keep);
        HIP_CHECK(hipEventRecord(cx.done, stream));                  // (the context's "last work" now ends with the scatter)
    }
    int *n_flagged_host = cx.pinned_small + 1;
    HIP_CHECK(hipMemcpyAsync(n_flagged_host, &plan->n_noise_flagged, sizeof(int), hipMemcpyDeviceToHost, stream));
    auto tail = [=, &cx]() -> bool {
        HIP_CHECK(hipStreamSynchronize(stream));
        cx.timings[4] = 0;
        const int n_flagged = *n_flagged_host - n_beside;               // what the first pass's k_finalize flagged on top (top-K in the zone)
        if (n_flagged <= 0) return false;
        Call<T> c = c0;
        c.same_train_rows = true;
        noise_exact_pass<T>(c, flag, n_beside > 0 ? (const int *)snap : (const int *)nullptr, n_flagged, stream, cx);
        return true;
    };
    if (defer) *defer = tail;
    else tail();
}

template <class F> int guarded(F &&f)
{
    g_err.clear();
    try { f(); return RM_OK; }
    catch (const RmError &e) { g_err = e.msg; return e.code; }
    catch (const std::bad_alloc &) { g_err = "host allocation failed"; return RM_ERR_NOMEM; }
    catch (const std::exception &e) { g_err = e.what(); return RM_ERR_HIP; }
}

// ---- interruption (reference src/recometrics.hpp:114-174: SignalSwitcher) ---------------------------------------------
// While a host-pointer call runs, SIGINT sets a flag instead of killing the process; the call looks at the flag between its
// user batches, stops, restores the previous handler, re-raises the signal (so that the caller's own handler -- Python's
// KeyboardInterrupt -- sees it) and reports the reference's message.  As in the reference only the first of several
// concurrent calls owns the handler (:141-143).  rm_request_interrupt() sets the same flag without a signal.
volatile std::sig_atomic_t g_interrupt = 0, g_interrupt_by_signal = 0;
std::atomic<bool> g_handler_locked{false};
extern "C" void rm_on_sigint(int) { g_interrupt = 1; g_interrupt_by_signal = 1; }
struct SignalGuard {
    void (*old_handler)(int) = nullptr;
    bool active = false;
    SignalGuard()
    {
        bool expected = false;
        if (g_handler_locked.compare_exchange_strong(expected, true)) {
            g_interrupt = 0; g_interrupt_by_signal = 0;
            old_handler = std::signal(SIGINT, rm_on_sigint);
            active = true;
        }
    }
    void restore()
    {
        if (active) { std::signal(SIGINT, old_handler); active = false; g_handler_locked.store(false); }
    }
    ~SignalGuard() { const bool mine = active; restore(); if (mine) { g_interrupt = 0; g_interrupt_by_signal = 0; } }
    void check()                                  // reference :166-173
    {
        if (!g_interrupt) return;
        const bool by_signal = g_interrupt_by_signal != 0;
        const bool mine = active;
        restore();
        if (mine) { g_interrupt = 0; g_interrupt_by_signal = 0; }
        if (by_signal && mine) std::raise(SIGINT);
        throw RmError{RM_ERR_INTERRUPTED, "Error: procedure was interrupted.\n"};
    }
};

// devices of the calls sharded inside the library (rm_set_devices); empty = the calling thread's current device only
std::mutex g_dev_mu;
std::vector<int> g_devices;

template <class T> struct HostCall {              // one host-pointer call (reference signature, src/recometrics_signatures.hpp:48-98)
    const T *A; size_t lda; const T *B; size_t ldb; int m, n, k;
    const int *trp, *tri, *tep, *tei; const T *tev;
    int K; bool cumulative, noise; T *outs[10]; bool cold; int mip, mpt;
    int *topk_idx; T *topk_score; long long *pos_rank; int *status;
    unsigned long long seed;
    int nthreads = 0;                                 // host threads of the fall-back sort of unsorted CSR rows (0 = all)
};

// item factors of a sharded call: uploaded from the host by shard 0, copied device-to-device (xGMI) by the others
struct SharedItems {
    std::mutex mu; std::condition_variable cv;
    bool ready = false, failed = false;
    const void *src = nullptr; int src_device = 0;
    // shards that have not finished (or given up on) their device-to-device copy of `src` yet: shard 0 keeps its context --
    // and with it the buffer `src` points to -- locked until this is 0, so that no later call can overwrite or free the
    // buffer under a slow shard
    int copies_pending = 0;
    void copy_done() { std::lock_guard<std::mutex> lk(mu); copies_pending--; cv.notify_all(); }
};

// direct device-to-device copies (xGMI on a multi-GPU node): enabled once per ordered pair, by the copying side
inline void enable_peer_access(int dst_device, int src_device)
{
    static std::mutex mu;
    static std::map<std::pair<int, int>, bool> done;
    if (dst_device == src_device) return;
    std::lock_guard<std::mutex> lk(mu);
    bool &d = done[std::make_pair(dst_device, src_device)];
    if (d) return;
    d = true;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dst_device, src_device) == hipSuccess && can) {
        const hipError_t e = hipDeviceEnablePeerAccess(src_device, 0);          // (the current device is dst_device)
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // not fatal: hipMemcpyPeer stages through the host
        else (void)hipGetLastError();
    }
}


// Users [u0, u1) of a host call on the current device.  The inputs are staged into HBM batch by batch and the batches are
// pipelined: while the kernels of batch i run on the call's stream, the rows of batch i + 1 (user factors, train / test CSR
// rows) are uploaded on a second stream into their places in the range's device arrays -- a batch only ever reads its own
// rows, so nothing is double-buffered.  The first batch is small (its upload is the only one nothing hides) and the batches
// double in size from there; they are also what the interrupt flag is looked at between (reference :488-489).  Each batch's
// outputs go straight into the caller's arrays.  `shared` (sharded calls) says where the item factors come from.
template <class T>
void run_host_range(const HostCall<T> &h, int u0, int u1, Ctx &cx, hipStream_t stream, SharedItems *shared, int shard, unsigned long long tag)
{
    std::lock_guard<std::mutex> lk(cx.mu);
    Workspace &ws = cx.ws;
    const int m = u1 - u0, n = h.n, k = h.k, K = h.K;
    cx.acc[0] = cx.acc[1] = cx.acc[2] = cx.acc[3] = 0; cx.ev_recorded = false;
    g_last_ctx = &cx;
    // whatever happens below, the other shards must neither wait for item factors that never come nor copy from a buffer
    // that is being reused: shard 0 publishes (possibly a failure) and then waits for every copy; the others report theirs
    struct SharedGuard {
        SharedItems *sh; int shard; bool published = false, reported = false;
        ~SharedGuard()
        {
            if (!sh) return;
            if (shard == 0) {
                std::unique_lock<std::mutex> sl(sh->mu);
                if (!published) { sh->ready = true; sh->failed = true; sh->cv.notify_all(); }
                sh->cv.wait(sl, [&] { return sh->copies_pending <= 0; });
            } else if (!reported) sh->copy_done();
        }
    } guard{shared, shard};
    if (!cx.up_stream) {
        HIP_CHECK(hipStreamCreateWithFlags(&cx.up_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) HIP_CHECK(hipEventCreateWithFlags(&cx.up_ev[i], hipEventDisableTiming));
    }
    hipStream_t up = cx.up_stream;
    if (cx.ev_valid) HIP_CHECK(hipStreamWaitEvent(up, cx.done, 0));   // the previous call on this context may still read the buffers
    // item factors, dense rows of k
    T *dB = (T *)ws.get("in_B", sizeof(T) * (size_t)n * k);
    auto upload_items = [&]() {
        if (!shared || shard == 0) {
            // (dense rows: ONE plain copy -- a 2-D copy of pageable memory goes through a staging buffer and a second,
            // device-side pass: 1.1 ms more for BASELINE C2's A and B)
            hipError_t e = h.ldb == (size_t)k ? hipMemcpyAsync(dB, h.B, sizeof(T) * (size_t)n * k, hipMemcpyHostToDevice, up)
                                              : hipMemcpy2DAsync(dB, sizeof(T) * k, h.B, sizeof(T) * h.ldb, sizeof(T) * k, n, hipMemcpyHostToDevice, up);
            if (shared) {
                if (e == hipSuccess) e = hipStreamSynchronize(up);
                std::lock_guard<std::mutex> sl(shared->mu);
                shared->ready = true; shared->failed = e != hipSuccess; shared->src = dB; shared->src_device = cx.device;
                guard.published = true;
                shared->cv.notify_all();
            }
            HIP_CHECK(e);
        } else {
            {
                std::unique_lock<std::mutex> sl(shared->mu);
                shared->cv.wait(sl, [&] { return shared->ready; });
                if (shared->failed) throw RmError{RM_ERR_HIP, "upload of the item factors failed on the first device"};
            }
            enable_peer_access(cx.device, shared->src_device);
            HIP_CHECK(hipMemcpyPeerAsync(dB, cx.device, shared->src, shared->src_device, sizeof(T) * (size_t)n * k, up));
            HIP_CHECK(hipStreamSynchronize(up));                      // the source buffer is shard 0's: tell it when we are done with it
            guard.reported = true;
            shared->copy_done();
        }
    };
    if (m <= 0) { upload_items(); HIP_CHECK(hipStreamSynchronize(up)); return; }
    // this range's users: factors, CSR rows with the index pointers rebased to the range
    const long long tr0 = h.trp[u0], te0 = h.tep[u0];
    const long long nnz_tr = (long long)h.trp[u1] - tr0, nnz_te = (long long)h.tep[u1] - te0;
    T *dA = (T *)ws.get("in_A", sizeof(T) * (size_t)m * k);
    int *dtrp = (int *)ws.get("in_trp", sizeof(int) * (size_t)(m + 1));
    int *dtep = (int *)ws.get("in_tep", sizeof(int) * (size_t)(m + 1));
    int *dtri = (int *)ws.get("in_tri", sizeof(int) * (size_t)std::max<long long>(nnz_tr, 1));
    int *dtei = (int *)ws.get("in_tei", sizeof(int) * (size_t)std::max<long long>(nnz_te, 1));
    T *dtev = h.tev ? (T *)ws.get("in_tev", sizeof(T) * (size_t)std::max<long long>(nnz_te, 1)) : nullptr;
    std::vector<int> rb;                                              // rebased index pointers (only when the range does not start at 0)
    const int *trp = h.trp + u0, *tep = h.tep + u0;
    if (tr0 || te0) {
        rb.resize(2 * (size_t)(m + 1));
        for (int i = 0; i <= m; i++) { rb[i] = h.trp[u0 + i] - (int)tr0; rb[m + 1 + i] = h.tep[u0 + i] - (int)te0; }
        trp = rb.data(); tep = rb.data() + m + 1;
    }
    HIP_CHECK(hipMemcpyAsync(dtrp, trp, sizeof(int) * (size_t)(m + 1), hipMemcpyHostToDevice, up));
    HIP_CHECK(hipMemcpyAsync(dtep, tep, sizeof(int) * (size_t)(m + 1), hipMemcpyHostToDevice, up));
    const size_t per = h.cumulative ? (size_t)K : 1;                 // values per user of the eight top-K metrics
    size_t out_w = 0;                                               // values per user over all requested metrics
    for (int i = 0; i < 10; i++) if (h.outs[i]) out_w += i >= 8 ? 1 : per;
    int *d_topk_idx = nullptr, *d_status = nullptr; T *d_topk_score = nullptr; long long *d_pos_rank = nullptr;
    if (h.topk_idx) {
        d_topk_idx = (int *)ws.get("o_topk_idx", sizeof(int) * (size_t)m * K);
        d_topk_score = (T *)ws.get("o_topk_score", sizeof(T) * (size_t)m * K);
        d_pos_rank = (long long *)ws.get("o_pos_rank", sizeof(long long) * (size_t)std::max<long long>(nnz_te, 1));
        d_status = (int *)ws.get("o_status", sizeof(int) * (size_t)m);
    }
    // largest batch: ~0.4 s of device work at the rate the sweep sustains (2 n k flop per user), whole user blocks
    const double rate = std::is_same<T, float>::value ? 6.0e13 : 2.5e13;
    double bu = 0.4 * rate / (2.0 * (double)n * (double)k);
    const bool forced = g_sw.batch_users > 0;                        // tests: equal batches of this size
    if (forced) bu = g_sw.batch_users;
    long long batch = (long long)std::min<double>(std::max(bu, 1024.0), 2.0e9);
    batch = (batch + 1023) / 1024 * 1024;
    // k_metrics > 256: the lane buffers when they fit at all (run(): want_lane; one block's worth per 4 GU users on top of the grid's
    // floor -- lane_blocks_bound), sized so that run() finds every batch fitting; else score rows (ext_topk)
    const long long lane_blocks = K > 256 && lane_lists_possible<T>(K) ? lane_budget_bytes(ws) * 3 / 4 / lane_list_bytes<T>(K, 1) - lane_blocks_bound<T>(0) : 0;
    if (lane_blocks >= 8) batch = std::max<long long>(1024, std::min<long long>(batch, lane_blocks * 4 * Prec<T>::GU / 1024 * 1024));
    else if (K > 256) {                                            // one score row per user of the batch (run(): ext_topk),
        const long long row = (((long long)n + 191) / 192 * 192) * (long long)sizeof(T);    // with a margin for what run() allocates first
        batch = std::max<long long>(1, std::min<long long>(batch, stream_budget_bytes(ws) * 3 / 4 / row));
    }
    // batch boundaries: a first batch of a QUARTER of the users (whole kilo-users) when the range is large enough for the pipeline to
    // matter -- its upload is the only one nothing hides --, then three times the batch before (its upload hides behind the batch
    // before, whose kernels take ~2.5 x as long per user as the copy), at most `batch` users.  BASELINE C2: 34,816 + 103,677 users.
    // Every batch pays the fixed parts of preparation and finalisation again and its plan only comes back when the sweep in front
    // of it has drained, so fewer batches win once the first upload is paid for: round 5 (preparation 0.88 -> 0.56 ms) 9.6 ms per
    // call against 9.9 with (1/8, 3/8, 1/2), and with the tie noise 10.3 against 10.85 (one batch fewer for the exact pass to
    // wait behind; `profiles/r5_host_entry.txt`).  Round 3, when a batch's fixed parts cost twice as much: (1/8, 3/8, 1/2) 10.4,
    // (1/4, 3/4) 10.6, without the second context 11.7 (`profiles/r3_host_entry.txt`).
    std::vector<long long> cuts{0};
    {
        const int r0 = g_sw.ramp > 0 ? std::max(2, g_sw.ramp) : 4;    // (RM_DEBUG_RAMP, A/B timing: another first fraction; default: a quarter)
        long long next = (forced || m <= 16384) ? batch : std::min<long long>(batch, std::max<long long>(8192, ((long long)m / r0 + 1023) / 1024 * 1024));
        while (cuts.back() < m) {
            long long b1 = std::min<long long>(m, cuts.back() + next);
            if (m - b1 < 2048 && m - cuts.back() <= batch) b1 = m;                  // no crumb at the end
            cuts.push_back(b1);
            next = forced ? batch : std::min<long long>(batch, next * 3);
        }
    }
    const int n_batches = (int)cuts.size() - 1;
    long long mb_max = 0;
    for (int bi = 0; bi < n_batches; bi++) mb_max = std::max(mb_max, cuts[bi + 1] - cuts[bi]);
    // the metric block of a batch: [metric][users of the batch x width], one device buffer, one D2H copy into page-locked
    // staging, scattered into the caller's arrays by the host (ten pageable copies cost 35-140 us EACH in host time)
    // Batches alternate between this context and its peer (own workspace, events and stream): batch i + 1 is planned and
    // packed while batch i sweeps, and its sweep takes over the compute units as batch i's blocks drain.
    Ctx *ctxs[2] = {&cx, &cx};
    hipStream_t streams[2] = {stream, stream};
    std::unique_lock<std::mutex> peer_lock;
    // (not with k_metrics > 256: a batch is then sized by the score rows ONE context may hold -- a third of the free memory --
    // and a second context holding as much again leaves the first one's next batch short: RM_ERR_NOMEM on the third batch)
    const bool rows_bound = K > 256 || g_sw.ext_topk;             // (the lane buffers of k_metrics > 256 are as large)
    if (n_batches > 1 && !rows_bound && !g_sw.one_context) {
        Ctx &pc = peer_context(cx);
        peer_lock = std::unique_lock<std::mutex>(pc.mu);
        if (!pc.own_stream) HIP_CHECK(hipStreamCreateWithFlags(&pc.own_stream, hipStreamNonBlocking));
        if (cx.ev_valid) HIP_CHECK(hipStreamWaitEvent(pc.own_stream, cx.done, 0));       // (buffers of the range may still be read by an earlier call)
        ctxs[1] = &pc; streams[1] = pc.own_stream;
    }
    T *dblocks[2], *hblocks[2];
    for (int i = 0; i < 2; i++) {
        if (i == 1 && ctxs[1] == ctxs[0]) { dblocks[1] = dblocks[0]; hblocks[1] = hblocks[0]; break; }
        dblocks[i] = (T *)ctxs[i]->ws.get("o_block", sizeof(T) * std::max<size_t>(out_w * (size_t)mb_max, 1));
        hblocks[i] = out_w ? (T *)ctxs[i]->pinned_get(sizeof(T) * out_w * (size_t)mb_max) : nullptr;
    }
    const bool two_ctx = ctxs[1] != ctxs[0];
    // fp32 tie noise over several batches: every batch runs its first pass only and flags the users the noise can touch in its
    // slice of `range_flag`; ONE exact pass over the flagged users of the whole range follows the last batch.  (Per batch, the
    // exact pass costs two waits on the host -- for the flags, for its own plan -- during which the next batch is not enqueued:
    // 14.6 ms against 10.2 without noise at BASELINE C2, profiles/r4_host_entry.txt.)  Not with the ranking outputs of rm_rank_*.
    const bool range_noise = h.noise && std::is_same<T, float>::value && n_batches > 1 && !h.topk_idx && !RM_ABL_NOISE_OFF &&
                             !g_sw.noise_per_batch;
    int *range_flag = nullptr, *range_snap = nullptr;
    // ... and the users flagged by the time the LAST batch launches its sweep -- every user with a test item in the noise zone, in
    // practice all of them -- are evaluated exactly BESIDE that sweep on a context of their own (`besides` below), as the device entry
    // does (run_call); what remains behind the last batch is the host-side scatter of their values, and an exact pass over the few
    // users only a batch's k_finalize flagged (a top-K score in the zone), usually none
    const bool beside_last = range_noise && !g_sw.no_noise_beside_last;
    if (range_noise) {
        range_flag = (int *)ws.get("noise_flag_range", sizeof(int) * (size_t)m);
        HIP_CHECK(hipMemsetAsync(range_flag, 0, sizeof(int) * (size_t)m, up));      // (in front of batch 0's rows: every batch waits for its rows' event)
        if (beside_last) {
            range_snap = (int *)ws.get("noise_flag_range_snap", sizeof(int) * (size_t)m);
            if (!cx.pinned_small) {
                HIP_CHECK(hipHostMalloc((void **)&cx.pinned_small, 64, hipHostMallocDefault));
                HIP_CHECK(hipEventCreateWithFlags(&cx.flags_ev, hipEventDisableTiming));
                HIP_CHECK(hipEventCreateWithFlags(&cx.pass_ev, hipEventDisableTiming));
            }
        }
    }
    auto upload_users = [&](int bi) {                                 // rows [cuts[bi], cuts[bi + 1]) into their places, on `up`
        const long long b0 = cuts[bi], b1 = cuts[bi + 1];
        if (h.lda == (size_t)k) HIP_CHECK(hipMemcpyAsync(dA + (size_t)b0 * k, h.A + ((size_t)u0 + b0) * k, sizeof(T) * (size_t)(b1 - b0) * k, hipMemcpyHostToDevice, up));
        else HIP_CHECK(hipMemcpy2DAsync(dA + (size_t)b0 * k, sizeof(T) * k, h.A + ((size_t)u0 + b0) * h.lda, sizeof(T) * h.lda, sizeof(T) * k, (size_t)(b1 - b0),
                                        hipMemcpyHostToDevice, up));
        const long long r0 = trp[b0], r1 = trp[b1], e0 = tep[b0], e1 = tep[b1];      // range-relative entries of the batch
        if (r1 > r0) HIP_CHECK(hipMemcpyAsync(dtri + r0, h.tri + tr0 + r0, sizeof(int) * (size_t)(r1 - r0), hipMemcpyHostToDevice, up));
        if (e1 > e0) HIP_CHECK(hipMemcpyAsync(dtei + e0, h.tei + te0 + e0, sizeof(int) * (size_t)(e1 - e0), hipMemcpyHostToDevice, up));
        if (dtev && e1 > e0) HIP_CHECK(hipMemcpyAsync(dtev + e0, h.tev + te0 + e0, sizeof(T) * (size_t)(e1 - e0), hipMemcpyHostToDevice, up));
        HIP_CHECK(hipEventRecord(cx.up_ev[bi & 1], up));
    };
    // RM_HOST_TRACE (timing studies only): host-side time stamps of the pipeline, printed at the end of the call
    const bool trace = g_sw.host_trace;
    RM_TRACE_POINT("host entry: batches start");
    std::vector<std::pair<const char *, double>> stamps;
    const auto t_start = std::chrono::steady_clock::now();
    auto stamp = [&](const char *what) { if (trace) stamps.emplace_back(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count()); };
    // shard 0 sends the item factors first (the other shards are waiting for them); the others send their own rows first
    if (!shared || shard == 0) { upload_items(); stamp("items enqueued"); upload_users(0); }
    else { upload_users(0); upload_items(); HIP_CHECK(hipEventRecord(cx.up_ev[0], up)); }
    stamp("batch 0 rows enqueued");
    // (`tail`: what the fp32 tie noise still has to look at once the batch is through, run_call; `copy_out`: the batch's device-to-host
    // copies, enqueued behind the batch and once more when the tail rewrote outputs)
    long long users_copied = 0;                                       // users [0, users_copied) of the range have been handed over to the caller
    long long flagged_total = 0;                                      // fp32 tie noise over the range: users the batches' first passes flagged
    struct InFlight { bool on = false; bool counts_flags = false; long long b0 = 0; int mb = 0; size_t boff[10]; size_t bo = 0; int which = 0;
                      std::function<bool()> tail; std::function<void()> copy_out; } fl[2];
    auto finish = [&](int which) {                                    // wait for the batch in flight on context `which`, hand its outputs over
        InFlight &f = fl[which];
        if (!f.on) return;
        f.on = false;
        if (f.tail) { const bool again = f.tail(); f.tail = nullptr; if (again) f.copy_out(); }
        HIP_CHECK(hipStreamSynchronize(streams[which]));
        stamp("batch done");
        for (int i = 0; i < 10; i++) {
            const size_t w = i >= 8 ? 1 : per;
            if (h.outs[i]) std::memcpy(h.outs[i] + ((size_t)u0 + f.b0) * w, hblocks[which] + f.boff[i], sizeof(T) * (size_t)f.mb * w);
        }
        stamp("outputs scattered");
        users_copied = std::max<long long>(users_copied, f.b0 + f.mb);
        if (f.counts_flags) flagged_total += ctxs[which]->pinned_small[5];
        if (n_batches > 1) {                                          // more than one batch: add up the stage timings
            Ctx &c = *ctxs[which];
            float ta = 0, tb = 0, tc = 0, td = 0;
            (void)hipEventElapsedTime(&ta, c.ev[0], c.ev[1]); (void)hipEventElapsedTime(&tb, c.ev[1], c.ev[2]);
            (void)hipEventElapsedTime(&tc, c.ev[2], c.ev[3]); (void)hipEventElapsedTime(&td, c.ev[0], c.ev[3]);
            cx.acc[0] += ta; cx.acc[1] += tb; cx.acc[2] += tc; cx.acc[3] += td;
            c.ev_recorded = false; cx.ev_recorded = false;
        }
    };
    // ---- fp32 tie noise over the whole range (see `range_noise` above) ----
    auto range_call = [&]() {                                         // every user of the range, outputs not set
        Call<T> c{};
        c.A = dA; c.lda = k; c.B = dB; c.ldb = k; c.m = m; c.n = n; c.k = k;
        c.train_p = dtrp; c.train_i = dtri; c.nnz_train = nnz_tr;
        c.test_p = dtep; c.test_i = dtei; c.test_v = dtev; c.nnz_test = nnz_te;
        c.K = K; c.cumulative = h.cumulative; c.noise = true; c.cold = h.cold; c.min_items_pool = h.mip; c.min_pos_test = h.mpt;
        c.items_tag = tag; c.seed = h.seed; c.user0 = (long long)u0;
        c.csr_checked = true;                                         // (every batch has looked at its rows)
        return c;
    };
    size_t x_off[10], x_bo = 0;                                       // layout of an exact pass's metric block: [metric][users of the range x width]
    for (int i = 0; i < 10; i++) { x_off[i] = x_bo; if (h.outs[i]) x_bo += (size_t)m * (i >= 8 ? 1 : per); }
    // an exact pass's results leave the device packed: one record of `out_w` values per evaluated user (k_noise_gather)
    size_t rec_off[10], rec_w = 0;
    for (int i = 0; i < 10; i++) { rec_off[i] = rec_w; if (h.outs[i]) rec_w += i >= 8 ? 1 : per; }
    auto gather_exact = [&](Ctx &gc, const T *dx, const int *row_user, int count, hipStream_t st, T *&hx, int *&hu) {
        GatherArgs<T> ga{};
        for (int i = 0; i < 10; i++) { ga.src[i] = h.outs[i] ? dx + x_off[i] : nullptr; ga.width[i] = i >= 8 ? 1 : (int)per; ga.off[i] = (int)rec_off[i]; }
        ga.out_w = (int)rec_w;
        T *dg = (T *)gc.ws.get("o_exact_packed", sizeof(T) * rec_w * (size_t)count);
        hipLaunchKernelGGL(k_noise_gather<T>, dim3(cdiv(count, 256)), dim3(256), 0, st, count, row_user, ga, dg);
        hx = (T *)gc.pinned_get(sizeof(T) * rec_w * (size_t)count + sizeof(int) * (size_t)count);
        hu = (int *)(hx + rec_w * (size_t)count);
        HIP_CHECK(hipMemcpyAsync(hx, dg, sizeof(T) * rec_w * (size_t)count, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipMemcpyAsync(hu, row_user, sizeof(int) * (size_t)count, hipMemcpyDeviceToHost, st));
    };
    auto scatter_exact = [&](const T *hx, const int *hu, int count) {  // the flagged users' records -> the caller's arrays
        for (int f = 0; f < count; f++) {
            const T *rec = hx + (size_t)f * rec_w;
            for (int i = 0; i < 10; i++) {
                if (!h.outs[i]) continue;
                const size_t w = i >= 8 ? 1 : per;
                std::memcpy(h.outs[i] + ((size_t)u0 + hu[f]) * w, rec + rec_off[i], sizeof(T) * w);
            }
        }
    };
    struct Beside { bool ran = false; int count = 0; T *hx = nullptr; int *hu = nullptr; Ctx *pc = nullptr; hipStream_t stream = nullptr; std::unique_lock<std::mutex> lock; } bes;
    // the exact pass beside the last batch's sweep, on the NOISE_SLOT context and its stream: flags as they stand once the last batch
    // has scored its positives, noise rows for those users (from dense train rows built for them alone), the pipeline for them,
    // and the copy of their metric block to page-locked memory -- all of it enqueued behind `flags_ev`, none of it waited for here
    // except the count of the flagged users
    // It runs on the NOISE_SLOT context's high-priority streams (create_stream): at normal priority, kernels that reach the device after
    // a sweep which fills every compute unit only start when it drains -- the pass began when the last batch ended and cost its full
    // 0.9 ms (profiles/r5_host_entry.txt).
    auto beside_start = [&](long long b0_last, hipEvent_t other) {
        if (!out_w) return;
        Ctx &pc = peer_context(cx, NOISE_SLOT);
        bes.lock = std::unique_lock<std::mutex>(pc.mu);
        bes.pc = &pc;
        if (!pc.own_stream) HIP_CHECK(create_stream(&pc.own_stream, pc.high_priority));
        hipStream_t ps = pc.own_stream;
        bes.stream = ps;
        Workspace &pw = pc.ws;
        HIP_CHECK(hipStreamWaitEvent(ps, cx.flags_ev, 0));
        // (the batch before the last runs on the other context's stream: its positives -- the early flags -- are in place once its
        // own sweep has been launched, the event run() records as ev[1]; what its k_finalize flags later is the late pass's)
        if (other) HIP_CHECK(hipStreamWaitEvent(ps, other, 0));
        // (the earlier batches' flags as they stand: a flag their k_finalize sets later is the sequential pass's, below)
        if (b0_last > 0) HIP_CHECK(hipMemcpyAsync(range_snap, range_flag, sizeof(int) * (size_t)b0_last, hipMemcpyDeviceToDevice, ps));
        int *noise_row = (int *)pw.get("noise_row", sizeof(int) * (size_t)m);
        int *row_user = (int *)pw.get("noise_row_user", sizeof(int) * (size_t)m);
        int *counter = (int *)pw.get("noise_counter", sizeof(int));
        unsigned char *only = (unsigned char *)pw.get("noise_only", (size_t)m);
        HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(int), ps));
        hipLaunchKernelGGL(k_noise_assign_rows, dim3(cdiv(m, 256)), dim3(256), 0, ps, m, range_snap, (const int *)nullptr, noise_row, row_user, counter);
        HIP_CHECK(hipMemcpyAsync(cx.pinned_small + 3, counter, sizeof(int), hipMemcpyDeviceToHost, ps));
        HIP_CHECK(hipStreamSynchronize(ps));                          // (the last batch's positives are scored; its sweep has just been launched)
        const int n_early = cx.pinned_small[3];
        const NoiseGeom<T> g(pw, n);
        if (n_early <= 0 || n_early > g.cap) return;                  // nobody, or more rows than the budget holds at once: the sequential pass
        hipLaunchKernelGGL(k_noise_select, dim3(cdiv(m, 256)), dim3(256), 0, ps, m, noise_row, 0, n_early, only);
        unsigned *D = (unsigned *)pw.get("noise_draws", sizeof(unsigned) * (size_t)n_early * (size_t)g.d_ld);
        T *E = (T *)pw.get("noise_rows", sizeof(T) * (size_t)n_early * (size_t)g.e_ld);
        Call<T> c = range_call();
        const long long words = dense_row_words(n);
        const bool fit = std::is_same<T, float>::value && dense_rows_fit(pw, m, n);
        unsigned *bits = nullptr;
        if (fit) {                                                    // rows of the flagged users alone, train items only
            bits = (unsigned *)pw.get("train_bits", (size_t)m * (size_t)words * 4);
            pc.bits_tag = 0; pc.bits_ptr = nullptr;
            launch_train_bits(ps, m, n, (int)words, dtrp, dtri, (const int *)nullptr, dtei, bits, (const Plan *)nullptr, only);
            c.ext_bits = bits; c.ext_words = words; c.ext_masked = false;
        }
        c.dense_fit = fit ? 1 : 0;
        noise_make_rows<T>(pc, c, g, row_user, n_early, dtrp, (long long)u0, D, E, ps, bits, words);
        c.only_users = only; c.noise_row = noise_row; c.noise_row0 = 0; c.noise_E = E; c.noise_ld = g.e_ld; c.eval_users = n_early;
        T *dx = (T *)pw.get("o_exact", sizeof(T) * std::max<size_t>(x_bo, 1));
        for (int i = 0; i < 10; i++) c.out[i] = h.outs[i] ? dx + x_off[i] : nullptr;
        try { run<T>(c, ps, pc); }
        catch (...) {
            (void)hipStreamSynchronize(ps);
            if (pc.side_stream) (void)hipStreamSynchronize(pc.side_stream);
            g_last_ctx = &cx;
            throw;
        }
        g_last_ctx = &cx;
        gather_exact(pc, dx, row_user, n_early, ps, bes.hx, bes.hu);
        bes.ran = true; bes.count = n_early;
    };
    //  a call that stops between its first passes and the exact one -- interrupt, failure -- must not leave the first
    // pass's un-noised values of the flagged users in the caller's arrays looking like results: they become NaN
    auto nan_flagged = [&]() {
        if (!range_noise || !out_w || users_copied <= 0) return;
        std::vector<int> hf((size_t)users_copied);
        if (hipMemcpy(hf.data(), range_flag, sizeof(int) * (size_t)users_copied, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return; }
        const T nanv = std::numeric_limits<T>::quiet_NaN();
        for (long long u = 0; u < users_copied; u++) {
            if (!hf[(size_t)u]) continue;
            for (int i = 0; i < 10; i++) {
                if (!h.outs[i]) continue;
                const size_t w = i >= 8 ? 1 : per;
                for (size_t j = 0; j < w; j++) h.outs[i][((size_t)u0 + (size_t)u) * w + j] = nanv;
            }
        }
    };
    try {
    for (int bi = 0; bi < n_batches; bi++) {
        const long long b0 = cuts[bi];
        const int b1 = (int)cuts[bi + 1], mb = b1 - (int)b0;
        if (g_interrupt) break;                                      // reference :488-489: the remaining users are skipped
        const int which = two_ctx ? (bi & 1) : 0;
        Ctx &bc = *ctxs[which];
        hipStream_t bs = streams[which];
        finish(which);                                               // (the batch two back: its context and staging are free again)
        HIP_CHECK(hipStreamWaitEvent(bs, cx.up_ev[bi & 1], 0));
        Call<T> c{};
        c.A = dA + (size_t)b0 * k; c.lda = k; c.B = dB; c.ldb = k; c.m = mb; c.n = n; c.k = k;
        c.train_p = dtrp + b0; c.train_i = dtri; c.nnz_train = nnz_tr;
        c.test_p = dtep + b0; c.test_i = dtei; c.test_v = dtev; c.nnz_test = nnz_te;
        c.K = K; c.cumulative = h.cumulative; c.noise = h.noise; c.cold = h.cold; c.min_items_pool = h.mip; c.min_pos_test = h.mpt;
        InFlight &f = fl[which];
        f.b0 = b0; f.mb = mb; f.bo = 0; f.which = which;
        for (int i = 0; i < 10; i++) { f.boff[i] = f.bo; c.out[i] = h.outs[i] ? dblocks[which] + f.bo : nullptr; if (h.outs[i]) f.bo += (size_t)mb * (i >= 8 ? 1 : per); }
        if (h.topk_idx) {
            c.topk_idx = d_topk_idx + (size_t)b0 * K; c.topk_score = d_topk_score + (size_t)b0 * K;
            c.pos_rank = d_pos_rank; c.status = d_status + b0;
        }
        c.items_tag = tag;
        c.seed = h.seed; c.user0 = (long long)u0 + b0;
        if (range_noise) c.first_pass_flags = range_flag + b0;
        const bool snap_here = beside_last && bi == n_batches - 1;
        if (snap_here) {
            c.flag_snapshot = range_snap + b0; c.flag_count_host = cx.pinned_small; c.flags_event = cx.flags_ev;
        }
        // enqueued (one short plan read-back inside; with more than one batch the tie noise's last look at the batch is deferred to
        // finish(): the next batch is enqueued first)
        run_call<T>(c, bs, bc, n_batches > 1 ? &f.tail : nullptr);
        f.on = true;
        stamp("batch enqueued");
        const size_t bo = f.bo;
        const long long e0 = tep[b0], e1 = tep[b1];                 // this batch's test entries (range-relative)
        Plan *batch_plan = (Plan *)bc.ws.get("plan", sizeof(Plan));
        int *flagged_here = (range_noise && bc.pinned_small) ? bc.pinned_small + 5 : nullptr;
        f.counts_flags = flagged_here != nullptr;
        f.copy_out = [=, &h]() {
            if (flagged_here) HIP_CHECK(hipMemcpyAsync(flagged_here, &batch_plan->n_noise_flagged, sizeof(int), hipMemcpyDeviceToHost, bs));
            if (bo) HIP_CHECK(hipMemcpyAsync(hblocks[which], dblocks[which], sizeof(T) * bo, hipMemcpyDeviceToHost, bs));
            if (h.topk_idx) {
                HIP_CHECK(hipMemcpyAsync(h.topk_idx + ((size_t)u0 + b0) * K, c.topk_idx, sizeof(int) * (size_t)mb * K, hipMemcpyDeviceToHost, bs));
                HIP_CHECK(hipMemcpyAsync(h.topk_score + ((size_t)u0 + b0) * K, c.topk_score, sizeof(T) * (size_t)mb * K, hipMemcpyDeviceToHost, bs));
                HIP_CHECK(hipMemcpyAsync(h.status + u0 + b0, c.status, sizeof(int) * (size_t)mb, hipMemcpyDeviceToHost, bs));
                if (e1 > e0) HIP_CHECK(hipMemcpyAsync(h.pos_rank + te0 + e0, c.pos_rank + e0, sizeof(long long) * (size_t)(e1 - e0), hipMemcpyDeviceToHost, bs));
            }
        };
        f.copy_out();
        // the next batch's rows travel while this batch's sweep runs (the copies below block the host, not the device)
        if (bi + 1 < n_batches && !g_interrupt) upload_users(bi + 1);
        stamp("next rows enqueued");
        if (snap_here && !g_interrupt) {
            beside_start(b0, (two_ctx && bi > 0) ? ctxs[(bi - 1) & 1]->ev[1] : (hipEvent_t)nullptr);
            stamp("exact pass beside the last batch enqueued");
        }
        if (!two_ctx) finish(which);
    }
    // (the batch before the last one first: its outputs are copied to the caller while the last batch is still on the device)
    { const int older = two_ctx ? ((n_batches - 2) & 1) : 0; finish(older); finish(older ^ 1); }
    g_last_ctx = &cx;
    // (ONE look at the flag decides both: whether the exact passes run, and whether the flagged users' values are withdrawn -- an
    // interrupt that arrives while the passes finish must not turn their finished, correct values into NaN)
    const bool interrupted = g_interrupt;
    if (range_noise && !interrupted && out_w) {
        if (bes.ran) {                                                // (its results left the device while the last batch was sweeping)
            HIP_CHECK(hipStreamSynchronize(bes.stream));
            scatter_exact(bes.hx, bes.hu, bes.count);
            stamp("exact pass (beside) scattered");
        }
        // the exact pass over the users of the range that the first passes flagged and the pass beside the last batch has not seen
        // (all of them without it): into a metric block of its own, from which the host takes the flagged users' values
        Call<T> c = range_call();
        T *dx = (T *)ws.get("o_exact", sizeof(T) * std::max<size_t>(x_bo, 1));
        for (int i = 0; i < 10; i++) c.out[i] = h.outs[i] ? dx + x_off[i] : nullptr;
        // (every batch reports how many users its first pass flagged: when the pass beside the last batch has seen them all --
        // the usual case -- nothing is left to look for)
        const long long late = flagged_total - (bes.ran ? bes.count : 0);
        if (late > 0) {
            const int *row_user = nullptr;
            const int n_flagged = noise_exact_pass<T>(c, range_flag, bes.ran ? (const int *)range_snap : (const int *)nullptr, -1, stream, cx, &row_user);
            stamp("exact pass enqueued");
            if (n_flagged > 0) {
                T *hx = nullptr; int *hu = nullptr;
                gather_exact(cx, dx, row_user, n_flagged, stream, hx, hu);
                HIP_CHECK(hipStreamSynchronize(stream));
                scatter_exact(hx, hu, n_flagged);
                stamp("exact pass scattered");
            }
        }
    } else if (bes.pc) (void)hipStreamSynchronize(bes.stream);
    if (interrupted) nan_flagged();
    } catch (...) {
        for (int i = 0; i < 2; i++) (void)hipStreamSynchronize(streams[i]);      // nothing may still write into staging that goes away
        (void)hipStreamSynchronize(up);
        if (bes.pc) (void)hipStreamSynchronize(bes.stream);
        nan_flagged();
        throw;
    }
    HIP_CHECK(hipStreamSynchronize(up));                              // `rb` and the caller's arrays go out of use (also after an interrupt)
    if (trace) {
        std::string line = "rm host trace (" + std::to_string(n_batches) + " batches, ms):";
        for (auto &st : stamps) { char buf[96]; snprintf(buf, sizeof buf, " [%s %.3f]", st.first, st.second); line += buf; }
        fprintf(stderr, "%s\n", line.c_str());
        std::lock_guard<std::mutex> tl(g_run_trace.mu);
        std::string l2 = "rm run trace (ms since the library was loaded):";
        for (auto &pt : g_run_trace.pts) { char buf[112]; snprintf(buf, sizeof buf, " [%s %.3f]", pt.first, pt.second); l2 += buf; }
        fprintf(stderr, "%s\n", l2.c_str());
        g_run_trace.pts.clear();
    }
}

// (m == 0 is not an error: the reference's loop over users, src/recometrics.hpp:428-437, simply does not run; the entry
// points return RM_OK before they get here.  `tei` may be null when there is no test entry at all -- every user is then NaN,
// as in the reference -- which is what a binding hands over for an empty index array.)
template <class T>
void validate(const T *A, const T *B, int m, int n, int k, const int *trp, const int *tep, const int *tei, long long nnz_test, int K, size_t lda, size_t ldb)
{
    if (!A || !B || !trp || !tep || (!tei && nnz_test > 0)) throw RmError{RM_ERR_INVALID, "null input pointer"};
    if (m <= 0 || n <= 0 || k <= 0) throw RmError{RM_ERR_INVALID, "m, n, k must be positive"};
    if (K <= 0) throw RmError{RM_ERR_INVALID, "k_metrics must be positive"};
    if (lda < (size_t)k || ldb < (size_t)k) throw RmError{RM_ERR_INVALID, "leading dimension smaller than k"};
}

// host-pointer entry: one device (the calling thread's current one), or the devices of rm_set_devices with contiguous user
// ranges [m g / G, m (g + 1) / G), one host thread + stream + workspace per shard, no exchange between the shards
template <class T>
void run_host_once(const HostCall<T> &h)
{
    if (h.m == 0) return;                                            // reference :428-437: no user, nothing written
    if (h.m < 0 || !h.tep) throw RmError{RM_ERR_INVALID, h.m < 0 ? "m, n, k must be positive" : "null input pointer"};
    validate(h.A, h.B, h.m, h.n, h.k, h.trp, h.tep, h.tei, (long long)h.tep[h.m], h.K, h.lda, h.ldb);
    // the index pointers say how much of the caller's index arrays is read at all: they are looked at here, on the host (2 (m + 1)
    // integers), before anything is sized or copied by them; the indices themselves are validated on the device (run())
    for (const int *p : {h.trp, h.tep}) {
        int bad = p[0] < 0;
        for (int u = 0; u < h.m; u++) bad |= p[u + 1] < p[u];
        if (bad) {
            int u = 0;
            while (u < h.m && !(p[u] < 0 || p[u + 1] < p[u])) u++;
            throw RmError{RM_ERR_INVALID, std::string("CSR index pointers of row ") + std::to_string(u) + " are negative or decreasing (" + (p == h.trp ? "X_train" : "X_test") + ")"};
        }
    }
    if (h.trp[h.m] > 0 && !h.tri) throw RmError{RM_ERR_INVALID, "null train indices"};
    // (deviation D8: the reference's walk would use gain 0 for a null Xtest_csr, :620, but its normalisation dereferences
    // the pointer, :870-874 -- a crash there, an error here)
    if (h.outs[5] && !h.tev) throw RmError{RM_ERR_INVALID, "NDCG requested without test values"};
    std::vector<int> devs;
    { std::lock_guard<std::mutex> lk(g_dev_mu); devs = g_devices; }
    const unsigned long long tag = g_call_counter.fetch_add(1);
    SignalGuard guard;
    if (devs.size() <= 1) {
        int prev = -1;
        if (devs.size() == 1) { HIP_CHECK(hipGetDevice(&prev)); HIP_CHECK(hipSetDevice(devs[0])); }
        try { run_host_range<T>(h, 0, h.m, context(0), nullptr, nullptr, 0, tag); }
        catch (...) { if (prev >= 0) (void)hipSetDevice(prev); throw; }
        if (prev >= 0) HIP_CHECK(hipSetDevice(prev));
        guard.check();
        return;
    }
    const int G = (int)devs.size();
    SharedItems shared;
    shared.copies_pending = G - 1;
    std::vector<RmError> errs((size_t)G, RmError{RM_OK, ""});
    std::vector<std::thread> workers;
    std::vector<char> reached((size_t)G, 0);                          // shard g got as far as run_host_range (whose guard then owns the hand-shake)
    Ctx *first = nullptr;
    std::mutex first_mu;
    for (int g = 0; g < G; g++) {
        workers.emplace_back([&, g] {
            try {
                HIP_CHECK(hipSetDevice(devs[g]));
                Ctx &cx = context(g);
                if (g == 0) { std::lock_guard<std::mutex> lk(first_mu); first = &cx; }
                if (!cx.own_stream) HIP_CHECK(hipStreamCreateWithFlags(&cx.own_stream, hipStreamNonBlocking));
                const int u0 = (int)((long long)h.m * g / G), u1 = (int)((long long)h.m * (g + 1) / G);
                reached[g] = 1;
                run_host_range<T>(h, u0, u1, cx, cx.own_stream, &shared, g, tag);
            } catch (const RmError &e) { errs[g] = e; }
            catch (const std::bad_alloc &) { errs[g] = RmError{RM_ERR_NOMEM, "host allocation failed"}; }
            catch (const std::exception &e) { errs[g] = RmError{RM_ERR_HIP, e.what()}; }
            // (a shard that failed before it reached run_host_range's guard -- hipSetDevice, stream creation -- has to do the
            // guard's job here: shard 0 publishes the failure, the others give up their copy)
            if (errs[g].code != RM_OK) {
                std::lock_guard<std::mutex> sl(shared.mu);
                if (g == 0) { if (!shared.ready) { shared.ready = true; shared.failed = true; } }
                else if (!reached[g]) shared.copies_pending--;
                shared.cv.notify_all();
            }
        });
    }
    for (auto &w : workers) w.join();
    g_last_ctx = first;
    for (int g = 0; g < G; g++) if (errs[g].code != RM_OK && errs[g].code != RM_INTERNAL_UNSORTED) throw RmError{errs[g].code, "device " + std::to_string(devs[g]) + ": " + errs[g].msg};
    for (int g = 0; g < G; g++) if (errs[g].code == RM_INTERNAL_UNSORTED) throw errs[g];
    guard.check();
}

// The reference's callers hand over CSR rows with sorted column indices (recometrics/__init__.py:35-41 sorts them with SciPy, one
// thread walking every entry: 26-33 ms at BASELINE C2); here the rows are uploaded as they come and validated on the device (run():
// k_check_csr_*, ~40 us).  Only when some row turns out unsorted is a copy of the index / value arrays sorted on the host
// (csrc/rm_csr.cpp, multi-threaded, stable like SciPy's) and the call run again on the copy -- the caller's arrays are const.
template <class T>
void run_host(const HostCall<T> &h)
{
    try { run_host_once<T>(h); return; }
    catch (const RmError &e) { if (e.code != RM_INTERNAL_UNSORTED) throw; }
    if (h.topk_idx) throw RmError{RM_ERR_INVALID, "rm_rank_*: the CSR rows must be sorted (pos_rank is indexed by the caller's entry order)"};
    const size_t nnz_tr = (size_t)h.trp[h.m], nnz_te = (size_t)h.tep[h.m];
    std::vector<int> tri(h.tri, h.tri + nnz_tr), tei(h.tei, h.tei + nnz_te);
    std::vector<T> tev;
    if (h.tev) tev.assign(h.tev, h.tev + nnz_te);
    int rc = rm_csr_sort_rows(h.trp, tri.data(), nullptr, 0, h.m, h.nthreads);
    if (rc == RM_OK) rc = rm_csr_sort_rows(h.tep, tei.data(), h.tev ? (void *)tev.data() : nullptr, h.tev ? (int32_t)sizeof(T) : 0, h.m, h.nthreads);
    if (rc != RM_OK) throw RmError{rc, "sorting a copy of the CSR rows failed"};
    HostCall<T> h2 = h;
    h2.tri = tri.data(); h2.tei = tei.data(); h2.tev = h.tev ? tev.data() : nullptr;
    try { run_host_once<T>(h2); }
    catch (const RmError &e) { if (e.code != RM_INTERNAL_UNSORTED) throw; throw RmError{RM_ERR_INVALID, "CSR rows are not sorted"}; }
}

// device-pointer entry: the same fall-back through the host (the arrays are const device memory of the caller's: the sorted copies live in
// the context's workspace)
template <class T>
void run_dev(Call<T> c, hipStream_t stream, Ctx &cx)
{
    try { run_call<T>(c, stream, cx); return; }
    catch (const RmError &e) { if (e.code != RM_INTERNAL_UNSORTED) throw; }
    const size_t m1 = (size_t)c.m + 1, nnz_tr = (size_t)std::max<long long>(c.nnz_train, 0), nnz_te = (size_t)std::max<long long>(c.nnz_test, 0);
    std::vector<int> trp(m1), tep(m1), tri(std::max<size_t>(nnz_tr, 1)), tei(std::max<size_t>(nnz_te, 1));
    std::vector<T> tev(c.test_v ? std::max<size_t>(nnz_te, 1) : 0);
    HIP_CHECK(hipMemcpyAsync(trp.data(), c.train_p, sizeof(int) * m1, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(tep.data(), c.test_p, sizeof(int) * m1, hipMemcpyDeviceToHost, stream));
    if (nnz_tr) HIP_CHECK(hipMemcpyAsync(tri.data(), c.train_i, sizeof(int) * nnz_tr, hipMemcpyDeviceToHost, stream));
    if (nnz_te) HIP_CHECK(hipMemcpyAsync(tei.data(), c.test_i, sizeof(int) * nnz_te, hipMemcpyDeviceToHost, stream));
    if (c.test_v && nnz_te) HIP_CHECK(hipMemcpyAsync(tev.data(), c.test_v, sizeof(T) * nnz_te, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    int rc = rm_csr_sort_rows(trp.data(), tri.data(), nullptr, 0, c.m, 0);
    if (rc == RM_OK) rc = rm_csr_sort_rows(tep.data(), tei.data(), c.test_v ? (void *)tev.data() : nullptr, c.test_v ? (int32_t)sizeof(T) : 0, c.m, 0);
    if (rc != RM_OK) throw RmError{rc, "sorting a copy of the CSR rows failed"};
    int *dtri = (int *)cx.ws.get("sorted_tri", sizeof(int) * std::max<size_t>(nnz_tr, 1));
    int *dtei = (int *)cx.ws.get("sorted_tei", sizeof(int) * std::max<size_t>(nnz_te, 1));
    T *dtev = c.test_v ? (T *)cx.ws.get("sorted_tev", sizeof(T) * std::max<size_t>(nnz_te, 1)) : nullptr;
    if (nnz_tr) HIP_CHECK(hipMemcpyAsync(dtri, tri.data(), sizeof(int) * nnz_tr, hipMemcpyHostToDevice, stream));
    if (nnz_te) HIP_CHECK(hipMemcpyAsync(dtei, tei.data(), sizeof(int) * nnz_te, hipMemcpyHostToDevice, stream));
    if (dtev && nnz_te) HIP_CHECK(hipMemcpyAsync(dtev, tev.data(), sizeof(T) * nnz_te, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipStreamSynchronize(stream));                         // (the vectors go out of scope below)
    c.train_i = dtri; c.test_i = dtei; c.test_v = dtev;
    try { run_call<T>(c, stream, cx); }
    catch (const RmError &e) { if (e.code != RM_INTERNAL_UNSORTED) throw; throw RmError{RM_ERR_INVALID, "CSR rows are not sorted"}; }
}

template <class T>
void debug_scores(const T *A, size_t lda, const T *B, size_t ldb, int m, int n, int k, T *out)
{
    typedef Prec<T> P;
    constexpr int GU = P::GU;
    if (!A || !B || !out || m <= 0 || n <= 0 || k <= 0) throw RmError{RM_ERR_INVALID, "bad argument"};
    const int NG = P::supported_ng(k);
    if (NG < 0) throw RmError{RM_ERR_UNSUPPORTED, P::limit()};
    Ctx &cx = context(0);
    std::lock_guard<std::mutex> lk(cx.mu);
    Workspace &ws = cx.ws;
    cx.packed_tag = 0;                                       // the packed item image is overwritten below
    hipStream_t stream = nullptr;
    T *dA = (T *)ws.get("in_A", sizeof(T) * (size_t)m * k);
    T *dB = (T *)ws.get("in_B", sizeof(T) * (size_t)n * k);
    HIP_CHECK(hipMemcpy2DAsync(dA, sizeof(T) * k, A, sizeof(T) * lda, sizeof(T) * k, m, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpy2DAsync(dB, sizeof(T) * k, B, sizeof(T) * ldb, sizeof(T) * k, n, hipMemcpyHostToDevice, stream));
    const int n_groups = (m + GU - 1) / GU, n_ublocks = (n_groups + GROUPS_PER_BLOCK - 1) / GROUPS_PER_BLOCK;
    const int tiles_total = (n + TILE_ITEMS - 1) / TILE_ITEMS;
    int *slot_user = (int *)ws.get("slot_user", sizeof(int) * (size_t)m);
    int *zeros = (int *)ws.get("dbg_zeros", sizeof(int) * (size_t)(m + 1 + n_groups));
    long long *grow = (long long *)ws.get("grow", sizeof(long long) * (size_t)n_groups);
    HIP_CHECK(hipMemsetAsync(zeros, 0, sizeof(int) * (size_t)(m + 1 + n_groups), stream));
    HIP_CHECK(hipMemsetAsync(grow, 0, sizeof(long long) * (size_t)n_groups, stream));
    hipLaunchKernelGGL(k_iota<int>, dim3(cdiv(m, 256)), dim3(256), 0, stream, slot_user, m);
    const long long bp_units = P::items_units(tiles_total, NG), ap_units = P::users_units(n_groups, NG);
    typename P::PackT *Bp = (typename P::PackT *)ws.get("Bp", 16 * (size_t)bp_units);
    typename P::PackT *Ap = (typename P::PackT *)ws.get("Ap", 16 * (size_t)ap_units);
    pack_operands(dA, (size_t)k, dB, (size_t)k, n, k, NG, TILE_ITEMS, slot_user, m, Ap, ap_units, Bp, bp_units, stream);
    T *dump = (T *)ws.get("dbg_dump", sizeof(T) * (size_t)m * n);
    const int K = 1;
    typename P::ListT *glists = (typename P::ListT *)ws.get("glists", sizeof(typename P::ListT) * (size_t)n_ublocks * 8 * GU * (2 * K + 32));
    typename P::Args sa{};
    sa.n = n; sa.K = K; sa.ngt = NG; sa.n_slots = m; sa.n_groups = n_groups; sa.n_ublocks = n_ublocks; sa.n_splits = 1; sa.part_splits = 1;
    sa.tiles_total = tiles_total; sa.jmax = 0; sa.check_nan = 1; sa.Ap = (decltype(sa.Ap))Ap; sa.Bp = (decltype(sa.Bp))Bp;
    sa.slot_user = slot_user; sa.slot_chunk = zeros; sa.train_p = zeros; sa.train_i = zeros; sa.gj = zeros + m + 1; sa.grow = grow;
    sa.glists = glists; sa.dump = dump;
    P::set_sync(sa, (int)P::lds_b(NG));
    dispatch_sweep(false, true, false, 2, NG, dim3(n_ublocks), P::lds_b(NG) + SYNC_BYTES, stream, sa);
    HIP_CHECK(hipMemcpyAsync(out, dump, sizeof(T) * (size_t)m * n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
}

} // namespace

#ifdef RM_CSTATS
extern "C" int rm_debug_stats_collect(unsigned long long *out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(rm::g_cstats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(rm::g_cstats), z, sizeof(z)); }
    return 0;
}
#endif
// =====================================================================================================================
// C-ABI
// =====================================================================================================================
#define RM_HOST_ENTRY(SUFFIX, T)                                                                                        \
extern "C" int rm_calc_metrics_##SUFFIX(                                                                                \
    const T *A, size_t lda, const T *B, size_t ldb, int32_t m, int32_t n, int32_t k,                                    \
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i,                                                           \
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i, const T *Xtest_csr,                                         \
    int32_t k_metrics, int cumulative, int break_ties_with_noise,                                                       \
    T *p_at_k, T *tp_at_k, T *r_at_k, T *ap_at_k, T *tap_at_k, T *ndcg_at_k, T *hit_at_k, T *rr_at_k,                   \
    T *roc_auc, T *pr_auc, int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,                       \
    int32_t nthreads, uint64_t seed)                                                                                    \
{                                                                                                                       \
    return guarded([&] {                                                                                                \
        HostCall<T> h{A, lda, B, ldb, m, n, k, Xtrain_csr_p, Xtrain_csr_i, Xtest_csr_p, Xtest_csr_i, Xtest_csr,         \
                      k_metrics, cumulative != 0, break_ties_with_noise != 0,                                           \
                      {p_at_k, tp_at_k, r_at_k, ap_at_k, tap_at_k, ndcg_at_k, hit_at_k, rr_at_k, roc_auc, pr_auc},      \
                      consider_cold_start != 0, min_items_pool, min_pos_test, nullptr, nullptr, nullptr, nullptr, seed};\
        h.nthreads = nthreads > 0 ? nthreads : 0;                                                                       \
        run_host<T>(h);                                                                                                 \
    });                                                                                                                 \
}                                                                                                                       \
extern "C" int rm_calc_metrics_dev_##SUFFIX(                                                                            \
    const T *A, size_t lda, const T *B, size_t ldb, int32_t m, int32_t n, int32_t k,                                    \
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i, int64_t nnz_train,                                        \
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i, const T *Xtest_csr, int64_t nnz_test,                       \
    int32_t k_metrics, int cumulative, int break_ties_with_noise,                                                       \
    T *p_at_k, T *tp_at_k, T *r_at_k, T *ap_at_k, T *tap_at_k, T *ndcg_at_k, T *hit_at_k, T *rr_at_k,                   \
    T *roc_auc, T *pr_auc, int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,                       \
    uint64_t seed, void *stream)                                                                                        \
{                                                                                                                       \
    return guarded([&] {                                                                                                \
        if (m == 0) return;                                                                                             \
        validate(A, B, m, n, k, Xtrain_csr_p, Xtest_csr_p, Xtest_csr_i, (long long)nnz_test, k_metrics, lda, ldb);      \
        if (ndcg_at_k && !Xtest_csr) throw RmError{RM_ERR_INVALID, "NDCG requested without test values"};               \
        Call<T> c{};                                                                                                    \
        c.A = A; c.lda = lda; c.B = B; c.ldb = ldb; c.m = m; c.n = n; c.k = k;                                          \
        c.train_p = Xtrain_csr_p; c.train_i = Xtrain_csr_i; c.nnz_train = nnz_train;                                    \
        c.test_p = Xtest_csr_p; c.test_i = Xtest_csr_i; c.test_v = Xtest_csr; c.nnz_test = nnz_test;                    \
        c.K = k_metrics; c.cumulative = cumulative != 0; c.noise = break_ties_with_noise != 0;                          \
        T *outs[10] = {p_at_k, tp_at_k, r_at_k, ap_at_k, tap_at_k, ndcg_at_k, hit_at_k, rr_at_k, roc_auc, pr_auc};      \
        for (int i = 0; i < 10; i++) c.out[i] = outs[i];                                                                \
        c.cold = consider_cold_start != 0; c.min_items_pool = min_items_pool; c.min_pos_test = min_pos_test;            \
        Ctx &cx = context(0);                                                                                           \
        std::lock_guard<std::mutex> lk(cx.mu);                                                                          \
        cx.acc[0] = cx.acc[1] = cx.acc[2] = cx.acc[3] = 0;                                                              \
        c.seed = seed; c.user0 = 0;                                                                                     \
        run_dev<T>(c, (hipStream_t)stream, cx);                                                                         \
    });                                                                                                                 \
}                                                                                                                       \
extern "C" int rm_rank_##SUFFIX(                                                                                        \
    const T *A, size_t lda, const T *B, size_t ldb, int32_t m, int32_t n, int32_t k,                                    \
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i, const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i,   \
    int32_t k_metrics, int break_ties_with_noise, int consider_cold_start, int32_t min_items_pool,                      \
    int32_t min_pos_test, uint64_t seed, int32_t *topk_idx, T *topk_score, int64_t *pos_rank, int32_t *status)          \
{                                                                                                                       \
    return guarded([&] {                                                                                                \
        if (!topk_idx || !topk_score || !pos_rank || !status) throw RmError{RM_ERR_INVALID, "null output pointer"};     \
        std::vector<T> ap((size_t)std::max(m, 1)), roc((size_t)std::max(m, 1)), pr((size_t)std::max(m, 1));             \
        HostCall<T> h{A, lda, B, ldb, m, n, k, Xtrain_csr_p, Xtrain_csr_i, Xtest_csr_p, Xtest_csr_i, (const T *)nullptr,\
                      k_metrics, false, break_ties_with_noise != 0,                                                     \
                      {nullptr, nullptr, nullptr, ap.data(), nullptr, nullptr, nullptr, nullptr, roc.data(), pr.data()},\
                      consider_cold_start != 0, min_items_pool, min_pos_test,                                           \
                      topk_idx, topk_score, (long long *)pos_rank, status, seed};                                       \
        run_host<T>(h);                                                                                                 \
    });                                                                                                                 \
}

RM_HOST_ENTRY(f32, float)
RM_HOST_ENTRY(f64, double)

extern "C" int rm_debug_scores_f32(const float *A, size_t lda, const float *B, size_t ldb, int32_t m, int32_t n, int32_t k, float *out)
{
    return guarded([&] { debug_scores<float>(A, lda, B, ldb, m, n, k, out); });
}

extern "C" int rm_debug_scores_f64(const double *A, size_t lda, const double *B, size_t ldb, int32_t m, int32_t n, int32_t k, double *out)
{
    return guarded([&] { debug_scores<double>(A, lda, B, ldb, m, n, k, out); });
}

extern "C" int rm_has_openmp(void) { return 1; }

extern "C" const char *rm_last_error(void) { return g_err.c_str(); }

extern "C" int rm_device_count(void)
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

extern "C" int rm_set_device(int device)
{
    return guarded([&] { HIP_CHECK(hipSetDevice(device)); });
}

extern "C" int rm_set_devices(const int32_t *devices, int32_t n)
{
    return guarded([&] {
        if (n < 0 || (n > 0 && !devices)) throw RmError{RM_ERR_INVALID, "bad device list"};
        int count = 0;
        HIP_CHECK(hipGetDeviceCount(&count));
        std::vector<int> v;
        for (int i = 0; i < n; i++) {
            if (devices[i] < 0 || devices[i] >= count) throw RmError{RM_ERR_INVALID, "device " + std::to_string(devices[i]) + " does not exist (" + std::to_string(count) + " visible)"};
            v.push_back(devices[i]);
        }
        std::lock_guard<std::mutex> lk(g_dev_mu);
        g_devices = v;
    });
}

extern "C" int rm_get_devices(int32_t *devices, int32_t cap)
{
    std::lock_guard<std::mutex> lk(g_dev_mu);
    for (int i = 0; i < (int)g_devices.size() && i < cap; i++) devices[i] = g_devices[i];
    return (int)g_devices.size();
}

extern "C" void rm_request_interrupt(void) { g_interrupt = 1; }

extern "C" void rm_debug_reload_switches(void) { g_sw.load(); }

extern "C" int rm_get_timings(double *out, int n)
{
    Ctx *cx = g_last_ctx;
    if (!out || n <= 0 || !cx || !cx->ev_valid) return 0;
    float a = 0, b = 0, c = 0, d = 0;
    if (cx->ev_recorded) {
        if (hipEventSynchronize(cx->ev[3]) != hipSuccess) return 0;
        (void)hipEventElapsedTime(&a, cx->ev[0], cx->ev[1]);
        (void)hipEventElapsedTime(&b, cx->ev[1], cx->ev[2]);
        (void)hipEventElapsedTime(&c, cx->ev[2], cx->ev[3]);
        (void)hipEventElapsedTime(&d, cx->ev[0], cx->ev[3]);
    }
    cx->timings[0] = cx->acc[0] + a; cx->timings[1] = cx->acc[1] + b; cx->timings[2] = cx->acc[2] + c; cx->timings[3] = cx->acc[3] + d;
    const int cnt = n < 10 ? n : 10;
    for (int i = 0; i < cnt && i < 8; i++) out[i] = cx->timings[i];
    if (cnt > 8) out[8] = cx->timed_slots;
    if (cnt > 9) out[9] = cx->total_slots;
    return cnt;
}

extern "C" int rm_release_workspace(void)
{
    return guarded([&] {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        // (contexts are never destroyed, so the pointers stay valid; a context's mutex is NOT taken under g_ctx_mu: a running
        // call holds its context and may ask for a peer context, which takes g_ctx_mu -- the other order would deadlock)
        std::vector<Ctx *> mine;
        {
            std::lock_guard<std::mutex> lk(g_ctx_mu);
            for (auto &kv : g_ctx) if (kv.first.first == dev) mine.push_back(kv.second.get());
        }
        for (Ctx *c : mine) {
            std::lock_guard<std::mutex> cl(c->mu);
            c->ws.release();
            // (every marker of cached CONTENT goes with the buffers: a fresh allocation may come back at the old address)
            c->packed_tag = 0; c->packed_ptr = nullptr; c->bits_tag = 0; c->bits_ptr = nullptr; c->log2_ptr = nullptr; c->log2_K = 0;
        }
    });
}
