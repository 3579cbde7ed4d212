// rm_csr.cpp -- host-side normalisation of the CSR inputs in front of the metric call: "are the column indices of every row
// sorted?" and, when they are not, the in-place sort of every row's (index, value) pairs.
//
// The reference's Python caller does this with SciPy (recometrics/__init__.py:35-41 `_as_csr` -> `X.sort_indices()`, called at
// :553-562 on X_train and X_test), a single-threaded pass over every stored entry whenever SciPy does not already know the
// answer -- 26-33 ms for the two matrices of BASELINE C2 (20 M entries) in front of a device call of 10 ms.  Here the pass
// is cut into row ranges of equal entry counts over `nthreads` host threads (the one place where the reference's `nthreads`
// argument means something on this path): the check is a streaming read at memory bandwidth.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/recometrics_hip.h"

namespace {

// rows [r0, r1) of thread t when the entries are split evenly: row boundaries found by bisection on indptr
inline void row_ranges(const int32_t *p, int32_t m, int nt, std::vector<int32_t> &cuts)
{
    cuts.assign((size_t)nt + 1, 0);
    const int64_t nnz = p[m];
    for (int t = 1; t < nt; t++) {
        const int64_t want = nnz * t / nt;
        cuts[t] = (int32_t)(std::lower_bound(p, p + m + 1, (int32_t)want) - p);
        if (cuts[t] > m) cuts[t] = m;
        if (cuts[t] < cuts[t - 1]) cuts[t] = cuts[t - 1];
    }
    cuts[nt] = m;
}

inline int pick_threads(int32_t nthreads, int64_t nnz)
{
    int hw = (int)std::thread::hardware_concurrency();
    if (hw <= 0) hw = 1;
    int nt = nthreads > 0 ? std::min<int>(nthreads, hw) : hw;
    nt = std::min(nt, 16);                                        // a memory-bound pass: 1.25 ms on 16 threads, 1.66 on 32 for BASELINE C2's 20 M entries
    nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, nnz / (1 << 18)));     // thread start-up ~ 20 us: not for small inputs
    return std::max(nt, 1);
}

// Runs f(r0, r1) over row ranges on up to `nthreads` threads.  Nothing a worker throws leaves its thread (that would be
// std::terminate, i.e. the death of the Python process): the first exception is kept and rethrown on the calling thread once
// every worker has been joined.  A thread that cannot be started (std::system_error: thread or pid limit) is not fatal either:
// the threads already running are joined by the guard, and the ranges nobody took are run inline.
template <class F> void parallel_rows(const int32_t *p, int32_t m, int32_t nthreads, F &&f)
{
    const int nt = pick_threads(nthreads, p[m]);
    if (nt == 1) { f(0, m); return; }
    std::vector<int32_t> cuts;
    row_ranges(p, m, nt, cuts);
    std::mutex err_mu;
    std::exception_ptr err;
    auto guarded = [&](int t) {
        try { f(cuts[t], cuts[t + 1]); }
        catch (...) { std::lock_guard<std::mutex> lk(err_mu); if (!err) err = std::current_exception(); }
    };
    struct Joiner {
        std::vector<std::thread> th;
        ~Joiner() { for (auto &x : th) if (x.joinable()) x.join(); }
    } workers;
    workers.th.reserve((size_t)nt);
    int started = 1;                                             // range 0 is the calling thread's
    try {
        for (int t = 1; t < nt; t++) { workers.th.emplace_back(guarded, t); started = t + 1; }
    } catch (const std::system_error &) { /* no more threads: the rest runs inline below */ }
    guarded(0);
    for (int t = started; t < nt; t++) guarded(t);
    for (auto &x : workers.th) x.join();
    if (err) std::rethrow_exception(err);
}

template <class V> void sort_row(int32_t *idx, V *val, int32_t len, std::vector<std::pair<int32_t, V>> &tmp)
{
    if (std::is_sorted(idx, idx + len)) return;
    tmp.resize((size_t)len);
    for (int32_t e = 0; e < len; e++) tmp[e] = std::make_pair(idx[e], val[e]);
    // (stable: entries that repeat a column keep their order, as scipy's csr_sort_indices leaves them)
    std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, V> &a, const std::pair<int32_t, V> &b) { return a.first < b.first; });
    for (int32_t e = 0; e < len; e++) { idx[e] = tmp[e].first; val[e] = tmp[e].second; }
}

} // namespace

extern "C" int rm_csr_rows_sorted(const int32_t *indptr, const int32_t *indices, int32_t m, int32_t nthreads)
{
    if (!indptr || m < 0) return -1;
    if (m == 0 || indptr[m] == 0) return 1;
    if (!indices) return -1;
    std::atomic<int> ok{1};
    try {
    parallel_rows(indptr, m, nthreads, [&](int32_t r0, int32_t r1) {
        for (int32_t r = r0; r < r1 && ok.load(std::memory_order_relaxed); r++) {
            const int32_t *a = indices + indptr[r], *b = indices + indptr[r + 1];
            int bad = 0;
            for (const int32_t *q = a; q + 1 < b; q++) bad |= q[0] > q[1];       // (branch-free inner loop: vectorises)
            if (bad) { ok.store(0, std::memory_order_relaxed); return; }
        }
    });
    } catch (...) { return -1; }
    return ok.load();
}

extern "C" int rm_csr_sort_rows(const int32_t *indptr, int32_t *indices, void *values, int32_t value_bytes, int32_t m, int32_t nthreads)
{
    if (!indptr || m < 0 || (value_bytes != 0 && value_bytes != 4 && value_bytes != 8)) return RM_ERR_INVALID;
    if (m == 0 || indptr[m] == 0) return RM_OK;
    if (!indices || (value_bytes && !values)) return RM_ERR_INVALID;
    try {
        parallel_rows(indptr, m, nthreads, [&](int32_t r0, int32_t r1) {
            std::vector<std::pair<int32_t, uint32_t>> t4;
            std::vector<std::pair<int32_t, uint64_t>> t8;
            for (int32_t r = r0; r < r1; r++) {
                const int32_t o = indptr[r], len = indptr[r + 1] - o;
                if (len < 2) continue;
                if (value_bytes == 8) sort_row<uint64_t>(indices + o, (uint64_t *)values + o, len, t8);
                else if (value_bytes == 4) sort_row<uint32_t>(indices + o, (uint32_t *)values + o, len, t4);
                else std::sort(indices + o, indices + o + len);
            }
        });
    } catch (...) { return RM_ERR_NOMEM; }
    return RM_OK;
}
