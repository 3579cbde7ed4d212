/* recometrics_hip.h -- C-ABI of librecometrics_hip.so: the MI355X (gfx950) drop-in for the hot path of
 * david-cortes/recometrics.
 *
 * Boundary replaced (reference files, relative to the reference repository root):
 *   src/recometrics_signatures.hpp:46       bool get_has_openmp()
 *   src/recometrics_signatures.hpp:48-72    void calc_metrics_double(...30 parameters...)
 *   src/recometrics_signatures.hpp:74-98    void calc_metrics_float(...30 parameters...)
 * which recometrics/wrapper.pyx:282-304,381-403 (Cython) calls today and which src/Rwrapper.cpp:250-274 (Rcpp)
 * reaches through the template src/recometrics.hpp:359.  The entry points below keep the reference's PARAMETER
 * ORDER and meaning one-for-one; the differences are the ones a C-ABI needs: `bool` -> `int`, C linkage, and an
 * `int` status return (0 = ok) instead of C++ exceptions (std::bad_alloc / std::runtime_error in the reference,
 * src/recometrics.hpp:395-397,:171).  rm_last_error() returns the message for the calling thread.
 *
 * Ownership (same as the reference, src/recometrics.hpp:193-358): the caller owns every buffer; outputs are
 * pre-allocated by the binding ([m] or row-major [m x k_metrics] when `cumulative`); a NULL output pointer means
 * "metric not requested"; inputs are never modified; users that cannot be evaluated get NaN in every requested
 * output, so outputs need no initialisation.  CSR is 0-based int32.  The reference's callers sort the column indices of every row
 * first (recometrics/__init__.py:35-41, R/recometrics.R) and nobody checks them; here the arrays are VALIDATED on the device
 * before anything indexes by them: index pointers that are negative, decreasing or beyond the index array, and column indices
 * outside [0, n), are RM_ERR_INVALID with a message that names the row (on the CPU: a segfault); rows that are merely unsorted
 * are sorted by the library -- a copy, the caller's arrays are const -- and give the outputs of the sorted matrix (rm_rank_*
 * excepted: its pos_rank is indexed by the caller's entry order, unsorted rows are RM_ERR_INVALID there).  An item listed twice
 * in a row is tolerated as the reference tolerates it.
 * On a status other than RM_OK the contents of the output arrays are undefined, with one exception: RM_ERR_INTERRUPTED (below).
 * `nthreads` is accepted for signature compatibility (the device path has no host thread pool).  `seed` is USED:
 * with `break_ties_with_noise` the reference's tie-breaking noise -- std::mt19937(seed + user) through libstdc++'s
 * uniform_real_distribution, reference :528-534 -- is reproduced bit for bit (rm_noise.hpp; DESIGN.md, "tie noise"),
 * so pass the same seed the reference would get.
 * m == 0 returns RM_OK and writes nothing (the reference's loop over users does not run).  Xtest_csr_i may be NULL
 * when Xtest_csr_p[m] == 0.  Deviation: NDCG requested with Xtest_csr == NULL is RM_ERR_INVALID (the reference
 * dereferences the null pointer in its normalisation, :870-874).
 *
 * No torch / HIP types appear in any signature: pointers, sizes and scalars only.
 */
#ifndef RECOMETRICS_HIP_H
#define RECOMETRICS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define RM_OK 0
#define RM_ERR_INVALID 1     /* bad argument (null pointer, negative size, n or k == 0, k_metrics > n ...) */
#define RM_ERR_HIP 2         /* HIP runtime failure (message has the hipError string) */
#define RM_ERR_NOMEM 3       /* device or host allocation failed (the reference throws std::bad_alloc) */
#define RM_ERR_UNSUPPORTED 4 /* shape outside what the kernels are built for (message says which) */
#define RM_ERR_INTERRUPTED 5 /* SIGINT (or rm_request_interrupt) during a host-pointer call: "Error: procedure was interrupted."
                              * -- the std::runtime_error of src/recometrics.hpp:171.  Users of finished batches have their
                              * results; the rest of the output arrays is untouched, as in the reference (:488-489).  With
                              * break_ties_with_noise in fp32, users of finished batches whose exact (noise) evaluation had not
                              * run yet are set to NaN rather than left with their un-noised first-pass values */

/* replaces calc_metrics_float  (src/recometrics_signatures.hpp:74-98).  ALL pointers are HOST pointers. */
int rm_calc_metrics_f32(
    const float *A, size_t lda, const float *B, size_t ldb,
    int32_t m, int32_t n, int32_t k,
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i,
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i, const float *Xtest_csr,
    int32_t k_metrics, int cumulative, int break_ties_with_noise,
    float *p_at_k, float *tp_at_k, float *r_at_k, float *ap_at_k, float *tap_at_k,
    float *ndcg_at_k, float *hit_at_k, float *rr_at_k, float *roc_auc, float *pr_auc,
    int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,
    int32_t nthreads, uint64_t seed);

/* replaces calc_metrics_double (src/recometrics_signatures.hpp:48-72).  ALL pointers are HOST pointers. */
int rm_calc_metrics_f64(
    const double *A, size_t lda, const double *B, size_t ldb,
    int32_t m, int32_t n, int32_t k,
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i,
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i, const double *Xtest_csr,
    int32_t k_metrics, int cumulative, int break_ties_with_noise,
    double *p_at_k, double *tp_at_k, double *r_at_k, double *ap_at_k, double *tap_at_k,
    double *ndcg_at_k, double *hit_at_k, double *rr_at_k, double *roc_auc, double *pr_auc,
    int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,
    int32_t nthreads, uint64_t seed);

/* Same contracts with every pointer a DEVICE pointer on the current device (inputs already resident in HBM,
 * outputs written to HBM); `stream` is a hipStream_t passed as void* (NULL = default stream).  Asynchronous
 * apart from one small plan read-back; the caller synchronises the stream before reading the outputs.
 * `nnz_train` / `nnz_test` are the lengths of the index arrays (the host variants read them from indptr[m]). */
int rm_calc_metrics_dev_f32(
    const float *A, size_t lda, const float *B, size_t ldb,
    int32_t m, int32_t n, int32_t k,
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i, int64_t nnz_train,
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i, const float *Xtest_csr, int64_t nnz_test,
    int32_t k_metrics, int cumulative, int break_ties_with_noise,
    float *p_at_k, float *tp_at_k, float *r_at_k, float *ap_at_k, float *tap_at_k,
    float *ndcg_at_k, float *hit_at_k, float *rr_at_k, float *roc_auc, float *pr_auc,
    int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,
    uint64_t seed, void *stream);

int rm_calc_metrics_dev_f64(
    const double *A, size_t lda, const double *B, size_t ldb,
    int32_t m, int32_t n, int32_t k,
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i, int64_t nnz_train,
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i, const double *Xtest_csr, int64_t nnz_test,
    int32_t k_metrics, int cumulative, int break_ties_with_noise,
    double *p_at_k, double *tp_at_k, double *r_at_k, double *ap_at_k, double *tap_at_k,
    double *ndcg_at_k, double *hit_at_k, double *rr_at_k, double *roc_auc, double *pr_auc,
    int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,
    uint64_t seed, void *stream);

/* Ranking introspection (host pointers): the ordered top-K item ids / scores per user ([m x k_metrics], -1 / NaN
 * padded), the 1-based position of every test item in the user's FULL candidate ranking ([nnz_test], 0 when the
 * item is masked by the train row or the user is skipped) and status[m] (0 = ranked, 1 = user skipped).  These are
 * the quantities the parity contract pins bit-exactly (top-K index sets, hit counts, positive ranks). */
int rm_rank_f32(
    const float *A, size_t lda, const float *B, size_t ldb, int32_t m, int32_t n, int32_t k,
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i,
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i,
    int32_t k_metrics, int break_ties_with_noise, int consider_cold_start,
    int32_t min_items_pool, int32_t min_pos_test, uint64_t seed,
    int32_t *topk_idx, float *topk_score, int64_t *pos_rank, int32_t *status);

int rm_rank_f64(
    const double *A, size_t lda, const double *B, size_t ldb, int32_t m, int32_t n, int32_t k,
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i,
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i,
    int32_t k_metrics, int break_ties_with_noise, int consider_cold_start,
    int32_t min_items_pool, int32_t min_pos_test, uint64_t seed,
    int32_t *topk_idx, double *topk_score, int64_t *pos_rank, int32_t *status);

/* Dense score matrix out[m x n] (host) computed by the sweep's own MFMA contraction -- test hook that pins the
 * "bit-identical to the k-ordered fma chain" claim (reference src/recometrics.hpp:99-112). */
int rm_debug_scores_f32(const float *A, size_t lda, const float *B, size_t ldb,
                        int32_t m, int32_t n, int32_t k, float *out);
int rm_debug_scores_f64(const double *A, size_t lda, const double *B, size_t ldb,
                        int32_t m, int32_t n, int32_t k, double *out);

/* replaces get_has_openmp (src/recometrics_signatures.hpp:46): host threads are irrelevant here; always 1 so that
 * the reference's "built without multi-threading" warning (recometrics/__init__.py:524-529) never fires. */
int rm_has_openmp(void);

const char *rm_last_error(void);       /* message of the last failing call on this thread ("" if none) */
int rm_device_count(void);             /* visible HIP devices (0 when there is no GPU / driver) */
int rm_set_device(int device);         /* device used by subsequent calls on this thread */

/* Devices of the HOST-pointer calls (rm_calc_metrics_f32/f64, rm_rank_*) of this process: with n > 1 a call shards its users
 * inside the library -- contiguous ranges [m g / n, m (g + 1) / n), one host thread + stream + workspace per entry, the item
 * factors uploaded once and copied device-to-device, every shard's results written straight into the caller's arrays; no
 * exchange between shards (reference: the OpenMP loop over users, src/recometrics.hpp:428-437).  The same device may be
 * listed more than once.  n = 0 restores "the calling thread's current device".  Results do not depend on the list. */
int rm_set_devices(const int32_t *devices, int32_t n);
int rm_get_devices(int32_t *devices, int32_t cap);      /* returns the length of the list */

/* Asks the host-pointer calls in flight to stop at their next batch boundary (what SIGINT does during such a call,
 * src/recometrics.hpp:114-174); they return RM_ERR_INTERRUPTED. */
void rm_request_interrupt(void);

/* Test hook: the RM_DEBUG_* / RM_*_MB switches (tests and A/B timing; they choose among code paths with
 * identical results) are read from the environment once, when the library is loaded; this reads them again.  Not for use
 * while a call is running. */
void rm_debug_reload_switches(void);

/* Timings of the most recent successful call on this thread, milliseconds measured with HIP events on the call's
 * stream: out[0] plan+pack+positives, out[1] sweep kernel, out[2] finalize, out[3] whole device section;
 * out[4] = launches of the sweep kernel, out[5] = item splits, out[6] = sweep blocks, out[7] = dynamic LDS bytes,
 * out[8] = user lanes (slots) of the launch that out[1] brackets -- when the blocks of streamed users run beside the main
 * launch on a second stream, out[1] times the main launch only --, out[9] = user lanes of the whole call.
 * Forces a synchronisation of that stream.  Returns the number of values written. */
int rm_get_timings(double *out, int n);

/* Releases the cached device workspace of the current device. */
int rm_release_workspace(void);

/* ---- train / test splitting (host only; "next" row N3) ------------------------------------------------------------
 * Replaces split_data_selected_users_{float,double}, split_data_separate_users_* and split_data_joined_users_*
 * (src/recometrics_signatures.hpp:100-220), which hand their results back in std::vector& parameters; a C-ABI returns an
 * opaque result instead.  mode 0 = every row is split ("all"), 1 = separated (test users drawn at random; outputs
 * rem / train / test / users_test), 2 = joined (train = [train of the test users ; remaining users]).  `n_users_test`,
 * `consider_cold_start`, `min_items_pool`, `min_pos_test` are ignored in mode 0.  Results are bit-identical to the
 * reference's for the same seed (both consume std::mt19937 through std::shuffle in the same order).
 * Arrays of a result, `which`: 0-2 train (indptr, indices, values), 3-5 test, 6-8 rem, 9 users_test. */
int rm_split_f32(const int32_t *X_csr_p, const int32_t *X_csr_i, const float *X_csr, int32_t m, int32_t n, int mode,
                 int32_t n_users_test, double test_fraction, int consider_cold_start, int32_t min_items_pool,
                 int32_t min_pos_test, uint64_t seed, void **result);
int rm_split_f64(const int32_t *X_csr_p, const int32_t *X_csr_i, const double *X_csr, int32_t m, int32_t n, int mode,
                 int32_t n_users_test, double test_fraction, int consider_cold_start, int32_t min_items_pool,
                 int32_t min_pos_test, uint64_t seed, void **result);
int64_t rm_split_size(const void *result, int which);           /* number of elements of array `which` (-1 = bad argument) */
int rm_split_copy(const void *result, int which, void *dst);    /* copies array `which` into dst (caller-sized) */
void rm_split_free(void *result);
const char *rm_split_last_error(void);

/* ---- CSR normalisation in front of the metric call (host only) --------------------------------------------------------
 * Replaces the `X.sort_indices()` of the reference's Python caller (recometrics/__init__.py:35-41 `_as_csr`, applied to X_train
 * and X_test at :553-562; the R caller's `sort_sparse_indices`, R/recometrics.R): SciPy's single-threaded pass over every
 * stored entry becomes a pass over row ranges on `nthreads` host threads (<= 0: all).
 * rm_csr_rows_sorted: 1 = the column indices of every row ascend (equal neighbours allowed, as SciPy's has_sorted_indices),
 *   0 = some row does not, -1 = bad argument.  rm_csr_sort_rows: sorts every row's (index, value) pairs by index, in place,
 *   stable; `values` are `value_bytes` (0 = no values, 4, 8) wide. */
int rm_csr_rows_sorted(const int32_t *indptr, const int32_t *indices, int32_t m, int32_t nthreads);
int rm_csr_sort_rows(const int32_t *indptr, int32_t *indices, void *values, int32_t value_bytes, int32_t m, int32_t nthreads);

#ifdef __cplusplus
}
#endif
#endif /* RECOMETRICS_HIP_H */
