/* rm_r_shim.h -- the R-specific part of the Rcpp glue that does not need R's headers, so that it can be compiled and tested
 * without an R installation (tests/test_r_shim.py drives it through ctypes exactly the way Rwrapper_hip.cpp does).
 *
 * What is R-specific about the call (reference src/Rwrapper.cpp, src/recometrics.hpp):
 *   - the factor matrices arrive COLUMN-major [k, m] / [k, n] (R matrices, t(A) on the R side: Rwrapper.cpp:91-93), which is
 *     the row-major [m, k] the C-ABI wants with lda = ldb = k (Rwrapper.cpp:250-253);
 *   - float32 data lives in INTEGER vectors (package `float`: the bits of a float in an int, Rwrapper.cpp:51-59,:353-358),
 *     inputs and outputs alike;
 *   - Xtest_csr is always a double vector; the float path converts it, and only when NDCG is requested (Rwrapper.cpp:241-249);
 *   - double outputs carry R's NA_real_ (a NaN with payload 1954) instead of a plain quiet NaN for users that cannot be
 *     evaluated (src/recometrics.hpp:75-80 `NAN_` under _FOR_R; Rwrapper.cpp:605-612 C_NAN_to_R_NA); float outputs keep a
 *     plain NaN (the R side of package `float` has no NA bit pattern of its own).
 * Outputs are indexed p, tp, r, ap, tap, ndcg, hit, rr, roc_auc, pr_auc; a NULL entry = not requested. */
#ifndef RM_R_SHIM_H
#define RM_R_SHIM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RM_R_NA_REAL_BITS 0x7FF00000000007A2ULL      /* R's NA_real_: R_NaReal = {hi 0x7FF00000, lo 1954} (arithmetic.c) */

/* every NaN of x[0..count) becomes NA_real_ (what Rwrapper.cpp:605-612 does for the R caller) */
void rm_r_nan_to_na(double *x, size_t count);

/* calc_metrics_double of Rwrapper.cpp:292-350: A [k x m], B [k x n] column-major doubles */
int rm_r_calc_metrics_f64(const double *A, const double *B, int32_t m, int32_t n, int32_t k,
                          const int *Xtrain_csr_p, const int *Xtrain_csr_i, const int *Xtest_csr_p, const int *Xtest_csr_i,
                          const double *Xtest_csr, int32_t k_metrics, int cumulative, int break_ties_with_noise,
                          double *const outs[10], int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,
                          int32_t nthreads, uint64_t seed);

/* calc_metrics_float of Rwrapper.cpp:352-410: A, B and the outputs are float bits in int storage; Xtest_csr is double and
 * has `nnz_test` entries (Xtest_csr_p[m]) */
int rm_r_calc_metrics_f32(const int *A_bits, const int *B_bits, int32_t m, int32_t n, int32_t k,
                          const int *Xtrain_csr_p, const int *Xtrain_csr_i, const int *Xtest_csr_p, const int *Xtest_csr_i,
                          const double *Xtest_csr, int32_t k_metrics, int cumulative, int break_ties_with_noise,
                          int *const outs_bits[10], int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,
                          int32_t nthreads, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
