/* rm_r_shim.c -- see rm_r_shim.h.  Plain C over include/recometrics_hip.h; no R headers. */
#include "rm_r_shim.h"

#include <stdlib.h>
#include <string.h>

#include "recometrics_hip.h"

void rm_r_nan_to_na(double *x, size_t count)
{
    const uint64_t na = RM_R_NA_REAL_BITS;
    for (size_t i = 0; i < count; i++)
        if (x[i] != x[i]) memcpy(&x[i], &na, sizeof(na));
}

static size_t out_count(int which, int32_t m, int32_t k_metrics, int cumulative)
{
    return (which >= 8 || !cumulative) ? (size_t)m : (size_t)m * (size_t)k_metrics;
}

int rm_r_calc_metrics_f64(const double *A, const double *B, int32_t m, int32_t n, int32_t k,
                          const int *Xtrain_csr_p, const int *Xtrain_csr_i, const int *Xtest_csr_p, const int *Xtest_csr_i,
                          const double *Xtest_csr, int32_t k_metrics, int cumulative, int break_ties_with_noise,
                          double *const o[10], int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,
                          int32_t nthreads, uint64_t seed)
{
    /* a column-major [k x m] matrix IS the row-major [m x k] one: leading dimension k */
    const int rc = rm_calc_metrics_f64(A, (size_t)k, B, (size_t)k, m, n, k,
                                       (const int32_t *)Xtrain_csr_p, (const int32_t *)Xtrain_csr_i,
                                       (const int32_t *)Xtest_csr_p, (const int32_t *)Xtest_csr_i, Xtest_csr,
                                       k_metrics, cumulative, break_ties_with_noise,
                                       o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9],
                                       consider_cold_start, min_items_pool, min_pos_test, nthreads, seed);
    if (rc != RM_OK) return rc;
    for (int i = 0; i < 10; i++)
        if (o[i]) rm_r_nan_to_na(o[i], out_count(i, m, k_metrics, cumulative));
    return RM_OK;
}

int rm_r_calc_metrics_f32(const int *A_bits, const int *B_bits, int32_t m, int32_t n, int32_t k,
                          const int *Xtrain_csr_p, const int *Xtrain_csr_i, const int *Xtest_csr_p, const int *Xtest_csr_i,
                          const double *Xtest_csr, int32_t k_metrics, int cumulative, int break_ties_with_noise,
                          int *const o[10], int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test,
                          int32_t nthreads, uint64_t seed)
{
    float *tev = NULL;
    if (o[5] && m > 0) {                               /* the values are only read for NDCG (Rwrapper.cpp:245-249) */
        const size_t nnz = (size_t)Xtest_csr_p[m];
        tev = (float *)malloc(sizeof(float) * (nnz ? nnz : 1));
        if (!tev) return RM_ERR_NOMEM;
        for (size_t i = 0; i < nnz; i++) tev[i] = (float)Xtest_csr[i];
    }
    /* an int that holds the bits of a float is a float in memory: sizeof(int) == sizeof(float) == 4 is what package `float`
     * itself relies on */
    const int rc = rm_calc_metrics_f32((const float *)A_bits, (size_t)k, (const float *)B_bits, (size_t)k, m, n, k,
                                       (const int32_t *)Xtrain_csr_p, (const int32_t *)Xtrain_csr_i,
                                       (const int32_t *)Xtest_csr_p, (const int32_t *)Xtest_csr_i, tev,
                                       k_metrics, cumulative, break_ties_with_noise,
                                       (float *)o[0], (float *)o[1], (float *)o[2], (float *)o[3], (float *)o[4],
                                       (float *)o[5], (float *)o[6], (float *)o[7], (float *)o[8], (float *)o[9],
                                       consider_cold_start, min_items_pool, min_pos_test, nthreads, seed);
    free(tev);
    return rc;
}
