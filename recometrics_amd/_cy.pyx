# cython: language_level=3, boundscheck=False, wraparound=False
"""Cython binding of include/recometrics_hip.h -- the compiled counterpart of recometrics_amd/_binding.py (ctypes), and the
shape a maintainer of the reference would give recometrics/wrapper.pyx:226-495 (see INTEGRATION.md): raw buffers out of the
NumPy objects, outputs pre-allocated here (size 0 == NULL == "metric not requested"), the device call outside the GIL,
status codes turned into the exceptions `except +` produces in the reference.

Built in-tree by recometrics_amd.build.build_cython() (cython -> C -> gcc, linked against csrc/librecometrics_hip.so)."""
from libc.stdint cimport int32_t, uint64_t
import numpy as np

cdef extern from "recometrics_hip.h":
    int rm_calc_metrics_f32(
        const float *A, size_t lda, const float *B, size_t ldb, int32_t m, int32_t n, int32_t k,
        const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i,
        const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i, const float *Xtest_csr,
        int32_t k_metrics, int cumulative, int break_ties_with_noise,
        float *p_at_k, float *tp_at_k, float *r_at_k, float *ap_at_k, float *tap_at_k,
        float *ndcg_at_k, float *hit_at_k, float *rr_at_k, float *roc_auc, float *pr_auc,
        int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test, int32_t nthreads, uint64_t seed) nogil
    int rm_calc_metrics_f64(
        const double *A, size_t lda, const double *B, size_t ldb, int32_t m, int32_t n, int32_t k,
        const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i,
        const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i, const double *Xtest_csr,
        int32_t k_metrics, int cumulative, int break_ties_with_noise,
        double *p_at_k, double *tp_at_k, double *r_at_k, double *ap_at_k, double *tap_at_k,
        double *ndcg_at_k, double *hit_at_k, double *rr_at_k, double *roc_auc, double *pr_auc,
        int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test, int32_t nthreads, uint64_t seed) nogil
    int rm_has_openmp() nogil
    int rm_device_count() nogil
    const char *rm_last_error() nogil

METRIC_ORDER = ("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr", "roc", "pr")


def has_openmp():
    return bool(rm_has_openmp())


def device_count():
    return int(rm_device_count())


cdef _raise(int status):
    msg = (<bytes>rm_last_error()).decode(errors="replace")
    if status == 1:
        raise ValueError(msg)
    if status == 3:
        raise MemoryError(msg)
    if status == 5:
        raise RuntimeError(msg.strip())
    raise RuntimeError("recometrics_amd (status %d): %s" % (status, msg))


cdef inline const int32_t *_iptr(const int32_t[::1] a) noexcept nogil:
    return &a[0] if a.shape[0] > 0 else NULL


def calc_metrics(A, size_t lda, B, size_t ldb, const int32_t[::1] train_p, const int32_t[::1] train_i,
                 const int32_t[::1] test_p, const int32_t[::1] test_i, test_v, int32_t k_metrics, want,
                 bint cumulative, bint break_ties_with_noise, bint consider_cold_start,
                 int32_t min_items_pool, int32_t min_pos_test, int32_t nthreads, uint64_t seed):
    """Same contract as recometrics_amd._binding.calc_metrics: returns the 10-tuple in METRIC_ORDER, size-0 arrays for the
    metrics not requested, (m, k_metrics) arrays for the cumulative top-K metrics."""
    cdef int32_t m = A.shape[0], k = A.shape[1], n = B.shape[0]
    cdef size_t size_arr = <size_t>m * <size_t>k_metrics if cumulative else <size_t>m
    dtype = A.dtype
    outs = []
    for name in METRIC_ORDER:
        cnt = (m if name in ("roc", "pr") else size_arr) if want.get(name) else 0
        outs.append(np.empty(cnt, dtype=dtype))
    cdef int status
    cdef size_t pa, pb, pv
    cdef size_t po[10]
    for i in range(10):
        po[i] = <size_t>outs[i].ctypes.data if outs[i].shape[0] else 0
    pa = <size_t>A.ctypes.data
    pb = <size_t>B.ctypes.data
    pv = <size_t>test_v.ctypes.data if test_v is not None and test_v.shape[0] else 0
    cdef const int32_t *trp = _iptr(train_p)
    cdef const int32_t *tri = _iptr(train_i)
    cdef const int32_t *tep = _iptr(test_p)
    cdef const int32_t *tei = _iptr(test_i)
    if dtype == np.float32:
        with nogil:
            status = rm_calc_metrics_f32(<const float *>pa, lda, <const float *>pb, ldb, m, n, k, trp, tri, tep, tei, <const float *>pv,
                                         k_metrics, cumulative, break_ties_with_noise,
                                         <float *>po[0], <float *>po[1], <float *>po[2], <float *>po[3], <float *>po[4],
                                         <float *>po[5], <float *>po[6], <float *>po[7], <float *>po[8], <float *>po[9],
                                         consider_cold_start, min_items_pool, min_pos_test, nthreads, seed)
    elif dtype == np.float64:
        with nogil:
            status = rm_calc_metrics_f64(<const double *>pa, lda, <const double *>pb, ldb, m, n, k, trp, tri, tep, tei, <const double *>pv,
                                         k_metrics, cumulative, break_ties_with_noise,
                                         <double *>po[0], <double *>po[1], <double *>po[2], <double *>po[3], <double *>po[4],
                                         <double *>po[5], <double *>po[6], <double *>po[7], <double *>po[8], <double *>po[9],
                                         consider_cold_start, min_items_pool, min_pos_test, nthreads, seed)
    else:
        raise TypeError("factors must be float32 or float64")
    if status != 0:
        _raise(status)
    if cumulative:
        outs = [(o.reshape((m, k_metrics)) if o.size else o.reshape((0, 0))) if i < 8 else o for i, o in enumerate(outs)]
    return tuple(outs)
