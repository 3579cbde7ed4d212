// Rwrapper_hip.cpp -- Rcpp glue of the R package over the MI355X library: replaces the two exported functions of the
// reference's src/Rwrapper.cpp that reach calc_metrics<real_t> (calc_metrics_double :292-350, calc_metrics_float :352-410;
// template :61-290, the call itself :250-274) and R_has_openmp (:614-622).  Same names, same argument lists, same return
// value (a list of vectors, or [k_metrics x m] matrices when `cumulative`, plus "k"), so R/RcppExports.R, src/RcppExports.cpp
// and R/recometrics.R:576-731 stay as they are.  The splitting functions of Rwrapper.cpp are untouched (host code).
//
// NOT EXECUTED in this repository: the build image has no R toolchain.  Everything in here that is not an Rcpp type --
// column-major factors with lda = k, float bits in integer storage, the double conversion of the test values, NaN -> NA_real_
// -- lives in rm_r_shim.c and IS tested (tests/test_r_shim.py), through the same entry points this file calls.
#include <Rcpp.h>
#include <Rcpp/unwindProtect.h>
// [[Rcpp::plugins(cpp11)]]
#include <cstdint>

extern "C" {
#include "recometrics_hip.h"
#include "rm_r_shim.h"
}

namespace {

// one output: a vector [m], or a matrix [k_metrics x m] (column-major == the C-ABI's row-major [m x k_metrics])
template <class Vec, class Mat> struct Out {
    Vec vec; Mat mat; bool on = false, cum = false;
    void alloc(bool wanted, bool cumulative, int m, int k_metrics)
    {
        on = wanted; cum = cumulative;
        if (!on) return;
        if (cum) mat = Mat(k_metrics, m); else vec = Vec(m);
    }
    SEXP sexp() const { return cum ? (SEXP)mat : (SEXP)vec; }
};

inline double *data_of(Rcpp::NumericVector &v) { return REAL(v); }
inline double *data_of(Rcpp::NumericMatrix &v) { return REAL(v); }
inline int *data_of(Rcpp::IntegerVector &v) { return INTEGER(v); }
inline int *data_of(Rcpp::IntegerMatrix &v) { return INTEGER(v); }

template <class Elem, class Vec, class Mat, class Call>
Rcpp::List run(Mat A, Mat B, bool want[10], int k_metrics, bool cumulative, Call call)
{
    const int m = A.ncol(), k = A.nrow(), n = B.ncol();             // R hands over t(A): [k x m]
    if (B.nrow() != k) Rcpp::stop("'A' and 'B' must have the same number of factors.");
    Out<Vec, Mat> o[10];
    Elem *ptr[10];
    for (int i = 0; i < 10; i++) {
        o[i].alloc(want[i], cumulative && i < 8, m, k_metrics);
        ptr[i] = !want[i] ? nullptr : ((cumulative && i < 8) ? data_of(o[i].mat) : data_of(o[i].vec));
    }
    const int status = call(data_of(A), data_of(B), m, n, k, ptr);
    if (status != RM_OK) Rcpp::stop(rm_last_error());               // std::bad_alloc / std::runtime_error in the reference
    static const char *names[10] = {"p_at_k", "tp_at_k", "r_at_k", "ap_at_k", "tap_at_k", "ndcg_at_k", "hit_at_k", "rr_at_k", "roc_auc", "pr_auc"};
    Rcpp::List out;
    for (int i = 0; i < 10; i++) if (want[i]) out[names[i]] = o[i].sexp();
    out["k"] = k_metrics;
    return out;
}

} // namespace

// [[Rcpp::export(rng = false)]]
Rcpp::List calc_metrics_double(
    Rcpp::NumericMatrix A, Rcpp::NumericMatrix B,
    Rcpp::IntegerVector Xtrain_csr_p, Rcpp::IntegerVector Xtrain_csr_i,
    Rcpp::IntegerVector Xtest_csr_p, Rcpp::IntegerVector Xtest_csr_i, Rcpp::NumericVector Xtest_csr,
    bool calc_p_at_k = true, bool calc_tp_at_k = false, bool calc_r_at_k = false, bool calc_ap_at_k = true,
    bool calc_tap_at_k = false, bool calc_ndcg_at_k = true, bool calc_hit_at_k = false, bool calc_rr_at_k = false,
    bool calc_roc_auc = false, bool calc_pr_auc = false,
    int k_metrics = 10, bool break_ties_with_noise = true, int min_pos_test = 1, int min_items_pool = 2,
    bool consider_cold_start = 0, bool cumulative = 0, int nthreads = 1, uint64_t seed = 1)
{
    bool want[10] = {calc_p_at_k, calc_tp_at_k, calc_r_at_k, calc_ap_at_k, calc_tap_at_k, calc_ndcg_at_k, calc_hit_at_k, calc_rr_at_k,
                     calc_roc_auc, calc_pr_auc};
    return run<double, Rcpp::NumericVector, Rcpp::NumericMatrix>(A, B, want, k_metrics, cumulative,
        [&](double *a, double *b, int m, int n, int k, double **o) {
            return rm_r_calc_metrics_f64(a, b, m, n, k, INTEGER(Xtrain_csr_p), INTEGER(Xtrain_csr_i), INTEGER(Xtest_csr_p),
                                         INTEGER(Xtest_csr_i), REAL(Xtest_csr), k_metrics, cumulative, break_ties_with_noise, o,
                                         consider_cold_start, min_items_pool, min_pos_test, nthreads, seed);
        });
}

// [[Rcpp::export(rng = false)]]
Rcpp::List calc_metrics_float(
    Rcpp::IntegerMatrix A, Rcpp::IntegerMatrix B,
    Rcpp::IntegerVector Xtrain_csr_p, Rcpp::IntegerVector Xtrain_csr_i,
    Rcpp::IntegerVector Xtest_csr_p, Rcpp::IntegerVector Xtest_csr_i, Rcpp::NumericVector Xtest_csr,
    bool calc_p_at_k = true, bool calc_tp_at_k = false, bool calc_r_at_k = false, bool calc_ap_at_k = true,
    bool calc_tap_at_k = false, bool calc_ndcg_at_k = true, bool calc_hit_at_k = false, bool calc_rr_at_k = false,
    bool calc_roc_auc = false, bool calc_pr_auc = false,
    int k_metrics = 10, bool break_ties_with_noise = true, int min_pos_test = 1, int min_items_pool = 2,
    bool consider_cold_start = 0, bool cumulative = 0, int nthreads = 1, uint64_t seed = 1)
{
    bool want[10] = {calc_p_at_k, calc_tp_at_k, calc_r_at_k, calc_ap_at_k, calc_tap_at_k, calc_ndcg_at_k, calc_hit_at_k, calc_rr_at_k,
                     calc_roc_auc, calc_pr_auc};
    return run<int, Rcpp::IntegerVector, Rcpp::IntegerMatrix>(A, B, want, k_metrics, cumulative,
        [&](int *a, int *b, int m, int n, int k, int **o) {
            return rm_r_calc_metrics_f32(a, b, m, n, k, INTEGER(Xtrain_csr_p), INTEGER(Xtrain_csr_i), INTEGER(Xtest_csr_p),
                                         INTEGER(Xtest_csr_i), REAL(Xtest_csr), k_metrics, cumulative, break_ties_with_noise, o,
                                         consider_cold_start, min_items_pool, min_pos_test, nthreads, seed);
        });
}

// [[Rcpp::export(rng = false)]]
void C_NAN_to_R_NA(SEXP vec)                    // Rwrapper.cpp:605-612, kept for the R code that calls it
{
    rm_r_nan_to_na(REAL(vec), (size_t)Rf_xlength(vec));
}

// [[Rcpp::export(rng = false)]]
bool R_has_openmp()                             // Rwrapper.cpp:614-622: host threads do not matter here
{
    return rm_has_openmp() != 0;
}
