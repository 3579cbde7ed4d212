// rm_split.cpp -- host-side train/test splitting of implicit-feedback CSR data ("next" row N3 of SURVEY.md section 8f:
// the step BEFORE the metric hot path).  CPU only -- the work is a sequential, RNG-order-dependent pass over the rows
// and runs once per experiment, so there is nothing for the GPU here.
//
// Restates the behaviour of reference src/recometrics.hpp:1015-1117 (split_data_selected_users), :1201-1323
// (split_data_separate_users), :1325-1360 (concat_csr_matrices) and :1442-1505 (split_data_joined_users) behind the
// C-ABI of include/recometrics_hip.h.  Bit-compatibility with the reference's outputs depends on consuming
// std::mt19937 through std::shuffle exactly as it does (same seed, same call order); both come from the same
// libstdc++, which is what defines the reference's own results.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/recometrics_hip.h"

namespace {

thread_local std::string g_split_err;

template <class T> struct Csr {
    std::vector<int32_t> p, i;
    std::vector<T> v;
    void append_row(const int32_t *ci, const T *cv, int32_t len)
    {
        i.insert(i.end(), ci, ci + len);
        v.insert(v.end(), cv, cv + len);
        p.push_back((int32_t)i.size());
    }
};

template <class T> struct View { const int32_t *p, *i; const T *v; int32_t m, n; };

struct SplitBase { bool is_f32; };                  // what the untyped handle is first read as
template <class T> struct SplitResult : SplitBase {
    Csr<T> train, test, rem;
    std::vector<int32_t> users_test;
};

inline int32_t test_count(int32_t row_len, double fraction)       // reference :1043 (round half away from zero)
{
    return (int32_t)std::round(row_len * fraction);
}

// Every row is cut into a test part of round(len * fraction) randomly chosen entries and a train part with the rest;
// both keep ascending column order.  One generator for the whole matrix, advanced only by the rows that are really cut.
template <class T>
void split_every_row(const View<T> &X, double fraction, uint64_t seed, Csr<T> &train, Csr<T> &test)
{
    if (X.m == 0) return;
    if (X.m < 0 || X.n < 0) throw std::runtime_error("Passed negative dimensions.\n");
    train.p.assign(1, 0); test.p.assign(1, 0);
    std::mt19937 rng(seed);
    std::vector<int32_t> pick;
    for (int32_t u = 0; u < X.m; u++) {
        const int32_t len = X.p[u + 1] - X.p[u];
        const int32_t take = test_count(len, fraction);
        const int32_t *ci = X.i + X.p[u];
        const T *cv = X.v + X.p[u];
        if (len == 0 || take == 0) { train.append_row(ci, cv, len); test.append_row(ci, cv, 0); continue; }
        if (take == len)           { test.append_row(ci, cv, len); train.append_row(ci, cv, 0); continue; }
        pick.resize(len);
        std::iota(pick.begin(), pick.end(), (int32_t)0);
        std::shuffle(pick.begin(), pick.end(), rng);                   // first `take` positions go to test
        auto by_column = [ci](int32_t a, int32_t b) { return ci[a] < ci[b]; };
        std::sort(pick.begin(), pick.begin() + take, by_column);
        std::sort(pick.begin() + take, pick.end(), by_column);
        for (int32_t t = 0; t < take; t++) { test.i.push_back(ci[pick[t]]); test.v.push_back(cv[pick[t]]); }
        for (int32_t t = take; t < len; t++) { train.i.push_back(ci[pick[t]]); train.v.push_back(cv[pick[t]]); }
        test.p.push_back((int32_t)test.i.size());
        train.p.push_back((int32_t)train.i.size());
    }
}

// Random selection of test users (reference :1227-1272): users are visited in a shuffled order; an ineligible one is
// swapped out to the shrinking tail and the slot is retried with whatever was swapped in.
template <class T>
int32_t choose_test_users(const View<T> &X, int32_t n_users_test, double fraction, bool cold, int32_t min_items_pool,
                          int32_t min_pos_test, uint64_t seed, std::vector<int32_t> &order)
{
    if (n_users_test > X.m) throw std::runtime_error("Target number of test users is larger than available users.\n");
    if (min_items_pool >= X.n) throw std::runtime_error("Selected minimum number of items is larger than total number of items.\n");
    order.resize(X.m);
    std::iota(order.begin(), order.end(), (int32_t)0);
    std::mt19937 rng(seed);
    std::shuffle(order.begin(), order.end(), rng);
    int32_t taken = 0, end = X.m;
    do {
        const int32_t u = order[taken];
        const int32_t len = X.p[u + 1] - X.p[u];
        const int32_t nte = test_count(len, fraction);
        const bool eligible = len != 0 && nte >= min_pos_test && X.n - (len - nte) >= min_items_pool &&
                              (cold || nte != len) && len + 1 < X.n;
        if (eligible) taken++;
        else std::swap(order[taken], order[--end]);
    } while (taken < n_users_test && taken < end);
    if (taken == 0) throw std::runtime_error("No users satisfy criteria for test inclusion.\n");
    std::sort(order.begin(), order.begin() + taken);
    std::sort(order.begin() + taken, order.end());
    return taken;
}

template <class T>
void gather_rows(const View<T> &X, const int32_t *rows, int32_t count, Csr<T> &out)
{
    out.p.assign(1, 0);
    for (int32_t r = 0; r < count; r++) {
        const int32_t u = rows[r];
        out.append_row(X.i + X.p[u], X.v + X.p[u], X.p[u + 1] - X.p[u]);
    }
}

template <class T>
void split_separated(const View<T> &X, int32_t n_users_test, double fraction, bool cold, int32_t min_items_pool,
                     int32_t min_pos_test, uint64_t seed, SplitResult<T> &res)
{
    std::vector<int32_t> order;
    const int32_t taken = choose_test_users(X, n_users_test, fraction, cold, min_items_pool, min_pos_test, seed, order);
    res.users_test.assign(order.begin(), order.begin() + taken);
    Csr<T> chosen;
    gather_rows(X, order.data(), taken, chosen);
    const View<T> cv{chosen.p.data(), chosen.i.data(), chosen.v.data(), taken, X.n};
    split_every_row(cv, fraction, seed, res.train, res.test);           // same seed, fresh generator (reference :1297-1310)
    gather_rows(X, order.data() + taken, X.m - taken, res.rem);
}

template <class T>
void stack_below(Csr<T> &top, const Csr<T> &bottom)                     // reference :1325-1360
{
    const int32_t offset = top.p.back();
    for (size_t r = 1; r < bottom.p.size(); r++) top.p.push_back(offset + bottom.p[r]);
    top.i.insert(top.i.end(), bottom.i.begin(), bottom.i.end());
    top.v.insert(top.v.end(), bottom.v.begin(), bottom.v.end());
}

template <class T>
int run_split(const int32_t *p, const int32_t *i, const T *v, int32_t m, int32_t n, int mode, int32_t n_users_test,
              double fraction, int cold, int32_t min_items_pool, int32_t min_pos_test, uint64_t seed, void **result)
{
    g_split_err.clear();
    if (!result) { g_split_err = "null result pointer"; return RM_ERR_INVALID; }
    *result = nullptr;
    if (!p || (!i && p[m] > 0) || (!v && p[m] > 0)) { g_split_err = "null CSR pointer"; return RM_ERR_INVALID; }
    SplitResult<T> *res = nullptr;
    try {
        res = new SplitResult<T>();
        res->is_f32 = sizeof(T) == 4;
        const View<T> X{p, i, v, m, n};
        if (mode == 0) split_every_row(X, fraction, seed, res->train, res->test);
        else if (mode == 1 || mode == 2) {
            split_separated(X, n_users_test, fraction, cold != 0, min_items_pool, min_pos_test, seed, *res);
            if (mode == 2) { stack_below(res->train, res->rem); res->rem = Csr<T>(); }
        } else throw std::runtime_error("unknown split mode");
    } catch (const std::bad_alloc &) {
        delete res; g_split_err = "host allocation failed"; return RM_ERR_NOMEM;
    } catch (const std::exception &e) {
        delete res; g_split_err = e.what(); return RM_ERR_INVALID;
    }
    *result = res;
    return RM_OK;
}

template <class T> const void *array_of(const SplitResult<T> &r, int which, int64_t &count, size_t &esize)
{
    const Csr<T> *c = which < 3 ? &r.train : which < 6 ? &r.test : &r.rem;
    if (which == 9) { count = (int64_t)r.users_test.size(); esize = 4; return r.users_test.data(); }
    switch (which % 3) {
        case 0: count = (int64_t)c->p.size(); esize = 4; return c->p.data();
        case 1: count = (int64_t)c->i.size(); esize = 4; return c->i.data();
        default: count = (int64_t)c->v.size(); esize = sizeof(T); return c->v.data();
    }
}

} // namespace

extern "C" int rm_split_f32(const int32_t *X_csr_p, const int32_t *X_csr_i, const float *X_csr, int32_t m, int32_t n, int mode,
                            int32_t n_users_test, double test_fraction, int consider_cold_start, int32_t min_items_pool,
                            int32_t min_pos_test, uint64_t seed, void **result)
{
    return run_split<float>(X_csr_p, X_csr_i, X_csr, m, n, mode, n_users_test, test_fraction, consider_cold_start,
                            min_items_pool, min_pos_test, seed, result);
}

extern "C" int rm_split_f64(const int32_t *X_csr_p, const int32_t *X_csr_i, const double *X_csr, int32_t m, int32_t n, int mode,
                            int32_t n_users_test, double test_fraction, int consider_cold_start, int32_t min_items_pool,
                            int32_t min_pos_test, uint64_t seed, void **result)
{
    return run_split<double>(X_csr_p, X_csr_i, X_csr, m, n, mode, n_users_test, test_fraction, consider_cold_start,
                             min_items_pool, min_pos_test, seed, result);
}

extern "C" int64_t rm_split_size(const void *result, int which)
{
    if (!result || which < 0 || which > 9) return -1;
    int64_t count = 0; size_t es = 0;
    if (static_cast<const SplitBase *>(result)->is_f32) array_of(*static_cast<const SplitResult<float> *>(result), which, count, es);
    else array_of(*static_cast<const SplitResult<double> *>(result), which, count, es);
    return count;
}

extern "C" int rm_split_copy(const void *result, int which, void *dst)
{
    if (!result || which < 0 || which > 9 || !dst) return RM_ERR_INVALID;
    int64_t count = 0; size_t es = 0; const void *src;
    if (static_cast<const SplitBase *>(result)->is_f32) src = array_of(*static_cast<const SplitResult<float> *>(result), which, count, es);
    else src = array_of(*static_cast<const SplitResult<double> *>(result), which, count, es);
    if (count > 0) std::memcpy(dst, src, (size_t)count * es);
    return RM_OK;
}

extern "C" void rm_split_free(void *result)
{
    if (!result) return;
    if (static_cast<SplitBase *>(result)->is_f32) delete static_cast<SplitResult<float> *>(result);
    else delete static_cast<SplitResult<double> *>(result);
}

extern "C" const char *rm_split_last_error(void) { return g_split_err.c_str(); }
