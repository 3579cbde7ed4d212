/* ---------------------------------------------------------------------------------------------
 * recometrics_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.
 *
 * CPU restatement of the per-user ranking-metric algorithm of david-cortes/recometrics
 * (`calc_metrics<real_t>`, reference src/recometrics.hpp:359-965, with `dot1` :84-112).
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg have
 * something to CHECK the HIP path against.  Nothing under recometrics_amd/ may import, link
 * or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file bit-for-bit against
 *   (a) the known-answer cases of the reference's own tests (tests/testthat/test-ndcg.R,
 *       test-auc.R) and
 *   (b) outputs of the real reference compiled here from /root/reference by oracle/Makefile
 *       (oracle/_ref/librecometrics_ref.so), captured as fixtures in tests/golden/*.npz by
 *       tests/golden/make_golden.py.
 *
 * Written from the behavioural specification, structured differently
 * from the reference: candidates are a predicate + compacted list (no swap-out + sort), the
 * ranking is a total order (score descending, item id ascending) so that results are defined
 * on ties too, and each metric family is its own function.
 *
 * Deliberate deviations from the reference (all in territory where the reference's own result
 * is unspecified or uninitialised; documented in DESIGN.md):
 *   D1  Hit@K / RR@K requested WITHOUT any of P/TP/R/AP/TAP/NDCG are computed (the reference
 *       leaves those output slots uninitialised, recometrics.hpp:420,605).
 *   D2  PR-AUC requested WITHOUT ROC-AUC is computed from the full ranking (the reference walks a
 *       partially sorted list, recometrics.hpp:537,851-860).
 *   D3  break_ties_with_noise=false and a NaN score among the candidates: all outputs NaN (the
 *       reference feeds NaN to std::sort's comparator, which is undefined behaviour).
 *   D4  exact score ties are ordered by ascending item id (the reference inherits libstdc++'s
 *       unspecified introsort/heap order).
 * Everything else -- including the quirks Q1 (min_pos_test clamp uses min), Q4 (cumulative
 * k_leq_n columns kept when the walk ran), Q5 (cumulative NDCG divides the already rounded DCG)
 * and Q6 (cumulative NDCG tail overwritten when npos < K) -- follows the reference.
 * ------------------------------------------------------------------------------------------- */
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <numeric>
#include <random>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

template <class T> struct Outputs {
    T *p, *tp, *r, *ap, *tap, *ndcg, *hit, *rr, *roc, *pr;
};

template <class T> struct Problem {
    const T *A; size_t lda; const T *B; size_t ldb;
    int32_t m, n, k;
    const int32_t *train_p, *train_i, *test_p, *test_i; const T *test_v;
    int32_t K; bool cumulative, noise;
    Outputs<T> out;
    bool cold; int32_t min_items_pool, min_pos_test; uint64_t seed;
    /* optional ranking dump (rmo_rank_*): */
    int32_t *topk_idx; T *topk_score; int64_t *pos_rank; int32_t *status;
};

/* reference recometrics.hpp:99-112 (canonical build: strict index-order fma chain from +0) */
inline float  chain_dot(const float  *x, const float  *y, int32_t k) { float  s = 0; for (int32_t t = 0; t < k; t++) s = __builtin_fmaf(x[t], y[t], s); return s; }
inline double chain_dot(const double *x, const double *y, int32_t k) { double s = 0; for (int32_t t = 0; t < k; t++) s = __builtin_fma (x[t], y[t], s); return s; }

template <class T> struct Scratch {
    std::vector<T> score;            /* [n], indexed by item id */
    std::vector<int32_t> rank;       /* candidate ids, ranked */
    std::vector<uint8_t> in_train;   /* [n] */
    std::vector<int32_t> vord;       /* test-value ordering for ideal DCG */
};

template <class T> constexpr T nan_v() { return std::numeric_limits<T>::quiet_NaN(); }

/* reference recometrics.hpp:450-476 */
template <class T>
void fill_user_nan(const Problem<T> &q, int32_t u)
{
    const Outputs<T> &o = q.out;
    T *top8[8] = {o.p, o.tp, o.r, o.ap, o.tap, o.ndcg, o.hit, o.rr};
    for (T *arr : top8) {
        if (!arr) continue;
        if (!q.cumulative) arr[u] = nan_v<T>();
        else std::fill(arr + (size_t)u * q.K, arr + (size_t)(u + 1) * q.K, nan_v<T>());
    }
    if (o.roc) o.roc[u] = nan_v<T>();
    if (o.pr)  o.pr[u]  = nan_v<T>();
}

template <class T>
void eval_user(const Problem<T> &q, int32_t u, Scratch<T> &s)
{
    const Outputs<T> &o = q.out;
    const int32_t n = q.n, K = q.K;
    const int32_t tr0 = q.train_p[u], tr1 = q.train_p[u + 1];
    const int32_t te0 = q.test_p[u],  te1 = q.test_p[u + 1];
    const int32_t ntr = tr1 - tr0, npos = te1 - te0;
    const int32_t ncand = n - ntr;
    if (q.status) q.status[u] = 1;
    if (q.topk_idx)
        for (int32_t i = 0; i < K; i++) { q.topk_idx[(size_t)u * K + i] = -1; q.topk_score[(size_t)u * K + i] = nan_v<T>(); }
    if (q.pos_rank) for (int32_t t = te0; t < te1; t++) q.pos_rank[t] = 0;

    /* ---- eligibility (recometrics.hpp:439-448) ---- */
    if (npos <= 0 || (ntr + npos >= n && !o.ndcg) || ncand < q.min_items_pool ||
        (!q.cold && ntr == 0) || npos < q.min_pos_test) {
        fill_user_nan(q, u); return;
    }
    const bool only_ndcg = (ntr + npos) >= n;           /* :479-482 */
    const bool k_leq_n  = ncand <= K;                   /* :483 */
    if (k_leq_n && !o.roc && !o.pr && !o.ap && !o.tap && !o.rr) { fill_user_nan(q, u); return; }   /* :485 */

    /* ---- candidates = items outside the train row, ascending (:491-497) ---- */
    for (int32_t t = tr0; t < tr1; t++) s.in_train[q.train_i[t]] = 1;
    s.rank.clear();
    for (int32_t j = 0; j < n; j++) if (!s.in_train[j]) s.rank.push_back(j);
    for (int32_t t = tr0; t < tr1; t++) s.in_train[q.train_i[t]] = 0;
    const int32_t C = (int32_t)s.rank.size();           /* == ncand for valid CSR */
    if (C <= 0) { fill_user_nan(q, u); return; }

    /* ---- scores (:499-512) ---- */
    const T *Au = q.A + (size_t)u * q.lda;
    T *sc = s.score.data();
    bool any_nan = false;
    T smax = std::numeric_limits<T>::lowest(), smin = std::numeric_limits<T>::max();
    for (int32_t c = 0; c < C; c++) {
        const int32_t j = s.rank[c];
        const T v = chain_dot(Au, q.B + (size_t)j * q.ldb, q.k);
        sc[j] = v;
        any_nan |= std::isnan(v);
        smax = (smax < v) ? v : smax;
        smin = (smin > v) ? v : smin;
    }
    if (any_nan) { fill_user_nan(q, u); return; }       /* :517-518 with noise; deviation D3 without */

    /* ---- tie noise (:514-535) ---- */
    if (q.noise) {
        if (smax == smin || std::isinf(smax) || std::isinf(smin)) { fill_user_nan(q, u); return; }
        std::mt19937 rng(q.seed + (uint64_t)u);
        std::uniform_real_distribution<T> runif((T)(-1e-12), (T)1e-12);
        for (int32_t c = 0; c < C; c++) sc[s.rank[c]] += runif(rng);
    }

    /* ---- ranking (:537-563); total order = deviation D4 ---- */
    auto before = [sc](int32_t a, int32_t b) { return sc[a] > sc[b] || (sc[a] == sc[b] && a < b); };
    const bool ref_full = (o.roc && !only_ndcg) || K >= C;
    const bool need_full = ref_full || (o.pr && !only_ndcg);
    if (need_full) std::sort(s.rank.begin(), s.rank.end(), before);
    else           std::partial_sort(s.rank.begin(), s.rank.begin() + K, s.rank.end(), before);
    const int32_t *R = s.rank.data();
    if (!q.noise) {
        const T hi = sc[R[0]];
        const T lo = ref_full ? sc[R[C - 1]] : sc[R[K - 1]];
        if (std::isinf(hi) || std::isinf(lo) || hi == lo) { fill_user_nan(q, u); return; }
    }
    if (q.status) q.status[u] = 0;

    const int32_t *ti = q.test_i + te0;
    const T *tv = q.test_v ? q.test_v + te0 : nullptr;
    const int32_t W = std::min(K, C);

    /* ranking dump for the index-set parity tests */
    if (q.topk_idx) {
        for (int32_t i = 0; i < K; i++) {
            q.topk_idx[(size_t)u * K + i]   = i < W ? R[i] : -1;
            q.topk_score[(size_t)u * K + i] = i < W ? sc[R[i]] : nan_v<T>();
        }
    }
    if (q.pos_rank) {
        for (int32_t t = 0; t < npos; t++) q.pos_rank[te0 + t] = 0;
        if (need_full)
            for (int32_t i = 0; i < C; i++) {
                const int32_t *f = std::lower_bound(ti, ti + npos, R[i]);
                if (f != ti + npos && *f == R[i]) q.pos_rank[te0 + (f - ti)] = (int64_t)i + 1;
            }
    }

    /* ---- top-K walk (:589-748) ---- */
    T *cp = nullptr, *ctp = nullptr, *cr = nullptr, *cap = nullptr, *ctap = nullptr, *cndcg = nullptr, *chit = nullptr, *crr = nullptr;
    if (q.cumulative) {
        const size_t st = (size_t)u * K;
        cp = o.p ? o.p + st : nullptr;       ctp = o.tp ? o.tp + st : nullptr;       cr = o.r ? o.r + st : nullptr;
        cap = o.ap ? o.ap + st : nullptr;    ctap = o.tap ? o.tap + st : nullptr;    cndcg = o.ndcg ? o.ndcg + st : nullptr;
        chit = o.hit ? o.hit + st : nullptr; crr = o.rr ? o.rr + st : nullptr;
    }
    const bool top = o.p || o.tp || o.r || o.ap || o.tap || o.ndcg || o.hit || o.rr;   /* deviation D1: hit, rr included */
    bool walked = false;
    int32_t hits = 0, first = std::numeric_limits<int32_t>::max();
    double avg_p = 0, dcg = 0;
    if (top && (!k_leq_n || o.ap || o.tap || o.rr || o.ndcg)) {
        walked = true;
        for (int32_t ix = 0; ix < W; ix++) {
            const int32_t *f = std::lower_bound(ti, ti + npos, R[ix]);
            if (f != ti + npos && *f == R[ix]) {
                hits++;
                avg_p += hits / (double)(ix + 1);
                dcg += tv ? ((double)tv[f - ti] / std::log2(ix + 2)) : 0.;
                first = std::min(first, ix);
            }
            if (q.cumulative) {
                if (cp)    cp[ix]    = hits / (double)(ix + 1);
                if (ctp)   ctp[ix]   = hits / (double)std::min(ix + 1, npos);
                if (cr)    cr[ix]    = hits / (double)npos;
                if (cap)   cap[ix]   = avg_p / (double)npos;
                if (ctap)  ctap[ix]  = avg_p / (double)std::min(ix + 1, npos);
                if (cndcg) cndcg[ix] = dcg;
                if (chit)  chit[ix]  = hits > 0;
                if (crr)   crr[ix]   = hits ? ((double)1 / (double)(first + 1)) : 0.;
            }
        }
        if (!q.cumulative) {
            if (o.p)   o.p[u]   = (double)hits / (double)K;
            if (o.tp)  o.tp[u]  = (double)hits / (double)std::min(K, npos);
            if (o.r)   o.r[u]   = (double)hits / (double)npos;
            if (o.ap)  o.ap[u]  = avg_p / (double)npos;
            if (o.tap) o.tap[u] = avg_p / (double)std::min(K, npos);
            if (o.hit) o.hit[u] = hits > 0;
            if (o.rr)  o.rr[u]  = hits ? (1. / (double)(first + 1)) : 0.;
        } else if (K > C) {          /* :712-746, unreachable for valid CSR (C >= min_items_pool >= K) */
            T *nanfill[4] = {cp, ctp, cr, chit};
            for (T *a : nanfill) if (a) std::fill(a + C, a + K, nan_v<T>());
            T *carry[4] = {cap, ctap, crr, cndcg};
            for (T *a : carry) if (a) std::fill(a + C, a + K, a[C - 1]);
        }
    }

    /* ---- NaN overrides (:750-788) ---- */
    if (k_leq_n) {
        if (!q.cumulative) {
            T *a4[4] = {o.p, o.tp, o.r, o.hit};
            for (T *a : a4) if (a) a[u] = nan_v<T>();
        } else if (!walked) {                                                    /* quirk Q4 */
            T *a4[4] = {cp, ctp, cr, chit};
            for (T *a : a4) if (a) std::fill_n(a, K, nan_v<T>());
        }
    } else if (only_ndcg) {
        if (!q.cumulative) {
            T *a7[7] = {o.p, o.tp, o.r, o.ap, o.tap, o.hit, o.rr};
            for (T *a : a7) if (a) a[u] = nan_v<T>();
        } else {
            T *a7[7] = {cp, ctp, cr, cap, ctap, chit, crr};
            for (T *a : a7) if (a) std::fill_n(a, K, nan_v<T>());
        }
    }

    /* ---- ROC-AUC / PR-AUC over the full ranking (:795-865); PR alone = deviation D2 ---- */
    if (only_ndcg) {
        if (o.roc) o.roc[u] = nan_v<T>();
        if (o.pr)  o.pr[u]  = nan_v<T>();
    } else if (o.roc || o.pr) {
        uint64_t sum_ranks = 0; int32_t h = 0; double ap_full = 0;
        const uint64_t P = (uint64_t)npos, Nneg = (uint64_t)C - P;
        for (int32_t ix = 0; ix < C; ix++) {
            const int32_t *f = std::lower_bound(ti, ti + npos, R[ix]);
            if (f != ti + npos && *f == R[ix]) {
                sum_ranks += (uint64_t)(ix + 1);
                h++;
                ap_full += (double)h / (double)(ix + 1);
                if (h == npos) break;
            }
        }
        if (o.roc) o.roc[u] = 1. - (long double)(sum_ranks - (P * (P + 1)) / 2) / (long double)(P * Nneg);
        if (o.pr)  o.pr[u]  = ap_full / (double)npos;
    }

    /* ---- NDCG normalisation (:868-961) ---- */
    if (o.ndcg) {
        const int32_t L = std::min(K, npos);
        s.vord.resize(npos);
        std::iota(s.vord.begin(), s.vord.end(), 0);
        std::partial_sort(s.vord.begin(), s.vord.begin() + L, s.vord.end(),
                          [tv](int32_t a, int32_t b) { return tv[a] > tv[b]; });
        const T vmax = tv[s.vord[0]], vlast = tv[s.vord[L - 1]];
        if (std::isnan(vmax) || std::isinf(vmax) || std::isnan(vlast) || std::isinf(vlast) || vmax <= 0) {
            if (!q.cumulative) o.ndcg[u] = nan_v<T>(); else std::fill_n(cndcg, K, nan_v<T>());
            return;
        }
        double idcg = 0, val = 0;
        if (!q.cumulative) {
            if (vlast >= 0) {
                for (int32_t ix = 0; ix < L; ix++) idcg += (double)tv[s.vord[ix]] / std::log2(ix + 2);
            } else {
                for (int32_t ix = 0; ix < L; ix++) { val = tv[s.vord[ix]]; if (val <= 0) break; idcg += val / std::log2(ix + 2); }
            }
            o.ndcg[u] = dcg / idcg;
        } else {
            if (vlast >= 0) {
                for (int32_t ix = 0; ix < L; ix++) { idcg += (double)tv[s.vord[ix]] / std::log2(ix + 2); cndcg[ix] /= idcg; }   /* quirk Q5 */
            } else {
                int32_t ix = 0;
                for (; ix < L; ix++) { val = tv[s.vord[ix]]; if (std::isnan(val) || val < 0) break; idcg += val / std::log2(ix + 2); cndcg[ix] /= idcg; }
                if (std::isnan(val)) std::fill(cndcg + ix, cndcg + L, nan_v<T>());
                else if (val < 0) for (; ix < L; ix++) cndcg[ix] /= idcg;
            }
            if (npos < K) std::fill(cndcg + npos, cndcg + std::min(K, C), cndcg[npos - 1]);      /* quirk Q6 */
        }
    }
}

template <class T>
int run(Problem<T> q, int32_t nthreads)
{
    nthreads = std::max(nthreads, 1);                         /* :390 */
    q.min_items_pool = std::max(std::max(q.min_items_pool, q.K), 2);   /* :391-392 */
    q.min_pos_test = std::min(q.min_pos_test, 1);             /* :393, quirk Q1 */
    try {
        #pragma omp parallel num_threads(nthreads)
        {
            Scratch<T> s;
            s.score.resize((size_t)q.n); s.in_train.assign((size_t)q.n, 0); s.rank.reserve((size_t)q.n);
            #pragma omp for schedule(dynamic)
            for (int32_t u = 0; u < q.m; u++) eval_user(q, u, s);
        }
    } catch (...) { return 1; }
    return 0;
}

template <class T>
Problem<T> make_problem(const T *A, size_t lda, const T *B, size_t ldb, int32_t m, int32_t n, int32_t k,
                        const int32_t *trp, const int32_t *tri, const int32_t *tep, const int32_t *tei, const T *tev,
                        int32_t K, int cumulative, int noise, Outputs<T> out,
                        int cold, int32_t min_items_pool, int32_t min_pos_test, uint64_t seed)
{
    Problem<T> q{};
    q.A = A; q.lda = lda; q.B = B; q.ldb = ldb; q.m = m; q.n = n; q.k = k;
    q.train_p = trp; q.train_i = tri; q.test_p = tep; q.test_i = tei; q.test_v = tev;
    q.K = K; q.cumulative = cumulative != 0; q.noise = noise != 0; q.out = out;
    q.cold = cold != 0; q.min_items_pool = min_items_pool; q.min_pos_test = min_pos_test; q.seed = seed;
    q.topk_idx = nullptr; q.topk_score = nullptr; q.pos_rank = nullptr; q.status = nullptr;
    return q;
}

} // namespace

#define RMO_DEFINE(SUFFIX, T)                                                                                         \
extern "C" int rmo_calc_metrics_##SUFFIX(                                                                             \
    const T *A, size_t lda, const T *B, size_t ldb, int32_t m, int32_t n, int32_t k,                                  \
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i,                                                         \
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i, const T *Xtest_csr,                                       \
    int32_t k_metrics, int cumulative, int break_ties_with_noise,                                                     \
    T *p_at_k, T *tp_at_k, T *r_at_k, T *ap_at_k, T *tap_at_k, T *ndcg_at_k, T *hit_at_k, T *rr_at_k,                 \
    T *roc_auc, T *pr_auc,                                                                                            \
    int consider_cold_start, int32_t min_items_pool, int32_t min_pos_test, int32_t nthreads, uint64_t seed)           \
{                                                                                                                     \
    Outputs<T> out{p_at_k, tp_at_k, r_at_k, ap_at_k, tap_at_k, ndcg_at_k, hit_at_k, rr_at_k, roc_auc, pr_auc};        \
    return run(make_problem<T>(A, lda, B, ldb, m, n, k, Xtrain_csr_p, Xtrain_csr_i, Xtest_csr_p, Xtest_csr_i,         \
                               Xtest_csr, k_metrics, cumulative, break_ties_with_noise, out,                          \
                               consider_cold_start, min_items_pool, min_pos_test, seed), nthreads);                   \
}                                                                                                                     \
/* Ranking dump: top-K item ids/scores per user, 1-based full-ranking position of every test item (0 when the     */ \
/* item is masked by the train row), status 0 = ranked / 1 = user skipped.  AUC-style full ranking is forced.     */ \
extern "C" int rmo_rank_##SUFFIX(                                                                                     \
    const T *A, size_t lda, const T *B, size_t ldb, int32_t m, int32_t n, int32_t k,                                  \
    const int32_t *Xtrain_csr_p, const int32_t *Xtrain_csr_i,                                                         \
    const int32_t *Xtest_csr_p, const int32_t *Xtest_csr_i,                                                           \
    int32_t k_metrics, int break_ties_with_noise, int consider_cold_start, int32_t min_items_pool,                    \
    int32_t min_pos_test, int32_t nthreads, uint64_t seed,                                                            \
    int32_t *topk_idx, T *topk_score, int64_t *pos_rank, int32_t *status)                                             \
{                                                                                                                     \
    std::vector<T> roc((size_t)m), pr((size_t)m), apk((size_t)m);                                                     \
    Outputs<T> out{nullptr, nullptr, nullptr, apk.data(), nullptr, nullptr, nullptr, nullptr, roc.data(), pr.data()}; \
    Problem<T> q = make_problem<T>(A, lda, B, ldb, m, n, k, Xtrain_csr_p, Xtrain_csr_i, Xtest_csr_p, Xtest_csr_i,     \
                                   (const T *)nullptr, k_metrics, 0, break_ties_with_noise, out,                      \
                                   consider_cold_start, min_items_pool, min_pos_test, seed);                          \
    q.topk_idx = topk_idx; q.topk_score = topk_score; q.pos_rank = pos_rank; q.status = status;                       \
    return run(q, nthreads);                                                                                          \
}

RMO_DEFINE(f32, float)
RMO_DEFINE(f64, double)

/* dense score matrix out[m x n] with the canonical chain (test hook for the MFMA bit-exactness check) */
extern "C" int rmo_scores_f32(const float *A, size_t lda, const float *B, size_t ldb, int32_t m, int32_t n, int32_t k, float *out)
{
    #pragma omp parallel for schedule(static)
    for (int32_t u = 0; u < m; u++)
        for (int32_t j = 0; j < n; j++) out[(size_t)u * n + j] = chain_dot(A + (size_t)u * lda, B + (size_t)j * ldb, k);
    return 0;
}
extern "C" int rmo_scores_f64(const double *A, size_t lda, const double *B, size_t ldb, int32_t m, int32_t n, int32_t k, double *out)
{
    #pragma omp parallel for schedule(static)
    for (int32_t u = 0; u < m; u++)
        for (int32_t j = 0; j < n; j++) out[(size_t)u * n + j] = chain_dot(A + (size_t)u * lda, B + (size_t)j * ldb, k);
    return 0;
}

extern "C" int rmo_has_openmp(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}
