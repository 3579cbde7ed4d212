// rm_list.hpp -- streaming top-K list shared by the fp32 and fp64 sweeps.
//
// UNSORTED list of the K best candidates seen so far plus the position of its worst entry ("replace the minimum"):
// an insert is one store and K independent loads, no dependent shifting chain; the list is sorted once at the end
// of the sweep.  The list of user u lives at L[i * GU] (i = 0..K-1), i.e. [K][GU users], so consecutive lanes touch
// consecutive entries (conflict-free in LDS, coalesced in HBM).  Exactly one lane per user (the "owner") touches it.
// Total order everywhere: score descending, then item id ascending (deviation D4).
#pragma once
#include "rm_device.hpp"

namespace rm {

#ifdef RM_STATS
__device__ unsigned long long g_stats[16];      // timing builds only (scratch/build_abl.py)
#define RM_SEL_STAT(IDX, VAL) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_stats[IDX], (unsigned long long)(VAL)); } while (0)
#else
#define RM_SEL_STAT(IDX, VAL) do {} while (0)
#endif

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <class S> struct ListRaw;
template <> struct ListRaw<float> {
    typedef u32x2 raw;                                        // (score bits, item id)
    __device__ static __forceinline__ raw pack(float s, int idx) { raw q; q.x = __float_as_uint(s); q.y = (unsigned)idx; return q; }
    __device__ static __forceinline__ void unpack(raw q, float &s, int &idx) { s = __uint_as_float(q.x); idx = (int)q.y; }
    // L1-bypassing load (sc1): entries other lanes of this wave stored must be read back through L2
    __device__ static __forceinline__ raw load_l2(const raw *p)
    {
        const unsigned long long b = __hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        raw q; q.x = (unsigned)b; q.y = (unsigned)(b >> 32); return q;
    }
};
template <> struct ListRaw<double> {
    typedef u32x4 raw;                                        // (score lo, score hi, item id, pad)
    __device__ static __forceinline__ raw pack(double s, int idx)
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(s);
        raw q; q.x = (unsigned)b; q.y = (unsigned)(b >> 32); q.z = (unsigned)idx; q.w = 0; return q;
    }
    __device__ static __forceinline__ void unpack(raw q, double &s, int &idx)
    {
        s = __longlong_as_double((long long)(((unsigned long long)q.y << 32) | q.x)); idx = (int)q.z;
    }
    __device__ static __forceinline__ raw load_l2(const raw *p)
    {
        const unsigned long long *pp = (const unsigned long long *)p;
        const unsigned long long b0 = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b1 = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        raw q; q.x = (unsigned)b0; q.y = (unsigned)(b0 >> 32); q.z = (unsigned)b1; q.w = (unsigned)(b1 >> 32); return q;
    }
};

template <class S> __device__ __forceinline__ bool entry_before(S s, int idx, S s2, int idx2)
{
    return s > s2 || (s == s2 && idx < idx2);
}

// P = pointer to ListRaw<S>::raw in LDS (address_space(3)) or in HBM (generic)
template <class S, int GU, class P>
__device__ __forceinline__ void list_find_worst(P L, int K, S &ws, int &widx, int &wpos)
{
    ListRaw<S>::unpack(L[0], ws, widx);
    wpos = 0;
    for (int i = 1; i < K; i++) {
        S s; int idx;
        ListRaw<S>::unpack(L[i * GU], s, idx);
        const bool worse = s < ws || (s == ws && idx > widx);
        ws = worse ? s : ws; widx = worse ? idx : widx; wpos = worse ? i : wpos;
    }
}

// unsorted list, replace-the-minimum: 1 store + K loads per insert (used for lists in HBM, where a shifting chain of
// dependent round trips would be slow)
template <class S, int GU, class P>
__device__ __forceinline__ void list_offer(P L, int K, S s, int item, S &ws, int &widx, int &wpos)
{
    if (s > ws || (s == ws && item < widx)) {
        L[wpos * GU] = ListRaw<S>::pack(s, item);
        list_find_worst<S, GU>(L, K, ws, widx, wpos);
    }
}

// descending sorted list, insertion from the bottom: a new entrant lands on average K/2 places up, so this is ~K/2
// (load, compare, store) steps -- fewer instructions than the K-entry rescan above, which is what matters in LDS where
// the issue slots are shared with the f32 MFMA.  (ws, widx) = the K-th entry.
template <class S, int GU, class P>
__device__ __forceinline__ void list_offer_sorted(P L, int K, S s, int item, S &ws, int &widx)
{
    if (s > ws || (s == ws && item < widx)) {
        int i = K - 1;
        S ps = ws; int pidx = widx;                    // entry that will end up K-th if the new one lands higher
        while (i > 0) {
            S qs; int qidx;
            ListRaw<S>::unpack(L[(i - 1) * GU], qs, qidx);
            if (qs > s || (qs == s && qidx < item)) break;
            L[i * GU] = ListRaw<S>::pack(qs, qidx);
            if (i == K - 1) { ps = qs; pidx = qidx; }
            i--;
        }
        L[i * GU] = ListRaw<S>::pack(s, item);
        if (i == K - 1) { ps = s; pidx = item; }
        ws = ps; widx = pidx;
    }
}

template <class S, int GU, class P>
__device__ __forceinline__ void list_sort_desc(P L, int K)
{
    for (int i = 1; i < K; i++) {
        S es; int eidx;
        ListRaw<S>::unpack(L[i * GU], es, eidx);
        int j = i;
        while (j > 0) {
            S qs; int qidx;
            ListRaw<S>::unpack(L[(j - 1) * GU], qs, qidx);
            if (entry_before<S>(qs, qidx, es, eidx)) break;
            L[j * GU] = ListRaw<S>::pack(qs, qidx);
            j--;
        }
        L[j * GU] = ListRaw<S>::pack(es, eidx);
    }
}

// ---- fp32 LDS lists as packed 64-bit keys: (order-preserving score key << 32) | ~item.  "Better" in the total order
// (score desc, item asc) is simply "larger key", so the worst-entry scan is one 64-bit compare + selects per entry. ----
__device__ __forceinline__ unsigned long long pack_key(float s, int item)
{
    return ((unsigned long long)ord_key(s) << 32) | (unsigned)(~item);
}
__device__ __forceinline__ void unpack_key(unsigned long long k, float &s, int &item)
{
    if ((k >> 32) == 0) { s = __int_as_float(0xff800000); item = IDX_EMPTY; }      // never written
    else { s = ord_unkey((unsigned)(k >> 32)); item = ~(int)(unsigned)k; }
}
template <int GU, class P>
__device__ __forceinline__ void keylist_find_worst(P L, int K, unsigned long long &wkey, int &wpos)
{
    wkey = L[0]; wpos = 0;
    for (int i = 1; i < K; i++) {
        const unsigned long long k = L[i * GU];
        const bool worse = k < wkey;
        wkey = worse ? k : wkey; wpos = worse ? i : wpos;
    }
}
template <int GU, class P>
__device__ __forceinline__ void keylist_offer(P L, int K, unsigned long long ck, unsigned long long &wkey, int &wpos)
{
    if (ck > wkey) {
        L[wpos * GU] = ck;
        keylist_find_worst<GU>(L, K, wkey, wpos);
    }
}
template <int GU, class P>
__device__ __forceinline__ void keylist_sort_desc(P L, int K)
{
    for (int i = 1; i < K; i++) {
        const unsigned long long e = L[i * GU];
        int j = i;
        while (j > 0) {
            const unsigned long long q = L[(j - 1) * GU];
            if (q > e) break;
            L[j * GU] = q;
            j--;
        }
        L[j * GU] = e;
    }
}

// ---- K beyond the LDS lists: per-LANE append buffers in HBM + lane-parallel selection ---------------------------------------
// (round 6; replaces the per-USER buffers whose compaction took the whole wave for one user at a time: 32 users x 7 compactions
// per wave and item range at BASELINE C2's shape with K = 100, each ~1,000 instructions with the three partner waves of the
// sub-tile waiting at the next barrier -- the sweep went 6.7 -> 22.7 ms from K = 10 to K = 100, profiles/r5_ab_c2.txt)
//
// Every lane owns a buffer of `cap` entries, laid out [entry][64 lanes] (fp32: (score, item) pairs; fp64: the scores of a wave, then
// its item ids at the same places), and appends each of ITS OWN scores that reaches the user's bound: a compare, a store or two under
// the lane mask and an add per score register -- no ballots, no shuffles to an owner lane.  A user's candidates are the union of the buffers of its
// lanes (two in the fp32 sweep, four in the fp64 one).  When some lane of the wave is a tile away from full, ALL users of the
// wave raise their bounds together, each lane working on its own buffer (the loops are wave-uniform, the data per lane):
//   * a bisection for a bound T with K <= #(entries >= T) <= K + slack -- three probes per pass over the buffers (one load,
//     three compares and adds per entry), first in value space (quartiles of [lo, hi]: the scores of a user are smooth), then in
//     key space (which ends within 16 / 32 passes whatever the data); the counts of a user's lanes are added by a shuffle;
//   * when the bisection ends on an exact score with more than K + slack entries at or above it (many exact ties), the ties
//     are cut by item id (a second bisection; items arrive in ascending order, so a later entry with that score can never
//     displace a kept one and the lane's own bound becomes strict);
//   * every lane packs its survivors to the front of its own buffer, in place (slot j <= slot i: no hazard).
// Nothing is sorted and nothing exact is needed here: ANY T with at least K entries at or above it is a valid lower bound of
// the user's final K-th best.  The exact, ordered top-K is k_collect_topk's job after the sweep (rm_finalize.hpp): a block per
// user, all item ranges and sub-tile waves of the user at once.
// A buffer that holds K + slack entries after a selection refills after ~(cap - that) more: with cap = 2K + 16 the stream
// position grows ~10x between selections (2 per item range at 27k items, 4 at 10M) instead of 2x between compactions.
template <class S> struct LaneSel;
template <> struct LaneSel<float> {
    typedef unsigned Key;
    static constexpr int LPU = 2;                                   // lanes per user: lane and lane ^ 32
    static constexpr Key KEY_TOP = 0xff800001u;                     // ord_key(+inf) + 1
    static __device__ __forceinline__ int usum(int x) { return x + __shfl_xor(x, 32); }
    static __device__ __forceinline__ float umax(float x) { return __builtin_fmaxf(x, __shfl_xor(x, 32)); }
    static __device__ __forceinline__ float umin(float x) { return __builtin_fminf(x, __shfl_xor(x, 32)); }
    // (through L2: the entries were stored by this wave, the L1 may hold an older copy of the line)
    static __device__ __forceinline__ float load(const float *p) { return __uint_as_float(__hip_atomic_load((const unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
    // fp32 entries are (score, item) PAIRS, [entry][64 lanes] x 8 bytes: an append is ONE 8-byte store and the packing moves an entry
    // with one load and one store -- the sweep's epilogues and the selections of a CU's eight waves go through one address unit, and
    // what they cost there is the number of memory instructions (round 6: scores and items in separate arrays were 10,500 of them per
    // wave and item range at BASELINE C2's shape with K = 100, two thirds of them the appends' masked 4-byte stores)
    static constexpr int SS = 128, IS = 128;                        // floats / ints from one entry of a lane to its next
    static constexpr int BATCH = 32, NSAMPLE = 32;                  // loads in flight per pass over a buffer; entries of the in-register sample
    static __device__ __forceinline__ void load_entry(const float *sc, const int *, size_t i, float &x, int &id)
    {
        const unsigned long long b = __hip_atomic_load((const unsigned long long *)(sc + i * SS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = __uint_as_float((unsigned)b); id = (int)(unsigned)(b >> 32);
    }
    static __device__ __forceinline__ void store_entry(float *sc, int *, size_t j, float x, int id)
    {
        *(unsigned long long *)(sc + j * SS) = (unsigned long long)__float_as_uint(x) | ((unsigned long long)(unsigned)id << 32);
    }
    static __device__ __forceinline__ float nan() { return __uint_as_float(0xffffffffu); }
    static __device__ __forceinline__ float ninf() { return __uint_as_float(0xff800000u); }
    static __device__ __forceinline__ float pinf() { return __uint_as_float(0x7f800000u); }
    static __device__ __forceinline__ bool finite(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }
};
template <> struct LaneSel<double> {
    typedef unsigned long long Key;
    static constexpr int LPU = 4;                                   // lane, lane ^ 16, lane ^ 32, lane ^ 48
    static constexpr Key KEY_TOP = 0xfff0000000000001ull;
    static __device__ __forceinline__ int usum(int x) { x += __shfl_xor(x, 16); return x + __shfl_xor(x, 32); }
    static __device__ __forceinline__ double umax(double x) { x = __builtin_fmax(x, __shfl_xor(x, 16)); return __builtin_fmax(x, __shfl_xor(x, 32)); }
    static __device__ __forceinline__ double umin(double x) { x = __builtin_fmin(x, __shfl_xor(x, 16)); return __builtin_fmin(x, __shfl_xor(x, 32)); }
    static __device__ __forceinline__ double load(const double *p) { return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
    // fp64: the scores of a wave ([entry][64 lanes] x 8 bytes), then its item ids ([entry][64 lanes] x 4 bytes)
    static constexpr int SS = 64, IS = 64;
    // (half of the fp32 sizes: a score is two registers, and the selection -- a real call out of a kernel with a hundred live registers --
    // must leave room for them: with 32 + 32 doubles the callee took 209 VGPRs and the 128-factor-chunk kernels faulted, round 6)
    static constexpr int BATCH = 16, NSAMPLE = 16;
    static __device__ __forceinline__ void load_entry(const double *sc, const int *it, size_t i, double &x, int &id)
    {
        x = load(sc + i * SS); id = (int)__hip_atomic_load((const unsigned *)(it + i * IS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ void store_entry(double *sc, int *it, size_t j, double x, int id) { sc[j * SS] = x; it[j * IS] = id; }
    static __device__ __forceinline__ double nan() { return __longlong_as_double(-1ll); }
    static __device__ __forceinline__ double ninf() { return __longlong_as_double((long long)0xfff0000000000000ull); }
    static __device__ __forceinline__ double pinf() { return __longlong_as_double(0x7ff0000000000000ll); }
    static __device__ __forceinline__ bool finite(double x) { return ((unsigned long long)__double_as_longlong(x) & 0x7ff0000000000000ull) != 0x7ff0000000000000ull; }
};
__device__ __forceinline__ int lane_sel_load_item(const int *p) { return (int)__hip_atomic_load((const unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int wave_max_i32(int x)
{
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(x, d); x = o > x ? o : x; }
    return __builtin_amdgcn_readfirstlane(x);
}
// slack of a selection: the bisection stops at the first bound that leaves between K and K + slack entries
// (K / 2: the first pass over the whole buffer probes three guesses from a sample that are a standard deviation of the estimate
// apart -- ~K / 4 entries at K = 100 -- and one of them lands in a window of K / 2 nineteen times out of twenty; a window of K / 4
// took 3.7 passes per selection, each 57 KB of a wave's buffers through a memory system every wave of the chip is asking)
__host__ __device__ inline int lane_sel_slack(int K) { return K / 2 > 8 ? K / 2 : 8; }

// counts of the lane's entries [0, cend) at or above each of three probes (cend a multiple of 16; the slots behind the lane's
// own count hold NaN, which no compare accepts).  A pass is bound by the latency of its loads -- the buffers of a CU's waves are
// far larger than its share of the L2, a round trip is ~2,000 cycles (profiles/r6_ab_c2.txt) -- so 32 are in flight at a time.
template <class S>
__device__ __forceinline__ void lane_count3(const S *sc, int cend, S p1, S p2, S p3, int &c1, int &c2, int &c3)
{
    typedef LaneSel<S> L;
    c1 = 0; c2 = 0; c3 = 0;
    int i = 0;
    constexpr int B = L::BATCH;
    for (; i + B <= cend; i += B) {
        S x[B];
        #pragma unroll
        for (int t = 0; t < B; t++) x[t] = L::load(sc + (size_t)(i + t) * L::SS);
        #pragma unroll
        for (int t = 0; t < B; t++) { c1 += x[t] >= p1; c2 += x[t] >= p2; c3 += x[t] >= p3; }
    }
    if (B > 16 && i < cend) {
        S x[16];
        #pragma unroll
        for (int t = 0; t < 16; t++) x[t] = L::load(sc + (size_t)(i + t) * L::SS);
        #pragma unroll
        for (int t = 0; t < 16; t++) { c1 += x[t] >= p1; c2 += x[t] >= p2; c3 += x[t] >= p3; }
    }
}

// sc / it: the lane's columns (entry i at [i * 64]).  cnt: the lane's entries, in and out.  thr_in: the lane's present bound
// (every entry is at or above its predecessor in float order -- see below).  hi_hint: a score no entry of the USER exceeds (+inf
// if unknown).  Out: thr_out = the lane's new bound, kth_key = order-preserving key of the bound that may be published to the
// user's other partial lists (0: nothing new).  All lanes of the wave call this together.
template <class S>
__device__ __forceinline__ void lane_select(S *sc, int *it, int &cnt, const int K, const bool primary, const S thr_in, const S hi_hint,
                                            const int n_items, S &thr_out, typename LaneSel<S>::Key &kth_key)
{
    typedef LaneSel<S> L;
    typedef typename L::Key Key;
    const int slack = lane_sel_slack(K);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the appends have reached L2
    kth_key = 0; thr_out = thr_in;
    const int C = L::usum(cnt);
    const bool active = primary && C > K + slack;
    if (!wave_any(active)) return;
    const int cmax = wave_max_i32(active ? cnt : 0), cmin = -wave_max_i32(active ? -cnt : -0x7fffffff);
    const int cend = (cmax + 15) & ~15;                             // (cap is a multiple of 16)
    for (int i = cmin; i < cend; i++) if (active && i >= cnt) sc[(size_t)i * L::SS] = L::nan();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- bracket: #(>= unkey(lo)) >= K, #(>= unkey(hi)) < K.  Every entry is >= the float below the lane's bound: the bound is
    // either the last selection's T, or the float above it when that selection cut exact ties by item (the kept ties sit AT T)
    Key lo = ord_key(thr_in) - 1, hi = L::finite(hi_hint) ? ord_key(hi_hint) + 1 : L::KEY_TOP;
    const bool unbounded = active && !(thr_in > L::ninf());       // no bound yet (-inf): the smallest entry is the bracket's low end
    if (wave_any(unbounded)) {
        S mn = L::pinf();
        for (int i = 0; i < cend; i += 16) {
            S x[16];
            #pragma unroll
            for (int t = 0; t < 16; t++) x[t] = L::load(sc + (size_t)(i + t) * L::SS);
            #pragma unroll
            for (int t = 0; t < 16; t++) mn = x[t] < mn ? x[t] : mn;      // (NaN: compare false)
        }
        mn = L::umin(mn);
        if (unbounded) lo = ord_key(mn);
    }
    const Key lo0 = lo;
    int cT = C;                                                     // entries at or above unkey(lo) (at lo0: an upper bound, see below)
    bool done = !active;
    // ---- the first pass's probes from a SAMPLE: every (cend / 32)-th entry of the lane (fp64: every (cend / 16)-th), one round trip.  The sample's
    // order statistics at the ranks that correspond to ~K + slack / 2 entries of the whole and a standard deviation of the estimate to
    // either side are found in registers (three bisections in lock-step, no memory traffic).  Whatever the sample says only chooses
    // WHERE the first pass over the whole buffer probes: the bracket is updated from the true counts alone, so a misleading sample
    // costs passes, never correctness.  (Without it: ~8 passes of 14 dependent round trips each; with it 2.)
    Key sk1 = 0, sk2 = 0, sk3 = 0;
    {
        constexpr int NS = L::NSAMPLE;
        const int st = cend / NS > 0 ? cend / NS : 1;
        S xs[NS];
        #pragma unroll
        for (int t = 0; t < NS; t++) xs[t] = t * st < cend ? L::load(sc + (size_t)(t * st) * L::SS) : L::nan();
        int ns = 0;
        #pragma unroll
        for (int t = 0; t < NS; t++) ns += xs[t] == xs[t];
        ns = L::usum(ns);
        // sample ranks: t2 ~ (K + slack / 2) ns / C, t1 = t2 + e (more entries: the lower key), t3 = t2 - e
        const float scale = (float)ns / (float)(C > 0 ? C : 1);
        int t2 = (int)((float)(K + slack / 2) * scale + 0.5f);
        t2 = t2 < 1 ? 1 : t2;
        int e = (int)(__builtin_sqrtf((float)t2) + 0.5f);
        e = e < 1 ? 1 : e;
        const int tg1 = t2 + e, tg2 = t2, tg3 = t2 - e > 0 ? t2 - e : 1;
        // three brackets over the sample: #(>= lo_j) >= tg_j > #(>= hi_j)
        Key l1 = lo, l2 = lo, l3 = lo, h1 = hi, h2 = hi, h3 = hi;
        bool sdone = !active || ns < tg1;                           // (too few samples at or above the bracket's low end: no guess)
        for (int sp = 0; sp < 40 && wave_any(!sdone); sp++) {
            auto mid = [&](Key l, Key h) -> Key {
                const Key dk = h - l;
                if (dk <= 1) return l;
                if (sp < 12) {
                    const S lf = ord_unkey(l), hf = ord_unkey(h - 1);
                    if (L::finite(lf) && L::finite(hf)) { const Key m = ord_key(lf + (hf - lf) * (S)0.5); if (m > l && m < h) return m; }
                }
                return l + (dk >> 1);
            };
            const Key m1 = mid(l1, h1), m2 = mid(l2, h2), m3 = mid(l3, h3);
            const S f1 = ord_unkey(m1), f2 = ord_unkey(m2), f3 = ord_unkey(m3);
            int a1 = 0, a2 = 0, a3 = 0;
            #pragma unroll
            for (int t = 0; t < NS; t++) { a1 += xs[t] >= f1; a2 += xs[t] >= f2; a3 += xs[t] >= f3; }
            a1 = L::usum(a1); a2 = L::usum(a2); a3 = L::usum(a3);
            // (a probe with exactly the wanted count IS an answer: any key between two neighbouring sample values serves)
            if (h1 - l1 > 1) { if (a1 == tg1) { l1 = m1; h1 = m1 + 1; } else if (a1 > tg1) l1 = m1; else h1 = m1; }
            if (h2 - l2 > 1) { if (a2 == tg2) { l2 = m2; h2 = m2 + 1; } else if (a2 > tg2) l2 = m2; else h2 = m2; }
            if (h3 - l3 > 1) { if (a3 == tg3) { l3 = m3; h3 = m3 + 1; } else if (a3 > tg3) l3 = m3; else h3 = m3; }
            RM_SEL_STAT(1, 1);
            sdone = sdone || (h1 - l1 <= 1 && h2 - l2 <= 1 && h3 - l3 <= 1);
            if (sp >= 7) sdone = true;              // (seven halvings of the value range: the probes only say where the first pass looks)
        }
        if (active && ns >= tg1) { sk1 = l1; sk2 = l2; sk3 = l3; }
    }
    for (int pass = 0; wave_any(!done); pass++) {
        Key k1, k2, k3;
        const Key d = hi - lo;                                      // >= 2 while not done
        bool by_value = false;
        if (pass == 0 && sk1 > lo && sk2 >= sk1 && sk3 >= sk2 && sk3 < hi) { k1 = sk1; k2 = sk2; k3 = sk3; by_value = true; }
        else if (pass < 8) {                                        // quartiles in value space
            const S lf = ord_unkey(lo), hf = ord_unkey(hi - 1);
            if (L::finite(lf) && L::finite(hf)) {
                const S w = hf - lf;
                k1 = ord_key(lf + w * (S)0.25); k2 = ord_key(lf + w * (S)0.5); k3 = ord_key(lf + w * (S)0.75);
                by_value = k1 > lo && k2 >= k1 && k3 >= k2 && k3 < hi;
            }
        }
        if (!by_value) {                                            // quartiles in key space: (d >> 2) * 3 < d
            const Key q = d >> 2;
            k1 = lo + (q ? q : (Key)1); k2 = lo + (q ? 2 * q : (Key)1); k3 = lo + (q ? 3 * q : (Key)1);
        }
        int c1, c2, c3;
        RM_SEL_STAT(0, 1);
        lane_count3<S>(sc, cend, ord_unkey(k1), ord_unkey(k2), ord_unkey(k3), c1, c2, c3);
        c1 = L::usum(c1); c2 = L::usum(c2); c3 = L::usum(c3);
        if (!done) {
            if (c3 >= K) { lo = k3; cT = c3; }
            else if (c2 >= K) { lo = k2; cT = c2; hi = k3; }
            else if (c1 >= K) { lo = k1; cT = c1; hi = k2; }
            else hi = k1;
            done = cT <= K + slack || hi - lo <= 1;
        }
    }
    S T = ord_unkey(lo);
    // The lane's bound may have come from ANOTHER partial list of the user since the entries were appended (the shared bound): then
    // some entries lie below it and possibly fewer than K at or above it -- no probe found K, the bracket closed on its low end,
    // whose count was never taken.  Take it: with fewer than K the lane keeps what reaches the bound it has and learns nothing new.
    const bool below = active && lo == lo0 && !unbounded;
    bool own_short = false;
    if (wave_any(below)) {
        RM_SEL_STAT(3, 1);
        int c0 = 0;
        for (int i = 0; i < cend; i++) { const S x = L::load(sc + (size_t)i * L::SS); c0 += x >= T; }
        c0 = L::usum(c0);
        if (below) { cT = c0; own_short = c0 < K; }
    }
    if (own_short) T = thr_in;
    // ---- exact ties: more than K + slack entries at or above the exact K-th best score -> cut the ties by item id ----
    const bool ties = active && !own_short && cT > K + slack;
    int item_max = IDX_EMPTY;
    if (wave_any(ties)) {
        RM_SEL_STAT(2, 1);
        const S above = ord_unkey(lo + 1);
        int ngt = 0;
        for (int i = 0; i < cend; i++) { const S x = L::load(sc + (size_t)i * L::SS); ngt += x >= above; }
        const int need = K - L::usum(ngt);                          // >= 1 ties to keep: those with the smallest items
        int ilo = -1, ihi = n_items - 1;                            // #(ties with item <= ilo) < need <= #(... <= ihi)
        while (wave_any(ties && ihi - ilo > 1)) {
            const int im = ilo + ((ihi - ilo) >> 1);
            int c = 0;
            for (int i = 0; i < cend; i++) {
                const S x = L::load(sc + (size_t)i * L::SS);
                const int id = lane_sel_load_item(it + (size_t)i * L::IS);
                c += (x == T) && id <= im;
            }
            c = L::usum(c);
            if (ties && ihi - ilo > 1) { if (c >= need) ihi = im; else ilo = im; }
        }
        if (ties) item_max = ihi;
    }
    // ---- pack the survivors to the front of the lane's own buffer ----
    int j = 0;
    for (int i = 0; i < cend; i += 16) {
        S x[16]; int id[16];
        #pragma unroll
        for (int t = 0; t < 16; t++) L::load_entry(sc, it, (size_t)(i + t), x[t], id[t]);
        #pragma unroll
        for (int t = 0; t < 16; t++) {
            const bool keep = active && x[t] >= T && (!ties || x[t] > T || id[t] <= item_max);
            if (keep) {
                if (j != i + t) L::store_entry(sc, it, (size_t)j, x[t], id[t]);
                j++;
            }
        }
    }
    if (active) cnt = j;
    if (active && !own_short) {
        const S tn = ties ? ord_unkey(lo + 1) : T;
        thr_out = tn > thr_in ? tn : thr_in;
        kth_key = lo;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // survivors in place before anyone appends behind them
}
// Out of line (both sweeps): a selection runs two or three times per item range, its working set (a 32-entry sample, 32 loads in
// flight) would otherwise be added to the register budget of the tile loop -- inlined into the fp32 sweep it pushed the kernels of
// 24-40 factors from 13 to 110 spilled SGPRs.  Results come back by value (in registers), not through references (scratch).
template <class S> struct LaneSelResult { int cnt; S thr; typename LaneSel<S>::Key kth_key; };
template <class S>
__device__ __attribute__((noinline)) LaneSelResult<S> lane_select_call(S *sc, int *it, int cnt, const int K, const bool primary, const S thr_in, const S hi_hint,
                                                                       const int n_items)
{
    LaneSelResult<S> r;
    r.cnt = cnt;
    lane_select<S>(sc, it, r.cnt, K, primary, thr_in, hi_hint, n_items, r.thr, r.kth_key);
    return r;
}

} // namespace rm
