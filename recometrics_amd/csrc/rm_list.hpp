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

// ---- large K: append buffer + wave-cooperative compaction (lists that do not fit LDS live in HBM) -----------------------
// A user's buffer holds up to CAP = 2K + 32 raw entries, user-major (contiguous) in HBM.  The owner lane appends every
// candidate that passes its threshold (one store); when more than 2K have accumulated, the whole wave selects the K best:
// every lane holds cnt/64 entries, the buffer is replayed 64 entries at a time through lane broadcasts, each entry gets
// its rank in the (score desc, item asc) order by counting, and entries ranked < K are written back at [rank] -- so the
// survivors are also sorted.  The threshold only rises at a compaction, so the stream position doubles between
// compactions: ~log2(n/K) of them per user and ~K of appends per compaction.
template <class S>
__device__ __forceinline__ void wave_compact(typename ListRaw<S>::raw *Gu, int cnt, int K, int lane, S &kth_s, int &kth_idx)
#include "rm_compact_body.inc"

// Compaction DURING the sweep does not need the survivors sorted, only to be the K best: the K-th best key is found by
// bisection over the bits of the order-preserving key (count of entries at or above the candidate by ballots: 64 steps of a
// few instructions per held entry, against cnt x entries for the rank count above -- about a quarter of the instructions at
// K = 100), then the survivors are packed to the front in their old order.  The waves of a block wait for one another at
// every tile, so the time a wave spends here is paid by all of them: compaction skew, not the appends, is what holds the
// append-buffer variants back (profiles/r2: 35-48 % of their wave cycles parked).
// Order: (score desc, item asc); `hi` = order-preserving score key, ties by the smaller item.
__device__ __forceinline__ unsigned long long sel_score_key(float s) { return (unsigned long long)ord_key(s); }
__device__ __forceinline__ unsigned long long sel_score_key(double s) { return ord_key(s); }
// (E = entries per lane as a template constant, loads unconditional with a clamped index: with a run-time E and predicated
// loads the compiler shuffled the register arrays through 240 VGPRs of copies)
template <class S, int E>
__device__ __forceinline__ void wave_select_e(typename ListRaw<S>::raw *Gu, int cnt, int K, int lane, S &kth_s, int &kth_idx)
{
    unsigned long long hi[E]; int it[E];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the owner lane's appends have reached L2
    #pragma unroll
    for (int t = 0; t < E; t++) {
        const int i = lane + t * WAVE;
        S es; int ei;
        ListRaw<S>::unpack(ListRaw<S>::load_l2(Gu + (i < cnt ? i : 0)), es, ei);
        hi[t] = i < cnt ? sel_score_key(es) : 0ull;          // key 0 is below every real score's key
        it[t] = i < cnt ? ei : IDX_EMPTY;
    }
    // K-th largest score key
    constexpr int BITS = sizeof(S) == 4 ? 32 : 64;
    unsigned long long T = 0ull;
    for (int bit = BITS - 1; bit >= 0; bit--) {
        const unsigned long long cand = T | (1ull << bit);
        int c = 0;
        #pragma unroll
        for (int t = 0; t < E; t++) c += __popcll(__ballot(hi[t] >= cand));
        if (c >= K) T = cand;
    }
    int n_gt = 0, n_eq = 0;
    #pragma unroll
    for (int t = 0; t < E; t++) { n_gt += __popcll(__ballot(hi[t] > T)); n_eq += __popcll(__ballot(hi[t] == T)); }
    // among the entries that tie the K-th score: the (K - n_gt) smallest items (usually there is exactly one such entry)
    const int need = K - n_gt;
    int item_max = IDX_EMPTY;                                 // keep equal-scored entries with item <= item_max
    if (n_eq > need) {
        unsigned I = 0u;                                      // the need-th smallest item among the ties, bit by bit
        for (int bit = 30; bit >= 0; bit--) {
            const unsigned cand = I | (1u << bit);
            int c = 0;
            #pragma unroll
            for (int t = 0; t < E; t++) c += __popcll(__ballot(hi[t] == T && (unsigned)it[t] < cand));
            if (c < need) I = cand;
        }
        item_max = (int)I;
    }
    // pack the survivors to the front (all entries are in registers: a write can only hit a slot already read)
    int base = 0, worst_item = -1;
    #pragma unroll
    for (int t = 0; t < E; t++) {
        const bool keep = hi[t] > T || (hi[t] == T && it[t] <= item_max);
        const unsigned long long m = __ballot(keep);
        if (keep) {
            S es;
            if (sizeof(S) == 4) es = (S)ord_unkey((unsigned)hi[t]); else es = (S)ord_unkey(hi[t]);
            Gu[base + __popcll(m & ((1ull << lane) - 1ull))] = ListRaw<S>::pack(es, it[t]);
        }
        base += __popcll(m);
        if (hi[t] == T && it[t] <= item_max && it[t] > worst_item) worst_item = it[t];
    }
    #pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(worst_item, d); worst_item = o > worst_item ? o : worst_item; }
    if (sizeof(S) == 4) kth_s = (S)ord_unkey((unsigned)T); else kth_s = (S)ord_unkey(T);
    kth_idx = worst_item;                                     // the K-th best = the tie with the largest kept item
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // survivors written before anyone appends behind them
}
template <class S>
__device__ __forceinline__ void wave_select(typename ListRaw<S>::raw *Gu, int cnt, int K, int lane, S &kth_s, int &kth_idx)
{
    switch ((cnt + WAVE - 1) / WAVE) {                        // (2 * 256 + 32) / 64 rounded up = 9 at most
        case 1: wave_select_e<S, 1>(Gu, cnt, K, lane, kth_s, kth_idx); break;
        case 2: wave_select_e<S, 2>(Gu, cnt, K, lane, kth_s, kth_idx); break;
        case 3: wave_select_e<S, 3>(Gu, cnt, K, lane, kth_s, kth_idx); break;
        case 4: wave_select_e<S, 4>(Gu, cnt, K, lane, kth_s, kth_idx); break;
        case 5: wave_select_e<S, 5>(Gu, cnt, K, lane, kth_s, kth_idx); break;
        case 6: wave_select_e<S, 6>(Gu, cnt, K, lane, kth_s, kth_idx); break;
        case 7: wave_select_e<S, 7>(Gu, cnt, K, lane, kth_s, kth_idx); break;
        case 8: wave_select_e<S, 8>(Gu, cnt, K, lane, kth_s, kth_idx); break;
        default: wave_select_e<S, 9>(Gu, cnt, K, lane, kth_s, kth_idx); break;
    }
}
template <class S>
__device__ __attribute__((noinline)) void wave_select_call(typename ListRaw<S>::raw *Gu, int cnt, int K, int lane, S &kth_s, int &kth_idx)
{
    wave_select<S>(Gu, cnt, K, lane, kth_s, kth_idx);
}

// Out-of-line form for the fp64 sweep: inlined, the nine-entry-per-lane working set is added to a register budget that
// is already full (256 factors) and the hot loop spills; as a real call only the live registers around this rare path
// are saved.  The fp32 sweep has the room and keeps the inlined form (a call there costs more than it saves).
template <class S>
__device__ __attribute__((noinline)) void wave_compact_call(typename ListRaw<S>::raw *Gu, int cnt, int K, int lane, S &kth_s, int &kth_idx)
#include "rm_compact_body.inc"

} // namespace rm
