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

// Out-of-line form for the fp64 sweep: inlined, the nine-entry-per-lane working set is added to a register budget that
// is already full (256 factors) and the hot loop spills; as a real call only the live registers around this rare path
// are saved.  The fp32 sweep has the room and keeps the inlined form (a call there costs more than it saves).
template <class S>
__device__ __attribute__((noinline)) void wave_compact_call(typename ListRaw<S>::raw *Gu, int cnt, int K, int lane, S &kth_s, int &kth_idx)
#include "rm_compact_body.inc"

} // namespace rm
