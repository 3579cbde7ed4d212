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

} // namespace rm
