"""ctypes access to the CHECKERS -- test infrastructure only.

* ``Oracle``    : this repo's CPU restatement (oracle/recometrics_oracle.cpp -> librecometrics_oracle.so)
* ``Reference`` : the real reference compiled by oracle/Makefile from /root/reference
                  (oracle/_ref/librecometrics_ref.so, canonical build; ``fast=True`` picks the
                  -march=x86-64-v3 build used only as a CPU timing baseline)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (recometrics_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
METRICS = ("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr", "roc", "pr")
NAMES = {"p": "P@K", "tp": "TP@K", "r": "R@K", "ap": "AP@K", "tap": "TAP@K", "ndcg": "NDCG@K",
         "hit": "Hit@K", "rr": "RR@K", "roc": "ROC_AUC", "pr": "PR_AUC"}

_REF_SYMS = {
    np.float32: b"_Z18calc_metrics_floatPKfmS0_miiiPKiS2_S2_PiS0_ibbPfS4_S4_S4_S4_S4_S4_S4_S4_S4_biiim",
    np.float64: b"_Z19calc_metrics_doublePKdmS0_miiiPKiS2_S2_PiS0_ibbPdS4_S4_S4_S4_S4_S4_S4_S4_S4_biiim",
}


def build(quiet=True):
    """(Re)build the checker libraries with oracle/Makefile (oracle/_ref only when /root/reference exists)."""
    subprocess.run(["make", "-C", _HERE] + (["-s"] if quiet else []), check=True,
                   stdout=subprocess.DEVNULL if quiet else None, stderr=subprocess.DEVNULL if quiet else None)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _prep(A, B, train, test, dtype):
    """train/test: (indptr, indices[, data]) tuples or scipy CSR."""
    def parts(X, want_data):
        if hasattr(X, "indptr"):
            X = (X.indptr, X.indices, X.data if want_data else None)
        p = np.ascontiguousarray(X[0], dtype=np.int32)
        i = np.ascontiguousarray(X[1], dtype=np.int32)
        v = None
        if want_data and len(X) > 2 and X[2] is not None:
            v = np.ascontiguousarray(X[2], dtype=dtype)
        return p, i, v
    A = np.ascontiguousarray(A, dtype=dtype)
    B = np.ascontiguousarray(B, dtype=dtype)
    trp, tri, _ = parts(train, False)
    tep, tei, tev = parts(test, True)
    return A, B, trp, tri, tep, tei, tev


class _Lib:
    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.lib = C.CDLL(path)

    def _calc(self, fn, boolt, A, B, train, test, k, metrics, cumulative, noise, cold,
              min_items_pool, min_pos_test, nthreads, seed, dtype):
        A, B, trp, tri, tep, tei, tev = _prep(A, B, train, test, dtype)
        m, kf = A.shape
        n = B.shape[0]
        assert B.shape[1] == kf and trp.shape[0] == m + 1 and tep.shape[0] == m + 1
        if tev is None:
            tev = np.ones(tei.shape[0], dtype=dtype)
        if tri.shape[0] == 0:
            tri = np.zeros(1, dtype=np.int32)
        outs = {}
        for name in METRICS:
            if name in metrics:
                shape = (m, k) if (cumulative and name not in ("roc", "pr")) else (m,)
                outs[name] = np.full(shape, -777.0, dtype=dtype)
            else:
                outs[name] = None
        fn.restype = C.c_int if boolt is C.c_int else None
        fn.argtypes = ([C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32]
                       + [C.c_void_p] * 5 + [C.c_int32, boolt, boolt] + [C.c_void_p] * 10
                       + [boolt, C.c_int32, C.c_int32, C.c_int32, C.c_uint64])
        rc = fn(_ptr(A), A.shape[1], _ptr(B), B.shape[1], m, n, kf,
                _ptr(trp), _ptr(tri), _ptr(tep), _ptr(tei), _ptr(tev),
                k, bool(cumulative), bool(noise),
                *[_ptr(outs[name]) for name in METRICS],
                bool(cold), min_items_pool, min_pos_test, nthreads, seed)
        if rc not in (None, 0):
            raise RuntimeError("checker returned status %r" % rc)
        return {NAMES[kname]: v for kname, v in outs.items() if v is not None}


class Oracle(_Lib):
    """This repo's restatement; also offers the ranking dump used by the index-set parity tests."""

    def __init__(self):
        # RECOMETRICS_ORACLE_LIB: another build of the same restatement (the sanitizer build of oracle/Makefile)
        path = os.environ.get("RECOMETRICS_ORACLE_LIB") or os.path.join(_HERE, "librecometrics_oracle.so")
        if not os.path.exists(path):
            build()
        super().__init__(path)

    def calc(self, A, B, train, test, k, metrics=METRICS, cumulative=False, noise=False, cold=True,
             min_items_pool=2, min_pos_test=1, nthreads=1, seed=1, dtype=np.float32):
        fn = self.lib.rmo_calc_metrics_f32 if dtype == np.float32 else self.lib.rmo_calc_metrics_f64
        return self._calc(fn, C.c_int, A, B, train, test, k, metrics, cumulative, noise, cold,
                          min_items_pool, min_pos_test, nthreads, seed, dtype)

    def rank(self, A, B, train, test, k, noise=False, cold=True, min_items_pool=2, min_pos_test=1,
             nthreads=1, seed=1, dtype=np.float32):
        A, B, trp, tri, tep, tei, _ = _prep(A, B, train, test, dtype)
        m, kf = A.shape
        n = B.shape[0]
        if tri.shape[0] == 0:
            tri = np.zeros(1, dtype=np.int32)
        idx = np.empty((m, k), dtype=np.int32)
        sc = np.empty((m, k), dtype=dtype)
        pr = np.empty(tei.shape[0], dtype=np.int64)
        st = np.empty(m, dtype=np.int32)
        fn = self.lib.rmo_rank_f32 if dtype == np.float32 else self.lib.rmo_rank_f64
        fn.restype = C.c_int
        fn.argtypes = ([C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32]
                       + [C.c_void_p] * 4 + [C.c_int32, C.c_int, C.c_int, C.c_int32, C.c_int32, C.c_int32, C.c_uint64]
                       + [C.c_void_p] * 4)
        rc = fn(_ptr(A), kf, _ptr(B), kf, m, n, kf, _ptr(trp), _ptr(tri), _ptr(tep), _ptr(tei),
                k, int(noise), int(cold), min_items_pool, min_pos_test, nthreads, seed,
                _ptr(idx), _ptr(sc), _ptr(pr), _ptr(st))
        if rc:
            raise RuntimeError("oracle rank returned %d" % rc)
        return {"topk_idx": idx, "topk_score": sc, "pos_rank": pr, "status": st}


    def scores(self, A, B, dtype=np.float32):
        A = np.ascontiguousarray(A, dtype=dtype)
        B = np.ascontiguousarray(B, dtype=dtype)
        out = np.empty((A.shape[0], B.shape[0]), dtype=dtype)
        fn = self.lib.rmo_scores_f32 if dtype == np.float32 else self.lib.rmo_scores_f64
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        fn(_ptr(A), A.shape[1], _ptr(B), B.shape[1], A.shape[0], B.shape[0], A.shape[1], _ptr(out))
        return out


class Reference(_Lib):
    """The real reference (C++-linkage symbols called through their mangled names)."""

    def __init__(self, fast=False, v4=False):
        name = "librecometrics_ref_v4.so" if v4 else ("librecometrics_ref_fast.so" if fast else "librecometrics_ref.so")
        path = os.path.join(_HERE, "_ref", name)
        if not os.path.exists(path) and os.path.exists("/root/reference/src/recometrics.hpp"):
            build()
        super().__init__(path)

    def calc(self, A, B, train, test, k, metrics=METRICS, cumulative=False, noise=False, cold=True,
             min_items_pool=2, min_pos_test=1, nthreads=1, seed=1, dtype=np.float32):
        fn = getattr(self.lib, _REF_SYMS[np.float32 if dtype == np.float32 else np.float64].decode())
        return self._calc(fn, C.c_bool, A, B, train, test, k, metrics, cumulative, noise, cold,
                          min_items_pool, min_pos_test, nthreads, seed, dtype)


def reference_available(fast=False, v4=False):
    name = "librecometrics_ref_v4.so" if v4 else ("librecometrics_ref_fast.so" if fast else "librecometrics_ref.so")
    return os.path.exists(os.path.join(_HERE, "_ref", name))


def host_has_avx512():
    """the x86-64-v4 build needs AVX-512 F / BW / CD / DQ / VL: only run it on a host whose CPU lists them"""
    try:
        flags = set()
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                flags = set(line.split(":", 1)[1].split())
                break
        return {"avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl"} <= flags
    except OSError:
        return False
