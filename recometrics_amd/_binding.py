"""ctypes binding of include/recometrics_hip.h (the C-ABI of librecometrics_hip.so).

Plays the role of the reference's Cython layer (recometrics/wrapper.pyx:226-495): pulls the raw buffers out of
the NumPy / SciPy objects, pre-allocates the outputs (size 0 == "not requested" == NULL pointer,
wrapper.pyx:208-224,:270-280) and turns status codes into the exceptions `except +` produces there.
There is NO fallback: if the HIP library is missing or there is no device, calls raise.
"""
import ctypes as C
import os

import numpy as np

_LIB_PATH = os.environ.get("RECOMETRICS_HIP_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "csrc", "librecometrics_hip.so")
_lib = None

METRIC_ORDER = ("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr", "roc", "pr")
EXPORTS = (
    "rm_calc_metrics_f32", "rm_calc_metrics_f64", "rm_calc_metrics_dev_f32", "rm_calc_metrics_dev_f64",
    "rm_rank_f32", "rm_rank_f64", "rm_debug_scores_f32", "rm_debug_scores_f64", "rm_has_openmp",
    "rm_last_error", "rm_device_count", "rm_set_device", "rm_set_devices", "rm_get_devices", "rm_request_interrupt",
    "rm_get_timings", "rm_release_workspace", "rm_debug_reload_switches",
    "rm_split_f32", "rm_split_f64", "rm_split_size", "rm_split_copy", "rm_split_free", "rm_split_last_error",
    "rm_csr_rows_sorted", "rm_csr_sort_rows",
)


class HipLibraryMissing(ImportError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Loads the HIP library (once).  Raises HipLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise HipLibraryMissing(
            "recometrics_amd: %s is missing -- build it with `python -m recometrics_amd.build` "
            "(there is no CPU fallback)" % _LIB_PATH)
    lib = C.CDLL(_LIB_PATH)
    vp, i32, i64, u64, sz, ci = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_size_t, C.c_int
    host = [vp, sz, vp, sz, i32, i32, i32, vp, vp, vp, vp, vp, i32, ci, ci] + [vp] * 10 + [ci, i32, i32, i32, u64]
    dev = [vp, sz, vp, sz, i32, i32, i32, vp, vp, i64, vp, vp, vp, i64, i32, ci, ci] + [vp] * 10 + [ci, i32, i32, u64, vp]
    rank_sig = [vp, sz, vp, sz, i32, i32, i32, vp, vp, vp, vp, i32, ci, ci, i32, i32, u64, vp, vp, vp, vp]
    for suf in ("f32", "f64"):
        getattr(lib, "rm_calc_metrics_" + suf).argtypes = host
        getattr(lib, "rm_calc_metrics_dev_" + suf).argtypes = dev
        getattr(lib, "rm_rank_" + suf).argtypes = rank_sig
        getattr(lib, "rm_debug_scores_" + suf).argtypes = [vp, sz, vp, sz, i32, i32, i32, vp]
        for fn in ("rm_calc_metrics_", "rm_calc_metrics_dev_", "rm_rank_", "rm_debug_scores_"):
            getattr(lib, fn + suf).restype = ci
    split_sig = [vp, vp, vp, i32, i32, ci, i32, C.c_double, ci, i32, i32, u64, C.POINTER(vp)]
    lib.rm_split_f32.argtypes = split_sig
    lib.rm_split_f64.argtypes = split_sig
    lib.rm_split_f32.restype = lib.rm_split_f64.restype = ci
    lib.rm_split_size.argtypes = [vp, ci]
    lib.rm_split_size.restype = i64
    lib.rm_split_copy.argtypes = [vp, ci, vp]
    lib.rm_split_copy.restype = ci
    lib.rm_split_free.argtypes = [vp]
    lib.rm_split_free.restype = None
    lib.rm_split_last_error.restype = C.c_char_p
    lib.rm_csr_rows_sorted.argtypes = [vp, vp, i32, i32]
    lib.rm_csr_rows_sorted.restype = ci
    lib.rm_csr_sort_rows.argtypes = [vp, vp, vp, i32, i32, i32]
    lib.rm_csr_sort_rows.restype = ci
    lib.rm_last_error.restype = C.c_char_p
    lib.rm_get_timings.argtypes = [C.POINTER(C.c_double), ci]
    lib.rm_set_device.argtypes = [ci]
    lib.rm_set_devices.argtypes = [C.POINTER(i32), i32]
    lib.rm_get_devices.argtypes = [C.POINTER(i32), i32]
    lib.rm_request_interrupt.restype = None
    lib.rm_debug_reload_switches.restype = None
    _lib = lib
    return lib


_split_lib = None


def _load_split():
    """The library that provides rm_split_* (host-only code): librecometrics_hip.so, or -- RECOMETRICS_SPLIT_LIB -- another
    build of csrc/rm_split.cpp alone (the sanitizer build of oracle/Makefile, which has no device code to link)."""
    global _split_lib
    alt = os.environ.get("RECOMETRICS_SPLIT_LIB")
    if not alt:
        return load()
    if _split_lib is None:
        lib = C.CDLL(alt)
        vp, i32, i64, u64, ci = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_int
        sig = [vp, vp, vp, i32, i32, ci, i32, C.c_double, ci, i32, i32, u64, C.POINTER(vp)]
        lib.rm_split_f32.argtypes = lib.rm_split_f64.argtypes = sig
        lib.rm_split_f32.restype = lib.rm_split_f64.restype = ci
        lib.rm_split_size.argtypes = [vp, ci]
        lib.rm_split_size.restype = i64
        lib.rm_split_copy.argtypes = [vp, ci, vp]
        lib.rm_split_copy.restype = ci
        lib.rm_split_free.argtypes = [vp]
        lib.rm_split_free.restype = None
        lib.rm_split_last_error.restype = C.c_char_p
        _split_lib = lib
    return _split_lib


def _raise(lib, rc):
    msg = (lib.rm_last_error() or b"").decode(errors="replace")
    if rc == 1:
        raise ValueError(msg)
    if rc == 3:
        raise MemoryError(msg)
    if rc == 5:
        # the reference throws std::runtime_error("Error: procedure was interrupted.") after re-raising SIGINT
        # (src/recometrics.hpp:166-173); Cython's `except +` turns it into this RuntimeError, and Python's own handler
        # raises KeyboardInterrupt for the re-raised signal
        raise RuntimeError(msg.strip())
    raise RuntimeError("recometrics_amd (status %d): %s" % (rc, msg))


def _p(a):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


def has_openmp():
    return bool(load().rm_has_openmp())


def device_count():
    return int(load().rm_device_count())


def set_device(i):
    lib = load()
    rc = lib.rm_set_device(int(i))
    if rc:
        _raise(lib, rc)


def set_devices(devices):
    """Devices of the host-pointer calls of this process (rm_set_devices): a list of device ids, the same id may repeat;
    an empty list restores "the current device"."""
    lib = load()
    arr = (C.c_int32 * max(len(devices), 1))(*devices)
    rc = lib.rm_set_devices(arr, len(devices))
    if rc:
        _raise(lib, rc)


def get_devices():
    lib = load()
    buf = (C.c_int32 * 64)()
    n = lib.rm_get_devices(buf, 64)
    return [int(buf[i]) for i in range(min(n, 64))]


def request_interrupt():
    load().rm_request_interrupt()


def reload_switches():
    """Test hook: the library reads its RM_DEBUG_* switches from the environment when it is loaded; this reads them again."""
    load().rm_debug_reload_switches()


def timings():
    lib = load()
    buf = (C.c_double * 10)()
    n = lib.rm_get_timings(buf, 10)
    keys = ("prep_ms", "sweep_ms", "finalize_ms", "device_ms", "sweep_launches", "item_splits", "sweep_blocks", "lds_bytes",
            "timed_slots", "total_slots")
    return {k: buf[i] for i, k in enumerate(keys[:n])}


def _csr_lib():
    """librecometrics_hip.so, or -- RECOMETRICS_SPLIT_LIB -- the sanitizer build of the host-only units (oracle/Makefile)"""
    lib = _load_split()
    if not getattr(lib, "_csr_ready", False):
        vp, i32, ci = C.c_void_p, C.c_int32, C.c_int
        lib.rm_csr_rows_sorted.argtypes = [vp, vp, i32, i32]
        lib.rm_csr_rows_sorted.restype = ci
        lib.rm_csr_sort_rows.argtypes = [vp, vp, vp, i32, i32, i32]
        lib.rm_csr_sort_rows.restype = ci
        lib._csr_ready = True
    return lib


def csr_rows_sorted(indptr, indices, nthreads=0):
    """True when the column indices of every row of an int32 CSR ascend (rm_csr_rows_sorted: multi-threaded pass)."""
    lib = _csr_lib()
    assert indptr.dtype == np.int32 and indices.dtype == np.int32
    rc = lib.rm_csr_rows_sorted(_p(indptr), _p(indices), indptr.shape[0] - 1, int(nthreads))
    if rc < 0:
        raise ValueError("rm_csr_rows_sorted: bad argument")
    return bool(rc)


def csr_sort_rows(indptr, indices, data, nthreads=0):
    """Sorts every row's (index, value) pairs by index, in place (rm_csr_sort_rows); `data` may be None."""
    lib = _csr_lib()
    assert indptr.dtype == np.int32 and indices.dtype == np.int32 and indices.flags.writeable
    vb = 0 if data is None else data.dtype.itemsize
    if vb not in (0, 4, 8):
        raise ValueError("values must be 4 or 8 bytes wide")
    rc = lib.rm_csr_sort_rows(_p(indptr), _p(indices), None if data is None else _p(data), vb, indptr.shape[0] - 1, int(nthreads))
    if rc:
        raise (MemoryError if rc == 3 else ValueError)("rm_csr_sort_rows failed (status %d)" % rc)


def _suffix(dtype):
    return "f32" if dtype == np.float32 else "f64"


def calc_metrics(A, lda, B, ldb, train_p, train_i, test_p, test_i, test_v, k_metrics, want, cumulative,
                 break_ties_with_noise, consider_cold_start, min_items_pool, min_pos_test, nthreads, seed, outs=None):
    """Host-array entry.  `want`: dict metric-name -> bool in METRIC_ORDER.  Returns the 10-tuple of arrays the
    reference's cpp_funs.calc_reco_metrics returns (wrapper.pyx:306-323): size-0 arrays for metrics not requested,
    (m, k) arrays for cumulative top-K metrics."""
    lib = load()
    dtype = A.dtype.type
    m, k = A.shape
    n = B.shape[0]
    size_arr = m * k_metrics if cumulative else m
    if outs is None:                                   # `outs`: ten caller-owned flat arrays (size 0 = not requested)
        outs = []
        for name in METRIC_ORDER:
            cnt = (m if name in ("roc", "pr") else size_arr) if want.get(name) else 0
            outs.append(np.empty(cnt, dtype=dtype))
    fn = getattr(lib, "rm_calc_metrics_" + _suffix(dtype))
    rc = fn(_p(A), lda, _p(B), ldb, m, n, k, _p(train_p), _p(train_i), _p(test_p), _p(test_i), _p(test_v),
            k_metrics, int(bool(cumulative)), int(bool(break_ties_with_noise)), *[_p(o) for o in outs],
            int(bool(consider_cold_start)), min_items_pool, min_pos_test, nthreads, seed)
    if rc:
        _raise(lib, rc)
    if cumulative:
        outs = [(o.reshape((m, k_metrics)) if o.size else o.reshape((0, 0))) if i < 8 else o for i, o in enumerate(outs)]
    return tuple(outs)


def calc_metrics_device(dtype, A, lda, B, ldb, m, n, k, train_p, train_i, nnz_train, test_p, test_i, test_v, nnz_test,
                        k_metrics, outs, cumulative=False, break_ties_with_noise=False, consider_cold_start=True,
                        min_items_pool=2, min_pos_test=1, seed=1, stream=0):
    """Device-pointer entry: every array argument is an integer device address (0 == NULL); `outs` is a sequence
    of 10 addresses in METRIC_ORDER.  Asynchronous on `stream` apart from one small plan read-back."""
    lib = load()
    fn = getattr(lib, "rm_calc_metrics_dev_" + _suffix(dtype))

    def vp(x):
        return C.c_void_p(int(x)) if x else None
    rc = fn(vp(A), lda, vp(B), ldb, m, n, k, vp(train_p), vp(train_i), nnz_train, vp(test_p), vp(test_i), vp(test_v), nnz_test,
            k_metrics, int(bool(cumulative)), int(bool(break_ties_with_noise)), *[vp(o) for o in outs],
            int(bool(consider_cold_start)), min_items_pool, min_pos_test, seed, vp(stream))
    if rc:
        _raise(lib, rc)


def rank(A, B, train_p, train_i, test_p, test_i, k_metrics, break_ties_with_noise=False, consider_cold_start=True,
         min_items_pool=2, min_pos_test=1, seed=1):
    lib = load()
    dtype = A.dtype.type
    m, k = A.shape
    n = B.shape[0]
    idx = np.empty((m, k_metrics), dtype=np.int32)
    sc = np.empty((m, k_metrics), dtype=dtype)
    pr = np.zeros(max(int(test_i.shape[0]), 1), dtype=np.int64)
    st = np.empty(m, dtype=np.int32)
    fn = getattr(lib, "rm_rank_" + _suffix(dtype))
    A = np.ascontiguousarray(A)                  # the raw pointer is passed with lda = k: rows must be dense
    B = np.ascontiguousarray(B)
    rc = fn(_p(A), A.shape[1], _p(B), B.shape[1], m, n, k, _p(train_p), _p(train_i), _p(test_p), _p(test_i),
            k_metrics, int(bool(break_ties_with_noise)), int(bool(consider_cold_start)), min_items_pool, min_pos_test, seed,
            _p(idx), _p(sc), _p(pr), _p(st))
    if rc:
        _raise(lib, rc)
    return {"topk_idx": idx, "topk_score": sc, "pos_rank": pr[:test_i.shape[0]], "status": st}


def debug_scores(A, B):
    lib = load()
    dtype = A.dtype.type
    A = np.ascontiguousarray(A)
    B = np.ascontiguousarray(B)
    out = np.empty((A.shape[0], B.shape[0]), dtype=dtype)
    fn = getattr(lib, "rm_debug_scores_" + _suffix(dtype))
    rc = fn(_p(A), A.shape[1], _p(B), B.shape[1], A.shape[0], B.shape[0], A.shape[1], _p(out))
    if rc:
        _raise(lib, rc)
    return out


def split_csr(indptr, indices, data, n_items, mode, n_users_test=0, test_fraction=0.3, consider_cold_start=False,
              min_items_pool=2, min_pos_test=1, seed=1):
    """Host-side split (rm_split_*): returns a dict of raw arrays -- "train" / "test" / "rem" as (indptr, indices, data)
    tuples and "users_test".  mode: 0 every row, 1 separated, 2 joined (reference wrapper.pyx:523-818)."""
    lib = _load_split()
    dtype = data.dtype.type
    fn = lib.rm_split_f32 if dtype == np.float32 else lib.rm_split_f64
    handle = C.c_void_p()
    m = indptr.shape[0] - 1
    rc = fn(_p(indptr), _p(indices), _p(data), m, n_items, mode, n_users_test, float(test_fraction),
            int(bool(consider_cold_start)), min_items_pool, min_pos_test, seed, C.byref(handle))
    if rc:
        msg = (lib.rm_split_last_error() or b"").decode(errors="replace")
        if rc == 3:
            raise MemoryError(msg)
        raise RuntimeError(msg)
    try:
        def arr(which, dt):
            cnt = lib.rm_split_size(handle, which)
            out = np.empty(max(cnt, 0), dtype=dt)
            if cnt > 0:
                lib.rm_split_copy(handle, which, out.ctypes.data_as(C.c_void_p))
            return out
        res = {}
        for name, base in (("train", 0), ("test", 3), ("rem", 6)):
            res[name] = (arr(base, np.int32), arr(base + 1, np.int32), arr(base + 2, dtype))
        res["users_test"] = arr(9, np.int32)
        return res
    finally:
        lib.rm_split_free(handle)
