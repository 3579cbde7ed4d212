#!/usr/bin/env python3
"""Generates tests/golden/split/*.npz -- inputs + outputs of the REAL reference's split functions
(split_data_selected_users / separate_users / joined_users, reference src/recometrics.hpp:1015-1505), reached through
oracle/ref_split_shim.cpp compiled with the reference's own source into oracle/_ref/librecometrics_ref_split.so.
Run in the build container only:   python tests/golden/make_golden_split.py"""
import ctypes as C
import json
import os
import sys

import numpy as np
from scipy.sparse import random as sprand

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "librecometrics_ref_split.so"))
lib.ref_split_f64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_int32, C.c_double, C.c_int,
                              C.c_int32, C.c_int32, C.c_uint64, C.c_void_p]
lib.ref_split_copy.argtypes = [C.c_int, C.c_void_p]
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "split")
NAMES = ["train_p", "train_i", "train_v", "test_p", "test_i", "test_v", "rem_p", "rem_i", "rem_v", "users_test"]


def ref_split(X, mode, n_users_test=0, frac=0.3, cold=False, min_items_pool=2, min_pos_test=1, seed=1):
    p = X.indptr.astype(np.int32); i = X.indices.astype(np.int32); v = X.data.astype(np.float64)
    sizes = np.zeros(10, np.int64)
    rc = lib.ref_split_f64(p.ctypes.data, i.ctypes.data, v.ctypes.data, X.shape[0], X.shape[1], mode, n_users_test, frac,
                           int(cold), min_items_pool, min_pos_test, seed, sizes.ctypes.data)
    if rc:
        return None
    out = {}
    for w, nm in enumerate(NAMES):
        arr = np.empty(int(sizes[w]), dtype=np.float64 if nm.endswith("_v") else np.int32)
        if arr.size:
            lib.ref_split_copy(w, arr.ctypes.data)
        out[nm] = arr
    return out


def main():
    cases = []
    rng = np.random.default_rng(7)
    for ci, (m, n, dens) in enumerate([(60, 40, 0.2), (200, 90, 0.08), (35, 500, 0.05), (300, 25, 0.3)]):
        X = sprand(m, n, density=dens, format="csr", random_state=100 + ci, dtype=np.float64)
        X.data = np.round(X.data * 9 + 1)
        X.sort_indices()
        # a few empty and a few nearly full rows
        variants = []
        for seed in (1, 12345):
            for frac in (0.3, 0.5, 0.77):
                variants.append(dict(mode=0, frac=frac, seed=seed))
            for mode in (1, 2):
                for (nut, frac, cold, mip, mpt) in ((max(2, m // 10), 0.3, False, 2, 1), (m // 2, 0.5, True, 5, 2), (m, 0.2, False, 2, 1)):
                    variants.append(dict(mode=mode, n_users_test=nut, frac=frac, cold=cold, min_items_pool=mip, min_pos_test=mpt, seed=seed))
        store = {"X_p": X.indptr.astype(np.int32), "X_i": X.indices.astype(np.int32), "X_v": X.data.astype(np.float64),
                 "shape": np.array([m, n], np.int32)}
        kept = []
        for vi, kw in enumerate(variants):
            res = ref_split(X, **kw)
            kw = dict(kw); kw["raised"] = res is None
            if res is not None:
                for nm, arr in res.items():
                    store["v%d__%s" % (vi, nm)] = arr
            kept.append(kw)
        store["meta"] = np.frombuffer(json.dumps(kept).encode(), dtype=np.uint8)
        path = os.path.join(OUT, "split_%d.npz" % ci)
        np.savez_compressed(path, **store)
        print(path, os.path.getsize(path) // 1024, "KB", len(kept), "variants", sum(k["raised"] for k in kept), "raised")


if __name__ == "__main__":
    main()
