#!/usr/bin/env python3
"""Generates tests/golden/api/*.npz + api.json -- G9 fixtures: what the REFERENCE's Python API returns.

Build-container only: imports the reference package (copied to /tmp and built in place with its own setup.py, SURVEY.md
Appendix B) and captures, for a dozen keyword combinations of recometrics.calc_reco_metrics (reference
recometrics/__init__.py:44-628): the `as_df=False` dict (keys, shapes, dtypes, values), the DataFrame's columns and
dtypes, and the warnings / exceptions raised.  Only inputs and outputs are stored.

The factors are 20-bit dyadic numbers: every dot product is exact in any summation order, so the reference's default
-march=native build (vectorised, reassociated sums) and the canonical build give the same scores, and the stored values can
be compared bit for bit (ROC-AUC apart).

    cp -r /root/reference /tmp/pyref && (cd /tmp/pyref && python3 setup.py build_ext --inplace)
    python tests/golden/make_golden_api.py /tmp/pyref
"""
import json
import os
import sys
import warnings

import numpy as np
from scipy.sparse import csr_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "api")


def problem(m, n, k, dtype, seed):
    from recometrics_amd.synth import make_interactions
    rng = np.random.default_rng(seed)
    A = (rng.integers(-512, 513, (m, k)) / 1024.0).astype(dtype)
    B = (rng.integers(-512, 513, (n, k)) / 1024.0 + rng.integers(0, 2, (n, k)) / 65536.0).astype(dtype)
    trp, tri, tep, tei, tev = make_interactions(m, n, 30, dtype, seed)
    Xtr = csr_matrix((np.ones(tri.shape[0], dtype), tri, trp), shape=(m, n))
    Xte = csr_matrix((tev, tei, tep), shape=(m, n))
    return A, B, Xtr, Xte


CASES = [
    # name, problem kwargs, call kwargs
    ("defaults_f32", dict(dtype="float32"), dict()),
    ("defaults_f64", dict(dtype="float64"), dict()),
    ("all_metrics_k7", dict(dtype="float32"), dict(k=7, all_metrics=True)),
    ("cumulative_k4", dict(dtype="float64"), dict(k=4, cumulative=True, hit=True, rr=True, recall=True)),
    ("no_rename", dict(dtype="float32"), dict(k=3, rename_k=False, roc_auc=True, pr_auc=True)),
    ("only_auc", dict(dtype="float64"), dict(precision=False, average_precision=False, ndcg=False, roc_auc=True, pr_auc=True)),
    ("noise_off_seed", dict(dtype="float32"), dict(k=6, break_ties_with_noise=False, trunc_precision=True, trunc_average_precision=True)),
    ("item_biases", dict(dtype="float32"), dict(k=5, item_biases="linspace")),
    ("min_pos_pool_cold", dict(dtype="float64"), dict(k=5, min_pos_test=3, min_items_pool=50, consider_cold_start=False)),
    ("more_users_in_A", dict(dtype="float32", extra_users=3), dict(k=5)),                      # warning: 'A' has more users ...
    ("mixed_dtypes", dict(dtype="float32", b_dtype="float64"), dict(k=5)),                      # float32 only if both are
    ("fortran_B", dict(dtype="float64", fortran_b=True), dict(k=5, roc_auc=True)),
    ("k_too_large", dict(dtype="float32"), dict(k=10_000)),                                     # ValueError
    ("negative_min_pos", dict(dtype="float32"), dict(min_pos_test=-1)),
]


def main():
    ref_dir = sys.argv[1] if len(sys.argv) > 1 else "/tmp/pyref"
    sys.path.insert(0, ref_dir)
    import recometrics as ref
    assert os.path.abspath(ref.__file__).startswith(os.path.abspath(ref_dir)), ref.__file__
    os.makedirs(OUT, exist_ok=True)
    index = []
    for ci, (name, pk, ck) in enumerate(CASES):
        dtype = np.dtype(pk["dtype"]).type
        A, B, Xtr, Xte = problem(150, 700, 12, dtype, 500 + ci)
        if pk.get("extra_users"):
            A = np.r_[A, A[:pk["extra_users"]]]
        if pk.get("b_dtype"):
            B = B.astype(pk["b_dtype"])
        if pk.get("fortran_b"):
            B = np.asfortranarray(B)
        ck = dict(ck)
        if ck.get("item_biases") == "linspace":
            ck["item_biases"] = (np.arange(700) % 64 / 64.0 - 0.5).astype(dtype)
        store = dict(A=A, B=B, trp=Xtr.indptr, tri=Xtr.indices, trv=Xtr.data, tep=Xte.indptr, tei=Xte.indices, tev=Xte.data)
        entry = dict(name=name, shape=[int(Xtr.shape[0]), int(Xtr.shape[1])], kwargs={k: v for k, v in ck.items() if k != "item_biases"},
                     has_item_biases="item_biases" in ck, fortran_b=bool(pk.get("fortran_b")))
        if "item_biases" in ck:
            store["item_biases"] = ck["item_biases"]
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            try:
                d = ref.calc_reco_metrics(Xtr.copy(), Xte.copy(), A, B, as_df=False, **ck)
                df = ref.calc_reco_metrics(Xtr.copy(), Xte.copy(), A, B, as_df=True, **ck)
                entry["error"] = None
            except Exception as e:      # noqa: BLE001
                entry["error"] = [type(e).__name__, str(e)]
                d, df = None, None
        entry["warnings"] = sorted({str(w.message) for w in wlist})
        if d is not None:
            entry["dict_keys"] = list(d.keys())
            entry["K"] = int(d["K"])
            for key, val in d.items():
                if key != "K":
                    store["out__" + key] = np.asarray(val)
            entry["df_columns"] = [str(c) for c in df.columns]
            entry["df_dtypes"] = [str(t) for t in df.dtypes]
            entry["df_shape"] = list(df.shape)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **store)
        index.append(entry)
        print("%-22s error=%s warnings=%d keys=%s" % (name, entry["error"], len(entry["warnings"]), entry.get("dict_keys")))
    json.dump(index, open(os.path.join(OUT, "api.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
