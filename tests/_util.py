"""Shared helpers for the tests (golden loading, bitwise comparison)."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    dtype = np.dtype(meta["dtype"]).type
    inputs = dict(A=z["A"], B=z["B"], train=(z["train_p"], z["train_i"]), test=(z["test_p"], z["test_i"], z["test_v"]))
    variants = []
    for vi, kw in enumerate(meta["variants"]):
        kw = dict(kw)
        kw["metrics"] = tuple(kw["metrics"])
        expected = {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith("v%d__" % vi)}
        variants.append((kw, expected))
    return dtype, inputs, variants


def bits(x):
    return x.view(np.uint32 if x.dtype == np.float32 else np.uint64)


def same_bits(x, y):
    """elementwise: identical bit patterns, or both NaN"""
    return (bits(np.ascontiguousarray(x)) == bits(np.ascontiguousarray(y))) | (np.isnan(x) & np.isnan(y))


def assert_same_bits(got, want, what=""):
    assert got.shape == want.shape, (what, got.shape, want.shape)
    ok = same_bits(got, want)
    if not ok.all():
        bad = np.argwhere(~ok)[:5]
        raise AssertionError("%s: %d of %d differ, e.g. %s" % (
            what, (~ok).sum(), ok.size, [(tuple(i), got[tuple(i)], want[tuple(i)]) for i in bad]))


def assert_close(got, want, tol, what=""):
    assert got.shape == want.shape, (what, got.shape, want.shape)
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert (nan_g == nan_w).all(), "%s: NaN pattern differs at %s" % (what, np.argwhere(nan_g != nan_w)[:5].tolist())
    w64 = np.where(nan_w, 0, want).astype(np.float64)
    d = np.abs(np.where(nan_g, 0, got).astype(np.float64) - w64)
    # (relative beyond magnitude 1: a user whose test items are all train items gets a meaningless ROC-AUC of ~ -2e14 from the
    # reference's formula -- reproduced to the last bit or two, which is far more than `tol` in absolute terms)
    d = d / np.maximum(1.0, np.abs(w64))
    assert d.max(initial=0) <= tol, "%s: max diff %g > %g at %s" % (what, d.max(), tol, np.unravel_index(d.argmax(), d.shape))
