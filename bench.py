#!/usr/bin/env python3
"""bench.py -- users/sec evaluated (all metrics, K=10) on N MI355X, next to the CPU reference on the host cores.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|NS|C1|...] [--users M]

One "step" = one pass of the hot path (rm_calc_metrics_dev_f32: plan + pack + positives + sweep + finalize, all ten
metrics) over one batch of synthetic users whose inputs are already resident in HBM.  N > 1 (launched by
torch.distributed.run, one rank per GPU): users are sharded, every rank holds a replica of the item factors and
evaluates its own shard of the same size (weak scaling); the only exchange is one RCCL all-gather of the per-user
metric block per step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, chip table: dense f32-input MFMA
PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
BARE_CHAIN_TFLOPS = {"f32": 148.4, "f64": 65.6}      # profiles/r4_peak_f64_mfma.txt: v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 alone


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--users", type=int, default=0, help="override the number of users per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the north-star-shape extra measurement")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = every GPU evaluates a full-size shard; strong = the workload's users are split over the GPUs")
    ap.add_argument("--parity-users", type=int, default=2048, help="users of the timed outputs compared with the reference (0 = skip)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-pointer (PCIe-inclusive) measurement")
    ap.add_argument("--e2e-child", action="store_true", help=argparse.SUPPRESS)      # internal: the host-pointer leg in a process of its own
    ap.add_argument("--e2e-only", action="store_true", help=argparse.SUPPRESS)       # internal: the child measures the host-pointer call alone (the large shapes)
    ap.add_argument("--sharded-child", type=int, default=0, help=argparse.SUPPRESS)  # internal: rm_set_devices([0..N-1]) against the unsharded call
    ap.add_argument("--no-other", action="store_true", help="skip the compact legs over the other BASELINE configs (C3, C4, C5)")
    return ap.parse_args()


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: start torch.distributed.run as a CHILD before anything in
    this process touches the GPU, and leave with its exit code (never exec from a GPU-initialised process)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def host_problem(m, n, k, mean_c, seed, dtype=np.float32, shard=0):
    """The synthetic workload as host arrays (the same for the device-resident problem and for the host-pointer leg)."""
    from recometrics_amd.synth import make_factors, make_interactions
    if (n * k) >> 28:                                                    # C4's 10M x 128: drawn in row chunks (bounded host memory)
        rng = np.random.default_rng(seed)
        B = np.empty((n, k), dtype=dtype)
        sc = np.float32(1.0 / np.sqrt(k))
        for r0 in range(0, n, 1 << 20):
            r1 = min(n, r0 + (1 << 20))
            B[r0:r1] = rng.standard_normal((r1 - r0, k), dtype=np.float32) * sc
    else:
        _, B = make_factors(1, n, k, dtype, seed)                        # item factors: the same replica on every rank
    A, _ = make_factors(m, 1, k, dtype, seed + 1 + 1000 * shard)         # this rank's user shard
    trp, tri, tep, tei, tev = make_interactions(m, n, mean_c, dtype, seed + 1000 * shard)
    return dict(A=A, B=B, train=(trp, tri), test=(tep, tei, tev))


class DeviceProblem:
    """Synthetic workload resident in HBM (torch tensors are only the memory owner; the hot path gets raw pointers)."""

    def __init__(self, torch, dev, m, n, k, mean_c, seed, K, dtype=np.float32, shard=0, cumulative=False, host=None, metrics=None):
        self.m, self.n, self.k, self.K, self.dtype = m, n, k, K, dtype
        self.metrics = tuple(metrics) if metrics else None                   # None = all ten; else the names asked for (_binding.METRIC_ORDER's spelling)
        self.host = host if host is not None else host_problem(m, n, k, mean_c, seed, dtype, shard)
        A, B = self.host["A"], self.host["B"]
        (trp, tri), (tep, tei, tev) = self.host["train"], self.host["test"]
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
        self.A, self.B = t(A), t(B)
        self.trp, self.tri, self.tep, self.tei, self.tev = t(trp), t(tri if tri.size else np.zeros(1, np.int32)), t(tep), t(tei), t(tev)
        self.nnz_tr, self.nnz_te = int(tri.shape[0]), int(tei.shape[0])
        self.cumulative = bool(cumulative)
        self.noise = False                                                   # break_ties_with_noise of step() unless it says otherwise
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        from recometrics_amd._binding import METRIC_ORDER as _ALL
        asked = [self.metrics is None or name in self.metrics for name in _ALL]
        if not self.cumulative:
            self.out = torch.empty((10, m), dtype=tdt, device=dev)           # per-user metric block
            self.out_ptrs = lambda o: [o[i].data_ptr() if asked[i] else 0 for i in range(10)]     # noqa: E731  (0 = not requested)
        else:                                                                # eight [m, K] blocks, then ROC / PR-AUC [m]
            self.out = torch.empty((8 * K + 2) * m, dtype=tdt, device=dev)
            es = self.out.element_size()
            self.out_ptrs = lambda o: [o.data_ptr() + es * (i * K * m if i < 8 else 8 * K * m + (i - 8) * m) for i in range(10)]   # noqa: E731

    def metric(self, o, i):
        """metric i of the output block `o` as a [m] / [m, K] tensor"""
        if not self.cumulative:
            return o[i]
        K, m = self.K, self.m
        return o[i * K * m:(i + 1) * K * m].view(m, K) if i < 8 else o[8 * K * m + (i - 8) * m:8 * K * m + (i - 7) * m]

    def step(self, binding, stream, out=None, noise=None):
        o = self.out if out is None else out
        noise = self.noise if noise is None else noise
        binding.calc_metrics_device(
            self.dtype, self.A.data_ptr(), self.k, self.B.data_ptr(), self.k, self.m, self.n, self.k,
            self.trp.data_ptr(), self.tri.data_ptr(), self.nnz_tr, self.tep.data_ptr(), self.tei.data_ptr(),
            self.tev.data_ptr(), self.nnz_te, self.K, self.out_ptrs(o),
            cumulative=self.cumulative, break_ties_with_noise=noise, stream=stream)


def load_traffic(workload, users):
    """HBM bytes per sweep launch from the committed PMC run (scratch/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 passes, FETCH_SIZE doubled per the gfx950 correction) and the file it was read from -- a constant of
    that profile, not a counter of this run; (None, None) when no matching profile exists."""
    for rnd in ("r6", "r5", "r4", "r3", "r2"):                      # the newest committed profile of this workload and user count
        rel = os.path.join("profiles", "%s_traffic_%s.json" % (rnd, workload))
        try:
            d = json.load(open(os.path.join(ROOT, rel)))
            if int(d.get("users", -1)) == int(users):
                return d["hbm_bytes"], rel
        except Exception:      # noqa: BLE001
            pass
    return None, None


def stratified_users(host, n_users, seed=0):
    """A sample of users that covers the code paths the timed launch takes: the heaviest test rows (streamed users: more than
    63 test items), users without train items (cold start), users without test items (skipped), the first and the last user
    block of the launch order, and a random remainder.  Sorted, unique."""
    trp, tep = host["train"][0], host["test"][0]
    m = len(tep) - 1
    if n_users >= m:
        return np.arange(m)
    ntest, ntrain = np.diff(tep), np.diff(trp)
    q = max(1, n_users // 8)
    rng = np.random.default_rng(seed)
    pick = [np.argsort(-ntest, kind="stable")[:q],                               # heaviest rows
            np.flatnonzero(ntest > 63)[:q], np.flatnonzero(ntrain == 0)[:q], np.flatnonzero(ntest == 0)[:q],
            np.flatnonzero((ntest > 31) & (ntest <= 63))[:q],                       # deepest LDS tables
            np.arange(min(q, m)), np.arange(max(0, m - q), m)]
    got = np.unique(np.concatenate(pick))
    rest = np.setdiff1d(np.arange(m), got)
    extra = rng.choice(rest, size=max(0, min(rest.shape[0], n_users - got.shape[0])), replace=False)
    return np.unique(np.concatenate([got, extra]))


def sub_problem(host, users):
    """Rows `users` of the host copy of the workload as a problem of its own (CSR rows gathered, pointers rebuilt)."""
    trp, tri = host["train"]
    tep, tei, tev = host["test"]

    def gather(p, *arrs):
        cnt = (p[users + 1] - p[users]).astype(np.int64)
        newp = np.zeros(users.shape[0] + 1, np.int64)
        np.cumsum(cnt, out=newp[1:])
        idx = np.repeat(p[users].astype(np.int64) - newp[:-1], cnt) + np.arange(newp[-1])
        return (newp.astype(np.int32),) + tuple(a[idx] for a in arrs)
    ntrp, ntri = gather(trp, tri)
    ntep, ntei, ntev = gather(tep, tei, tev)
    if ntri.shape[0] == 0:
        ntri = np.zeros(1, np.int32)
    return host["A"][users], host["B"], (ntrp, ntri), (ntep, ntei, ntev)


def masked_problem(host, users):
    """The WHOLE workload with the test rows of every user outside `users` emptied: the reference skips a user without test items at
    once (src/recometrics.hpp:439-448), so it only evaluates the sample -- and every sampled user keeps its ORIGINAL index, which is
    what seeds its tie noise there (mt19937(seed + user), :531).  Train rows and user factors are passed whole (never copied)."""
    tep, tei, tev = host["test"]
    m = len(tep) - 1
    keep = np.zeros(m, bool)
    keep[users] = True
    cnt = np.where(keep, np.diff(tep), 0).astype(np.int64)
    newp = np.zeros(m + 1, np.int64)
    np.cumsum(cnt, out=newp[1:])
    idx = np.repeat(tep[:-1].astype(np.int64)[keep] - newp[:-1][keep], cnt[keep]) + np.arange(newp[-1])
    tri = host["train"][1]
    return host["A"], host["B"], (host["train"][0], tri if tri.shape[0] else np.zeros(1, np.int32)), (newp.astype(np.int32), tei[idx], tev[idx])


def parity_check(prob, out, n_users, noise=False, seed=1, cpu_seconds=15.0, binding=None):
    """verify parity on the same inputs in the same run before accepting a number -- a stratified sample of
    the users of the timed outputs (`out`: the metric block on the device) against the REAL reference compiled by
    oracle/Makefile (oracle/_ref, canonical build) when it is present, else against the restatement: metrics within 1e-5,
    identical NaN pattern.  With `noise` the reference draws its mt19937(seed + user) tie noise per ORIGINAL user index: it is
    handed the whole workload with the test rows of the users outside the sample emptied (`masked_problem`), so the same
    stratified sample keeps its indices.
    Every sampled user whose metrics differ from the reference's in any BIT (ROC-AUC apart: x87 long double there) must have
    an exactly tied score on one of its positives (deviation D4: the reference leaves tied scores in libstdc++'s order,
    oracle/ties.py) -- `tie_users` counts them, a differing user without such a tie fails the check."""
    from oracle import oracle as orc
    from oracle.ties import tie_pairs_per_user
    host = prob.host
    # a bounded amount of CPU work: the reference evaluates ~7e9 (item x factor) products per second on the box's cores
    # (4,300 users/s at C2), so ~15 s allow 1e11 / (n k) users -- 2,048 at C2 and C3, ~800 at the north-star shape, ~90 at C4
    n_users = int(max(16, min(n_users, cpu_seconds * 7.0e9 / (float(prob.n) * float(prob.k)))))
    users = stratified_users(host, n_users)
    # (with the tie noise the reference must see every sampled user under its own index: the whole workload, the other users' test rows emptied)
    A, B, tr, te = masked_problem(host, users) if noise else sub_problem(host, users)
    impl, kind = (orc.Reference(), "reference") if orc.reference_available() else (orc.Oracle(), "port")
    # (the reference addresses its per-thread scratch as thread * n in int32, src/recometrics.hpp:499: at n = 10M more than 214
    # threads overflow it)
    nthreads = max(1, min(256, os.cpu_count() or 1, (2 ** 31 - 1) // int(B.shape[0])))
    asked = getattr(prob, "metrics", None) or orc.METRICS
    want = impl.calc(A, B, tr, te, prob.K, metrics=tuple(asked), nthreads=nthreads, noise=noise, seed=seed, dtype=prob.dtype, cumulative=prob.cumulative)
    uidx = torch_index(out, users)
    if noise:                                                    # from here on the sample as a problem of its own, like the noise-free case
        want = {name: arr[users] for name, arr in want.items()}
        A, B, tr, te = sub_problem(host, users)
    info = {"users": int(users.shape[0]), "checker": kind, "streamed_users": int((np.diff(te[0]) > 63).sum()),
            "cold_users": int((np.diff(tr[0]) == 0).sum()), "sample": "stratified" + (" (original user indices: the reference's noise seeds)" if noise else "")}
    worst = 0.0
    differing = np.zeros(users.shape[0], bool)
    for i, name in enumerate(orc.METRICS):
        if name not in asked:
            continue
        w = want[orc.NAMES[name]]
        g = prob.metric(out, i)[uidx].cpu().numpy()
        if not (np.isnan(w) == np.isnan(g)).all():
            return dict(info, ok=False, what="NaN pattern of %s" % name)
        d = float(np.nanmax(np.abs(w.astype(np.float64) - g.astype(np.float64)), initial=0.0))
        worst = max(worst, d)
        if d > 1e-5:
            return dict(info, ok=False, what="%s differs by %g" % (name, d))
        if name != "roc":
            bits = np.uint32 if g.dtype == np.float32 else np.uint64
            same = (np.ascontiguousarray(g).view(bits) == np.ascontiguousarray(w).view(bits)) | (np.isnan(g) & np.isnan(w))
            differing |= ~same.reshape(users.shape[0], -1).all(axis=1)
    info["bitwise_differing_users"] = int(differing.sum())
    if differing.any() and binding is not None and kind == "reference":
        who = np.flatnonzero(differing)[:128]                  # (a dense score row per user: bounded)
        sc = binding.debug_scores(np.ascontiguousarray(A[who]), np.ascontiguousarray(B))
        pairs = tie_pairs_per_user(sc, tr, te, who, noise_zone=(2.0 ** -14 if noise and prob.dtype == np.float32 else None))
        info["tie_users_checked"] = int(who.shape[0])
        info["tie_users"] = int((pairs > 0).sum())
        if (pairs == 0).any():
            return dict(info, ok=False, what="users %s differ from the reference in a bit without an exact tie on a positive" % users[who[pairs == 0]][:5].tolist())
    else:
        info["tie_users"] = 0
    return dict(info, ok=True, max_abs_diff=worst)


def default_build_check(prob, out, n_users, binding, cpu_seconds=10.0):
    """SURVEY.md 8(c) contract item (4): the timed outputs against the reference's DEFAULT build -- vectorised, reassociated dot
    products (oracle/_ref/librecometrics_ref_fast.so: -march=x86-64-v3, AVX2 + FMA) -- on the stratified sample: every metric
    within 1e-5 except for users with a near tie, and every user beyond 1e-5 must be one (oracle/ties.py)."""
    from oracle import oracle as orc
    from oracle.ties import compare_with_default_build
    if not orc.reference_available(fast=True):
        return {"ok": None, "what": "oracle/_ref/librecometrics_ref_fast.so is not on this box"}
    host = prob.host
    n_users = int(max(16, min(n_users, cpu_seconds * 1.0e10 / (float(prob.n) * float(prob.k)))))
    users = stratified_users(host, n_users)
    A, B, tr, te = sub_problem(host, users)
    A = np.ascontiguousarray(A)
    nthreads = max(1, min(256, os.cpu_count() or 1, (2 ** 31 - 1) // int(B.shape[0])))
    want = orc.Reference(fast=True).calc(A, B, tr, te, prob.K, nthreads=nthreads, noise=False, dtype=prob.dtype, cumulative=prob.cumulative)
    uidx = torch_index(out, users)
    got = {orc.NAMES[name]: prob.metric(out, i)[uidx].cpu().numpy() for i, name in enumerate(orc.METRICS)}
    amax = float(np.abs(A).max() * np.abs(B).max())
    rec = compare_with_default_build(got, want, lambda who: binding.debug_scores(np.ascontiguousarray(A[who]), B), tr, te, prob.K, prob.dtype,
                                     prob.k, amax)
    rec["checker"] = "reference, default-style build (-march=x86-64-v3: vectorised dot products)"
    rec["sample"] = "stratified"
    return rec


def torch_index(out, users):
    import torch
    return torch.from_numpy(users.astype(np.int64)).to(out.device)


def e2e_host_measure(binding, host, k, K, dtype, reps=5, cumulative=False):
    """host arrays in -> host arrays out through rm_calc_metrics_* (H2D of A/B/CSR, device work, D2H
    of the metric block), first call (workspace allocation) and steady state (median)."""
    trp, tri = host["train"]
    tep, tei, tev = host["test"]
    want = {name: True for name in binding.METRIC_ORDER}
    m = host["A"].shape[0]

    def call():
        t0 = time.perf_counter()
        binding.calc_metrics(host["A"], k, host["B"], k, trp, tri if tri.size else np.zeros(1, np.int32), tep, tei, tev,
                             K, want, cumulative, False, True, 2, 1, 1, 1)
        return (time.perf_counter() - t0) * 1e3
    first = call()
    call()                                                     # (the second call still builds things: the peer context of the batch pipeline, its workspace)
    rest = sorted(call() for _ in range(reps + 2))
    steady = rest[len(rest) // 2]
    bytes_in = sum(int(x.nbytes) for x in (host["A"], host["B"], trp, tri, tep, tei, tev))
    return {"first_call_ms": first, "steady_ms": steady, "users_per_s": m / (steady * 1e-3), "bytes_in": bytes_in,
            "what": "rm_calc_metrics_%s: host pointers in, host pointers out (PCIe-inclusive), in a process of its own WITHOUT torch -- as the "
                    "Cython / Rcpp wrappers would call it (torch's bundled HIP runtime moves pageable memory 3-4x slower than the system's); "
                    "never `value`" % ("f32" if dtype == np.float32 else "f64")}


def api_default_measure(host, k, K, dtype, reps=5):
    """The call a user makes: recometrics_amd.calc_reco_metrics(X_train, X_test, A, B, k=10, all_metrics=True, as_df=False)
    -- the API defaults otherwise: break_ties_with_noise=True, nthreads=-1 -- with SciPy CSR matrices in (reference
    recometrics/__init__.py:553-562).  Every repetition gets FRESH csr_array objects over the same buffers, so SciPy does not
    know whether their indices are sorted and the library's multi-threaded check runs each time (csrc/rm_csr.cpp)."""
    import scipy.sparse as sp
    import recometrics_amd
    trp, tri = host["train"]
    tep, tei, tev = host["test"]
    m, n = host["A"].shape[0], host["B"].shape[0]
    ones = np.ones(tri.shape[0], dtype)

    def fresh():
        return (sp.csr_array((ones, tri, trp), shape=(m, n), copy=False), sp.csr_array((tev, tei, tep), shape=(m, n), copy=False))

    def call(mats, as_df=False):
        t0 = time.perf_counter()
        out = recometrics_amd.calc_reco_metrics(mats[0], mats[1], host["A"], host["B"], k=K, all_metrics=True, as_df=as_df)
        return (time.perf_counter() - t0) * 1e3, out
    first, _ = call(fresh())
    call(fresh())
    times = sorted(call(fresh())[0] for _ in range(reps + 2))
    mats = fresh()
    call(mats)
    same = sorted(call(mats)[0] for _ in range(reps))          # the same objects again: SciPy's flag is set, no check at all
    call(fresh(), True)
    frame = sorted(call(fresh(), True)[0] for _ in range(reps))  # the literal default: as_df=True (a pandas DataFrame over the outputs' own storage)
    t0 = time.perf_counter()
    xs = fresh()
    xs[0].sort_indices(); xs[1].sort_indices()
    scipy_ms = (time.perf_counter() - t0) * 1e3
    return {"first_call_ms": first, "ms": times[len(times) // 2], "same_objects_ms": same[len(same) // 2], "as_df_ms": frame[len(frame) // 2],
            "users_per_s": m / (times[len(times) // 2] * 1e-3), "scipy_sort_indices_ms": scipy_ms,
            "what": "recometrics_amd.calc_reco_metrics(X_train, X_test, A, B, k=%d, all_metrics=True, as_df=False): noise on, SciPy CSR in, "
                    "fresh matrix objects per call (sortedness unknown to SciPy), torch-free process; `scipy_sort_indices_ms` = what "
                    "SciPy's own single-threaded pass over the two matrices takes on this host (the reference's path); never `value`" % K}


def e2e_child_main(args):
    """`bench.py --e2e-child`: the host-pointer leg and the Python-API leg, no torch in the process; prints one JSON object."""
    from recometrics_amd import _binding as binding
    from recometrics_amd.synth import CONFIGS
    m, n, k, dtype, K, mean_c, seed = CONFIGS[args.workload]
    if args.workload == "C3":
        m = m // 8
    if args.users:
        m = args.users
    binding.load()
    host = host_problem(m, n, k, mean_c, seed, dtype)
    if args.e2e_only:                                          # (the large shapes: three repetitions of a call of a tenth of a second or more)
        print(json.dumps({"e2e_host": e2e_host_measure(binding, host, k, K, dtype, reps=1, cumulative=args.workload == "C3")}))
        return
    res = {"e2e_host": e2e_host_measure(binding, host, k, K, dtype)}
    try:
        res["api_default"] = api_default_measure(host, k, K, dtype)
    except Exception as e:      # noqa: BLE001
        res["api_default"] = {"error": repr(e)}
    print(json.dumps(res))


def sharded_child_main(args):
    """`bench.py --sharded-child N`: the workload through rm_set_devices([0 .. N-1]) in ONE process (one host thread, stream and
    workspace per device, item factors fanned out device to device) against the unsharded call on device 0: every output
    array must be bit-identical.  No torch in the process."""
    from recometrics_amd import _binding as binding
    from recometrics_amd.synth import CONFIGS
    m, n, k, dtype, K, mean_c, seed = CONFIGS[args.workload]
    if args.workload == "C3":
        m = m // 8
    if args.users:
        m = args.users
    binding.load()
    ndev = binding.device_count()
    devices = [i % max(ndev, 1) for i in range(args.sharded_child)]
    host = host_problem(m, n, k, mean_c, seed, dtype)
    trp, tri = host["train"]
    tep, tei, tev = host["test"]
    want = {name: True for name in binding.METRIC_ORDER}

    def call():
        t0 = time.perf_counter()
        o = binding.calc_metrics(host["A"], k, host["B"], k, trp, tri if tri.size else np.zeros(1, np.int32), tep, tei, tev,
                                 K, want, False, False, True, 2, 1, 1, 1)
        return (time.perf_counter() - t0) * 1e3, o
    binding.set_devices([0])
    call()
    t_one, base = call()
    binding.set_devices(devices)
    call()
    t_sh, got = call()
    binding.set_devices([])
    bits = np.uint32 if dtype == np.float32 else np.uint64
    same = all(((a.view(bits) == b.view(bits)) | (np.isnan(a) & np.isnan(b))).all() for a, b in zip(base, got))
    print(json.dumps({"devices": devices, "distinct_devices": len(set(devices)), "visible_devices": ndev, "users": m,
                      "bitwise_equal_to_unsharded": bool(same), "unsharded_ms": t_one, "sharded_ms": t_sh,
                      "what": "rm_calc_metrics_* (host pointers) with rm_set_devices(%s) against rm_set_devices([0]); second call of each" % devices}))


BIG_HOST_LEGS = (("NS", 32768), ("C3", 125000), ("C4", 8192), ("C5", 16384))


def host_leg(early, wname, step_ms):
    """the `e2e_host` record of a large shape next to its device-resident step: budget = 1.15 x step + input bytes at 45 GB/s"""
    child = early.get("big_" + wname)
    if not child:
        return None
    rec = dict(child.get("e2e_host", child))
    if "steady_ms" in rec and step_ms:
        rec["device_step_ms"] = step_ms
        rec["budget_ms"] = 1.15 * step_ms + rec.get("bytes_in", 0) / 45e9 * 1e3
        rec["within_budget"] = rec["steady_ms"] <= rec["budget_ms"]
    return rec


def run_child(args, m, extra, timeout=900, workload=None):
    """Runs a leg as a child process (started from this one, which keeps running: never an exec)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", workload or args.workload, "--users", str(m)] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if res.returncode != 0 or not lines:
        return {"error": "child failed (%d): %s" % (res.returncode, res.stderr[-400:])}
    return json.loads(lines[-1])


def cpu_baseline(host, K, n_users_total, budget_s, dtype=np.float32):
    """The CPU path on this host's cores over a bounded sample of the same workload."""
    from oracle import oracle as orc
    ncores = os.cpu_count() or 1
    # the reference addresses its per-thread scratch as `buffer + omp_get_thread_num() * n` in int32
    # (src/recometrics.hpp:499): more than (2^31 - 1) / n threads overflow it (214 at n = 10M)
    ncores = max(1, min(ncores, (2 ** 31 - 1) // int(host["B"].shape[0])))
    march = None
    if orc.reference_available(v4=True) and orc.host_has_avx512():
        impl, kind, march = orc.Reference(v4=True), "reference", "x86-64-v4 (AVX-512: what the reference's default -march=native gives on this host)"
    elif orc.reference_available(fast=True):
        impl, kind, march = orc.Reference(fast=True), "reference", "x86-64-v3 (its default user build is -march=native)"
    else:
        impl, kind = orc.Oracle(), "port"
    A, B = host["A"], host["B"]
    trp, tri = host["train"]
    tep, tei, tev = host["test"]

    def run(nu):
        sub_tr = (trp[:nu + 1], tri[:trp[nu]] if trp[nu] else np.zeros(1, np.int32))
        sub_te = (tep[:nu + 1], tei[:tep[nu]], tev[:tep[nu]])
        t0 = time.perf_counter()
        impl.calc(A[:nu], B, sub_tr, sub_te, K, nthreads=ncores, noise=False, dtype=dtype)
        return time.perf_counter() - t0

    # (first sample sized for ~2 s at the rate the reference sustains on 256 threads, ~7e9 item x factor products per second: at
    # n = 10M a fixed 2 x cores users would already take 100 s)
    probe = int(min(n_users_total, max(8, min(max(ncores * 2, 64), 1.5e10 / (float(B.shape[0]) * float(B.shape[1]))))))
    run(min(probe, 32))                                    # page in
    t_probe = run(probe)
    nu = int(min(n_users_total, max(probe, probe * budget_s / max(t_probe, 1e-6))))
    t = run(nu) if nu > probe else t_probe
    if t < 0.6 * budget_s and nu < n_users_total:          # the small probe overestimates the per-user cost (thread start-up): once more
        nu = int(min(n_users_total, nu * budget_s / max(t, 1e-6)))
        t = run(nu)
    how = ("the reference built by oracle/Makefile with -march=" + march) if kind == "reference" else "oracle/ restatement"
    return {"value": nu / t, "unit": "users/s", "cores": ncores, "kind": kind,
            "sample": "%d of %d users of the same workload, all metrics, K=%d, %d threads, %.1f s; %s" % (nu, n_users_total, K, ncores, t, how)}


def measure(torch, dist, binding, prob, steps, warmup, world, gather_buf):
    stream = torch.cuda.current_stream().cuda_stream
    sweep_ms, prep_ms, fin_ms = [], [], []
    # N > 1: the all-gather of step i's metric block runs on RCCL's stream while step i + 1 computes; two metric blocks and
    # two gather buffers in rotation, and a block is not written again before the gather that reads it has finished
    outs = [prob.out, torch.empty_like(prob.out)] if world > 1 else [prob.out]
    gbufs = [gather_buf, torch.empty_like(gather_buf)] if world > 1 else [None]
    pending = [None, None]
    count = [0]

    def one():
        i = count[0] % len(outs)
        count[0] += 1
        if pending[i] is not None:
            pending[i].wait()                               # stream-level: the compute stream waits, the host does not
            pending[i] = None
        prob.step(binding, stream, outs[i])
        if world > 1:
            pending[i] = dist.all_gather_into_tensor(gbufs[i], outs[i], async_op=True)

    def drain():
        for i, h in enumerate(pending):
            if h is not None:
                h.wait()
                pending[i] = None
    for _ in range(warmup):
        one()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
        tm = binding.timings()                              # HIP events recorded on the stream the kernels ran on
        sweep_ms.append(tm["sweep_ms"]); prep_ms.append(tm["prep_ms"]); fin_ms.append(tm["finalize_ms"])
    drain()                                                 # every step's gather is inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        last = (count[0] - 1) % len(outs)
        if last != 0:
            prob.out.copy_(outs[last])                      # parity_check reads prob.out
    measure.last_sweep_ms = [float(x) for x in sweep_ms]      # per-step launch durations of the run just timed (the compact legs quote them)
    return dt, float(np.mean(sweep_ms)), float(np.mean(prep_ms)), float(np.mean(fin_ms)), binding.timings()


def main():
    args = parse()
    if args.e2e_child:                                       # (before torch is imported: that is the point)
        e2e_child_main(args)
        return
    if args.sharded_child:
        sharded_child_main(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    # The host-pointer legs (`e2e_host`, `api_default`, the tutorial's API call) run FIRST, each in a torch-free child process, while
    # THIS process has not touched the GPU yet: with the parent's HIP context alive beside it (torch's streams and queues, idle but
    # mapped) the same child measured 0.5-0.8 ms more per call than alone (profiles/r5_host_entry.txt against the round's first bench
    # lines) -- two processes' hardware queues are time-sliced.  A wrapper's process is alone on its GPU; so is the child here.
    early = {}
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.no_e2e:
        from recometrics_amd.synth import CONFIGS as _CFG
        m0 = _CFG[args.workload][0] // (8 if args.workload == "C3" else 1)
        if args.users:
            m0 = args.users
        try:
            early["main"] = run_child(args, m0, ["--e2e-child"])
        except Exception as e:      # noqa: BLE001
            early["main"] = {"error": repr(e)}
        if not args.no_extra and args.workload != "TUT":
            try:
                early["TUT"] = run_child(args, _CFG["TUT"][0], ["--e2e-child"], workload="TUT")
            except Exception as e:      # noqa: BLE001
                early["TUT"] = {"error": repr(e)}
        # SURVEY.md 8(d)(i) -- host arrays in, host arrays out -- where the inputs are BIG: the north-star shape (512 MB of item factors),
        # C3's shard of one GPU (64 MB of user factors, 81 MB of outputs), the C4 and C5 slices of the compact legs (5.12 GB / 1.02 GB of
        # item factors).  Each call is held against 1.15 x the device-resident step of the same shape + its input bytes at 45 GB/s.
        if not args.no_extra and not args.no_other:
            for wname, mo in BIG_HOST_LEGS:
                if wname == args.workload:
                    continue
                try:
                    early["big_" + wname] = run_child(args, mo, ["--e2e-child", "--e2e-only"], workload=wname)
                except Exception as e:      # noqa: BLE001
                    early["big_" + wname] = {"error": repr(e)}
    import torch
    import torch.distributed as dist
    from recometrics_amd import _binding as binding
    from recometrics_amd.synth import CONFIGS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RM_BENCH_BACKEND=gloo (tests on a one-GPU box): the ranks share the devices there are; RCCL wants one GPU per rank
    backend = os.environ.get("RM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend, rank=rank, world_size=world)
    # what a SCALE record can be checked against: the ranks the collective library really has (0 = no process group)
    rccl_ranks = dist.get_world_size() if (world > 1 and dist.is_initialized()) else 0
    comm_backend = dist.get_backend() if (world > 1 and dist.is_initialized()) else None
    torch.cuda.set_device(local_rank)
    binding.load()
    binding.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # A multi-GPU line is only worth something if the ranks really are N processes on N DISTINCT devices talking through RCCL:
    # every rank reports (host, device identity), and the run is refused otherwise (RM_BENCH_BACKEND=gloo, the tests' stand-in
    # on a one-GPU box, is reported as such and not refused)
    self_check = None
    if world > 1:
        import socket
        props = torch.cuda.get_device_properties(local_rank)
        ident = (socket.gethostname(), str(getattr(props, "uuid", "")), str(getattr(props, "pci_bus_id", "")), local_rank)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        distinct_hw = len({(h, u, b) for h, u, b, i in idents})       # by hardware identity (uuid / PCI bus id), where the runtime reports one
        distinct_idx = len({(h, i) for h, _, _, i in idents})          # by (host, device index)
        # (a runtime that reports the same placeholder identity for every device -- distinct_hw == 1 -- says nothing: the indices decide)
        hw_ok = distinct_hw == world or distinct_hw == 1
        ok = (rccl_ranks == args.gpus == world) and comm_backend == "nccl" and distinct_idx == world and hw_ok and torch.cuda.device_count() >= min(world, 8)
        self_check = {"ok": bool(ok), "rccl_ranks": rccl_ranks, "gpus_asked": args.gpus, "comm_backend": comm_backend,
                      "distinct_devices": distinct_idx if hw_ok else min(distinct_hw, distinct_idx), "distinct_hw_identities": distinct_hw,
                      "devices": ["%s:%d %s" % (h, i, b or u) for h, u, b, i in idents]}
        if not ok and backend == "nccl":
            if rank == 0:
                print(json.dumps({"error": "bench.py --gpus %d: not %d RCCL ranks on %d distinct devices" % (args.gpus, args.gpus, args.gpus),
                                  "multi_gpu_self_check": self_check}))
            dist.destroy_process_group()
            sys.exit(3)

    m, n, k, dtype, K, mean_c, seed = CONFIGS[args.workload]
    if args.workload == "C3":
        m = m // 8                                           # C3 is quoted user-sharded over 8 GPUs
    if args.users:
        m = args.users
    m_total = m
    if world > 1 and args.scaling == "strong":                # the workload's users split over the ranks (contiguous ranges)
        from recometrics_amd.sharding import user_range
        if args.workload == "C3" and not args.users:
            m_total = CONFIGS["C3"][0]
        lo, hi = user_range(m_total, world, rank)
        m = hi - lo
    prob = DeviceProblem(torch, dev, m, n, k, mean_c, seed, K, dtype, shard=rank)
    m_cap = m if args.scaling == "weak" else -(-m_total // world)       # all_gather_into_tensor wants equal shards: pad
    if world > 1 and m_cap != m:
        prob.out = torch.zeros((10, m_cap), dtype=prob.out.dtype, device=dev)
    gather_buf = torch.empty((world * 10, m_cap), dtype=prob.out.dtype, device=dev) if world > 1 else None
    peak = PEAK_FP32_MFMA_TFLOPS if dtype == np.float32 else PEAK_FP64_MFMA_TFLOPS
    esize = 4.0 if dtype == np.float32 else 8.0
    dname = "f32" if dtype == np.float32 else "f64"

    dt, sweep_ms, prep_ms, fin_ms, tm = measure(torch, dist, binding, prob, args.steps, args.warmup, world, gather_buf)
    users_done = (world * m) if args.scaling == "weak" else m_total
    users_per_s = users_done * args.steps / dt
    # 2*n*k per user x users of one launch (`share` = the slots the timed launch covers: all of them).
    share = (tm.get("timed_slots") or 0) / tm["total_slots"] if tm.get("total_slots") else 1.0
    flops_per_launch = 2.0 * n * k * m * share
    achieved_tf = flops_per_launch / (sweep_ms * 1e-3) / 1e12
    line = {
        "metric": "users/sec evaluated (all metrics, K=%d)" % K, "value": users_per_s, "unit": "users/s",
        "n_gpus": world, "rccl_ranks": rccl_ranks, "comm_backend": comm_backend,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": dname, "data": "synthetic",
        "config": {"workload": "%s: %d users/GPU x %d items, %d factors %s, K=%d, all 10 metrics, noise off"
                               % (args.workload, m, n, k, dname, K),
                   "users_per_gpu": m, "n_items": n, "n_factors": k, "k_metrics": K,
                   "sharding": "users sharded, item factors replicated, 1 all-gather of the metric block per step (overlapped with the next step)"},
        "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved_tf / peak, "traffic": load_traffic(args.workload, m)[0],
                     "traffic_source": ("%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of the same command, committed; a constant of that "
                                        "profile, not a counter of this run)" % load_traffic(args.workload, m)[1]) if load_traffic(args.workload, m)[1] else None,
                     "kernel": "k_sweep (main launch: %.1f %% of the users)" % (100 * share), "avg_launch_ms": sweep_ms, "flops_per_launch": flops_per_launch,
                     "hbm_equiv_GBs": n * k * esize * m * share / (sweep_ms * 1e-3) / 1e9,
                     "hbm_equiv_frac": n * k * esize * m * share / (sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                     # context, never `frac`: what a bare chain of this matrix instruction sustains on this part under its own power
                     # limit (a constant of a committed microbenchmark, not a measurement of this run)
                     "bare_mfma_chain_TFLOPs": BARE_CHAIN_TFLOPS[dname], "frac_of_bare_chain": achieved_tf / BARE_CHAIN_TFLOPS[dname],
                     "bare_chain_source": "profiles/r4_peak_f64_mfma.txt (scratch/peak_f64.hip on one MI355X)"},
        "stage_ms": {"prep": prep_ms, "sweep": sweep_ms, "finalize": fin_ms, "item_splits": tm.get("item_splits"),
                     "sweep_blocks": tm.get("sweep_blocks"), "lds_bytes": tm.get("lds_bytes")},
    }

    failed = False
    if rank == 0 and args.parity_users > 0:
        try:
            pc = parity_check(prob, prob.out, args.parity_users, binding=binding)
        except Exception as e:      # noqa: BLE001
            pc = {"users": 0, "ok": False, "what": repr(e)}
        line["parity_checked"] = pc["users"] if pc["ok"] else 0
        line["parity"] = pc
        failed = not pc["ok"]
        if world == 1 and not args.no_extra:
            try:
                line["parity_vs_default_build"] = default_build_check(prob, prob.out, args.parity_users, binding)
                failed = failed or line["parity_vs_default_build"].get("ok") is False
            except Exception as e:      # noqa: BLE001
                line["parity_vs_default_build"] = {"ok": False, "what": repr(e)}
                failed = True
    if rank == 0 and world == 1 and not args.no_e2e:
        child = early.get("main", {"error": "not run"})
        line["e2e_host"] = child.get("e2e_host", child)
        line["api_default"] = child.get("api_default", {"error": "no result"})

    if rank == 0 and world == 1 and not args.no_extra:
        # the API's default, break_ties_with_noise=True (exact mt19937 noise: a second pass over the users it can touch);
        # `value` above is with the noise off, as the config string says
        try:
            stream = torch.cuda.current_stream().cuda_stream
            scratch_out = torch.empty_like(prob.out)
            for _ in range(2):
                prob.step(binding, stream, scratch_out, noise=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                prob.step(binding, stream, scratch_out, noise=True)
            torch.cuda.synchronize()
            dtn = (time.perf_counter() - t0) / 10
            line["noise_on"] = {"users_per_s": m / dtn, "ms_per_step": dtn * 1e3,
                                "steps": 10, "warmup": 2,
                                "what": "same workload with break_ties_with_noise=True (the API default), seed 1; never `value`"}
            if args.parity_users > 0:
                line["noise_on"]["parity"] = parity_check(prob, scratch_out, min(args.parity_users, 1024), noise=True, seed=1, binding=binding)
                failed = failed or not line["noise_on"]["parity"]["ok"]
            del scratch_out
        except Exception as e:      # noqa: BLE001
            line["noise_on"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_extra and args.workload != "NS":
        # the north-star shape (n = 1M items, 128 factors): B = 512 MB does not fit the Infinity Cache
        try:
            m2, n2, k2, _, K2, c2, s2 = CONFIGS["NS"]
            m2 = 32768
            del prob.A, prob.B
            p2 = DeviceProblem(torch, dev, m2, n2, k2, c2, s2, K2)
            ns_steps, ns_warmup = 5, 2
            dt2, sw2, pr2, fi2, _ = measure(torch, dist, binding, p2, ns_steps, ns_warmup, 1, None)
            tm2 = binding.timings()
            sh2 = (tm2.get("timed_slots") or 0) / tm2["total_slots"] if tm2.get("total_slots") else 1.0
            tf2 = 2.0 * n2 * k2 * m2 * sh2 / (sw2 * 1e-3) / 1e12
            line["north_star_shape"] = {
                "workload": "NS: %d users x %d items, %d factors fp32, K=%d, all metrics" % (m2, n2, k2, K2),
                "users_per_s": m2 * ns_steps / dt2, "steps": ns_steps, "warmup": ns_warmup, "ms_per_step": dt2 / ns_steps * 1e3,
                "sweep_ms": sw2, "prep_ms": pr2, "finalize_ms": fi2,
                "mfma_TFLOPs": tf2, "mfma_frac": tf2 / PEAK_FP32_MFMA_TFLOPS,
                "main_launch_share_of_users": sh2,
                "hbm_equiv_GBs": n2 * k2 * 4.0 * m2 * sh2 / (sw2 * 1e-3) / 1e9,
                "hbm_equiv_frac": n2 * k2 * 4.0 * m2 * sh2 / (sw2 * 1e-3) / 1e9 / PEAK_HBM_GBS}
            tr_b, tr_src = load_traffic("NS", m2)
            if tr_b:
                td = json.load(open(os.path.join(ROOT, tr_src)))
                line["north_star_shape"].update({"traffic": tr_b, "traffic_source": tr_src, "hbm_read_GBs": td["hbm_read_bytes"] / (sw2 * 1e-3) / 1e9,
                                                 "hbm_read_frac_of_peak": td["hbm_read_bytes"] / (sw2 * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                 "traffic_counts": "L2-miss bytes of the sweep launch: FETCH_SIZE x 2 (the factor for this LDS-DMA pattern: profiles/r6_fetch_calibration.txt) + WRITE_SIZE; Infinity-Cache hits are counted, so this is fabric traffic, an upper bound of the HBM bytes"})
            hl = host_leg(early, "NS", dt2 / ns_steps * 1e3)
            if hl:
                line["north_star_shape"]["e2e_host"] = hl
            if args.parity_users > 0:
                line["north_star_shape"]["parity"] = parity_check(p2, p2.out, min(args.parity_users, 512), binding=binding)
                failed = failed or not line["north_star_shape"]["parity"]["ok"]
            del p2
        except Exception as e:      # noqa: BLE001
            line["north_star_shape"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_extra and args.workload != "TUT":
        # the only realistic workload the reference documents (examples/recometrics_example.ipynb cells 3, 5, 9, 11; BASELINE.md
        # section 1): 10,000 users x 160,112 items x 50 factors, k = 5, all metrics, API defaults -- tie noise ON.  The small-m
        # regime (79 user blocks for 256 CUs: the grid is filled by item ranges) and a factor count that is not a multiple of 8
        # (50 -> 56: seven factor groups, one half empty).  Device-resident step, then the call a user types (SciPy in, dict out).
        try:
            mt, nt, kt, dtt, Kt, ct, st = CONFIGS["TUT"]
            if "prob" in locals() and hasattr(prob, "A"):
                del prob.A, prob.B
            torch.cuda.empty_cache()
            pt = DeviceProblem(torch, dev, mt, nt, kt, ct, st, Kt, dtt)
            pt.noise = True
            t_steps, t_warm = 10, 2
            dtt_, swt, prt, fit, _ = measure(torch, dist, binding, pt, t_steps, t_warm, 1, None)
            tmt = binding.timings()
            # (with the noise on, rm_get_timings reports the main pass -- every user on the plain scores; the exact pass of the
            # flagged users runs beside it)
            tft = 2.0 * nt * kt * mt / (swt * 1e-3) / 1e12
            line["tutorial"] = {
                "workload": "TUT: %d users x %d items, %d factors fp32, k=%d, all metrics, break_ties_with_noise=True (the notebook's call)" % (mt, nt, kt, Kt),
                "users_per_s": mt * t_steps / dtt_, "steps": t_steps, "warmup": t_warm, "ms_per_step": dtt_ / t_steps * 1e3,
                "sweep_ms": swt, "prep_ms": prt, "finalize_ms": fit, "item_splits": tmt.get("item_splits"), "sweep_blocks": tmt.get("sweep_blocks"),
                "mfma_TFLOPs": tft, "mfma_frac": tft / PEAK_FP32_MFMA_TFLOPS,
                "mfma_frac_padded_factors": tft / PEAK_FP32_MFMA_TFLOPS * (((kt + 7) // 8 * 8) / kt)}
            if args.parity_users > 0:
                line["tutorial"]["parity"] = parity_check(pt, pt.out, min(args.parity_users, 1024), noise=True, seed=1, cpu_seconds=8.0, binding=binding)
                failed = failed or not line["tutorial"]["parity"]["ok"]
            del pt
            binding.load().rm_release_workspace()
            torch.cuda.empty_cache()
            if not args.no_e2e:
                child = early.get("TUT", {})
                api = child.get("api_default", {})
                line["tutorial"]["api_ms"] = api.get("ms")
                line["tutorial"]["api_users_per_s"] = api.get("users_per_s")
                line["tutorial"]["host_entry_ms"] = child.get("e2e_host", {}).get("steady_ms")
        except Exception as e:      # noqa: BLE001
            line["tutorial"] = {"error": repr(e)}

    if rank == 0 and world == 1 and not args.no_extra and not args.no_other:
        # The requests WITHOUT ROC / PR-AUC (the k_sweep<..., AUC = false, ...> family): the API's literal default -- precision,
        # average_precision, ndcg, tie noise on (recometrics/__init__.py:50-62) -- at BASELINE C2's shape (k = 10) and at the
        # tutorial's (k = 5), and BASELINE.md section 2's "NS, no AUC" row (the eight top-K metrics at the north-star shape).
        noauc = {}
        legs = [("defaults_C2", "C2", None, ("p", "ap", "ndcg"), True, 10, 2, "C2's shape, precision + average_precision + ndcg, k = 10, tie noise on (the API's default request)"),
                ("defaults_TUT", "TUT", None, ("p", "ap", "ndcg"), True, 10, 2, "the tutorial's shape, precision + average_precision + ndcg, k = 5, tie noise on"),
                ("NS_no_auc", "NS", 32768, ("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr"), False, 5, 2, "north-star shape, the eight top-K metrics, K = 10 (BASELINE.md section 2)")]
        for lname, wname, mo, mets, noise_on, lsteps, lwarm, what in legs:
            try:
                m_, no, ko, dto, Ko, co, so = CONFIGS[wname]
                mo = mo or m_
                if "prob" in locals() and hasattr(prob, "A"):
                    del prob.A, prob.B
                torch.cuda.empty_cache()
                pn = DeviceProblem(torch, dev, mo, no, ko, co, so, Ko, dto, metrics=mets)
                pn.noise = noise_on
                dtn_, swn, prn, fin, _ = measure(torch, dist, binding, pn, lsteps, lwarm, 1, None)
                tfn = 2.0 * no * ko * mo / (swn * 1e-3) / 1e12
                noauc[lname] = {"workload": what, "users": mo, "n_items": no, "n_factors": ko, "k_metrics": Ko, "metrics": list(mets), "noise": noise_on,
                                "steps": lsteps, "warmup": lwarm, "users_per_s": mo * lsteps / dtn_, "ms_per_step": dtn_ / lsteps * 1e3,
                                "sweep_ms": swn, "prep_ms": prn, "finalize_ms": fin, "mfma_TFLOPs": tfn, "mfma_frac": tfn / PEAK_FP32_MFMA_TFLOPS}
                if args.parity_users > 0:
                    noauc[lname]["parity"] = parity_check(pn, pn.out, min(args.parity_users, 1024), noise=noise_on, seed=1, cpu_seconds=5.0, binding=binding)
                    failed = failed or not noauc[lname]["parity"]["ok"]
                del pn
                binding.load().rm_release_workspace()
                torch.cuda.empty_cache()
            except Exception as e:      # noqa: BLE001
                noauc[lname] = {"error": repr(e)}
        line["no_auc"] = noauc

    if rank == 0 and world == 1 and not args.no_extra and not args.no_other:
        # every other BASELINE config in the same driver-run line, as compact legs: 3 timed steps after 1 warm-up, the sweep's
        # fraction of the MFMA peak, and a parity sample against the compiled reference bounded to ~5 s of CPU work each
        others = {}
        specs = [("C3", 125000, True, "C3: 1M users over 8 GPUs = 125,000 users per GPU, cumulative K = 1..20"),
                 ("C4", 8192, False, "C4: a slice of 8,192 of its 100,000 users at the full 10M items, K = 100 + AUC"),
                 ("C5", 16384, False, "C5: a slice of 16,384 of its 200,000 users (50,000 per GPU on 4), 256 factors fp64, K = 50")]
        for wname, mo, cum, what in specs:
            if wname == args.workload:
                continue
            try:
                _, no, ko, dto, Ko, co, so = CONFIGS[wname]
                if "prob" in locals() and hasattr(prob, "A"):
                    del prob.A, prob.B
                torch.cuda.empty_cache()
                po = DeviceProblem(torch, dev, mo, no, ko, co, so, Ko, dto, cumulative=cum)
                dto_, swo, pro, fio, _ = measure(torch, dist, binding, po, 3, 1, 1, None)
                tmo = binding.timings()
                sho = (tmo.get("timed_slots") or 0) / tmo["total_slots"] if tmo.get("total_slots") else 1.0
                pk = PEAK_FP32_MFMA_TFLOPS if dto == np.float32 else PEAK_FP64_MFMA_TFLOPS
                per_step = list(getattr(measure, "last_sweep_ms", [swo]))
                tfo = 2.0 * no * ko * mo * sho / (swo * 1e-3) / 1e12
                others[wname] = {"workload": what, "users": mo, "n_items": no, "n_factors": ko, "k_metrics": Ko, "cumulative": cum,
                                 "dtype": "f32" if dto == np.float32 else "f64", "steps": 3, "warmup": 1,
                                 "users_per_s": mo * 3 / dto_, "ms_per_step": dto_ / 3 * 1e3, "sweep_ms": swo, "prep_ms": pro, "finalize_ms": fio,
                                 "mfma_TFLOPs": tfo, "mfma_peak_TFLOPs": pk, "mfma_frac": tfo / pk,
                                 # (three steps: one slow step -- seen once at C5, 108 ms among 86 ms ones -- moves the mean by 8 %)
                                 "sweep_ms_per_step": per_step, "mfma_frac_best_step": tfo / pk * swo / min(per_step)}
                # HBM traffic of this config's sweep launch from its own committed PMC profile (B is larger than the 256 MiB
                # Infinity Cache at C4 and C5: the read rate against the 8 TB/s peak, next to the MFMA fraction)
                tr_b, tr_src = load_traffic(wname, mo)
                if tr_b:
                    td = json.load(open(os.path.join(ROOT, tr_src)))
                    others[wname].update({"traffic": tr_b, "traffic_source": tr_src,
                                          "hbm_read_GBs": td["hbm_read_bytes"] / (swo * 1e-3) / 1e9,
                                          "hbm_read_frac_of_peak": td["hbm_read_bytes"] / (swo * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                          "traffic_counts": "L2-miss bytes of the sweep launch: FETCH_SIZE x 2 (the factor for this LDS-DMA pattern: profiles/r6_fetch_calibration.txt) + WRITE_SIZE; Infinity-Cache hits are counted, so this is fabric traffic, an upper bound of the HBM bytes"})
                hl = host_leg(early, wname, dto_ / 3 * 1e3)
                if hl:
                    others[wname]["e2e_host"] = hl
                if args.parity_users > 0:
                    others[wname]["parity"] = parity_check(po, po.out, min(args.parity_users, 512), cpu_seconds=5.0, binding=binding)
                    failed = failed or not others[wname]["parity"]["ok"]
                del po
                binding.load().rm_release_workspace()
                torch.cuda.empty_cache()
            except Exception as e:      # noqa: BLE001
                others[wname] = {"error": repr(e)}
        line["other_configs"] = others

    if world > 1:
        # the first real multi-GPU run checks itself: the same workload through rm_set_devices([0 .. N-1]) in ONE (torch-free)
        # process must equal the unsharded call bit for bit -- two distinct devices, peer copies of the item factors over xGMI
        line["multi_gpu_self_check"] = self_check
        # (every device idle first; the other ranks then wait for the leg on the process group's store -- a host-side wait -- and
        # not inside a collective, whose kernel would spin on the very devices the child is about to use)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        store = None
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:       # noqa: BLE001
            store = None
        if rank == 0:
            try:
                # (a separate leg: its verdict is in the line -- `bitwise_equal_to_unsharded` -- and does not void the timed number)
                # (bounded: an informational leg must not cost the timed line its place in the driver's record if a first-ever
                # device-to-device copy stalls; it takes ~15 s)
                line["sharded_host"] = run_child(args, m, ["--sharded-child", str(world)], timeout=240)
            except Exception as e:      # noqa: BLE001
                line["sharded_host"] = {"error": repr(e)}
            if store is not None:
                try:
                    store.set("rm_bench_sharded_leg", "done")
                except Exception:       # noqa: BLE001
                    pass
        elif store is not None:
            try:
                import datetime
                store.wait(["rm_bench_sharded_leg"], datetime.timedelta(seconds=300))
            except Exception:       # noqa: BLE001
                pass
        dist.barrier()

    if rank == 0 and world == 1 and not args.no_cpu:
        try:
            line["cpu_baseline"] = cpu_baseline(prob.host, K, m, args.cpu_seconds, dtype)
        except Exception as e:      # noqa: BLE001
            line["cpu_baseline"] = {"value": None, "unit": "users/s", "cores": os.cpu_count(), "kind": "port", "sample": "failed: %r" % (e,)}
    elif rank == 0:
        line["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    if failed:
        sys.exit("bench.py: the timed outputs do NOT match the oracle -- the number above is invalid")


if __name__ == "__main__":
    main()
