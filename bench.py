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
    return ap.parse_args()


class DeviceProblem:
    """Synthetic workload resident in HBM (torch tensors are only the memory owner; the hot path gets raw pointers)."""

    def __init__(self, torch, dev, m, n, k, mean_c, seed, K, dtype=np.float32, shard=0):
        from recometrics_amd.synth import make_factors, make_interactions
        self.m, self.n, self.k, self.K, self.dtype = m, n, k, K, dtype
        _, B = make_factors(1, n, k, dtype, seed)                        # item factors: the same replica on every rank
        A, _ = make_factors(m, 1, k, dtype, seed + 1 + 1000 * shard)     # this rank's user shard
        trp, tri, tep, tei, tev = make_interactions(m, n, mean_c, dtype, seed + 1000 * shard)
        self.host = dict(A=A, B=B, train=(trp, tri), test=(tep, tei, tev))
        t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)  # noqa: E731
        self.A, self.B = t(A), t(B)
        self.trp, self.tri, self.tep, self.tei, self.tev = t(trp), t(tri if tri.size else np.zeros(1, np.int32)), t(tep), t(tei), t(tev)
        self.nnz_tr, self.nnz_te = int(tri.shape[0]), int(tei.shape[0])
        self.out = torch.empty((10, m), dtype=torch.float32 if dtype == np.float32 else torch.float64, device=dev)   # per-user metric block

    def step(self, binding, stream):
        o = self.out
        binding.calc_metrics_device(
            self.dtype, self.A.data_ptr(), self.k, self.B.data_ptr(), self.k, self.m, self.n, self.k,
            self.trp.data_ptr(), self.tri.data_ptr(), self.nnz_tr, self.tep.data_ptr(), self.tei.data_ptr(),
            self.tev.data_ptr(), self.nnz_te, self.K, [o[i].data_ptr() for i in range(10)],
            cumulative=False, break_ties_with_noise=False, stream=stream)


def load_traffic(workload, users):
    """HBM bytes per sweep launch from the committed PMC run (scratch/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in
    separate rocprofv3 passes, FETCH_SIZE doubled per the gfx950 correction); None when no matching profile exists."""
    path = os.path.join(ROOT, "profiles", "r1_traffic_%s.json" % workload)
    try:
        d = json.load(open(path))
        return d["hbm_bytes"] if int(d.get("users", -1)) == int(users) else None
    except Exception:      # noqa: BLE001
        return None


def cpu_baseline(host, K, n_users_total, budget_s, dtype=np.float32):
    """The CPU path on this host's cores over a bounded sample of the same workload."""
    from oracle import oracle as orc
    ncores = os.cpu_count() or 1
    if orc.reference_available(fast=True):
        impl, kind = orc.Reference(fast=True), "reference"
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

    probe = min(n_users_total, max(ncores * 2, 64))
    run(min(probe, 32))                                    # page in
    t_probe = run(probe)
    nu = int(min(n_users_total, max(probe, probe * budget_s / max(t_probe, 1e-6))))
    t = run(nu) if nu > probe else t_probe
    return {"value": nu / t, "unit": "users/s", "cores": ncores, "kind": kind,
            "sample": "%d of %d users of the same workload, all metrics, K=%d, %d threads, %.1f s" % (nu, n_users_total, K, ncores, t)}


def measure(torch, dist, binding, prob, steps, warmup, world, gather_buf):
    stream = torch.cuda.current_stream().cuda_stream
    sweep_ms, prep_ms, fin_ms = [], [], []

    def one():
        prob.step(binding, stream)
        if world > 1:
            dist.all_gather_into_tensor(gather_buf, prob.out)
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
        tm = binding.timings()                              # HIP events recorded on the stream the kernels ran on
        sweep_ms.append(tm["sweep_ms"]); prep_ms.append(tm["prep_ms"]); fin_ms.append(tm["finalize_ms"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, float(np.mean(sweep_ms)), float(np.mean(prep_ms)), float(np.mean(fin_ms)), binding.timings()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from recometrics_amd import _binding as binding
    from recometrics_amd.synth import CONFIGS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    binding.load()
    binding.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    m, n, k, dtype, K, mean_c, seed = CONFIGS[args.workload]
    if args.workload == "C3":
        m = m // 8                                           # C3 is quoted user-sharded over 8 GPUs
    if args.users:
        m = args.users
    prob = DeviceProblem(torch, dev, m, n, k, mean_c, seed, K, dtype, shard=rank)
    gather_buf = torch.empty((world * 10, m), dtype=prob.out.dtype, device=dev) if world > 1 else None
    peak = PEAK_FP32_MFMA_TFLOPS if dtype == np.float32 else PEAK_FP64_MFMA_TFLOPS
    esize = 4.0 if dtype == np.float32 else 8.0
    dname = "f32" if dtype == np.float32 else "f64"

    dt, sweep_ms, prep_ms, fin_ms, tm = measure(torch, dist, binding, prob, args.steps, args.warmup, world, gather_buf)
    users_per_s = world * m * args.steps / dt
    flops_per_launch = 2.0 * n * k * m                       # 2*n*k per user x users of one launch
    achieved_tf = flops_per_launch / (sweep_ms * 1e-3) / 1e12
    line = {
        "metric": "users/sec evaluated (all metrics, K=%d)" % K, "value": users_per_s, "unit": "users/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dname, "data": "synthetic",
        "config": {"workload": "%s: %d users/GPU x %d items, %d factors %s, K=%d, all 10 metrics, noise off"
                               % (args.workload, m, n, k, dname, K),
                   "users_per_gpu": m, "n_items": n, "n_factors": k, "k_metrics": K,
                   "sharding": "users sharded, item factors replicated, 1 all-gather of the metric block per step"},
        "roofline": {"bound": "mfma", "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved_tf / peak, "traffic": load_traffic(args.workload, m),
                     "kernel": "k_sweep", "avg_launch_ms": sweep_ms, "flops_per_launch": flops_per_launch,
                     "hbm_equiv_GBs": n * k * esize * m / (sweep_ms * 1e-3) / 1e9,
                     "hbm_equiv_frac": n * k * esize * m / (sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
        "stage_ms": {"prep": prep_ms, "sweep": sweep_ms, "finalize": fin_ms, "item_splits": tm.get("item_splits"),
                     "sweep_blocks": tm.get("sweep_blocks"), "lds_bytes": tm.get("lds_bytes")},
    }

    if rank == 0 and world == 1 and not args.no_extra and args.workload != "NS":
        # the north-star shape (n = 1M items, 128 factors): B = 512 MB does not fit the Infinity Cache
        try:
            m2, n2, k2, _, K2, c2, s2 = CONFIGS["NS"]
            m2 = 32768
            del prob.A, prob.B
            p2 = DeviceProblem(torch, dev, m2, n2, k2, c2, s2, K2)
            dt2, sw2, pr2, fi2, _ = measure(torch, dist, binding, p2, 2, 1, 1, None)
            tf2 = 2.0 * n2 * k2 * m2 / (sw2 * 1e-3) / 1e12
            line["north_star_shape"] = {
                "workload": "NS: %d users x %d items, %d factors fp32, K=%d, all metrics" % (m2, n2, k2, K2),
                "users_per_s": m2 * 2 / dt2, "sweep_ms": sw2, "prep_ms": pr2, "finalize_ms": fi2,
                "mfma_TFLOPs": tf2, "mfma_frac": tf2 / PEAK_FP32_MFMA_TFLOPS,
                "hbm_equiv_GBs": n2 * k2 * 4.0 * m2 / (sw2 * 1e-3) / 1e9,
                "hbm_equiv_frac": n2 * k2 * 4.0 * m2 / (sw2 * 1e-3) / 1e9 / PEAK_HBM_GBS}
            del p2
        except Exception as e:      # noqa: BLE001
            line["north_star_shape"] = {"error": repr(e)}

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


if __name__ == "__main__":
    main()
