"""GPU tests of what sits around the kernels: sharding inside the library (rm_set_devices), the torch.distributed helper
with the HIP binding as its compute, interruption between user batches, calls from several threads.

One MI355X is visible on the test box, so the in-library shards are VIRTUAL (the same device listed several times: every
shard has its own host thread, stream and workspace, exactly as on distinct devices)."""
import os
import signal
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from _util import assert_same_bits

pytestmark = pytest.mark.gpu


def _same(x, y):
    from _util import same_bits
    return same_bits(x, y)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL = {name: True for name in ("p", "tp", "r", "ap", "tap", "ndcg", "hit", "rr", "roc", "pr")}


@pytest.fixture(scope="module")
def hip():
    from recometrics_amd import _binding
    _binding.load()
    assert _binding.device_count() > 0, "no HIP device visible"
    return _binding


def _calc(hip, pr, K, cumulative=False, dtype=np.float32, outs=None, noise=False):
    trp, tri = pr["train"]
    tep, tei, tev = pr["test"]
    return hip.calc_metrics(np.ascontiguousarray(pr["A"], dtype), pr["A"].shape[1], np.ascontiguousarray(pr["B"], dtype), pr["B"].shape[1],
                            trp, tri, tep, tei, tev.astype(dtype), K, ALL, cumulative, noise, True, 2, 1, 1, 1, outs=outs)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], [0]])
@pytest.mark.parametrize("dtype,cumulative", [(np.float32, False), (np.float32, True), (np.float64, False)])
def test_in_library_shards_equal_the_single_device_call(hip, devices, dtype, cumulative):
    """rm_set_devices: users sharded inside one call (uneven ranges: 1,003 users over 2 and 3 shards), every output
    bit-identical to the unsharded call; the ranking outputs of rm_rank_* too"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(1003, 6000, 48, dtype, mean_c=70, seed=12)
    hip.set_devices([])
    want = _calc(hip, pr, 10, cumulative, dtype)
    want_rank = hip.rank(pr["A"], pr["B"], pr["train"][0], pr["train"][1], pr["test"][0], pr["test"][1], 10)
    try:
        hip.set_devices(devices)
        assert hip.get_devices() == devices
        got = _calc(hip, pr, 10, cumulative, dtype)
        got_rank = hip.rank(pr["A"], pr["B"], pr["train"][0], pr["train"][1], pr["test"][0], pr["test"][1], 10)
    finally:
        hip.set_devices([])
    for name, g, w in zip(hip.METRIC_ORDER, got, want):
        assert_same_bits(g, w, "%s over devices %s" % (name, devices))
    for key in want_rank:
        assert (got_rank[key] == want_rank[key]).all() or key == "topk_score", key
    assert_same_bits(got_rank["topk_score"], want_rank["topk_score"], "top-K scores")


def test_set_devices_rejects_unknown_devices(hip):
    with pytest.raises(ValueError, match="does not exist"):
        hip.set_devices([0, 99])
    assert hip.get_devices() == []


def test_user_batches_equal_one_batch(hip, monkeypatch):
    """host-pointer calls evaluate their users in batches (so that an interrupt is seen within a fraction of a second);
    the batch size must not change any result (RM_BATCH_USERS forces many small batches)"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(5000, 3000, 32, np.float32, mean_c=120, seed=4)
    want = _calc(hip, pr, 7, True)
    monkeypatch.setenv("RM_BATCH_USERS", "1024")
    got = _calc(hip, pr, 7, True)
    for name, g, w in zip(hip.METRIC_ORDER, got, want):
        assert_same_bits(g, w, name + " in batches of 1024 users")
    rk = hip.rank(pr["A"], pr["B"], pr["train"][0], pr["train"][1], pr["test"][0], pr["test"][1], 7)
    monkeypatch.delenv("RM_BATCH_USERS")
    rk1 = hip.rank(pr["A"], pr["B"], pr["train"][0], pr["train"][1], pr["test"][0], pr["test"][1], 7)
    for key in rk:
        assert (rk[key] == rk1[key]).all() or key == "topk_score", key


def _interruptible_problem(copies=6):
    """245,760 users (one synthetic problem of 40 k users stacked `copies` times): at 1,024 users per batch the call takes a
    few hundred batches, ~0.1 s -- it outlasts the 30 ms timers below by a wide margin"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(40 * 1024, 20000, 64, np.float32, mean_c=40, seed=8)

    def stack(p, *arrs):
        ptr = np.concatenate([p[:1]] + [p[1:].astype(np.int64) + i * int(p[-1]) for i in range(copies)]).astype(np.int32)
        return (ptr,) + tuple(np.tile(a, copies) for a in arrs)
    return {"A": np.tile(pr["A"], (copies, 1)), "B": pr["B"], "train": stack(*pr["train"]), "test": stack(*pr["test"])}


def test_interrupt_between_batches(hip, monkeypatch):
    """reference src/recometrics.hpp:114-174,:488-489,:964: an interrupt stops the evaluation (the remaining users are
    skipped), the call fails with "Error: procedure was interrupted.", what was finished is intact"""
    pr = _interruptible_problem()
    m = pr["A"].shape[0]
    full = _calc(hip, pr, 10)
    monkeypatch.setenv("RM_BATCH_USERS", "1024")         # 240 batches: the call outlasts the timer by a wide margin
    outs = [np.full(m, -7.0, np.float32) for _ in range(10)]
    t = threading.Timer(0.03, hip.request_interrupt)
    t.start()
    with pytest.raises(RuntimeError, match="procedure was interrupted"):
        _calc(hip, pr, 10, outs=outs)
    t.join()
    done = int((outs[0] != -7.0).sum() if not np.isnan(outs[0]).any() else (~(outs[0] == -7.0)).sum())
    assert 1024 <= done < m and done % 1024 == 0, "finished users: %d of %d" % (done, m)
    for name, g, w in zip(hip.METRIC_ORDER, outs, full):
        assert_same_bits(g[:done], w[:done], name + " of the users finished before the interrupt")
        assert (g[done:] == -7.0).all(), name + ": users after the interrupt must be untouched"
    # the flag is cleared: the next call runs to the end
    again = _calc(hip, pr, 10)
    for name, g, w in zip(hip.METRIC_ORDER, again, full):
        assert_same_bits(g, w, name + " after an interrupted call")


def test_interrupt_with_tie_noise_never_leaves_un_noised_values(hip, monkeypatch):
    """fp32 + break_ties_with_noise over user batches: every batch first evaluates on the plain scores and flags the users the noise
    can touch; their exact evaluation follows the last batch.  A call interrupted in between must not hand out the first pass's
    values of flagged users as results: every user of a finished batch either carries its final (noise-on) value, bit for bit, or
    NaN; users behind the interrupt are untouched.  (Cold items -- zero factors -- put many test scores into the noise zone.)"""
    pr = _interruptible_problem()
    pr = dict(pr, B=pr["B"].copy())
    pr["B"][::5] = 0                                         # every fifth item scores exactly 0 for everybody
    m = pr["A"].shape[0]
    full = _calc(hip, pr, 10, noise=True)
    plain = _calc(hip, pr, 10, noise=False)
    changed = np.zeros(m, bool)
    for g, w in zip(plain, full):
        changed |= ~(_same(g, w))
    assert changed.sum() > 100, "the problem must have users whose metrics the noise changes (%d)" % changed.sum()
    monkeypatch.setenv("RM_BATCH_USERS", "1024")
    outs = [np.full(m, -7.0, np.float32) for _ in range(10)]
    t = threading.Timer(0.03, hip.request_interrupt)
    t.start()
    with pytest.raises(RuntimeError, match="procedure was interrupted"):
        _calc(hip, pr, 10, outs=outs, noise=True)
    t.join()
    touched = np.zeros(m, bool)
    for g in outs:
        touched |= ~(g == -7.0)
    done = int(touched.nonzero()[0].max()) + 1 if touched.any() else 0
    assert 1024 <= done < m, "finished users: %d of %d" % (done, m)
    for name, g, w in zip(hip.METRIC_ORDER, outs, full):
        ok = _same(g[:done], w[:done]) | np.isnan(g[:done])
        assert ok.all(), "%s: %d finished users carry neither their final value nor NaN" % (name, (~ok).sum())
        assert (g[done:] == -7.0).all(), name + ": users after the interrupt must be untouched"
    # a user whose first-pass value differs from its final one must be NaN (its exact evaluation never ran)
    for name, g, w, pl in zip(hip.METRIC_ORDER, outs, full, plain):
        differs = ~_same(pl[:done], w[:done])
        assert (np.isnan(g[:done][differs]) | _same(g[:done][differs], w[:done][differs])).all(), name


def test_sigint_during_a_call_becomes_keyboard_interrupt(hip, monkeypatch):
    """a real SIGINT: the library's handler takes it during the call, then restores Python's handler and re-raises the
    signal, so that the caller sees KeyboardInterrupt (reference :166-173 + recometrics/wrapper.pyx `except +`)"""
    pr = _interruptible_problem()
    monkeypatch.setenv("RM_BATCH_USERS", "1024")          # 240 batches: the signal arrives while the call is running
    before = signal.getsignal(signal.SIGINT)
    t = threading.Timer(0.03, lambda: os.kill(os.getpid(), signal.SIGINT))
    t.start()
    with pytest.raises((KeyboardInterrupt, RuntimeError)) as ei:
        _calc(hip, pr, 10)
        time.sleep(0.5)            # (the pending KeyboardInterrupt surfaces here if the call itself returned an error first)
    t.join()
    assert ei.type is KeyboardInterrupt or "interrupted" in str(ei.value)
    assert signal.getsignal(signal.SIGINT) is before, "the caller's SIGINT handler must be back in place"


def test_calls_from_two_threads_share_a_device(hip):
    """two host threads calling at the same time on one device: each call has the context (workspace, events) to itself
    while it runs -- results equal the sequential ones"""
    from recometrics_amd.synth import make_problem
    prs = [make_problem(3000, 4000, 24, np.float32, mean_c=60, seed=s) for s in (1, 2)]
    want = [_calc(hip, pr, 10) for pr in prs]
    got = [None, None]

    def work(i):
        for _ in range(3):
            got[i] = _calc(hip, prs[i], 10)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [th.start() for th in ths]
    [th.join() for th in ths]
    for i in range(2):
        for name, g, w in zip(hip.METRIC_ORDER, got[i], want[i]):
            assert_same_bits(g, w, "thread %d %s" % (i, name))


def test_set_device_on_two_devices_from_one_thread(hip):
    if hip.device_count() < 2:
        pytest.skip("one device visible")
    from recometrics_amd.synth import make_problem
    pr = make_problem(500, 3000, 16, np.float32, mean_c=40, seed=3)
    hip.set_device(0)
    want = _calc(hip, pr, 5)
    hip.set_device(1)
    got = _calc(hip, pr, 5)
    hip.set_device(0)
    for name, g, w in zip(hip.METRIC_ORDER, got, want):
        assert_same_bits(g, w, name + " on device 1")


_RANK_SCRIPT = r"""
import os, sys, numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from recometrics_amd import _binding as hip
from recometrics_amd.sharding import calc_metrics_sharded
from recometrics_amd.synth import make_problem
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
hip.load(); hip.set_device(0)                      # both ranks share GPU 0 (one device on the test box)
pr = make_problem(777, 5000, 32, np.float32, mean_c=50, seed=5)
names = {"p": "P@K", "tp": "TP@K", "r": "R@K", "ap": "AP@K", "tap": "TAP@K", "ndcg": "NDCG@K", "hit": "Hit@K", "rr": "RR@K", "roc": "ROC_AUC", "pr": "PR_AUC"}
def compute(A, B, train, test, k, want, **kw):
    outs = hip.calc_metrics(np.ascontiguousarray(A), A.shape[1], B, B.shape[1], train[0], train[1], test[0], test[1], test[2], k,
                            {n: True for n in hip.METRIC_ORDER}, False, False, True, 2, 1, 1, 1)
    return {names[n]: o for n, o in zip(hip.METRIC_ORDER, outs)}
full = calc_metrics_sharded(pr["A"], pr["B"], pr["train"], pr["test"], 10, None, compute, world, rank, dist=dist)
if rank == 0:
    np.savez(%(out)r, **full)
dist.destroy_process_group()
"""


def test_torch_distributed_helper_with_the_hip_binding(hip, tmp_path):
    """recometrics_amd.sharding.calc_metrics_sharded (one process per GPU, one all-gather of the metric block) with the
    HIP binding as its compute: two gloo ranks sharing GPU 0 == the single-process call"""
    from recometrics_amd.synth import make_problem
    out = str(tmp_path / "sharded.npz")
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % {"root": ROOT, "out": out})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    got = np.load(out)
    pr = make_problem(777, 5000, 32, np.float32, mean_c=50, seed=5)
    want = _calc(hip, pr, 10)
    names = {"p": "P@K", "tp": "TP@K", "r": "R@K", "ap": "AP@K", "tap": "TAP@K", "ndcg": "NDCG@K", "hit": "Hit@K", "rr": "RR@K", "roc": "ROC_AUC", "pr": "PR_AUC"}
    for n, w in zip(hip.METRIC_ORDER, want):
        assert_same_bits(got[names[n]].astype(np.float32), w, "sharded over 2 ranks: " + names[n])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_with_two_ranks(hip, scaling):
    """bench.py --gpus 2 end to end (it starts torch.distributed.run itself): two ranks, the metric blocks gathered while the
    next step computes, one JSON line from rank 0 with the parity check green.  gloo and a shared GPU stand in for RCCL
    and two GPUs on the one-GPU test box (RM_BENCH_BACKEND)."""
    import json
    env = dict(os.environ, RM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--users", "6001",
                          "--no-cpu", "--no-extra", "--scaling", scaling], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == scaling
    assert line["value"] > 0 and line["parity_checked"] > 0 and line["parity"]["ok"]
    users = 2 * 6001 if scaling == "weak" else 6001
    assert abs(line["value"] - users * 3 / (line["ms_per_step"] * 3e-3)) < 1e-6 * line["value"]
    # the run checks itself: here the stand-in backend and the shared GPU are REPORTED (with RCCL they would be refused, exit 3)
    sc = line["multi_gpu_self_check"]
    assert sc["rccl_ranks"] == 2 and sc["comm_backend"] == "gloo" and sc["ok"] is False and sc["distinct_devices"] == 1
    sh = line["sharded_host"]
    assert sh["bitwise_equal_to_unsharded"] is True and len(sh["devices"]) == 2, sh


def test_bench_refuses_a_multi_gpu_run_that_is_not_one():
    """bench.py --gpus 2 with RCCL on a box with ONE GPU: two ranks cannot sit on two distinct devices -- the run must end
    non-zero without a result line (whatever fails first: RCCL's own refusal of a shared device, or the self-check)"""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RM_BENCH_BACKEND", None)
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box with exactly one GPU")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--users", "2048",
                          "--no-cpu", "--no-extra"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode != 0
    for ln in res.stdout.splitlines():
        if ln.startswith("{"):
            assert "value" not in json.loads(ln)


def test_cython_binding_equals_the_ctypes_binding(hip, oracle=None):
    """recometrics_amd/_cy.pyx (compiled for real): the same C-ABI through Cython -- outputs
    bit-identical to the ctypes binding, single and cumulative, both precisions; status codes become the same exceptions"""
    from recometrics_amd import build as rb
    from recometrics_amd.synth import make_problem
    assert os.path.exists(rb.cython_module_path()), "build the Cython binding first (python -m recometrics_amd.build)"
    from recometrics_amd import _cy
    assert _cy.has_openmp() and _cy.device_count() > 0
    for dtype in (np.float32, np.float64):
        pr = make_problem(400, 3000, 24, dtype, mean_c=50, seed=9)
        trp, tri = pr["train"]
        tep, tei, tev = pr["test"]
        for cumulative in (False, True):
            want = hip.calc_metrics(pr["A"], 24, pr["B"], 24, trp, tri, tep, tei, tev, 8, ALL, cumulative, True, True, 2, 1, 1, 123)
            got = _cy.calc_metrics(pr["A"], 24, pr["B"], 24, trp, tri, tep, tei, tev, 8, ALL, cumulative, True, True, 2, 1, 1, 123)
            for name, g, w in zip(hip.METRIC_ORDER, got, want):
                assert g.dtype == w.dtype and g.shape == w.shape
                assert_same_bits(g, w, "cython vs ctypes: %s" % name)
    with pytest.raises(ValueError, match="k_metrics"):
        _cy.calc_metrics(pr["A"], 24, pr["B"], 24, trp, tri, tep, tei, tev, 0, ALL, False, False, True, 2, 1, 1, 1)


def test_rccl_all_gather_of_the_metric_block_with_one_rank():
    """backend "nccl" IS RCCL on ROCm: with the one GPU of the test box, build a world of one rank, evaluate a shard with the
    HIP binding and push its metric block through sharding.all_gather_rows on the DEVICE -- the library loads, the
    communicator is built and the collective runs, which is everything a multi-GPU run adds to the tested path except
    the second device."""
    code = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from recometrics_amd import _binding, sharding
from recometrics_amd.synth import make_problem
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% int(os.environ["PORT"]), world_size=1, rank=0, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
_binding.load(); _binding.set_device(0)
pr = make_problem(300, 2000, 32, np.float32, mean_c=40, seed=3)
trp, tri = pr["train"]; tep, tei, tev = pr["test"]
want = {name: True for name in _binding.METRIC_ORDER}
outs = _binding.calc_metrics(pr["A"], 32, pr["B"], 32, trp, tri, tep, tei, tev, 10, want, False, False, True, 2, 1, 1, 1)
block = torch.from_numpy(np.stack(outs, axis=1)).to("cuda:0")
full = sharding.all_gather_rows(block, 300, 1, dist=dist, always=True)
torch.cuda.synchronize()
assert full.is_cuda and full.shape == block.shape
a, b = full.cpu().numpy(), block.cpu().numpy()
assert ((a == b) | (np.isnan(a) & np.isnan(b))).all()
t = torch.ones(4, device="cuda:0"); dist.all_reduce(t); torch.cuda.synchronize(); assert float(t.sum()) == 4.0
dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version())
""" % ROOT
    env = dict(os.environ, PORT=str(29500 + os.getpid() % 2000), HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "RCCL_OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


@pytest.mark.parametrize("noise", [False, True])
def test_k_metrics_beyond_the_lists_in_user_batches(hip, monkeypatch, noise):
    """k_metrics > 256 keeps one score row per user of a batch in HBM: several batches of one host call (and, with the tie
    noise, the exact second pass over the flagged users) must all fit the budget that sized the first one -- the rows of an
    earlier batch are still cached in the workspace and count as available.  No RM_STREAM_BUDGET_MB: the real budget."""
    from recometrics_amd.synth import make_problem
    pr = make_problem(3000, 2500, 24, np.float32, mean_c=50, seed=14)
    pr["B"] = pr["B"].copy()
    pr["B"][np.random.default_rng(2).random(2500) < 0.1] = 0              # zero scores: users the noise can reorder
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]

    def call():
        return hip.calc_metrics(pr["A"], 24, pr["B"], 24, trp, tri, tep, tei, tev, 300, ALL, True, noise, True, 2, 1, 1, 5)
    want = call()
    monkeypatch.setenv("RM_BATCH_USERS", "1024")
    got = call()
    for name, g, w in zip(hip.METRIC_ORDER, got, want):
        assert_same_bits(g, w, name + " (K = 300, three batches, noise=%s)" % noise)


def test_pipelined_uploads_equal_one_batch(hip, monkeypatch):
    """large user ranges are uploaded and evaluated as a ramp of batches (a quarter of the users, then three times the batch before:
    the rows of batch i + 1 travel while batch i computes); every output equals the single-batch call bit for bit, also with the API-default tie noise and through
    rm_rank_*"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(40000, 1200, 16, np.float32, mean_c=30, seed=15)     # > 16,384 users: 10,240 + 29,760, on two contexts
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    for noise in (False, True):
        got = hip.calc_metrics(pr["A"], 16, pr["B"], 16, trp, tri, tep, tei, tev, 5, ALL, False, noise, True, 2, 1, 1, 3)
        monkeypatch.setenv("RM_BATCH_USERS", "65536")                      # one batch
        want = hip.calc_metrics(pr["A"], 16, pr["B"], 16, trp, tri, tep, tei, tev, 5, ALL, False, noise, True, 2, 1, 1, 3)
        monkeypatch.delenv("RM_BATCH_USERS")
        for name, g, w in zip(hip.METRIC_ORDER, got, want):
            assert_same_bits(g, w, name + " (ramp of batches, noise=%s)" % noise)
    rk = hip.rank(pr["A"], pr["B"], trp, tri, tep, tei, 5)
    monkeypatch.setenv("RM_BATCH_USERS", "65536")
    rk1 = hip.rank(pr["A"], pr["B"], trp, tri, tep, tei, 5)
    for key in rk:
        assert (rk[key] == rk1[key]).all() or key == "topk_score", key


@pytest.mark.parametrize("env", [{"RM_DEBUG_NOISE_SEQUENTIAL": "1"}, {"RM_DEBUG_ONE_CONTEXT": "1"}, {"RM_DEBUG_NO_TEST_MASK": "1"}, {"RM_DEBUG_NOISE_PER_BATCH": "1"},
                                 {"RM_DEBUG_NOISE_PER_BATCH": "1", "RM_DEBUG_NOISE_SEQUENTIAL": "1"}, {}])
def test_noise_and_batch_pipelines_agree(hip, env, monkeypatch):
    """the API default (tie noise on) through the host entry with > 16,384 users: the exact pass of the flagged users beside the
    first sweep on a peer context, batches alternating between two contexts, test items masked by the dense rows -- against
    the sequential exact pass, one context, the round-2 dense rows: bit for bit"""
    from recometrics_amd.synth import make_problem
    pr = make_problem(20000, 900, 12, np.float32, mean_c=40, seed=23)
    pr["B"] = pr["B"].copy()
    pr["B"][np.random.default_rng(4).random(900) < 0.05] = 0               # zero scores: users the noise reorders
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]

    def call():
        return hip.calc_metrics(pr["A"], 12, pr["B"], 12, trp, tri, tep, tei, tev, 8, ALL, True, True, True, 2, 1, 1, 2 ** 34 + 1)
    monkeypatch.setenv("RM_DEBUG_NOISE_SEQUENTIAL", "1"); monkeypatch.setenv("RM_DEBUG_ONE_CONTEXT", "1"); monkeypatch.setenv("RM_DEBUG_NO_TEST_MASK", "1")
    monkeypatch.setenv("RM_DEBUG_NOISE_PER_BATCH", "1")           # (default: ONE exact pass over the flagged users of the whole range behind the last batch)
    want = call()
    for key in ("RM_DEBUG_NOISE_SEQUENTIAL", "RM_DEBUG_ONE_CONTEXT", "RM_DEBUG_NO_TEST_MASK", "RM_DEBUG_NOISE_PER_BATCH"):
        monkeypatch.delenv(key)
    for key, val in env.items():
        monkeypatch.setenv(key, val)
    got = call()
    for name, g, w in zip(hip.METRIC_ORDER, got, want):
        assert_same_bits(g, w, "%s under %s" % (name, env))


@pytest.mark.parametrize("noise", [False, True])
def test_k_metrics_beyond_the_lists_when_memory_sizes_the_batches(hip, monkeypatch, noise):
    """k_metrics > 256 with a memory budget that BINDS: RM_DEBUG_FREE_MB makes the library see a device with 3 GB (what its
    cached workspaces hold is subtracted), rows are 0.8 MB (200 k items), so a batch is ~900 users and 2,600 users take three.
    Such a call used to alternate its batches between two contexts, each keeping a full set of score rows: the third batch
    found a third of the remaining memory too small and failed with RM_ERR_NOMEM.  Now: one context, and every output equals
    the same users evaluated in separate one-batch calls, bit for bit."""
    from recometrics_amd.synth import make_problem
    pr = make_problem(2600, 200_000, 8, np.float32, mean_c=40, seed=31)
    trp, tri = pr["train"]; tep, tei, tev = pr["test"]
    K = 300
    hip.load().rm_release_workspace()

    def call(u0, u1):
        p0, p1 = trp[u0:u1 + 1] - trp[u0], tep[u0:u1 + 1] - tep[u0]
        return hip.calc_metrics(pr["A"][u0:u1], 8, pr["B"], 8, p0.astype(np.int32), tri[trp[u0]:trp[u1]], p1.astype(np.int32), tei[tep[u0]:tep[u1]],
                                tev[tep[u0]:tep[u1]], K, ALL, False, noise, True, 2, 1, 1, 9)
    cuts = [0, 650, 1300, 1950, 2600]
    if noise:
        parts = None            # (the reference's noise is seeded per ORIGINAL user index: slices starting elsewhere are other streams)
    else:
        parts = [call(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    hip.load().rm_release_workspace()
    monkeypatch.setenv("RM_DEBUG_FREE_MB", "3000")
    monkeypatch.setenv("RM_HOST_TRACE", "1")
    got = call(0, 2600)                                       # raised MemoryError before the fix
    monkeypatch.delenv("RM_DEBUG_FREE_MB")
    hip.load().rm_release_workspace()
    if parts is not None:
        for i, name in enumerate(hip.METRIC_ORDER):
            assert_same_bits(got[i], np.concatenate([p[i] for p in parts]), name + " (K = 300, memory-bound batches)")
    else:
        again = call(0, 2600)                                 # without the cap: one batch
        for i, name in enumerate(hip.METRIC_ORDER):
            assert_same_bits(got[i], again[i], name + " (K = 300, memory-bound batches, noise on)")


def test_release_workspace_beside_running_calls_does_not_deadlock():
    """rm_release_workspace() from one thread while another runs host calls that alternate between two contexts (the call
    holds its context's mutex and asks for the peer context under the registry's mutex; the release used to take the two in
    the opposite order: ABBA).  In a process of its own, with a time limit."""
    code = r"""
import sys, threading, numpy as np
sys.path.insert(0, %r)
from recometrics_amd import _binding as hip
from recometrics_amd.synth import make_problem
hip.load()
pr = make_problem(20000, 600, 8, np.float32, mean_c=20, seed=3)
trp, tri = pr["train"]; tep, tei, tev = pr["test"]
ALL = {name: True for name in hip.METRIC_ORDER}
call = lambda: hip.calc_metrics(pr["A"], 8, pr["B"], 8, trp, tri, tep, tei, tev, 5, ALL, False, True, True, 2, 1, 1, 3)
want = call()
stop = threading.Event()
def releaser():
    while not stop.is_set():
        hip.load().rm_release_workspace()
th = threading.Thread(target=releaser); th.start()
for _ in range(25):
    got = call()
    for g, w in zip(got, want):
        assert ((g == w) | (np.isnan(g) & np.isnan(w))).all()
stop.set(); th.join()
print("NO_DEADLOCK")
""" % ROOT
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "NO_DEADLOCK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
