"""N > 1 path on CPU: world_size-2 `gloo` processes shard the users, evaluate their shard (the oracle stands in for the
HIP binding -- there is no GPU here) and all-gather the metric block; result must equal the single-process evaluation."""
import os
import socket

import numpy as np
import pytest

from _util import assert_same_bits


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_compute(A, B, train, test, k, want, **kw):
    from oracle.oracle import Oracle
    metrics = tuple(nm for nm, on in want.items() if on)
    return Oracle().calc(A, B, train, test, k, metrics=metrics, **kw)


def _worker(rank, world, port, cumulative, q):
    import torch.distributed as dist
    from recometrics_amd.sharding import calc_metrics_sharded
    from recometrics_amd.synth import make_problem
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pr = make_problem(101, 700, 12, np.float32, mean_c=20, seed=4)
    want = {nm: True for nm in ("p", "ap", "ndcg", "rr", "roc", "pr")}
    out = calc_metrics_sharded(pr["A"], pr["B"], pr["train"], pr["test"], 5, want, _oracle_compute, world, rank,
                               dist=dist, cumulative=cumulative)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cumulative", [False, True])
def test_two_rank_user_sharding_matches_single_process(cumulative):
    import torch.multiprocessing as mp
    from oracle.oracle import Oracle
    from recometrics_amd.synth import make_problem
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cumulative, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    pr = make_problem(101, 700, 12, np.float32, mean_c=20, seed=4)
    want = Oracle().calc(pr["A"], pr["B"], pr["train"], pr["test"], 5, metrics=("p", "ap", "ndcg", "rr", "roc", "pr"),
                         cumulative=cumulative)
    assert set(got) == set(want)
    for name in want:
        assert_same_bits(np.ascontiguousarray(got[name]), want[name], name)


def test_user_range_partition_is_exact():
    from recometrics_amd.sharding import slice_csr, user_range
    for m in (1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            edges = [user_range(m, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    p = np.array([0, 2, 2, 5, 9], np.int32); i = np.arange(9, dtype=np.int32); v = np.arange(9, dtype=np.float32)
    sp, si, sv = slice_csr(p, i, v, 1, 3)
    assert sp.tolist() == [0, 0, 3] and si.tolist() == [2, 3, 4] and sv.tolist() == [2, 3, 4]
