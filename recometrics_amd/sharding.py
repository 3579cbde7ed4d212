"""User sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Users are independent units of the hot path (reference src/recometrics.hpp:437: one OpenMP iteration per user), so the
path shards with NO data-path collective: rank r evaluates the contiguous user range [m*r/W, m*(r+1)/W) against a full
replica of the item factors, and the only exchange is ONE all-gather of the per-user metric block at the end
(payload = users x requested metrics x sizeof(real_t); ~0.65 GB at BASELINE config C3, bandwidth-irrelevant on xGMI).
"""
import numpy as np


def user_range(m, world, rank):
    """Contiguous, balanced partition of m users over `world` ranks."""
    return (m * rank) // world, (m * (rank + 1)) // world


def slice_csr(indptr, indices, data, lo, hi):
    """Rows [lo, hi) of a CSR matrix with the index pointer rebased to 0."""
    indptr = np.asarray(indptr)
    a, b = int(indptr[lo]), int(indptr[hi])
    p = (indptr[lo:hi + 1] - indptr[lo]).astype(np.int32)
    return p, np.ascontiguousarray(indices[a:b]), (None if data is None else np.ascontiguousarray(data[a:b]))


def all_gather_rows(local, m, world, dist=None, group=None, always=False):
    """`local`: torch tensor [m_local, ...] holding this rank's user rows -> tensor [m, ...] on every rank.

    Shards are padded to ceil(m / world) rows so that a single all_gather_into_tensor (one RCCL ring over xGMI) moves
    everything; the padding is dropped when the ranges are re-assembled.  With one rank there is nothing to exchange and
    the collective is skipped unless `always` (the one-GPU test of the RCCL path)."""
    import torch
    if dist is None:
        import torch.distributed as dist  # noqa: PLC0415
    if world == 1 and not always:
        return local
    cap = -(-m // world)
    rank = dist.get_rank(group)
    lo, hi = user_range(m, world, rank)
    assert local.shape[0] == hi - lo
    padded = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:hi - lo] = local
    out = torch.empty((world * cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    parts = []
    for r in range(world):
        a, b = user_range(m, world, r)
        parts.append(out[r * cap: r * cap + (b - a)])
    return torch.cat(parts, dim=0)


def calc_metrics_sharded(A, B, train, test, k, want, compute, world, rank, dist=None, group=None, device="cpu", **kw):
    """Evaluates this rank's user shard with `compute` and all-gathers the metric block.

    compute(A_shard, B, (trp, tri), (tep, tei, tev), k, want, **kw) -> dict name -> array [m_local] or [m_local, k]
    (the product passes the HIP binding; the gloo tests pass the CPU oracle).  Returns dict name -> full array [m, ...]."""
    import torch
    m = A.shape[0]
    lo, hi = user_range(m, world, rank)
    trp, tri, _ = slice_csr(train[0], train[1], None, lo, hi)
    tep, tei, tev = slice_csr(test[0], test[1], test[2] if len(test) > 2 else None, lo, hi)
    local = compute(A[lo:hi], B, (trp, tri), (tep, tei, tev), k, want, **kw)
    names = sorted(local)
    cols = [np.asarray(local[nm]).reshape(hi - lo, -1) for nm in names]
    widths = [c.shape[1] for c in cols]
    block = torch.from_numpy(np.ascontiguousarray(np.concatenate(cols, axis=1))).to(device)
    full = all_gather_rows(block, m, world, dist=dist, group=group).cpu().numpy()
    out, at = {}, 0
    for nm, w, c in zip(names, widths, cols):
        arr = full[:, at:at + w]
        out[nm] = arr[:, 0] if np.asarray(local[nm]).ndim == 1 else arr
        at += w
    return out
