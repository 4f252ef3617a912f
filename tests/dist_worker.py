"""Worker for tests/test_dist_cpu.py: one rank of the contraction-sharded half-step on CPU (gloo).

Each rank takes ITS slab of the contraction from the product's own partition function
(nnlm_shard_range via nnlm_amd._lib.shard_range), forms the partial [Gram | cross-product] buffer with numpy,
sums it with ONE all_reduce (the step RCCL performs on the GPU), then runs the per-column solver replicated.
The per-column arithmetic is the oracle's (this is a test of the shard math, not of the kernels)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nnlm_amd import _lib  # noqa: E402
from oracle import nnlm_oracle as npo  # noqa: E402


def sharded_half_step(which, A, Wt, H, reg, inner, tol, method, rank, world, prec):
    n, m = A.shape
    k = H.shape[0]
    b, e = _lib.shard_range(n, m, prec, which, rank, world)
    if which == 1:   # solve H: contraction over rows i of A
        Y, X, B = Wt[:, b:e], H, A[b:e, :]
    else:            # solve W: contraction over columns j of A
        Y, X, B = H[:, b:e], Wt, A[:, b:e].T
    buf = np.concatenate([(Y @ Y.T).ravel(), (Y @ B).ravel()])
    t = torch.from_numpy(buf)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)          # the single collective of the half-step
    G = buf[:k * k].reshape(k, k)
    C = buf[k * k:].reshape(k, -1)
    G = npo._gram_edits(G.copy(), reg)
    X = X.copy()
    total = 0
    for j in range(X.shape[1]):
        if method == 1:
            mu = G @ X[:, j] - C[:, j]
            if reg[2] != 0:
                mu += reg[2]
            total += npo.scd_ls_update(X[:, j], G, mu, None, inner, tol)
        else:
            total += npo.lee_ls_update(X[:, j], G, C[:, j].copy(), reg[2], None, inner, tol)
    return X, total, (b, e)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)
    n, m, k = 300, 170, 6
    A = rng.random((n, m))
    Wt, H = rng.random((k, n)), rng.random((k, m))
    res = {}
    for prec in (0, 1):
        for method in (1, 2):
            reg = [0.02, 0.01, 0.03]
            Wn, t0, r0 = sharded_half_step(0, A, Wt, H, reg, 4, 1e-9, method, rank, world, prec)
            Hn, t1, r1 = sharded_half_step(1, A, Wn, H, reg, 4, 1e-9, method, rank, world, prec)
            res[f"W_{prec}_{method}"], res[f"H_{prec}_{method}"] = Wn, Hn
            res[f"it_{prec}_{method}"] = np.array([t0, t1])
            res[f"rng_{prec}_{method}"] = np.array([r0, r1])
    np.savez(f"{out}.rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
