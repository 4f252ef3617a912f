"""Worker for tests/test_dist_cpu.py: one rank of the SHIPPED multi-GPU exchange on CPU (gloo), driven by the product's own
partition functions exported through the C ABI (nnlm_shard_range, nnlm_shard_cols; no GPU needed for those).

Dense square-loss half-step, form "reduce" (nnlm_comm_set_form(NNLM_FORM_REDUCE); nnlm_mi355x.hip half_step / half_step_solve):
    1. every rank contracts ITS slab of the contraction (nnlm_shard_range) into one buffer [G (k x k) | C (k x cols)];
    2. ONE all_reduce(sum) of that buffer                                       (ncclAllReduce);
    3. every rank solves ITS columns [col0, col1) (nnlm_shard_cols) into a packed slab [k][cpr], zero padded;
    4. ONE all_gather of the slabs                                              (ncclAllGather);
    5. unpack: column rr*cpr + lc of the factor = entry lc of rank rr's slab    (shard_unpack_kernel).
Column-sharded form (missing values, KL methods, and the DEFAULT for dense square loss): a rank does all the work of its columns
over the WHOLE contraction (step 1-2 disappear: no all-reduce), then 3-5.
The per-column arithmetic is the oracle's (this tests the exchange and its index arithmetic, not the kernels); every rank
must end with bit-identical factors."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nnlm_amd import _lib  # noqa: E402
from oracle import nnlm_oracle as npo  # noqa: E402


def gather_unpack(slab, k, ncols, cpr, world):
    """steps 4-5: all_gather of the packed [k][cpr] slabs + shard_unpack_kernel's index arithmetic."""
    send = torch.from_numpy(np.ascontiguousarray(slab))
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    X = np.zeros((k, ncols))
    for rr in range(world):
        blk = recv[rr].numpy()
        for lc in range(cpr):
            col = rr * cpr + lc
            if col < ncols:
                X[:, col] = blk[:, lc]
    return X


def dense_half_step(which, A, Wt, H, reg, inner, tol, method, rank, world, prec):
    n, m = A.shape
    k = H.shape[0]
    b, e = _lib.shard_range(n, m, prec, which, rank, world)
    if which == 1:   # solve H: contraction over rows i of A
        Y, X, B = Wt[:, b:e], H, A[b:e, :]
    else:            # solve W: contraction over columns j of A
        Y, X, B = H[:, b:e], Wt, A[:, b:e].T
    ncols = X.shape[1]
    buf = np.concatenate([(Y @ Y.T).ravel(), (Y @ B).ravel()])
    dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.SUM)   # step 2 (in place on the numpy buffer)
    G = npo._gram_edits(buf[:k * k].reshape(k, k).copy(), reg)
    C = buf[k * k:].reshape(k, -1)
    cpr, c0, c1 = _lib.shard_cols(ncols, rank, world)
    slab = np.zeros((k, cpr))
    total = 0
    for j in range(c0, c1):                                        # step 3: own columns only
        x = X[:, j].copy()
        if method == 1:
            mu = G @ x - C[:, j]
            if reg[2] != 0:
                mu += reg[2]
            total += npo.scd_ls_update(x, G, mu, None, inner, tol)
        else:
            total += npo.lee_ls_update(x, G, C[:, j].copy(), reg[2], None, inner, tol)
        slab[:, j - c0] = x
    Xn = gather_unpack(slab, k, ncols, cpr, world)
    t = torch.tensor([total], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                       # sweep counters: one integer all-reduce
    return Xn, int(t[0]), (b, e), (c0, c1)


def column_half_step(which, A, Wt, H, reg, inner, tol, method, rank, world):
    """Column-sharded form (dense and missing values, methods 1-4): no all-reduce, the rank's columns over the whole contraction."""
    if which == 1:
        Yt, X, B = Wt, H, A
    else:
        Yt, X, B = H, Wt, A.T
    k, ncols = X.shape
    cpr, c0, c1 = _lib.shard_cols(ncols, rank, world)
    slab = np.zeros((k, cpr))
    total = 0
    if c1 > c0:
        Xs = np.asfortranarray(X[:, c0:c1].copy())
        Bs = np.asfortranarray(B[:, c0:c1])
        fn = npo.update_with_missing if not np.isfinite(Bs).all() else npo.update
        total = fn(Xs, np.asfortranarray(Yt), Bs, None, np.array(reg, dtype=float), inner, tol, method)
        slab[:, :c1 - c0] = Xs
    Xn = gather_unpack(slab, k, ncols, cpr, world)
    t = torch.tensor([total], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return Xn, int(t[0]), (c0, c1)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)
    n, m, k = 300, 170, 6
    A = rng.random((n, m))
    Wt, H = rng.random((k, n)), rng.random((k, m))
    reg = [0.02, 0.01, 0.03]
    res = {}
    for prec in (0, 1):
        for method in (1, 2):
            Wn, t0, r0, c0 = dense_half_step(0, A, Wt, H, reg, 4, 1e-9, method, rank, world, prec)
            Hn, t1, r1, c1 = dense_half_step(1, A, Wn, H, reg, 4, 1e-9, method, rank, world, prec)
            res[f"W_{prec}_{method}"], res[f"H_{prec}_{method}"] = Wn, Hn
            res[f"it_{prec}_{method}"] = np.array([t0, t1])
            res[f"rng_{prec}_{method}"] = np.array([r0, r1])
            res[f"cols_{prec}_{method}"] = np.array([c0, c1])
    Ana = A.copy()
    Ana.ravel()[np.random.default_rng(7).choice(A.size, A.size // 10, replace=False)] = np.nan
    for tag, Amat, method, inner in (("dense1", A, 1, 4), ("dense2", A, 2, 4), ("na1", Ana, 1, 4), ("na2", Ana, 2, 4), ("kl3", A, 3, 2), ("kl4", A, 4, 2),
                                     ("nakl", Ana, 4, 1)):
        Wn, t0, _ = column_half_step(0, Amat, Wt, H, reg, inner, 1e-9, method, rank, world)
        Hn, t1, _ = column_half_step(1, Amat, Wn, H, reg, inner, 1e-9, method, rank, world)
        res[f"W_{tag}"], res[f"H_{tag}"], res[f"it_{tag}"] = Wn, Hn, np.array([t0, t1])
    np.savez(f"{out}.rank{rank}.npz", **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
