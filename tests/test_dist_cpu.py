"""World-size-2 gloo test of the multi-GPU path's exchange on CPU (no GPU needed).

GPU side (nnlm_amd/csrc/nnlm_mi355x.hip half_step / half_step_solve / half_step_kl), two forms.  Column-sharded (default for
everything: dense, missing values, KL): a rank does all the work of ITS columns over the whole contraction into a packed [k][cpr]
slab -> ONE ncclAllGather -> unpack.  "reduce" (dense square loss with nnlm_comm_set_form(NNLM_FORM_REDUCE), north_star's wording):
contraction-sharded [Gram | cross-product] partials -> ONE ncclAllReduce -> column-sharded solve -> ONE ncclAllGather -> unpack.  tests/dist_worker.py runs exactly that exchange with torch.distributed/gloo, taking every range from the product's own
partition functions (nnlm_shard_range, nnlm_shard_cols through the C ABI); the results must equal the unsharded oracle."""
import os
import socket
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import relF  # noqa: E402
from nnlm_amd import _lib  # noqa: E402
from oracle import ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_ranges_partition_the_contraction():
    for n, m in ((20000, 10000), (300, 170), (5, 3), (257, 129)):
        for prec in (0, 1):
            for which in (0, 1):
                ext = n if which == 1 else m
                for world in (1, 2, 3, 4, 8):
                    r = [_lib.shard_range(n, m, prec, which, rk, world) for rk in range(world)]
                    assert r[0][0] == 0 and r[-1][1] == ext
                    assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
                    assert all(b <= e for b, e in r)


def test_shard_cols_partition_the_columns():
    for ncols in (20000, 10000, 300, 170, 5, 257, 1):
        for world in (1, 2, 3, 4, 8):
            r = [_lib.shard_cols(ncols, rk, world) for rk in range(world)]
            cpr = r[0][0]
            assert all(x[0] == cpr for x in r) and cpr % 256 == 0 and cpr * world >= ncols
            assert r[0][1] == 0 and max(x[2] for x in r) == ncols
            for rk, (_, c0, c1) in enumerate(r):
                assert c0 == min(rk * cpr, ncols) and c1 == min(c0 + cpr, ncols)  # rank rk's slab starts at rk * cpr: what the unpack assumes


def _run_workers(tmp_path, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(out + f".rank{r}.npz") for r in range(world)]


def test_two_rank_sharded_half_steps_equal_unsharded_oracle(tmp_path):
    zs = _run_workers(tmp_path, 2)
    z0, z1 = zs
    rng = np.random.default_rng(42)
    n, m, k = 300, 170, 6
    A = rng.random((n, m))
    Wt, H = rng.random((k, n)), rng.random((k, m))
    reg = [0.02, 0.01, 0.03]
    for prec in (0, 1):
        for method in (1, 2):
            Wn_ref, it0 = ref.update(Wt, H, A.T.copy(), None, reg, 4, 1e-9, method)
            Hn_ref, it1 = ref.update(H, Wn_ref, A, None, reg, 4, 1e-9, method)
            for z in zs:
                assert relF(z[f"W_{prec}_{method}"], Wn_ref) < 1e-11 and relF(z[f"H_{prec}_{method}"], Hn_ref) < 1e-11
                assert list(z[f"it_{prec}_{method}"]) == [it0, it1]  # per-rank sweep counts, summed by the integer all-reduce
            # both ranks hold bit-identical factors after the all-gather; each contracted and solved a different share
            assert np.array_equal(z0[f"W_{prec}_{method}"], z1[f"W_{prec}_{method}"])
            assert np.array_equal(z0[f"H_{prec}_{method}"], z1[f"H_{prec}_{method}"])
            assert not np.array_equal(z0[f"rng_{prec}_{method}"], z1[f"rng_{prec}_{method}"])
            assert not np.array_equal(z0[f"cols_{prec}_{method}"], z1[f"cols_{prec}_{method}"])
    # column-sharded form: dense square loss (the default), missing values (per-column Grams) and the KL methods
    Ana = A.copy()
    Ana.ravel()[np.random.default_rng(7).choice(A.size, A.size // 10, replace=False)] = np.nan
    for tag, Amat, method, inner in (("dense1", A, 1, 4), ("dense2", A, 2, 4), ("na1", Ana, 1, 4), ("na2", Ana, 2, 4), ("kl3", A, 3, 2),
                                     ("kl4", A, 4, 2), ("nakl", Ana, 4, 1)):
        Wn_ref, it0 = ref.update(Wt, H, Amat.T.copy(), None, reg, inner, 1e-9, method)
        Hn_ref, it1 = ref.update(H, Wn_ref, Amat, None, reg, inner, 1e-9, method)
        for z in zs:
            assert relF(z[f"W_{tag}"], Wn_ref) < 1e-11 and relF(z[f"H_{tag}"], Hn_ref) < 1e-11, tag
            assert list(z[f"it_{tag}"]) == [it0, it1], tag
        assert np.array_equal(z0[f"W_{tag}"], z1[f"W_{tag}"]) and np.array_equal(z0[f"H_{tag}"], z1[f"H_{tag}"])
