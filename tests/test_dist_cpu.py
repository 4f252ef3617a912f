"""World-size-2 gloo test of the multi-GPU path's shard arithmetic on CPU (no GPU needed).

GPU side (nnlm_amd/csrc/nnlm_mi355x.hip half_step): each rank contracts the slab nnlm_shard_range() gives it, folds its
split-K slabs into one [Gram | cross-product] buffer, ONE ncclAllReduce sums it, the sweep runs replicated.  Here the
same partition function drives a numpy/gloo restatement, and the result must equal the unsharded oracle."""
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


def test_two_rank_sharded_half_steps_equal_unsharded_oracle(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "res")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    z0, z1 = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    rng = np.random.default_rng(42)
    n, m, k = 300, 170, 6
    A = rng.random((n, m))
    Wt, H = rng.random((k, n)), rng.random((k, m))
    reg = [0.02, 0.01, 0.03]
    for prec in (0, 1):
        for method in (1, 2):
            Wn_ref, it0 = ref.update(Wt, H, A.T.copy(), None, reg, 4, 1e-9, method)
            Hn_ref, it1 = ref.update(H, Wn_ref, A, None, reg, 4, 1e-9, method)
            for z in (z0, z1):
                assert relF(z[f"W_{prec}_{method}"], Wn_ref) < 1e-11 and relF(z[f"H_{prec}_{method}"], Hn_ref) < 1e-11
                assert list(z[f"it_{prec}_{method}"]) == [it0, it1]
            # both ranks hold bit-identical factors after the all-reduce (replicated sweep)
            assert np.array_equal(z0[f"W_{prec}_{method}"], z1[f"W_{prec}_{method}"])
            assert not np.array_equal(z0[f"rng_{prec}_{method}"], z1[f"rng_{prec}_{method}"])
