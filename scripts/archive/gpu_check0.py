"""First GPU bring-up check (run via gpurun): half-steps and short runs vs the oracle, then a timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nnlm_amd
from nnlm_amd import _lib
from oracle import ref

def relF(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

rng = np.random.default_rng(1)
ok = True
for prec, pname, tol in ((_lib.PREC_F64, "f64", 1e-11), (_lib.PREC_F32, "f32", 2e-5)):
    for (n, m, k) in ((200, 100, 5), (300, 260, 17), (515, 131, 50), (260, 700, 64)):
        A = rng.random((n, m)); W0 = rng.random((n, k)) * 0.01; H0 = rng.random((k, m)) * 0.01
        for method in (1, 2):
            with nnlm_amd.Handle(0, prec) as h:
                h.set_matrix(A)
                h.set_factors(k, W0, H0)
                reg = [0.02, 0.01, 0.03]
                h.half_step(0, reg, 7, 1e-9, method)
                W1, H1 = h.get_factors()
                sw = h.take_sweeps()
                Wt_ref, it = ref.update(W0.T.copy(), H0, A.T.copy(), None, reg, 7, 1e-9, method)
                e1 = relF(W1, Wt_ref.T)
                h.half_step(1, reg, 7, 1e-9, method)
                W2, H2 = h.get_factors()
                sw2 = h.take_sweeps()
                H_ref, it2 = ref.update(H0, Wt_ref, A, None, reg, 7, 1e-9, method)
                e2 = relF(H2, H_ref)
                mse, kl, pen = h.errors()
                Ah = W2 @ H2
                mse_ref = float(np.mean((A - Ah) ** 2)); kl_ref = float(np.mean(-(A + 1e-16) * np.log(Ah + 1e-16) + Ah))
                good = e1 < tol and e2 < tol and abs(mse - mse_ref) < 1e-6 * mse_ref + 1e-12 and abs(kl - kl_ref) < 1e-5
                ok &= good
                print(f"{pname} n={n} m={m} k={k} method={method}: relF(W)={e1:.2e} relF(H)={e2:.2e} sweeps {sw}/{it} {sw2}/{it2} "
                      f"mse {mse:.8g}/{mse_ref:.8g} kl {kl:.8g}/{kl_ref:.8g} {'OK' if good else 'FAIL'}", flush=True)

# full driver vs oracle, config 1
n, m, k = 200, 100, 5
A = rng.random((n, m)); W0 = rng.random((n, k)) * 0.01; H0 = rng.random((k, m)) * 0.01
args = (A, k, W0, H0, None, None, [0, 0, 0], [0, 0, 0], 20, -1.0, 1, 0, True, 50, 1e-9, 1, 2)
for pname in ("f64", "f32"):
    os.environ["NNLM_PRECISION"] = pname
    r = nnlm_amd.c_nnmf(*args); o = ref.c_nnmf(*args)
    print(pname, "c_nnmf cfg1: relF W", relF(r["W"], o["W"]), "H", relF(r["H"], o["H"]), "mse", r["mse_error"][-1], o["mse_error"][-1],
          "epochs", r["average_epoch"][:3], o["average_epoch"][:3], "n_it", r["n_iteration"], o["n_iteration"], flush=True)

# timing at config 2
n, m, k = 20000, 10000, 50
rng = np.random.default_rng(20250928)
A = rng.random((n, m)); W0 = 0.01 * rng.random((n, k)); H0 = 0.01 * rng.random((k, m))
for prec, pname in ((_lib.PREC_F32, "f32"), (_lib.PREC_F64, "f64")):
    with nnlm_amd.Handle(0, prec) as h:
        t0 = time.time(); h.set_matrix(A); t1 = time.time()
        h.set_factors(k, W0, H0)
        print(pname, "upload+prep s", t1 - t0, h.matrix_info(), flush=True)
        z = [0, 0, 0]
        h.iterate(2, z, z, 50, 1e-9, 1); h.sync()
        h.profile_enable(True)
        t0 = time.time(); h.iterate(5, z, z, 50, 1e-9, 1); h.sync(); t1 = time.time()
        print(pname, "5 iterations s", t1 - t0, "it/s", 5 / (t1 - t0), "sweeps", h.take_sweeps() / (n + m) / 5)
        for nm in ("xprod_h", "xprod_w", "gram", "sweep_h", "sweep_w"):
            ms, cnt = h.profile_get(nm); print("   ", nm, ms / max(cnt, 1), "ms x", cnt)
        t0 = time.time(); mse, kl, pen = h.errors(); t1 = time.time()
        print(pname, "errors s", t1 - t0, mse, kl, flush=True)
print("ALL OK" if ok else "SOME FAILED")
