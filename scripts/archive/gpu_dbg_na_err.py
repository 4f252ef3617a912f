"""Debug: the traced error of iteration 0 from the fused kernel (max_iter = 2) against the separate error kernel (max_iter = 1), NA flow."""
import numpy as np, sys
sys.path.insert(0, ".")
import nnlm_amd

def run(A, k, W0, H0, it):
    return nnlm_amd.c_nnmf(A, k, W0, H0, None, None, [0.0, 0, 0.0], [0.0, 0, 0.0], it, -1.0, 1, 0, False, 50, 1e-9, 1, 1)

rng = np.random.default_rng(1)
for (n, m, k, miss) in [(200, 100, 5, "rand"), (255, 128, 5, "rand"), (256, 127, 5, "rand"), (129, 128, 5, "rand"), (300, 190, 20, "rand"), (513, 130, 40, "rand"),
                        (777, 333, 50, "rand"), (200, 100, 5, "none")]:
    A = rng.random((n, m)); W0 = rng.random((n, k)); H0 = rng.random((k, m))
    A5 = A.copy()
    if miss == "rand": A5.ravel()[rng.choice(A5.size, A5.size // 10, replace=False)] = np.nan
    elif miss == "one": A5[3, 7] = np.nan
    elif miss == "none": pass
    elif miss.startswith("at"): A5[int(miss[2:].split(",")[0]), int(miss[2:].split(",")[1])] = np.nan
    elif miss.startswith("row"): A5[int(miss[3:]), :] = np.nan; A5[int(miss[3:]), 0] = 0.5
    else: A5[:, int(miss[3:])] = np.nan; A5[0, int(miss[3:])] = 0.5
    r1 = run(A5, k, W0, H0, 1)
    r2 = run(A5, k, W0, H0, 2)
    e1, e2 = r1["mse_error"][0], r2["mse_error"][0]
    print(n, m, k, miss, "separate", e1, "fused", e2, "rel", (e2 - e1) / e1, "abs*N", (e2 - e1) * np.isfinite(A5).sum())
