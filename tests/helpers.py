import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def kat():
    return json.load(open(os.path.join(GOLDEN, "nnlm_kat.json")))


def kat_case1():
    c = kat()["case1"]
    A = np.array(c["A_colmajor"]).reshape(c["ncol"], c["nrow"]).T
    return A, np.array(c["b"], dtype=float), c["tolerance"]


def kat_case2():
    A, _, _ = kat_case1()
    c = kat()["case2"]
    b2 = np.array(c["b2_colmajor"], dtype=float).reshape(c["b2_ncol"], 5).T
    return A, b2, c["tolerance"]


def kat_case3():
    c = kat()["case3"]
    A2 = np.array(c["A2_colmajor"]).reshape(c["ncol"], c["nrow"]).T
    return A2, np.array(c["b3"], dtype=float), np.array(c["expected"]), c["tolerance"]


def relF(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def r_all_equal(a, b, tol=1.5e-8):
    """testthat::expect_equal for numerics: mean relative difference < tolerance."""
    a = np.asarray(a, dtype=float).ravel()
    b = np.asarray(b, dtype=float).ravel()
    xy = np.mean(np.abs(a - b))
    xn = np.mean(np.abs(b))
    return (xy / xn if xn > tol else xy) < tol
