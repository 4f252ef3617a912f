"""Reader for the reference's on-disk data format (`data/nsclc.rda`: bzip2 + RDX2 XDR), SURVEY.md section 8f rank 4.

tests/golden/nsclc.rda is the reference's own data file (a DATA fixture: 200 genes x 100 patients of log2 expression,
used by vignettes/Fast-And-Versatile-NMF.Rmd:281-296), kept byte for byte."""
import bz2
import gzip
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import GOLDEN  # noqa: E402
from nnlm_amd import rda  # noqa: E402


def test_nsclc_decodes_to_the_200x100_expression_matrix():
    objs = rda.load_rda(os.path.join(GOLDEN, "nsclc.rda"))
    assert list(objs) == ["nsclc"]
    o = objs["nsclc"]
    A = o.matrix()
    assert A.shape == (200, 100) and A.dtype == np.float64 and A.flags["F_CONTIGUOUS"]
    assert np.isfinite(A).all() and A.min() > 0          # non-negative: valid input for both losses
    assert abs(A.min() - 2.59627) < 1e-5 and abs(A.max() - 14.10538) < 1e-5
    assert abs(A.mean() - 7.020149196) < 1e-9
    rows, cols = o.dimnames()
    assert len(rows) == 200 and len(cols) == 100
    assert rows[:3] == ["PTK2B", "CTNS", "POLE"] and cols[:3] == ["P001", "P002", "P003"] and cols[-1] == "P100"


def _xdr_matrix(name, M, names=None):
    """A minimal RDX2/XDR stream: pairlist(name = REALSXP with dim (+ dimnames))."""
    def i32(v): return struct.pack(">i", v)
    def charsxp(s): return i32(0x00040009) + i32(len(s)) + s.encode()
    def sym(s): return i32(1) + charsxp(s)
    def ints(v): return i32(13) + i32(len(v)) + b"".join(i32(x) for x in v)
    def strs(v): return i32(16) + i32(len(v)) + b"".join(charsxp(s) for s in v)
    body = i32(14 | 0x200) + i32(M.size) + np.asarray(M, dtype=">f8").ravel(order="F").tobytes()
    attrs = i32(2 | 0x400) + sym("dim") + ints(list(M.shape))
    if names is not None:
        attrs += i32(2 | 0x400) + sym("dimnames") + i32(19) + i32(2) + strs(names[0]) + i32(254)
    attrs += i32(254)
    return b"RDX2\nX\n" + i32(2) + i32(0x030202) + i32(0x020300) + i32(2 | 0x400) + sym(name) + body + attrs + i32(254)


@pytest.mark.parametrize("wrap", [lambda b: b, bz2.compress, gzip.compress])
def test_roundtrip_of_a_hand_built_stream_in_every_compression(wrap):
    M = np.arange(12, dtype=float).reshape(3, 4) / 7
    objs = rda.loads_rda(wrap(_xdr_matrix("m", M, names=(["a", "b", "c"], None))))
    assert np.array_equal(objs["m"].matrix(), M)
    assert objs["m"].dimnames() == [["a", "b", "c"], None]


def test_malformed_streams_fail_loudly():
    good = _xdr_matrix("m", np.ones((2, 2)))
    with pytest.raises(ValueError, match="not an RDX2"):
        rda.loads_rda(b"RDA2\nX\n" + good[7:])
    with pytest.raises(ValueError, match="XDR"):
        rda.loads_rda(b"RDX2\nA\n" + good[7:])
    with pytest.raises(ValueError, match="truncated"):
        rda.loads_rda(good[:-20])
    with pytest.raises(ValueError, match="no dim"):
        rda.RObject(np.ones(3)).matrix()
