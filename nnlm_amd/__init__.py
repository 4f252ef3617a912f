"""nnlm_amd -- MI355X (gfx950) implementation of the nnmf()/nnlm() hot path of the R package NNLM.

Host-side mirror of the reference's R interface (nnmf, nnlm, predict_nnmf, mse_mkl) over the C ABI of
libnnlm_mi355x.so.  There is no CPU fallback: without the HIP library and a gfx950 device every
compute entry raises.
"""
from ._lib import Handle, NnlmError, PREC_F32, PREC_F64, c_nnlm, c_nnmf, comm_unique_id, load, make_callbacks  # noqa: F401
