# The two .Call wrappers the R layer uses (upstream: the generated R/RcppExports.R).  Same names and argument lists, so
# R/nnmf.R (c_nnmf, 17 arguments) and R/nnlm.R (c_nnlm, 9 arguments) call them unchanged; the routines are registered by
# R_init_NNLM in src/r_glue.c.

c_nnlm <- function(x, y, alpha, mask, beta0, max_iter, rel_tol, n_threads, method) {
	.Call(`_NNLM_c_nnlm`, x, y, alpha, mask, beta0, max_iter, rel_tol, n_threads, method)
	}

c_nnmf <- function(A, k, W, H, Wm, Hm, alpha, beta, max_iter, rel_tol, n_threads, verbose, show_warning,
	inner_max_iter, inner_rel_tol, method, trace) {
	.Call(`_NNLM_c_nnmf`, A, k, W, H, Wm, Hm, alpha, beta, max_iter, rel_tol, n_threads, verbose, show_warning,
		inner_max_iter, inner_rel_tol, method, trace)
	}
