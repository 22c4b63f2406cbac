"""Moore-Penrose pseudo-inverse on the GPU by Newton-Schulz iteration,

    X_0 = W^T / ||W||_F^2 ,   X_{k+1} = 2 X_k - X_k W X_k ,

which converges (quadratically, once the smallest eigenvalue of X_k W leaves 0) to W^+ for ANY W, needs nothing but GEMMs --
here the fp32-MFMA kernel of the C ABI -- and no host round trip except an occasional 2-float convergence check.
It is the on-device alternative to the reference's host call ``numpy.linalg.pinv`` (eval/matrix_approx_zeshel.py:47,49; LAPACK
SVD, rcond 1e-15).  Two routes live here: the fp64 iteration (pinv_newton_schulz_f64: the exact pseudo-inverse of the fp32 block,
rounded once -- what cur.py's default backend "auto" runs while the block is well conditioned; only backend "numpy", the same
LAPACK call as the reference, makes U bit-identical) and round 1's fp32 iteration (pinv_newton_schulz, backend "device32":
accuracy ~ cond(W) * 6e-8 like an fp32 SVD, singular values below ~1e-4 * sigma_max effectively truncated by the iteration cap).
The device routes matter for large anchor counts, where the host SVD dominates the index build (2048 x 1024: 1.4 s on the host, a
few ms here)."""
import torch

from . import ops


def pinv_newton_schulz(W, max_iters=80, check_every=4, rtol=1e-6):
	"""W: [m x n] fp32 / bf16 tensor on the GPU -> W^+ [n x m] fp32 on the GPU."""
	if W.dtype != torch.float32:
		W = ops.convert(W, torch.float32)
	m, n = W.shape
	fro = ops.sumsq(W)
	X = torch.empty((n, m), dtype=torch.float32, device=W.device)
	ops.scale_copy(W.t(), X, 1.0, divide_by=fro)            # X_0 = W^T / ||W||_F^2  (||W||_2^2 <= ||W||_F^2: contraction)
	Xn = torch.empty_like(X)
	tall = m >= n                                            # iterate with the smaller Gram side
	P = torch.empty((n, n) if tall else (m, m), dtype=torch.float32, device=W.device)
	norms = torch.zeros(2, dtype=torch.float32, device=W.device)
	prev = None
	for it in range(max_iters):
		if tall:
			ops.gemm(X, W, out=P)                            # P = X W            [n x n]
			ops.gemm(P, X, out=Xn, alpha=-1.0, beta=2.0, cin=X)   # X' = 2 X - P X
		else:
			ops.gemm(W, X, out=P)                            # P = W X            [m x m]
			ops.gemm(X, P, out=Xn, alpha=-1.0, beta=2.0, cin=X)   # X' = 2 X - X P
		X, Xn = Xn, X
		if it >= 8 and (it % check_every) == 0:              # ||X_k||_F grows monotonically towards ||W^+||_F
			cur = float(ops.sumsq(X, out=norms[0:1]).item())
			if prev is not None and abs(cur - prev) <= rtol * cur:
				break
			prev = cur
	return X


def _spectral_norm_sq(M, iters=8):
	"""||M||_2^2 by power iteration on M^T M (fp64 GEMVs on the device, one host read at the end).  A lower bound that is within
	a few per cent after 8 iterations for the spectra met here; callers add their own safety factor.
	M: [m x n] fp64.  The iterate is rescaled by 1 / ||.||^2 (a scaling is all power iteration needs; no square root kernel)."""
	m, n = M.shape
	v = torch.full((n, 1), 1.0 / n, dtype=torch.float64, device=M.device)
	v[::2] *= 1.5                                            # (not orthogonal to anything structured)
	t = torch.empty((m, 1), dtype=torch.float64, device=M.device)
	u = torch.empty((n, 1), dtype=torch.float64, device=M.device)
	nrm = torch.empty(2, dtype=torch.float64, device=M.device)
	for _ in range(iters):
		ops.gemm_f64(M, v, out=t)                            # t = M v
		ops.gemm_f64(M.t(), t, out=u)                        # u = M^T M v
		ops.diff_sumsq_f64(u, out=nrm)
		ops.convert_f64(u, v, 1.0, divide_by=nrm[1:2])       # v = u / ||u||^2
	ops.gemm_f64(M, v, out=t)
	num = float(ops.diff_sumsq_f64(t, out=nrm)[1].item())
	den = float(ops.diff_sumsq_f64(v, out=nrm)[1].item())
	return num / den if den > 0 else 0.0


def pinv_newton_schulz_f64(W, max_iters=64, check_from=4, rtol=1e-10, return_info=False):
	"""Parity-grade pseudo-inverse: Newton-Schulz in DOUBLE precision on the fp64 matrix cores (anncur_gemm_f64), the result
	rounded to fp32 once.  W: [m x n] fp32 / bf16 on the GPU -> W^+ [n x m] fp32 on the GPU.

	X_0 = W^T / (1.1 sigma_max^2) with sigma_max from a short power iteration (iterations ~ 2 log2 cond(W) + 4 instead of
	+ log2 rank with the Frobenius scaling).  In fp64 the iteration resolves singular values down to ~2^-(iters/2) of the
	largest, i.e. it converges to the exact pseudo-inverse of the fp32 matrix W for any conditioning the fp32 SVD of
	numpy.linalg.pinv can resolve; what separates the two is then numpy's own fp32 round-off (~cond(W) * 6e-8).  Stops when
	||X_k+1 - X_k||_F <= rtol ||X_k||_F (checked every second iteration: one 16-byte D2H each), followed by one more
	(quadratically convergent) step.  64 iterations resolve cond(W) up to ~1e8: a block that has not converged by then is singular
	to fp32 precision (numpy then inverts singular values that are fp32 round-off, and so would this iteration, to different
	noise) -- info["converged"] is False and the caller (cur._pinv) hands such a block to the host's LAPACK call, as the reference
	does.
	info: iterations, converged, cond_2 = ||W||_2 ||W^+||_2 (both by power iteration: the caller's handle on how far numpy's
	fp32 SVD can be trusted), cond_F."""
	m, n = W.shape
	Wd = torch.empty((m, n), dtype=torch.float64, device=W.device)
	ops.convert_f64(W, Wd)
	norms = ops.diff_sumsq_f64(Wd)                          # [0, ||W||_F^2]
	s2 = _spectral_norm_sq(Wd)
	w2 = float(norms[1].item())
	if not (s2 > 0.0):
		s2 = w2
	scale = torch.full((1,), min(1.1 * s2, w2) if w2 > 0 else 1.0, dtype=torch.float64, device=W.device)   # (||W||_F^2 always contracts)
	X = torch.empty((n, m), dtype=torch.float64, device=W.device)
	ops.convert_f64(Wd.t(), X, 1.0, divide_by=scale)        # X_0 = W^T / (1.1 sigma_max^2)
	Xn = torch.empty_like(X)
	tall = m >= n
	P = torch.empty((n, n) if tall else (m, m), dtype=torch.float64, device=W.device)
	chk = torch.empty(2, dtype=torch.float64, device=W.device)
	its, final = 0, False
	for it in range(max_iters):
		if tall:
			ops.gemm_f64(X, Wd, out=P)                                  # P = X W   [n x n]
			ops.gemm_f64(P, X, out=Xn, alpha=-1.0, beta=2.0, cin=X)     # X' = 2 X - P X
		else:
			ops.gemm_f64(Wd, X, out=P)                                  # P = W X   [m x m]
			ops.gemm_f64(X, P, out=Xn, alpha=-1.0, beta=2.0, cin=X)     # X' = 2 X - X P
		X, Xn = Xn, X
		its = it + 1
		if final:
			break
		if it >= check_from and (it - check_from) % 2 == 0:
			d2, x2 = ops.diff_sumsq_f64(X, Xn, out=chk).tolist()
			if not (x2 > 0.0) or not (x2 < 1e300) or d2 <= (rtol * rtol) * x2:
				final = x2 < 1e300
				if not final:
					break
	out = torch.empty((n, m), dtype=torch.float32, device=W.device)
	ops.convert_f64(X, out)
	if return_info:
		x2 = float(ops.diff_sumsq_f64(X, out=chk)[1].item())
		xs2 = _spectral_norm_sq(X) if final else float("inf")
		return out, {"iterations": its, "cond_F": (w2 * x2) ** 0.5, "cond_2": (s2 * xs2) ** 0.5, "converged": final}
	return out
