"""Moore-Penrose pseudo-inverse on the GPU by Newton-Schulz iteration,

    X_0 = W^T / ||W||_F^2 ,   X_{k+1} = 2 X_k - X_k W X_k ,

which converges (quadratically, once the smallest eigenvalue of X_k W leaves 0) to W^+ for ANY W, needs nothing but GEMMs --
here the fp32-MFMA kernel of the C ABI -- and no host round trip except an occasional 2-float convergence check.
It is the on-device alternative to the reference's host call ``numpy.linalg.pinv`` (eval/matrix_approx_zeshel.py:47,49; LAPACK
SVD, rcond 1e-15).  The default backend stays "numpy" because only the same LAPACK call makes U bit-identical to the reference;
the device backend matters for large anchor counts, where the host SVD dominates the index build (2048 x 1024: 1.4 s on the host,
a few ms here).  fp32 accuracy ~ cond(W) * 6e-8, like an fp32 SVD; rank-deficient W converges to the true pseudo-inverse, only
more slowly (singular values below ~1e-4 * sigma_max are effectively truncated by the iteration cap)."""
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
