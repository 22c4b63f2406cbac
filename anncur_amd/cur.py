"""CURApprox on the MI355X -- same constructor, attributes, methods and error behaviour as
the reference operator (eval/matrix_approx_zeshel.py:19-126), computed by the HIP kernels
behind include/anncur_hip.h.

    M (n x m)  ~=  C (n x kc) . U (kc x kr) . R (kr x m)

Differences from the reference, all deliberate:
  * the intersection check is the intended ``torch.equal`` (the reference's
    ``assert torch.eq(...)`` at :44 raises for any block larger than 1x1);
  * errors raise instead of opening an IPython shell;
  * tensors given on the CPU are moved to the GPU, results come back on the device of the
    tensor passed to the call (CPU in -> CPU out), so reference call sites run unchanged;
  * ``compute_dtype``: "fp32" (default for fp32 inputs: 1e-4 score parity with the CPU path)
    or "bf16" (default for bf16 inputs: the fused MFMA score+top-k kernel).
U = pinv(W) (:47,:49): ``pinv_backend="numpy"`` is the reference's own ``numpy.linalg.pinv`` call on the
host (U bit-identical); the default "auto" computes it on the GPU in fp64 (exact pseudo-inverse of the
fp32 block, within 1e-5 of numpy's on the golden cases) and keeps the host call for ill-conditioned
blocks, whose numpy result is an inverse of fp32 round-off.  Every product runs on the GPU.
"""
import logging

import numpy as np
import torch

from . import ops
from .ops import TopK

LOGGER = logging.getLogger(__name__)


def _is_sorted(idx_list):
	"""Strictly increasing (eval/matrix_approx_zeshel.py:53-55)."""
	a = np.asarray(idx_list)
	return bool(np.all(a[:-1] < a[1:]))


def _pinv_host(M):
	"""numpy.linalg.pinv (LAPACK SVD, rcond=1e-15) on the fp32 values, as the reference does."""
	return torch.from_numpy(np.linalg.pinv(M.detach().float().cpu().numpy()))


AUTO_COND_LIMIT = 1e3   # pinv_backend "auto": the device result is taken while cond_2(W) stays below this (numpy's fp32 SVD is then good to ~1e-4)
AUTO_COND_SAFETY = 1.25  # ... times this: the power-iteration estimate of cond_2 is a lower bound
AUTO_SQUARE_RATIO, AUTO_SQUARE_MIN = 0.9, 32   # "auto": blocks with min(shape) > 0.9 max(shape) (and at least 32 wide) go straight to the host


def _pinv(M, device, backend):
	"""backend "numpy": the reference's own call on the host (U bit-identical to the reference).
	backend "device": Newton-Schulz in fp64 on the GPU (anncur_amd/pinv.py), rounded to fp32 once: the exact pseudo-inverse of the
	  fp32 matrix; differs from numpy's fp32 LAPACK SVD by numpy's own round-off, ~cond(W) * 6e-8 (measured 1e-6..1e-5 on the
	  golden cases).  A block that is singular to fp32 precision (no convergence within the iteration budget) goes to the host
	  call: what numpy returns there is an inverse of round-off noise that no other algorithm reproduces.
	backend "auto": the device route, kept only where it is as good as pinned -- cond_2(W) = ||W||_2 ||W^+||_2 <= 1e3 (power iteration), where
	  the two agree far inside the 1e-4 score tolerance; an ill-conditioned block (e.g. as many anchor rows as anchor columns:
	  numpy inverts singular values that are fp32 noise, and the reference's numbers are made of that noise) goes to the host call.
	backend "device32": the fp32 iteration of round 1 (fast, ~1e-3)."""
	if backend == "numpy":
		return _pinv_host(M).to(device)
	if backend in ("device", "auto"):
		from .pinv import pinv_newton_schulz_f64
		if min(M.shape) == 0:
			return torch.zeros((M.shape[1], M.shape[0]), dtype=torch.float32, device=device)
		if backend == "auto" and min(M.shape) >= AUTO_SQUARE_MIN and min(M.shape) > AUTO_SQUARE_RATIO * max(M.shape):
			# (near-)square anchor block (entry A's n_ment_anchors == n_ent_anchors cells): its smallest singular value is ~1/n of the
			# largest, cond_2 far above the limit -- the iteration would burn its budget and end on the host anyway
			LOGGER.info("pinv auto: %d x %d block is (near-)square -> host numpy.linalg.pinv", M.shape[0], M.shape[1])
			return _pinv_host(M).to(device)
		# auto: a block with cond_2 <= 1e3 converges within ~2 log2(cond) + 6 = 26 steps; one that has not by 36 is beyond the limit
		X, info = pinv_newton_schulz_f64(M.to(device), return_info=True, max_iters=36 if backend == "auto" else 64)
		# cond_2 comes from two 8-step power iterations, each a LOWER bound (a few per cent): judged with a safety factor
		if info["converged"] and (backend == "device" or AUTO_COND_SAFETY * info["cond_2"] <= AUTO_COND_LIMIT):
			return X
		LOGGER.info("pinv %s: cond_2 = %.3g after %d iterations (converged: %s) -> host numpy.linalg.pinv", backend, info["cond_2"], info["iterations"], info["converged"])
		return _pinv_host(M).to(device)
	if backend == "device32":
		from .pinv import pinv_newton_schulz
		return pinv_newton_schulz(M.to(device))
	raise ValueError(f"pinv_backend = {backend} not supported")


def _is_full_range(idx, n):
	"""idx is exactly 0..n-1 in order (a permutation or a list with repeats is NOT the identity: the reference's
	latent_rows[row_idxs, :] honours order and duplicates)."""
	if torch.is_tensor(idx):
		idx = idx.detach().cpu().numpy()
	a = np.asarray(idx)
	return a.ndim == 1 and a.shape[0] == n and n > 0 and bool(np.array_equal(a, np.arange(n)))


class CURApprox(object):

	def __init__(self, rows, cols, row_idxs, col_idxs, approx_preference, A=None, compute_dtype=None, device=None, pinv_backend="auto"):
		super(CURApprox, self).__init__()
		self.pinv_backend = pinv_backend
		if device is None:
			device = rows.device if (torch.is_tensor(rows) and rows.is_cuda) else torch.device("cuda", torch.cuda.current_device())
		self.device = torch.device(device)
		self._home = rows.device if torch.is_tensor(rows) else torch.device("cpu")

		self.n = cols.shape[0]
		self.m = rows.shape[1]
		self.row_idxs = row_idxs
		self.col_idxs = col_idxs
		# device-resident operands (underscore names); the reference's attribute names C / R / U / latent_rows / latent_cols are
		# properties that hand out tensors on the device the caller's matrices live on (CPU in -> CPU attributes)
		self._C = self._to_dev(cols)  # n x kc
		self._R = self._to_dev(rows)  # kr x m
		self._home_cache = {}
		self.approx_preference = approx_preference
		if compute_dtype is None:
			compute_dtype = "bf16" if self._R.dtype == torch.bfloat16 else "fp32"
		if compute_dtype not in ("fp32", "bf16"):
			raise ValueError(f"compute_dtype = {compute_dtype} not supported")
		self.compute_dtype = compute_dtype

		assert _is_sorted(self.row_idxs), "row_idxs should be sorted"
		assert _is_sorted(self.col_idxs), "col_idxs should be sorted"
		assert len(row_idxs) == self._R.shape[0]
		assert len(col_idxs) == self._C.shape[1]

		intersect_mat = ops.gather_rows(self._C, row_idxs)  # kr x kc
		assert torch.equal(intersect_mat, ops.gather_cols(self._R, col_idxs)), \
			"Invalid rows and cols as their intersection does not match"

		if A is not None:  # oracle U = C^+ A R^+  (:46-47), products left to right on the GPU
			A_dev = self._to_dev(A)
			CpA = ops.gemm(_pinv(self._C, self.device, pinv_backend), A_dev)       # kc x m
			self._U = ops.gemm(CpA, _pinv(self._R, self.device, pinv_backend))     # kc x kr
		else:
			self._U = _pinv(intersect_mat, self.device, pinv_backend)             # kc x kr  (:49)

		self._Et = None   # [m x kc] fp32: latent_cols transposed ("rows" preference)
		self._Etp = None  # bf16 packed copy for the fused kernel
		self._Etp_sorted = self._item_ids = None
		self._latent_rows, self._latent_cols = self._build_latent_row_cols(self._C, self._U, self._R, self.approx_preference)

	# ------------------------------------------------------------------ the reference's attributes (matrix_approx_zeshel.py:36-51)
	def _attr(self, name):
		t = getattr(self, "_" + name)
		if t.device == self._home:
			return t
		if name not in self._home_cache:
			self._home_cache[name] = t.to(self._home)
		return self._home_cache[name]

	C = property(lambda self: self._attr("C"))
	R = property(lambda self: self._attr("R"))
	U = property(lambda self: self._attr("U"))
	latent_rows = property(lambda self: self._attr("latent_rows"))
	latent_cols = property(lambda self: self._attr("latent_cols"))

	# ------------------------------------------------------------------ helpers
	def _to_dev(self, t):
		if not torch.is_tensor(t):
			t = torch.as_tensor(np.asarray(t))
		if t.dtype not in (torch.float32, torch.bfloat16):
			t = t.float()
		return t.to(self.device)

	@staticmethod
	def _back(t, like):
		dev = like.device if torch.is_tensor(like) else torch.device("cpu")
		return t if t.device == dev else t.to(dev)

	@staticmethod
	def _is_sorted(idx_list):
		return _is_sorted(idx_list)

	def _build_latent_row_cols(self, C, U, R, approx_preference):
		if approx_preference == "cols":
			latent_rows = ops.gemm(C, U)  # n x kr
			latent_cols = R               # kr x m
		elif approx_preference == "rows":
			latent_rows = C               # n x kc
			# E^T = R^T U^T directly in the item-major layout the score kernels stream ([m x kc], rows contiguous)
			self._Et = ops.gemm(R.t(), U.t())
			latent_cols = self._Et.t()    # kc x m view (= U @ R)
			kp = ops.padded_k(self._Et.shape[1])
			if self.compute_dtype == "bf16" and kp is not None:
				self._Etp = ops.pack_bf16(self._Et, kp, row_multiple=32)               # item order: error terms, dense route
				self._Etp_sorted, self._item_ids = _norm_sorted_pack(self._Et, kp)     # norm order: fused top-k
		else:
			raise NotImplementedError(f"approx_preference = {approx_preference} not supported")
		return latent_rows, latent_cols

	def _take_rows(self, M, idx):
		return M if _is_full_range(idx, M.shape[0]) else ops.gather_rows(M, idx)

	def _take_cols(self, M, idx):
		if _is_full_range(idx, M.shape[1]):
			return M
		if M.stride(1) == 1 or M.shape[1] == 1:
			return ops.gather_cols(M, idx)
		return ops.gather_rows(M.t(), idx).t()  # column-major view (latent_cols = Et^T): gather item rows of Et

	# ------------------------------------------------------------------ reconstruction (a6)
	def get_rows(self, row_idxs):
		ans = ops.gemm(self._take_rows(self._latent_rows, row_idxs), self._latent_cols)
		return self._back(ans, self._home)

	def get_cols(self, col_idxs):
		ans = ops.gemm(self._latent_rows, self._take_cols(self._latent_cols, col_idxs))
		return self._back(ans, self._home)

	def get(self, row_idxs, col_idxs):
		ans = ops.gemm(self._take_rows(self._latent_rows, row_idxs), self._take_cols(self._latent_cols, col_idxs))
		return self._back(ans, self._home)

	def get_complete_col(self, sparse_cols):
		if self.approx_preference != "cols":
			raise NotImplementedError("This is not designed to give good approx of cols as U matrix is multiplied w/ R matrix. Build index w/ approx_preference = cols instead.")
		return self._back(ops.gemm(self._latent_rows, self._to_dev(sparse_cols)), sparse_cols)

	def topk_in_col(self, sparse_cols, k):
		if self.approx_preference != "cols":
			raise NotImplementedError("This is not designed to give good approx of cols as U matrix is multiplied w/ R matrix. Build index w/ approx_preference = cols instead.")
		dense = ops.gemm(self._latent_rows, self._to_dev(sparse_cols))
		v, i = ops.rowwise_topk(dense, k)
		return TopK(self._back(v, sparse_cols), self._back(i.long(), sparse_cols))

	def get_complete_row(self, sparse_rows):
		if self.approx_preference != "rows":
			raise NotImplementedError("This is not designed to give good approx of rows as C and U matrix are multiplied together. Build index w/ approx_preference = rows instead.")
		return self._back(ops.gemm(self._to_dev(sparse_rows), self._latent_cols), sparse_rows)

	# ------------------------------------------------------------------ retrieval (a6 + a7)
	def topk_in_row_device(self, sparse_rows, k):
		"""(values f32 [Q,k], indices int32 [Q,k]) on the GPU; S_hat is never materialised on the bf16 route."""
		if self.approx_preference != "rows":
			raise NotImplementedError("This is not designed to give good approx of rows as C and U matrix are multiplied together. Build index w/ approx_preference = rows instead.")
		X = self._to_dev(sparse_rows)
		Q = X.shape[0]
		if self._Etp is not None and ops.fused_supported(Q, self.m, self._Etp.shape[1], k):
			return ops.score_topk_fused(ops.pack_bf16(X, self._Etp.shape[1]), self._Etp_sorted, self.m, k, leading_sample=True, item_ids=self._item_ids)
		Et = self._Et if self.compute_dtype == "fp32" else (self._Etp[:self.m, :X.shape[1]] if self._Etp is not None else self._Et)
		if self.compute_dtype == "bf16" and X.dtype != torch.bfloat16:
			X = ops.convert(X, torch.bfloat16)
		return ops.score_topk_dense(X, Et, k)

	def topk_in_row(self, sparse_rows, k):
		v, i = self.topk_in_row_device(sparse_rows, k)
		return TopK(self._back(v, sparse_rows), self._back(i.long(), sparse_rows))

	def eval_rows(self, sparse_rows, exact_rows, k):
		"""One grid cell of entry point A for these query rows (crossenc.py:84,106,146-147): (approximate top-k of S_hat, per-row
		sum (S_hat - A)^2, per-row sum A^2).  On the bf16 route with Kp <= 256 and a bf16 exact matrix this is ONE sweep
		(ops.eval_fused: the kernel that streams the exact tile beside the MFMA chain also filters its accumulator for candidates);
		otherwise the two calls topk_in_row_device + approx_error_rows."""
		if self.approx_preference != "rows":
			raise NotImplementedError("This is not designed to give good approx of rows as C and U matrix are multiplied together. Build index w/ approx_preference = rows instead.")
		X, A = self._to_dev(sparse_rows), self._to_dev(exact_rows)
		if (self.compute_dtype == "bf16" and self._Etp is not None and ops.eval_fused_ok(self._Etp.shape[1], A, X.shape[0], self.m, k)):
			return ops.eval_fused(ops.pack_bf16(X, self._Etp.shape[1]), self._Etp, A, self.m, k, hint=self._Etp_sorted)
		approx = self.topk_in_row_device(sparse_rows, k)
		err, nrm = self.approx_error_rows(sparse_rows, exact_rows)
		return approx, err, nrm

	def approx_error_rows(self, sparse_rows, exact_rows):
		"""Per-row sum (S_hat - A)^2 and sum A^2 (a11) without materialising S_hat.  On the bf16 route S_hat is the one the retrieval
		ranks (bf16 item embeddings), computed on the sweep's MFMA loop."""
		X, A = self._to_dev(sparse_rows), self._to_dev(exact_rows)
		if self.compute_dtype == "bf16" and self._Etp is not None and ops.approx_error_packed_ok(self._Etp.shape[1], A):
			return ops.approx_error_packed(ops.pack_bf16(X, self._Etp.shape[1]), self._Etp, A, self.m)
		return ops.approx_error(X, self._Et, A)


def _norm_sorted_pack(Et, kp):
	"""(packed bf16 E^T with its rows ordered by descending norm, int32 map row -> item id): the index builder's hint for the fused
	top-k (anncur_score_topk_ex): items with a large ||E_i|| are the likeliest high scorers, so a threshold sampled from the leading
	rows -- and a sweep that meets them first -- lets far fewer elements through (-40 % on the synthetic protocol, whose norms vary
	by only 7 %).  The result is unchanged: the exact top-k of S_hat, reported with the original item ids."""
	order = ops.descending_norm_order(Et if Et.dtype == torch.float32 else ops.convert(Et, torch.float32))   # 256 norm buckets, stable inside a bucket
	return ops.pack_bf16(ops.gather_rows(Et, order), kp, row_multiple=32), order


class CURRowIndex(object):
	"""The "rows"-preference index alone: built from the anchor rows R [kr x m] and the anchor columns' ids, without the
	(n x kc) matrix of every query's anchor scores.  This is what a rank of a row-sharded evaluation holds: R assembled by one
	all-gather, U = pinv(R[:, col_idxs]) and E = U.R replicated, its own queries' anchor scores gathered locally.
	Same arithmetic as CURApprox(rows=R, cols=A[:, col_idxs], ...) (eval/matrix_approx_zeshel.py:42-65)."""

	def __init__(self, rows, col_idxs, compute_dtype=None, pinv_backend="auto"):
		self.R = rows
		self.m = rows.shape[1]
		self.col_idxs = col_idxs
		if compute_dtype is None:
			compute_dtype = "bf16" if rows.dtype == torch.bfloat16 else "fp32"
		self.compute_dtype = compute_dtype
		W = ops.gather_cols(rows, col_idxs)                      # kr x kc
		self.U = _pinv(W, rows.device, pinv_backend)             # kc x kr
		self._Et = ops.gemm(rows.t(), self.U.t())                # m x kc
		kp = ops.padded_k(self._Et.shape[1])
		self._Etp = self._Etp_sorted = self._item_ids = None
		if compute_dtype == "bf16" and kp is not None:
			self._Etp = ops.pack_bf16(self._Et, kp, row_multiple=32)
			self._Etp_sorted, self._item_ids = _norm_sorted_pack(self._Et, kp)

	def topk(self, X, k):
		"""X [q x kc]: the queries' exact scores against the anchor items -> (values f32, indices int32) on the GPU."""
		Q = X.shape[0]
		if self._Etp is not None and ops.fused_supported(Q, self.m, self._Etp.shape[1], k):
			return ops.score_topk_fused(ops.pack_bf16(X, self._Etp.shape[1]), self._Etp_sorted, self.m, k, leading_sample=True, item_ids=self._item_ids)
		Et = self._Et if self.compute_dtype == "fp32" or self._Etp is None else self._Etp[:self.m, :X.shape[1]]
		if self.compute_dtype == "bf16" and X.dtype != torch.bfloat16:
			X = ops.convert(X, torch.bfloat16)
		return ops.score_topk_dense(X, Et, k)

	def eval_topk(self, X, exact_rows, k, k_retvr):
		"""(exact top-k of exact_rows, approximate top-k_retvr of X) -- the two rankings of the reference's per-query loop
		(crossenc.py:97-106).  The exact scan (HBM-bound) runs on a second stream beside the retrieval (MFMA-bound) and is joined before
		the call returns; the retrieval is self.topk(), i.e. chunked over the queries above the workspace limit.  (ops.eval_topk's
		row-chunk co-scheduling is kept as an op but lost to this placement at every size measured: DESIGN 4.1 / bench.py --scan-mode.)"""
		dev = exact_rows.device
		Q = exact_rows.shape[0]
		ev = torch.empty((Q, k), dtype=torch.float32, device=dev)
		ei = torch.empty((Q, k), dtype=torch.int32, device=dev)   # (allocated on the launch stream: no record_stream needed after the join)
		main, side = torch.cuda.current_stream(dev), ops.aux_stream(dev)
		side.wait_stream(main)
		with torch.cuda.stream(side):
			exact = ops.rowwise_topk(exact_rows, k, out=(ev, ei))
		approx = self.topk(X, k_retvr)
		main.wait_stream(side)
		return exact, approx

	def eval_cell(self, X, exact_rows, k, k_retvr):
		"""Everything one grid cell of entry point A needs for these query rows (crossenc.py:97-106,146-147): (exact top-k of exact_rows,
		approximate top-k_retvr, per-row sum (S_hat - A)^2, per-row sum A^2).  The HBM-bound exact scan runs on a second stream beside
		ONE sweep that yields the candidates and the error sums (ops.eval_fused) where the fused route takes the cell; otherwise beside
		the retrieval, followed by the error kernel."""
		dev = exact_rows.device
		Q = exact_rows.shape[0]
		ev = torch.empty((Q, k), dtype=torch.float32, device=dev)
		ei = torch.empty((Q, k), dtype=torch.int32, device=dev)
		main, side = torch.cuda.current_stream(dev), ops.aux_stream(dev)
		side.wait_stream(main)
		with torch.cuda.stream(side):
			exact = ops.rowwise_topk(exact_rows, k, out=(ev, ei))
		if self.compute_dtype == "bf16" and self._Etp is not None and ops.eval_fused_ok(self._Etp.shape[1], exact_rows, Q, self.m, k_retvr):
			approx, err, nrm = ops.eval_fused(ops.pack_bf16(X, self._Etp.shape[1]), self._Etp, exact_rows, self.m, k_retvr, hint=self._Etp_sorted)
		else:
			approx = self.topk(X, k_retvr)
			err, nrm = self.approx_error_rows(X, exact_rows)
		main.wait_stream(side)
		return exact, approx, err, nrm

	def approx_error_rows(self, X, exact_rows):
		if self.compute_dtype == "bf16" and self._Etp is not None and ops.approx_error_packed_ok(self._Etp.shape[1], exact_rows):
			return ops.approx_error_packed(ops.pack_bf16(X, self._Etp.shape[1]), self._Etp, exact_rows, self.m)
		return ops.approx_error(X, self._Et, exact_rows)
