"""Tensor-level wrappers over the C ABI.  torch is used only for device memory and the
current HIP stream; all arithmetic happens in libanncur_hip.so.

Every function takes CUDA(HIP) tensors and raises on CPU tensors: there is no CPU path.
"""
from collections import namedtuple
import ctypes
import functools
import itertools

import numpy as np
import torch

from . import _lib
from ._lib import F32, BF16, check

TopK = namedtuple("TopK", ["values", "indices"])

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _dt(t):
	try:
		return _DT[t.dtype]
	except KeyError:
		raise TypeError(f"unsupported dtype {t.dtype}: anncur_amd kernels take float32 or bfloat16") from None


def _dev(*ts):
	for t in ts:
		if not (torch.is_tensor(t) and t.is_cuda):
			raise _lib.AnncurHipError("anncur_amd ops need tensors on the GPU (cuda/HIP device); there is no CPU fallback")


def _stream():
	return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _on_device(fn):
	"""Run the op with its tensors' GPU as the current device: the launch stream (_stream()), the library's per-device caches
	(dynamic-LDS attribute, CU count) and torch's own allocations / fills inside the op then all belong to the device that holds
	the operands, whatever the process' current device is (CURApprox(device=...), --device cuda:N).  Operands on two different
	GPUs are an error."""
	@functools.wraps(fn)
	def wrapped(*args, **kwargs):
		dev = None
		for a in itertools.chain(args, kwargs.values()):
			if torch.is_tensor(a) and a.is_cuda:
				if dev is None:
					dev = a.device
				elif a.device != dev:
					raise _lib.AnncurHipError(f"{fn.__name__}: operands live on different devices ({dev} and {a.device})")
		if dev is None or dev.index == torch.cuda.current_device():
			return fn(*args, **kwargs)
		with torch.cuda.device(dev):
			return fn(*args, **kwargs)
	return wrapped


def _p(t):
	return ctypes.c_void_p(t.data_ptr())


def _rowmajor(t):
	"""2-D tensor with unit column stride (rows may be padded)."""
	if t.dim() != 2:
		raise ValueError("expected a 2-D tensor")
	if t.shape[1] > 1 and t.stride(1) != 1:
		t = t.contiguous()
	if t.shape[0] > 1 and t.stride(0) < t.shape[1]:
		t = t.contiguous()
	return t


def _ld(t):
	return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def as_index(idx, device, n=None):
	"""Python list / numpy / tensor of indices -> int32 device tensor.  With `n` (the size of the indexed dimension) host-side
	indices get torch's indexing semantics: negative values wrap, anything outside [-n, n) raises IndexError (the kernels' own
	clamp -- zeros for a bad index -- is only a safety net).  Indices that already live on the GPU are taken as they are: checking
	them would cost a device synchronisation."""
	if torch.is_tensor(idx) and idx.is_cuda:
		return idx.to(device=device, dtype=torch.int32).contiguous()
	a = idx.detach().numpy() if torch.is_tensor(idx) else np.asarray(idx)
	a = a.astype(np.int64, copy=False).reshape(-1)
	if n is not None and a.size:
		lo, hi = int(a.min()), int(a.max())
		if lo < -n or hi >= n:
			raise IndexError(f"index {lo if lo < -n else hi} is out of bounds for dimension with size {n}")
		if lo < 0:
			a = np.where(a < 0, a + n, a)
	return torch.as_tensor(a, dtype=torch.int32).to(device)


# ------------------------------------------------------------------ a2
@_on_device
def gather_cols(A, col_idx, out_dtype=None):
	"""A[:, col_idx]  (reference: eval/run_retrieval_eval_wrt_exact_crossenc.py:74)."""
	_dev(A)
	A = _rowmajor(A)
	idx = as_index(col_idx, A.device, A.shape[1])
	out = torch.empty((A.shape[0], idx.numel()), dtype=out_dtype or A.dtype, device=A.device)
	check(_lib.load().anncur_gather_cols(_p(A), _dt(A), A.shape[0], A.shape[1], _ld(A), _p(idx), idx.numel(), _p(out), _dt(out),
										 max(idx.numel(), 1), _stream()), "gather_cols")
	return out


@_on_device
def gather_rows(A, row_idx, out_dtype=None):
	"""A[row_idx, :]  (reference: eval/run_retrieval_eval_wrt_exact_crossenc.py:73)."""
	_dev(A)
	A = _rowmajor(A)
	idx = as_index(row_idx, A.device, A.shape[0])
	out = torch.empty((idx.numel(), A.shape[1]), dtype=out_dtype or A.dtype, device=A.device)
	n = idx.numel()
	for s in range(0, n, 65535):  # grid.y limit
		e = min(n, s + 65535)
		check(_lib.load().anncur_gather_rows(_p(A), _dt(A), A.shape[0], A.shape[1], _ld(A), _p(idx[s:e]), e - s, _p(out[s:e]), _dt(out),
											 max(A.shape[1], 1), _stream()), "gather_rows")
	return out


@_on_device
def convert(src, dtype):
	_dev(src)
	src = _rowmajor(src)
	out = torch.empty(src.shape, dtype=dtype, device=src.device)
	check(_lib.load().anncur_convert(_p(src), _dt(src), _ld(src), _p(out), _dt(out), max(out.shape[1], 1), src.shape[0], src.shape[1],
									 _stream()), "convert")
	return out


# ------------------------------------------------------------------ a4/a5/a6
@_on_device
def gemm(A, B, out=None, out_dtype=torch.float32, alpha=1.0, beta=0.0, cin=None):
	"""C = alpha * A @ B (+ beta * cin) for 2-D tensors with ARBITRARY strides (transposed views cost nothing);
	fp32 products and sums (exact fmaf chains on the matrix cores).  cin: fp32 [M, N], may be `out` itself."""
	_dev(A, B)
	M, K = A.shape
	K2, N = B.shape
	if K != K2:
		raise ValueError(f"gemm: inner dimensions differ ({K} vs {K2})")
	if out is None:
		out = torch.empty((M, N), dtype=out_dtype, device=A.device)
	elif tuple(out.shape) != (M, N):
		raise ValueError("gemm: out has the wrong shape")
	lib = _lib.load()
	for m0 in range(0, max(M, 1), 65535 * 128):  # grid.y limit of one launch
		m1 = min(M, m0 + 65535 * 128)
		a, c = A[m0:m1], out[m0:m1]
		if cin is None and alpha == 1.0:
			check(lib.anncur_gemm(_p(a), _dt(A), A.stride(0), A.stride(1), _p(B), _dt(B), B.stride(0), B.stride(1), _p(c), _dt(out),
								  out.stride(0), out.stride(1), m1 - m0, N, K, _stream()), "gemm")
		else:
			if cin is not None and (cin.dtype != torch.float32 or tuple(cin.shape) != (M, N)):
				raise ValueError("gemm: cin must be fp32 [M, N]")
			ci = cin[m0:m1] if cin is not None else None
			check(lib.anncur_gemm_ex(_p(a), _dt(A), A.stride(0), A.stride(1), _p(B), _dt(B), B.stride(0), B.stride(1), _p(c), _dt(out),
									 out.stride(0), out.stride(1), m1 - m0, N, K, float(alpha), float(beta),
									 _p(ci) if ci is not None else None, ci.stride(0) if ci is not None else 0, ci.stride(1) if ci is not None else 0,
									 _stream()), "gemm_ex")
	return out


@_on_device
def sumsq(A, out=None):
	"""Frobenius norm squared of an fp32 matrix -> 1-element device tensor (no host sync)."""
	_dev(A)
	A = _rowmajor(A)
	if A.dtype != torch.float32:
		raise TypeError("sumsq takes fp32")
	out = torch.empty(1, dtype=torch.float32, device=A.device) if out is None else out
	check(_lib.load().anncur_sumsq(_p(A), A.shape[0], A.shape[1], _ld(A), _p(out), _stream()), "sumsq")
	return out


@_on_device
def scale_copy(src, dst, alpha=1.0, divide_by=None):
	"""dst[i, j] = alpha / divide_by[0] * src[i, j] for fp32 2-D tensors with arbitrary strides (e.g. a scaled transpose)."""
	_dev(src, dst)
	if src.dtype != torch.float32 or dst.dtype != torch.float32 or tuple(src.shape) != tuple(dst.shape):
		raise ValueError("scale_copy: fp32 tensors of equal shape")
	check(_lib.load().anncur_scale_copy(_p(src), src.stride(0), src.stride(1), _p(dst), dst.stride(0), dst.stride(1), src.shape[0], src.shape[1],
										float(alpha), _p(divide_by) if divide_by is not None else None, _stream()), "scale_copy")
	return dst


# ------------------------------------------------------------------ fp64 helpers of the on-device pseudo-inverse
@_on_device
def gemm_f64(A, B, out=None, alpha=1.0, beta=0.0, cin=None):
	"""C = alpha * A @ B (+ beta * cin) in fp64 on the matrix cores, arbitrary strides."""
	_dev(A, B)
	if A.dtype != torch.float64 or B.dtype != torch.float64:
		raise TypeError("gemm_f64 takes float64 tensors")
	M, K = A.shape
	K2, N = B.shape
	if K != K2:
		raise ValueError(f"gemm_f64: inner dimensions differ ({K} vs {K2})")
	if out is None:
		out = torch.empty((M, N), dtype=torch.float64, device=A.device)
	check(_lib.load().anncur_gemm_f64(_p(A), A.stride(0), A.stride(1), _p(B), B.stride(0), B.stride(1), _p(out), out.stride(0), out.stride(1),
									  M, N, K, float(alpha), float(beta), _p(cin) if cin is not None else None,
									  cin.stride(0) if cin is not None else 0, cin.stride(1) if cin is not None else 0, _stream()), "gemm_f64")
	return out


_DT64 = {torch.float32: F32, torch.bfloat16: BF16, torch.float64: _lib.F64}


@_on_device
def convert_f64(src, dst, alpha=1.0, divide_by=None):
	"""dst[i, j] = alpha / divide_by[0] * src[i, j]; f32 / bf16 / f64 -> f64, or f64 -> f32 (one rounding); any strides."""
	_dev(src, dst)
	if tuple(src.shape) != tuple(dst.shape) or src.dim() != 2:
		raise ValueError("convert_f64: 2-D tensors of equal shape")
	check(_lib.load().anncur_convert_f64(_p(src), _DT64[src.dtype], src.stride(0), src.stride(1), _p(dst), _DT64[dst.dtype], dst.stride(0), dst.stride(1),
										 src.shape[0], src.shape[1], float(alpha), _p(divide_by) if divide_by is not None else None, _stream()), "convert_f64")
	return dst


@_on_device
def diff_sumsq_f64(X, Y=None, out=None):
	"""-> device tensor [sum (X - Y)^2, sum X^2] (float64, no host sync); X, Y contiguous float64 of equal size."""
	_dev(X)
	if X.dtype != torch.float64 or not X.is_contiguous() or (Y is not None and (Y.dtype != torch.float64 or not Y.is_contiguous() or Y.numel() != X.numel())):
		raise ValueError("diff_sumsq_f64: contiguous float64 tensors of equal size")
	out = torch.empty(2, dtype=torch.float64, device=X.device) if out is None else out
	check(_lib.load().anncur_diff_sumsq_f64(_p(X), _p(Y) if Y is not None else None, X.numel(), _p(out), _stream()), "diff_sumsq_f64")
	return out


@_on_device
def approx_error(X, Et, A_exact):
	"""Per-row sum_i (X.E - A)^2 and sum_i A^2 without materialising X.E
	(reference: eval/run_retrieval_eval_wrt_exact_crossenc.py:146-147)."""
	_dev(X, Et, A_exact)
	X, Et, A_exact = _rowmajor(X), _rowmajor(Et), _rowmajor(A_exact)
	Q, K = X.shape
	I = Et.shape[0]
	if Et.shape[1] != K or tuple(A_exact.shape) != (Q, I):
		raise ValueError("approx_error: shape mismatch")
	err = torch.empty(Q, dtype=torch.float32, device=X.device)
	nrm = torch.empty(Q, dtype=torch.float32, device=X.device)
	lib = _lib.load()
	step = 65535 * 128
	for q0 in range(0, Q, step):
		q1 = min(Q, q0 + step)
		check(lib.anncur_approx_error(_p(X[q0:q1]), _dt(X), _ld(X), _p(Et), _dt(Et), _ld(Et), _p(A_exact[q0:q1]), _dt(A_exact),
									  _ld(A_exact), q1 - q0, I, K, _p(err[q0:q1]), _p(nrm[q0:q1]), _stream()), "approx_error")
	return err, nrm


def approx_error_packed_ok(Kp, A_exact):
	"""True if anncur_approx_error_packed takes these operands (bf16 MFMA loop of the sweep instead of the strided fp32 GEMM)."""
	return (Kp in (64, 128, 256, 512) and A_exact.dim() == 2 and A_exact.stride(1) == 1 and _ld(A_exact) % 4 == 0
			and _ld(A_exact) >= A_exact.shape[1] and A_exact.data_ptr() % 16 == 0)


@_on_device
def approx_error_packed(Xp, Etp, A_exact, n_items):
	"""a11 on the fused path's operands: Xp [Q x Kp] packed bf16, Etp [ceil32(I) x Kp] packed bf16, A_exact [Q x I] fp32 / bf16."""
	_dev(Xp, Etp, A_exact)
	Q, Kp = Xp.shape
	if Etp.shape[1] != Kp or tuple(A_exact.shape) != (Q, n_items) or Etp.shape[0] < (n_items + 31) // 32 * 32:
		raise ValueError("approx_error_packed: shape mismatch")
	if Xp.dtype != torch.bfloat16 or Etp.dtype != torch.bfloat16 or not approx_error_packed_ok(Kp, A_exact):
		raise ValueError("approx_error_packed: operands must be packed bf16 and the exact matrix 16-byte aligned with a row pitch multiple of 4")
	err = torch.empty(Q, dtype=torch.float32, device=Xp.device)
	nrm = torch.empty(Q, dtype=torch.float32, device=Xp.device)
	check(_lib.load().anncur_approx_error_packed(_p(Xp), _ld(Xp), _p(Etp), _ld(Etp), _p(A_exact), _dt(A_exact), _ld(A_exact), Q, n_items, Kp,
												 _p(err), _p(nrm), _stream()), "approx_error_packed")
	return err, nrm


def eval_fused_ok(Kp, A_exact, Q, I, k):
	"""True if anncur_eval_fused takes this cell: Kp <= 256, bf16 exact matrix with 16-byte aligned rows, shape inside the fused path."""
	if not (Kp in (64, 128, 256) and A_exact.dim() == 2 and A_exact.dtype == torch.bfloat16 and A_exact.stride(1) == 1 and _ld(A_exact) % 8 == 0
			and _ld(A_exact) >= A_exact.shape[1] and A_exact.data_ptr() % 16 == 0):
		return False
	# the kernel's exact-tile offsets are 32-bit (255 rows x pitch x 2 bytes, csrc/score_fused.hip exact_tile_offsets_fit): longer pitches take the two-kernel route
	if 255 * _ld(A_exact) * 2 + 64 >= 1 << 32:
		return False
	# one workspace for the whole cell (no query chunks here): cells above the limit take the chunked two-kernel route
	nbytes = _lib.load().anncur_eval_fused_workspace_bytes(Q, I, Kp, k)
	return 0 < nbytes <= FUSED_WS_LIMIT_BYTES


@_on_device
def eval_fused(Xp, Etp, A_exact, n_items, k, return_fallbacks=False, hint=None):
	"""One sweep for a grid cell of entry point A (reference: eval/run_retrieval_eval_wrt_exact_crossenc.py:84,106,146-147):
	(TopK of S_hat = Xp . Etp^T, err_sq [Q], norm_sq [Q]).  Xp [Q x Kp], Etp [ceil32(I) x Kp] packed bf16 in ITEM order, A_exact [Q x I] bf16.
	hint: a copy of Etp's rows in descending-norm order (same shape): the prepass samples ITS leading tiles -- a tighter first threshold, same results."""
	_dev(Xp, Etp, A_exact)
	Q, Kp = Xp.shape
	if Xp.dtype != torch.bfloat16 or Etp.dtype != torch.bfloat16 or Etp.shape[1] != Kp or not Etp.is_contiguous() or Etp.shape[0] < -(-n_items // 32) * 32 \
			or tuple(A_exact.shape) != (Q, n_items):
		raise ValueError("eval_fused: Xp [Q x Kp], Etp [ceil32(I) x Kp] packed bf16, A_exact [Q x I]")
	if not eval_fused_ok(Kp, A_exact, Q, n_items, k):
		raise _lib.AnncurHipError(f"eval_fused: cell (Q={Q}, I={n_items}, Kp={Kp}, k={k}, A {A_exact.dtype}) is outside the one-pass route")
	Xp = _rowmajor(Xp)
	lib = _lib.load()
	nbytes = lib.anncur_eval_fused_workspace_bytes(Q, n_items, Kp, k)
	ws = _Workspace.get(nbytes, Xp.device)
	val = torch.empty((Q, k), dtype=torch.float32, device=Xp.device)
	idx = torch.empty((Q, k), dtype=torch.int32, device=Xp.device)
	err = torch.empty(Q, dtype=torch.float32, device=Xp.device)
	nrm = torch.empty(Q, dtype=torch.float32, device=Xp.device)
	if hint is not None:
		_dev(hint)
		if hint.dtype != torch.bfloat16 or tuple(hint.shape) != tuple(Etp.shape) or not hint.is_contiguous():
			raise ValueError("eval_fused: hint must be a contiguous bf16 copy of Etp's rows (same shape)")
	check(lib.anncur_eval_fused_ex(_p(Xp), _ld(Xp), _p(Etp), Kp, _p(hint) if hint is not None else None, _p(A_exact), _dt(A_exact), _ld(A_exact), Q, n_items, Kp, k,
								   _p(val), _p(idx), _p(err), _p(nrm), _p(ws), nbytes, _stream()), "eval_fused")
	if return_fallbacks:
		return TopK(val, idx), err, nrm, ws[:4].view(torch.int32)
	return TopK(val, idx), err, nrm


# ------------------------------------------------------------------ a7/a8
@_on_device
def rowwise_topk(A, k, out=None):
	"""Exact torch.topk(A, k, dim=1) on the device: (values f32 [Q,k], indices int32 [Q,k]),
	sorted descending, ties -> smaller index.  out: (values, indices) contiguous [Q, k] tensors to write into (row slices of a larger
	result, when a matrix is scanned in several launches)."""
	_dev(A)
	A = _rowmajor(A)
	Q, I = A.shape
	if out is not None:
		val, idx = out
		if tuple(val.shape) != (Q, k) or tuple(idx.shape) != (Q, k) or val.dtype != torch.float32 or idx.dtype != torch.int32 or not val.is_contiguous() or not idx.is_contiguous():
			raise ValueError("rowwise_topk: out must be contiguous (float32 [Q, k], int32 [Q, k])")
	else:
		val = torch.empty((Q, k), dtype=torch.float32, device=A.device)
		idx = torch.empty((Q, k), dtype=torch.int32, device=A.device)
	check(_lib.load().anncur_rowwise_topk(_p(A), _dt(A), Q, I, _ld(A), k, _p(val), _p(idx), _stream()), "rowwise_topk")
	return TopK(val, idx)


@_on_device
def rowwise_topk_ragged(A, row_len, k):
	"""rowwise_topk over ragged rows: row q of A holds row_len[q] (int32 tensor, k <= row_len[q] <= A.shape[1]) elements; k <= 128."""
	_dev(A, row_len)
	A = _rowmajor(A)
	Q, I = A.shape
	if row_len.dtype != torch.int32 or row_len.numel() != Q or not row_len.is_contiguous():
		raise ValueError("rowwise_topk_ragged: row_len must be a contiguous int32 tensor with one entry per row")
	val = torch.empty((Q, k), dtype=torch.float32, device=A.device)
	idx = torch.empty((Q, k), dtype=torch.int32, device=A.device)
	check(_lib.load().anncur_rowwise_topk_ragged(_p(A), _dt(A), Q, I, _ld(A), _p(row_len), k, _p(val), _p(idx), _stream()), "rowwise_topk_ragged")
	return TopK(val, idx)


GatherTables = namedtuple("GatherTables", ["col_idx", "vec_tab", "n_items", "dtype"])


@_on_device
def gather_tables(col_idx, n_items, dtype):
	"""The per-anchor-set tables of rowwise_topk_gather: col_idx ascending, distinct columns (int32 / int64 tensor on the GPU)."""
	_dev(col_idx)
	if col_idx.dim() != 1 or col_idx.numel() < 1 or col_idx.numel() > 65535:
		raise ValueError("gather_tables: 1..65535 anchor columns")
	ci = col_idx.to(torch.int32).contiguous()
	if bool((ci[1:] <= ci[:-1]).any()) or int(ci[0]) < 0 or int(ci[-1]) >= n_items:
		raise ValueError("gather_tables: columns must be ascending, distinct and inside the matrix")
	vec = 8 if dtype == torch.bfloat16 else 4
	n_vec = -(-n_items // vec)
	tab = torch.empty(n_vec, dtype=torch.int32, device=ci.device)
	check(_lib.load().anncur_gather_tables(_p(ci), ci.numel(), n_items, BF16 if dtype == torch.bfloat16 else F32, _p(tab), _stream()), "gather_tables")
	return GatherTables(ci, tab, n_items, dtype)


def rowwise_topk_gather_ok(A, k):
	"""True if rowwise_topk_gather takes this matrix: wave-level scan (k <= 128), 16-byte aligned rows."""
	return (k <= 128 and A.dim() == 2 and A.stride(1) == 1 and A.data_ptr() % 16 == 0
			and (A.shape[0] == 1 or (_ld(A) * A.element_size()) % 16 == 0) and A.dtype in (torch.float32, torch.bfloat16))


@_on_device
def rowwise_topk_gather(A, k, tables, out=None, cq_out=None):
	"""rowwise_topk(A, k) and C_q = A[:, tables.col_idx] from ONE pass over A (reference: the slice test_scores[:, anchor_ent_idxs] of
	..._splits.py:297,300 folded into the exact top-k of :86).  Returns (TopK, C_q [Q x n_idx] of A's dtype)."""
	_dev(A)
	A = _rowmajor(A)
	Q, I = A.shape
	if tables.n_items != I or tables.dtype != A.dtype or not rowwise_topk_gather_ok(A, k):
		raise ValueError("rowwise_topk_gather: tables built for another matrix shape / dtype, k > 128, or rows not 16-byte aligned")
	n_idx = tables.col_idx.numel()
	if out is not None:
		val, idx = out
		if (tuple(val.shape) != (Q, k) or tuple(idx.shape) != (Q, k) or val.dtype != torch.float32 or idx.dtype != torch.int32
				or not val.is_contiguous() or not idx.is_contiguous() or val.device != A.device or idx.device != A.device):
			raise ValueError("rowwise_topk_gather: out must be contiguous (float32 [Q, k], int32 [Q, k]) on A's device")
	else:
		val = torch.empty((Q, k), dtype=torch.float32, device=A.device)
		idx = torch.empty((Q, k), dtype=torch.int32, device=A.device)
	cq = cq_out if cq_out is not None else torch.empty((Q, n_idx), dtype=A.dtype, device=A.device)
	if tuple(cq.shape) != (Q, n_idx) or cq.dtype != A.dtype or (n_idx > 1 and cq.stride(1) != 1) or cq.device != A.device or (Q > 1 and cq.stride(0) < n_idx):
		raise ValueError("rowwise_topk_gather: cq_out must be [Q x n_idx] of A's dtype on A's device, unit column stride")
	check(_lib.load().anncur_rowwise_topk_gather(_p(A), _dt(A), Q, I, _ld(A), k, _p(val), _p(idx), _p(tables.col_idx), n_idx,
												  _p(tables.vec_tab), _p(cq), _ld(cq), _stream()), "rowwise_topk_gather")
	return TopK(val, idx), cq


_KP_CHOICES = (64, 128, 256, 512)   # query operand resident in registers (score_kernel)
_KP_WIDE_MAX = 4096                  # beyond: LDS-tiled K-general kernel (wide_kernel), Kp a multiple of 128


def padded_k(K):
	"""Inner dimension the fused kernels take for a logical K (zero-padded), or None if K is too wide for them."""
	for kp in _KP_CHOICES:
		if K <= kp:
			return kp
	kp = -(-K // 128) * 128
	return kp if kp <= _KP_WIDE_MAX else None


def _kp_ok(Kp):
	return Kp in _KP_CHOICES or (512 < Kp <= _KP_WIDE_MAX and Kp % 128 == 0)


@_on_device
def pack_bf16(M, Kp, row_multiple=1):
	"""[n x K] (f32/bf16) -> zero-padded bf16 [ceil(n/row_multiple)*row_multiple x Kp], packed."""
	_dev(M)
	n, K = M.shape
	n_pad = -(-n // row_multiple) * row_multiple
	out = torch.zeros((n_pad, Kp), dtype=torch.bfloat16, device=M.device)
	M = _rowmajor(M)
	view = out[:n, :K]
	check(_lib.load().anncur_convert(_p(M), _dt(M), _ld(M), _p(view), BF16, Kp, n, K, _stream()), "pack_bf16")
	return out


FUSED_WS_LIMIT_BYTES = 32 << 30   # default workspace above this size -> score_topk_fused runs the queries in row chunks


class _Workspace:
	"""Grow-only device scratch for the fused kernel (one per device)."""
	_bufs = {}

	@classmethod
	def get(cls, nbytes, device):
		key = (device.type, device.index)
		buf = cls._bufs.get(key)
		if buf is None or buf.numel() < nbytes:
			cls._bufs[key] = buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
		off = (-buf.data_ptr()) % 256
		return buf[off:off + nbytes]


class _ScoreScratch:
	"""Grow-only fp32 scratch per device (the batched IVF search's score matrix)."""
	_bufs = {}

	@classmethod
	def get(cls, n, device):
		key = (device.type, device.index)
		buf = cls._bufs.get(key)
		if buf is None or buf.numel() < n:
			cls._bufs[key] = buf = torch.empty(n, dtype=torch.float32, device=device)
		return buf[:n]


def fused_supported(Q, I, Kp, k):
	return _kp_ok(Kp) and bool(_lib.load().anncur_score_topk_supported(Q, I, Kp, k))


def fused_workspace(Q, I, Kp, k, device):
	"""A private workspace for score_topk_fused (256-byte aligned uint8 tensor): calls that may run concurrently on different
	streams must not share the default one."""
	nbytes = _lib.load().anncur_score_topk_workspace_bytes(Q, I, Kp, k)
	if nbytes == 0:
		raise _lib.AnncurHipError(f"score_topk: shape (Q={Q}, I={I}, Kp={Kp}, k={k}) is outside the fused path")
	buf = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
	off = (-buf.data_ptr()) % 256
	return buf[off:off + nbytes]


def _item_ids_arg(item_ids, I, device):
	if item_ids is None:
		return None
	if item_ids.dtype != torch.int32 or item_ids.numel() != I or not item_ids.is_contiguous() or item_ids.device != device:
		raise ValueError("item_ids must be a contiguous int32 device tensor with one id per item")
	return item_ids


def _topk_flags(leading_sample=False, mfma16=False, qt1=False, mfma32=False, ring=False, staged=False):
	"""The flags word of anncur_score_topk_ex / _timed / _plan_ex."""
	return ((_lib.TOPK_LEADING_SAMPLE if leading_sample else 0) | (_lib.TOPK_MFMA16 if mfma16 else 0) | (_lib.TOPK_QT1 if qt1 else 0)
			| (_lib.TOPK_MFMA32 if mfma32 else 0) | (_lib.TOPK_RING if ring else 0) | (_lib.TOPK_STAGED if staged else 0))


@_on_device
def score_topk_fused(Xp, Etp, I, k, return_fallbacks=False, workspace=None, leading_sample=False, item_ids=None, mfma16=False, qt1=False, mfma32=False, ring=False, staged=False):
	"""Fused S_hat = X.E + top-k.  Xp [Q x Kp] bf16 packed, Etp [Ip x Kp] bf16 packed (see pack_bf16).
	workspace: from fused_workspace(); default = one grow-only buffer per device (one call in flight at a time).
	leading_sample / item_ids: the index builder's hints of anncur_score_topk_ex (rows of Etp ordered by descending norm, and the
	map from rows back to item ids); the result is the exact top-k either way.
	mfma16 / mfma32 / qt1: the sweep variants ANNCUR_TOPK_MFMA16 / _MFMA32 / _QT1 (fused_plan(..., mfma16=, ...) tells whether the shape
	takes them: "lg" == 1 / "lg" == 2 / "QT" == 1; the default for Kp <= 256 is the 16x16x32 body up to k = 128, 32x32x16 above)."""
	_dev(Xp, Etp)
	if Xp.dtype != torch.bfloat16 or Etp.dtype != torch.bfloat16:
		raise TypeError("score_topk_fused takes bf16 operands")
	Q, Kp = Xp.shape
	if Etp.shape[1] != Kp or not Etp.is_contiguous() or Etp.shape[0] < -(-I // 32) * 32:
		raise ValueError("Et must be packed [ceil(I/32)*32 x Kp]")
	Xp = _rowmajor(Xp)
	lib = _lib.load()
	nbytes = lib.anncur_score_topk_workspace_bytes(Q, I, Kp, k)
	if nbytes == 0:
		raise _lib.AnncurHipError(f"score_topk: shape (Q={Q}, I={I}, Kp={Kp}, k={k}) is outside the fused path")
	if workspace is None and nbytes > FUSED_WS_LIMIT_BYTES and Q > 512 and not return_fallbacks:
		# very many queries: the candidate segments grow with Q; run row chunks whose workspace stays under the limit
		qc = Q
		while qc > 512 and lib.anncur_score_topk_workspace_bytes(qc, I, Kp, k) > FUSED_WS_LIMIT_BYTES:
			qc = max(512, (qc // 2 + 255) // 256 * 256)
		val = torch.empty((Q, k), dtype=torch.float32, device=Xp.device)
		idx = torch.empty((Q, k), dtype=torch.int32, device=Xp.device)
		for q0 in range(0, Q, qc):   # (a chunk of 512 queries is accepted whatever its workspace size)
			part = score_topk_fused(Xp[q0:q0 + qc], Etp, I, k, leading_sample=leading_sample, item_ids=item_ids, mfma16=mfma16, qt1=qt1, mfma32=mfma32, ring=ring, staged=staged)
			val[q0:q0 + qc], idx[q0:q0 + qc] = part.values, part.indices
		return TopK(val, idx)
	if workspace is None:
		ws = _Workspace.get(nbytes, Xp.device)
	else:
		ws = workspace
		if ws.numel() < nbytes or ws.data_ptr() % 256 != 0 or ws.device != Xp.device:
			raise ValueError("score_topk_fused: workspace too small, misaligned or on another device (use fused_workspace())")
	val = torch.empty((Q, k), dtype=torch.float32, device=Xp.device)
	idx = torch.empty((Q, k), dtype=torch.int32, device=Xp.device)
	ids = _item_ids_arg(item_ids, I, Xp.device)
	check(lib.anncur_score_topk_ex(_p(Xp), _ld(Xp), _p(Etp), Kp, Q, I, Kp, k, _p(val), _p(idx), _p(ws), nbytes,
								   _topk_flags(leading_sample, mfma16, qt1, mfma32, ring, staged), _p(ids) if ids is not None else None, _stream()), "score_topk")
	if ring and not torch.cuda.is_current_stream_capturing():
		# ANNCUR_TOPK_RING (opt-in): a wave whose bounded spin on the tile ring's flags ran out stopped waiting and added 2^30 to the call's
		# fallback counter (csrc/score16r.hpp); its candidates are then not to be trusted.  Fatal here (one host sync, the variant is opt-in).
		if int(ws[:4].view(torch.int32).item()) >= 1 << 30:
			raise _lib.AnncurHipError("score_topk (ring body): a wave gave up waiting on the tile ring; the result of this call is invalid")
	if return_fallbacks:
		return TopK(val, idx), ws[:4].view(torch.int32)
	return TopK(val, idx)


_aux_streams = {}


def aux_stream(device):
	"""The second stream anncur_eval_topk forks the exact scan's row chunks onto (one per device, created on first use)."""
	key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
	if key not in _aux_streams:
		_aux_streams[key] = torch.cuda.Stream(device=device)
	return _aux_streams[key]


_cu_streams = {}


def cu_partition_streams(device, n_scan):
	"""Two streams that split the chip's compute units: (retrieval stream on the CUs the scan leaves, scan stream on `n_scan` CUs), by
	hipExtStreamCreateWithCUMask.  On MI355X the first n bits of the mask stand for n / 8 CUs on each of the 8 XCDs, in steps of 32
	bits (scripts/cumask_map.py; sparse masks are ignored by the driver), so n_scan is rounded down to a multiple of 32.  For an
	HBM-bound kernel (the exact scan) beside an MFMA-bound one (the fused retrieval) whose workgroups fill the register file: sharing a
	CU means time-slicing it, a partition lets both run for the whole step.  Created once per (device, n_scan); fails loudly if the
	runtime lacks the call."""
	device = torch.device(device)
	idx = device.index if device.index is not None else torch.cuda.current_device()
	# hipExtStreamCreateWithCUMask makes BLOCKING streams: work on the NULL stream synchronises with them implicitly, and graph replays on a
	# masked stream right after NULL-stream work ended in a GPU memory access fault when RCCL was in the process (round 4; never
	# reproduced without RCCL: scripts/r5/cumask_null_stream_repro.hip).  A caller whose current stream is the NULL stream is one torch op away
	# from that: refused here.  Run under `with torch.cuda.stream(torch.cuda.Stream()):` (bench.py does).
	if torch.cuda.current_stream(idx).cuda_stream == 0:
		raise _lib.AnncurHipError("cu_partition_streams: the current stream is the NULL stream, which synchronises implicitly with CU-masked (blocking) streams; "
								  "make a non-default stream current first (with torch.cuda.stream(torch.cuda.Stream()): ...)")
	n_cu = torch.cuda.get_device_properties(idx).multi_processor_count
	n_scan = (int(n_scan) // 32) * 32
	if not (0 < n_scan < n_cu):
		raise ValueError(f"cu_partition_streams: n_scan={n_scan} must leave CUs on both sides of {n_cu}")
	key = (idx, n_scan)
	if key not in _cu_streams:
		hip = ctypes.CDLL("libamdhip64.so")
		fn = hip.hipExtStreamCreateWithCUMask
		fn.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
		words = (n_cu + 31) // 32
		def make(lo, hi):
			mask = (ctypes.c_uint32 * words)(*[sum(1 << b for b in range(32) if lo <= 32 * w + b < hi) for w in range(words)])
			h = ctypes.c_void_p()
			with torch.cuda.device(idx):
				rc = fn(ctypes.byref(h), words, mask)
			if rc != 0 or not h.value:
				raise _lib.AnncurHipError(f"hipExtStreamCreateWithCUMask failed (code {rc})")
			return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", idx))
		_cu_streams[key] = (make(n_scan, n_cu), make(0, n_scan))
	return _cu_streams[key]


@_on_device
def eval_topk(A, k, Xp, Etp, I, k_retvr, workspace=None, leading_sample=False, item_ids=None, mfma16=False, qt1=False, aux=None, serial=False, mfma32=False, ring=False, staged=False):
	"""The per-query evaluation loop's two top-k's in one call (reference: eval/run_retrieval_eval_wrt_exact_crossenc.py:97-106):
	(exact = rowwise_topk(A, k), approx = score_topk_fused(Xp, Etp, I, k_retvr)), the exact scan's row chunks co-scheduled with the
	retrieval's latency-bound launches on a second stream (anncur_eval_topk).  serial=True: the same two results, one after the other."""
	_dev(A, Xp, Etp)
	if Xp.dtype != torch.bfloat16 or Etp.dtype != torch.bfloat16:
		raise TypeError("eval_topk takes bf16 retrieval operands")
	A = _rowmajor(A)
	Q, Kp = Xp.shape
	if A.shape[0] != Q or A.shape[1] != I or Etp.shape[1] != Kp or not Etp.is_contiguous() or Etp.shape[0] < -(-I // 32) * 32:
		raise ValueError("eval_topk: A must be [Q x I], Et packed [ceil(I/32)*32 x Kp]")
	Xp = _rowmajor(Xp)
	lib = _lib.load()
	nbytes = lib.anncur_score_topk_workspace_bytes(Q, I, Kp, k_retvr)
	if nbytes == 0:
		raise _lib.AnncurHipError(f"eval_topk: shape (Q={Q}, I={I}, Kp={Kp}, k={k_retvr}) is outside the fused path")
	ws = _Workspace.get(nbytes, Xp.device) if workspace is None else workspace
	if ws.numel() < nbytes or ws.data_ptr() % 256 != 0 or ws.device != Xp.device:
		raise ValueError("eval_topk: workspace too small, misaligned or on another device (use fused_workspace())")
	ev = torch.empty((Q, k), dtype=torch.float32, device=A.device)
	ei = torch.empty((Q, k), dtype=torch.int32, device=A.device)
	av = torch.empty((Q, k_retvr), dtype=torch.float32, device=A.device)
	ai = torch.empty((Q, k_retvr), dtype=torch.int32, device=A.device)
	ids = _item_ids_arg(item_ids, I, Xp.device)
	if serial:
		aux_p = None
	else:
		# (the chunks forked onto the auxiliary stream are joined back into the launch stream before the call returns: every tensor used
		#  there is ordered like launch-stream work, the caching allocator needs no record_stream)
		aux = aux or aux_stream(A.device)
		aux_p = ctypes.c_void_p(aux.cuda_stream)
	check(lib.anncur_eval_topk(_p(A), _dt(A), _ld(A), k, _p(ev), _p(ei), _p(Xp), _ld(Xp), _p(Etp), Kp, Q, I, Kp, k_retvr, _p(av), _p(ai), _p(ws), nbytes,
							   _topk_flags(leading_sample, mfma16, qt1, mfma32, ring, staged), _p(ids) if ids is not None else None, _stream(), aux_p), "eval_topk")
	return TopK(ev, ei), TopK(av, ai)


@_on_device
def score_topk_fused_timed(Xp, Etp, I, k, leading_sample=False, item_ids=None, mfma16=False, qt1=False, mfma32=False, ring=False, staged=False):
	"""Measurement only: (TopK, [prepass, threshold, sweep stage, select, sweep kernels only, n sweep launches, sweep launch 1, 2, 3]) in
	ms, from HIP events on the launch stream."""
	_dev(Xp, Etp)
	Q, Kp = Xp.shape
	lib = _lib.load()
	nbytes = lib.anncur_score_topk_workspace_bytes(Q, I, Kp, k)
	if nbytes == 0:
		raise _lib.AnncurHipError("score_topk_timed: unsupported shape")
	ws = _Workspace.get(nbytes, Xp.device)
	val = torch.empty((Q, k), dtype=torch.float32, device=Xp.device)
	idx = torch.empty((Q, k), dtype=torch.int32, device=Xp.device)
	ms = (ctypes.c_float * 9)()
	ids = _item_ids_arg(item_ids, I, Xp.device)
	check(lib.anncur_score_topk_timed(_p(Xp), _ld(Xp), _p(Etp), Kp, Q, I, Kp, k, _p(val), _p(idx), _p(ws), nbytes,
									  _topk_flags(leading_sample, mfma16, qt1, mfma32, ring, staged), _p(ids) if ids is not None else None, _stream(), ms),
		  "score_topk_timed")
	return TopK(val, idx), [float(x) for x in ms]


@_on_device
def fused_survivors(workspace, Q, I, Kp, k, leading_sample=False, mfma16=False, qt1=False, mfma32=False, ring=False, staged=False):
	"""Mean number of candidates per query the sweep of the last score_topk_fused call on `workspace` kept (diagnostics; synchronises)."""
	out = ctypes.c_double()
	check(_lib.load().anncur_score_topk_survivors(_p(workspace), Q, I, Kp, k, _topk_flags(leading_sample, mfma16, qt1, mfma32, ring, staged), ctypes.byref(out), _stream()),
		  "score_topk_survivors")
	return out.value


def fused_plan(Q, I, Kp, k, leading_sample=False, mfma16=False, qt1=False, mfma32=False, ring=False, staged=False):
	"""The plan a fused call with these flags runs.  "lg": candidate segments per (query, item split) -- 2 = the 32x32x16 body (per-lane
	rings; the default above k = 1024 -- above 384 under staged=True --, and for Kp = 512), 1 = the 16x16x32 body (one queue per wave; the default for Kp <= 256, k <= 1024),
	4 = the wide kernel (Kp > 512); "QT": 32-query sub-tiles per wave (1 = qt1 honoured, or Kp = 512);
	"stage_pred": body of each sweep stage -- 0 / 1 = 32x32x16 with the ballot / exec-mask filter, 2 = 16x16x32 (4-wave workgroups, barrier per
	tile), 3 / 4 = Kp = 512 with the wave queue on 32x32x16 / 16x16x32, 5 = 16x16x32 in 8-wave workgroups with the tile ring (ring=True)."""
	out = (ctypes.c_int32 * 19)()
	check(_lib.load().anncur_score_topk_plan_ex(Q, I, Kp, k, _topk_flags(leading_sample, mfma16, qt1, mfma32, ring, staged), out, 19), "score_topk_plan_ex")
	v = [int(x) for x in out]
	plan = dict(zip(("n_sample_tiles", "n_tiles", "splits", "segment_capacity", "group", "lg", "QT", "n_stages"), v[:8]))
	n = plan["n_stages"]
	plan["stage_end"], plan["stage_pred"], plan["stage_flush"] = v[8:8 + n], v[11:11 + n], v[14:14 + n]
	plan["ladder"], plan["ladder_top_rank"] = bool(v[17]), v[18]   # the sweep raises its thresholds in-launch (csrc/score16.hpp; staged=True switches it off)
	return plan


def _dense_scores(X, Et):
	"""S = X @ Et^T with fp32 products and sums: the strided fp32-MFMA kernel of this library (any strides, fp32 or bf16 operands).
	No vendor GEMM anywhere on the path."""
	return gemm(X, Et.t())


@_on_device
def score_topk_dense(X, Et, k, max_bytes=2 << 30):
	"""Unfused route: S = X @ Et^T in fp32 (row chunks), then the exact scan.  Any K, any dtype.  bf16 operands of a shape the
	fused kernels take never come here from the index classes (cur.py / nearest_nbr.py try score_topk_fused first); what is left
	are the fp32 route, tiny item sets and tests that want the unfused answer."""
	_dev(X, Et)
	Q, I = X.shape[0], Et.shape[0]
	val = torch.empty((Q, k), dtype=torch.float32, device=X.device)
	idx = torch.empty((Q, k), dtype=torch.int32, device=X.device)
	rows = max(1, min(Q, max_bytes // max(4 * I, 1)))
	for q0 in range(0, Q, rows):
		q1 = min(Q, q0 + rows)
		S = _dense_scores(X[q0:q1], Et)
		v, i = rowwise_topk(S, k)
		val[q0:q1], idx[q0:q1] = v, i
	return TopK(val, idx)


@_on_device
def rerank(A, approx_idx, k_retvr, k_out):
	"""The k_out best of approx_idx[:, :k_retvr] by exact score A (reference: ..._splits.py:93-96)."""
	_dev(A, approx_idx)
	A = _rowmajor(A)
	if approx_idx.dtype != torch.int32 or approx_idx.stride(1) != 1:
		approx_idx = approx_idx.to(torch.int32).contiguous()
	Q, I = A.shape
	val = torch.empty((Q, k_out), dtype=torch.float32, device=A.device)
	idx = torch.empty((Q, k_out), dtype=torch.int32, device=A.device)
	check(_lib.load().anncur_rerank(_p(A), _dt(A), Q, I, _ld(A), _p(approx_idx), _ld(approx_idx), k_retvr, k_out, _p(val), _p(idx),
									_stream()), "rerank")
	return TopK(val, idx)


@_on_device
def overlap_counts(a, b, pairs, mapped_host_out=None):
	"""common[p, q] = |set(a[q, :ka_p]) & set(b[q, :kb_p])| for pairs = [(ka, kb), ...] -> int32 [n_pairs, Q].
	mapped_host_out: a contiguous PINNED host tensor int32 [n_pairs, Q] (mapped into the device address space by the HIP runtime): the kernel writes
	the counts there itself -- no device buffer, no copy launch (graph-capturable; the caller synchronises before reading it)."""
	_dev(a, b)
	a = a.to(torch.int32).contiguous()
	b = b.to(torch.int32).contiguous()
	Q = a.shape[0]
	if mapped_host_out is not None:
		out = mapped_host_out
		if out.is_cuda or not out.is_pinned() or not out.is_contiguous() or out.dtype != torch.int32 or tuple(out.shape) != (len(pairs), Q):
			raise ValueError("overlap_counts: mapped_host_out must be a contiguous pinned int32 host tensor [n_pairs, Q]")
	else:
		out = torch.empty((len(pairs), Q), dtype=torch.int32, device=a.device)
	lib = _lib.load()
	for s in range(0, len(pairs), 64):
		chunk = pairs[s:s + 64]
		ka = (ctypes.c_int32 * len(chunk))(*[int(p[0]) for p in chunk])
		kb = (ctypes.c_int32 * len(chunk))(*[int(p[1]) for p in chunk])
		check(lib.anncur_overlap_counts(_p(a), a.shape[1], _p(b), b.shape[1], Q, ka, kb, len(chunk), _p(out[s:s + len(chunk)]), _stream()),
			  "overlap_counts")
	return out


# ------------------------------------------------------------------ f3: inverted file
@_on_device
def ivf_build_lists(assign, nlist):
	"""assign int32 [n] -> (counts int32 [nlist], offsets int32 [nlist + 1], ids int32 [n]: the points of each list, ascending)."""
	_dev(assign)
	assign = assign.to(torch.int32).contiguous()
	n = assign.numel()
	counts = torch.empty(nlist, dtype=torch.int32, device=assign.device)
	offsets = torch.empty(nlist + 1, dtype=torch.int32, device=assign.device)
	ids = torch.empty(n, dtype=torch.int32, device=assign.device)
	check(_lib.load().anncur_ivf_build_lists(_p(assign), n, nlist, _p(counts), _p(offsets), _p(ids), _stream()), "ivf_build_lists")
	return counts, offsets, ids


@_on_device
def descending_norm_order(M, n_buckets=256):
	"""Row ids of the fp32 matrix M in coarse descending-norm order (int32 device tensor): norm buckets + the stable counting sort
	of the inverted-file builder.  The index builder's ordering hint; no torch arithmetic involved."""
	_dev(M)
	M = _rowmajor(M)
	if M.dtype != torch.float32:
		raise TypeError("descending_norm_order takes fp32")
	n = M.shape[0]
	norms = torch.empty(n, dtype=torch.float32, device=M.device)
	mm = torch.empty(2, dtype=torch.int32, device=M.device)
	bucket = torch.empty(n, dtype=torch.int32, device=M.device)
	check(_lib.load().anncur_norm_buckets(_p(M), n, M.shape[1], _ld(M), n_buckets, _p(norms), _p(mm), _p(bucket), _stream()), "norm_buckets")
	return ivf_build_lists(bucket, n_buckets)[2]


@_on_device
def ivf_list_means(Xs, offsets, centroids):
	"""centroids[l] <- mean of the rows of list l of Xs (list-ordered fp32 vectors); empty lists keep theirs.  In place."""
	_dev(Xs, offsets, centroids)
	Xs = _rowmajor(Xs)
	if Xs.dtype != torch.float32 or centroids.dtype != torch.float32 or centroids.stride(1) != 1:
		raise TypeError("ivf_list_means takes fp32 row-major tensors")
	check(_lib.load().anncur_ivf_list_means(_p(Xs), _ld(Xs), centroids.shape[1], _p(offsets), centroids.shape[0], _p(centroids), _ld(centroids), _stream()),
		  "ivf_list_means")
	return centroids


@_on_device
def renorm_rows(M):
	"""Rows of the fp32 matrix M rescaled to unit L2 norm, in place (a zero row stays): FAISS' fvec_renorm_L2 (spherical k-means)."""
	_dev(M)
	if M.dtype != torch.float32 or M.dim() != 2 or (M.shape[1] > 1 and M.stride(1) != 1):
		raise TypeError("renorm_rows takes a row-major fp32 matrix")
	check(_lib.load().anncur_renorm_rows(_p(M), M.shape[0], M.shape[1], _ld(M), _stream()), "renorm_rows")
	return M


@_on_device
def ivf_scan(Xs, offsets, ids, Q, probe, k):
	"""Exact inner products inside the probed lists + top-k.  Xs [n x dp], Q [nq x dp] fp32 zero-padded to dp (multiple of 16)."""
	_dev(Xs, offsets, ids, Q, probe)
	nq, dp = Q.shape
	probe = probe.to(torch.int32).contiguous()
	val = torch.empty((nq, k), dtype=torch.float32, device=Q.device)
	idx = torch.empty((nq, k), dtype=torch.int32, device=Q.device)
	check(_lib.load().anncur_ivf_scan(_p(Xs), _ld(Xs), dp, _p(offsets), _p(ids), _p(Q), _ld(Q), nq, _p(probe), probe.shape[1], k, _p(val), _p(idx), _stream()),
		  "ivf_scan")
	return TopK(val, idx)


@_on_device
def ivf_scan_grouped(Xs, offsets, ids, sizes_host, Q, probe, k, max_bytes=8 << 30, lists_bf16=None, profile=None):
	"""The same search as ivf_scan for MANY queries: pairs (query, probe slot) sorted by list, every list one small fp32-MFMA GEMM against
	its pairs' queries (anncur_ivf_group_scores), then the exact scan over each query's nprobe lists side by side and the column -> id map.
	sizes_host: the lists' lengths on the host (numpy int64, known since add()); they bound the tile count of the launch, the worklist itself is
	built on the device (no synchronisation inside the call: the search is stream-ordered like every other op).
	lists_bf16: a bf16 copy of Xs (same layout) -> the per-list GEMMs run on the bf16 matrix cores (queries rounded to bf16 here).
	profile: a dict -> receives HIP events around the per-list GEMM launch and around the scan + id map ("events": [(e0, e1, e2), ...] per
	query chunk; measurement only: bench.py's kernel-only IVF figure)."""
	_dev(Xs, offsets, ids, Q, probe)
	lib = _lib.load()
	nq_all, dp = Q.shape
	probe = probe.to(torch.int32).contiguous()
	nprobe, nlist = probe.shape[1], sizes_host.shape[0]
	lmax = int(max(int(sizes_host.max()), 1))
	lmax = -(-lmax // 8) * 8
	k_eff = min(k, nprobe * lmax)
	val = torch.empty((nq_all, k), dtype=torch.float32, device=Q.device)
	idx = torch.empty((nq_all, k), dtype=torch.int32, device=Q.device)
	step = max(64, min(nq_all, max_bytes // (nprobe * lmax * 4)))
	vt = -(-sizes_host // 64)                                        # 64-vector tiles per list
	vt_sum, vt_max = int(vt.sum()), int(max(int(vt.max()), 1))
	for q0 in range(0, nq_all, step):
		q1 = min(nq_all, q0 + step)
		nq = q1 - q0
		pr = probe[q0:q1].contiguous()
		cnt, poff, pair_ids = ivf_build_lists(pr.reshape(-1).clamp(min=0), nlist)     # (probe slots are valid list ids here: nprobe <= nlist)
		# tile worklist built on the device (round 4: round 3 copied the pairs-per-list counts to the host here -- a synchronisation in the
		# middle of every search -- and built the triples in numpy): the launch takes an upper bound on the number of 64 x 64 tiles that
		# needs no look at the counts, sum_l ceil(pairs_l / 64) vt_l <= (pairs / 64) max vt + sum vt; workgroups past the last tile exit
		max_tiles = (nq * nprobe // 64) * vt_max + vt_sum
		tile_start = torch.empty(nlist + 1, dtype=torch.int32, device=Q.device)
		S = _ScoreScratch.get(nq * nprobe * lmax, Q.device).view(nq, nprobe * lmax)   # grow-only scratch: a fresh 100s-of-MB allocation per call cost more than the search
		S.fill_(float("-inf"))
		Qc = Q[q0:q1]
		if profile is not None:
			evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
			profile.setdefault("events", []).append(evs)
			profile.setdefault("max_tiles", []).append(int(max_tiles))
			profile.setdefault("tile_starts", []).append(tile_start)
			evs[0].record()
		if lists_bf16 is not None:
			Qb = convert(Qc, torch.bfloat16)
			check(lib.anncur_ivf_group_scores_dev(_p(lists_bf16), BF16, _ld(lists_bf16), dp, _p(offsets), nlist, _p(Qb), _ld(Qb), nprobe, _p(pair_ids), _p(poff),
												  _p(tile_start), max_tiles, lmax, _p(S), _stream()), "ivf_group_scores_dev")
		else:
			check(lib.anncur_ivf_group_scores_dev(_p(Xs), F32, _ld(Xs), dp, _p(offsets), nlist, _p(Qc), _ld(Qc), nprobe, _p(pair_ids), _p(poff),
												  _p(tile_start), max_tiles, lmax, _p(S), _stream()), "ivf_group_scores_dev")
		if profile is not None: evs[1].record()
		v, c = rowwise_topk(S, k_eff)
		out_i = idx[q0:q1, :k_eff] if k_eff == k else torch.empty((nq, k_eff), dtype=torch.int32, device=Q.device)
		check(lib.anncur_ivf_map_ids(_p(c), _p(v), nq, k_eff, lmax, _p(pr), nprobe, _p(offsets), _p(ids), _p(out_i), _stream()), "ivf_map_ids")
		if profile is not None: evs[2].record()
		val[q0:q1, :k_eff] = v
		if k_eff < k:
			idx[q0:q1, :k_eff] = out_i
			val[q0:q1, k_eff:] = float("-inf"); idx[q0:q1, k_eff:] = -1
	return TopK(val, idx)


IVF_GROUPED_MAX_K, IVF_GROUPED_MAX_NLIST = 128, 8192


def ivf_search_grouped_ok(k, nlist):
	"""True where ivf_search_grouped serves the search (the wave-per-row scan's k, list histograms in LDS); ivf_scan_grouped otherwise."""
	return k <= IVF_GROUPED_MAX_K and nlist <= IVF_GROUPED_MAX_NLIST


class _ByteScratch:
	"""Grow-only 256-byte aligned workspace per device (ivf_search_grouped)."""
	_bufs = {}

	@classmethod
	def get(cls, n, device):
		key = (device.type, device.index)
		buf = cls._bufs.get(key)
		if buf is None or buf.numel() < n + 256:
			cls._bufs[key] = buf = torch.empty(n + 256, dtype=torch.uint8, device=device)
		off = (-buf.data_ptr()) % 256
		return buf[off:off + n]


@_on_device
def ivf_search_grouped(lists, offsets, ids, sizes_host, Q, probe, k, max_bytes=8 << 30, _skip_gemm=False):
	"""The batched IVF search of ivf_scan_grouped as ONE library call per query chunk (round 5): pairs grouped by list on the device, one tile
	GEMM launch on the matrix cores (bf16 `lists` / Q with rows a multiple of 128 elements: 128 x 128 tiles), scores in PACKED rows (query q's
	probed lists back to back, nothing pre-filled), ragged scan, column -> id map.  lists / Q: fp32 or bf16 (the same for both), rows
	zero-padded to a multiple of 16 elements.  probe int32 [nq x nprobe].  k <= 128, nlist <= 8192 (ivf_search_grouped_ok).
	_skip_gemm: measurement only (bench.py times the call with and without its tile launch; the results are then meaningless)."""
	_dev(lists, offsets, ids, Q, probe)
	lib = _lib.load()
	if lists.dtype != Q.dtype:
		raise ValueError("ivf_search_grouped: lists and queries must have the same dtype")
	nq_all, dp = Q.shape
	probe = probe.to(torch.int32).contiguous()
	nprobe, nlist = probe.shape[1], sizes_host.shape[0]
	lmax = int(max(int(sizes_host.max()), 1))
	k_eff = min(k, nprobe * lmax)
	pitch = -(-max(nprobe * lmax, k_eff) // 8) * 8                 # any row fits: nprobe lists of at most lmax vectors
	val = torch.empty((nq_all, k), dtype=torch.float32, device=Q.device)
	idx = torch.empty((nq_all, k), dtype=torch.int32, device=Q.device)
	step = max(64, min(nq_all, max_bytes // (pitch * 4), ((1 << 32) - 1) // pitch))
	for q0 in range(0, nq_all, step):
		q1 = min(nq_all, q0 + step)
		nq = q1 - q0
		Qc = Q[q0:q1]
		T = int(lib.anncur_ivf_search_tile(_dt(lists), dp, _ld(lists), _ld(Qc), nq))
		vt = -(-sizes_host // T)
		max_tiles = (nq * nprobe // T) * int(max(int(vt.max()), 1)) + int(vt.sum())    # sum_l ceil(pairs_l / T) vt_l <= (pairs / T) max vt + sum vt
		if _skip_gemm: max_tiles = 0
		S = _ScoreScratch.get(nq * pitch, Q.device)
		nbytes = lib.anncur_ivf_search_workspace_bytes(nq, nprobe, nlist, k_eff, max_tiles)
		ws = _ByteScratch.get(nbytes, Q.device)
		pr = probe[q0:q1]
		if k_eff == k:
			ov, oi = val[q0:q1], idx[q0:q1]
		else:
			ov = torch.empty((nq, k_eff), dtype=torch.float32, device=Q.device)
			oi = torch.empty((nq, k_eff), dtype=torch.int32, device=Q.device)
		check(lib.anncur_ivf_search_grouped(_p(lists), _dt(lists), _ld(lists), dp, _p(offsets), _p(ids), nlist, _p(Qc), _ld(Qc), nq, _p(pr), nprobe, k_eff, max_tiles,
											_p(S), pitch, _p(ws), nbytes, _p(ov), _p(oi), _stream()), "ivf_search_grouped")
		if k_eff < k:
			val[q0:q1, :k_eff] = ov; idx[q0:q1, :k_eff] = oi
			val[q0:q1, k_eff:] = float("-inf"); idx[q0:q1, k_eff:] = -1
	return TopK(val, idx)


@_on_device
def copy_to_mapped_host(src, pinned_host):
	"""Device kernel copy of `src` (CUDA tensor) into a PINNED host tensor (mapped into the device address space by the HIP
	runtime): graph-capturable, no copy engine.  The caller synchronises (event) before reading `pinned_host`."""
	_dev(src)
	if pinned_host.is_cuda or not pinned_host.is_pinned() or not pinned_host.is_contiguous():
		raise ValueError("copy_to_mapped_host needs a contiguous pinned host tensor")
	src = src.contiguous()
	nbytes = src.numel() * src.element_size()
	if nbytes != pinned_host.numel() * pinned_host.element_size():
		raise ValueError("size mismatch")
	check(_lib.load().anncur_copy_bytes(_p(src), ctypes.c_void_p(pinned_host.data_ptr()), nbytes, _stream()), "copy_bytes")
