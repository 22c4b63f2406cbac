"""TEST INFRASTRUCTURE ONLY -- not shipped, not imported by ``anncur_amd``.

CPU (NumPy / torch-CPU) restatement of the reference algorithm for the CUR
nearest-neighbour path.  Every function cites the reference file:line it follows
(paths relative to the upstream repository root).

Pinning: this restatement is checked against outputs of the reference itself
(imported unmodified in the build container by ``oracle/make_golden.py``; the
vectors are committed under ``tests/golden/``) -- see
``tests/test_oracle_golden.py``.  Parity is therefore PINNED for everything in
this file except ``flat_ip_search`` (FAISS is a third-party dependency that is
neither vendored nor pinned by the reference: "parity unpinned" at that one
boundary, restated from FAISS's public contract).

Third-party arithmetic the reference reaches on this path and that this file
calls directly, exactly like the reference does: ``numpy.linalg.pinv``
(LAPACK gesdd, rcond=1e-15), ``torch.matmul`` (CPU sgemm), ``torch.topk``.
"""
from __future__ import annotations

import itertools
from collections import defaultdict

import numpy as np
import torch

NEG_FILL = -99999999999999  # eval/run_retrieval_eval_wrt_exact_crossenc.py:108 (fp32: -1.00000000376832e14)


# --------------------------------------------------------------------------- a1
def select_anchors(rng: np.random.Generator, n: int, size: int):
	"""sorted(rng.choice(n, size, replace=False)).

	eval/run_retrieval_eval_wrt_exact_crossenc.py:67-70 (rows first, then cols from
	the SAME generator) and ..._w_fixed_train_test_splits.py:289-295 (one generator
	consumed sequentially across the whole n_ent_anchors loop).
	"""
	return sorted(rng.choice(n, size=size, replace=False))


def _pinv(x):
	"""np.linalg.pinv on the fp32 data, default rcond=1e-15 (eval/matrix_approx_zeshel.py:47,49)."""
	return torch.from_numpy(np.linalg.pinv(x.numpy() if torch.is_tensor(x) else np.asarray(x)))


def is_strictly_increasing(idx) -> bool:
	"""eval/matrix_approx_zeshel.py:53-55."""
	idx = list(idx)
	return all(i < j for i, j in zip(idx[:-1], idx[1:]))


# --------------------------------------------------------------------------- a3-a7
class CURApproxOracle:
	"""eval/matrix_approx_zeshel.py:19-126, restated.

	Differences from the reference, on purpose: the intersection check uses
	``torch.equal`` (the reference's ``assert torch.eq(...)`` at :44 raises for any
	block larger than 1x1 and only survives under ``python -O``), and errors raise
	instead of dropping into an IPython shell.
	"""

	def __init__(self, rows, cols, row_idxs, col_idxs, approx_preference, A=None):
		self.n = cols.shape[0]                     # :25
		self.m = rows.shape[1]                     # :26
		self.row_idxs = row_idxs
		self.col_idxs = col_idxs
		self.C = cols                              # n x kc   :31
		self.R = rows                              # kr x m   :32
		self.approx_preference = approx_preference
		assert is_strictly_increasing(row_idxs), "row_idxs should be sorted"   # :36
		assert is_strictly_increasing(col_idxs), "col_idxs should be sorted"   # :37
		assert len(row_idxs) == self.R.shape[0]    # :39
		assert len(col_idxs) == self.C.shape[1]    # :40
		intersect = self.C[row_idxs, :]            # kr x kc  :42
		assert torch.equal(intersect, self.R[:, col_idxs]), "intersection mismatch"  # :44 (intended check)
		if A is not None:                          # :46-47  oracle U = C+ A R+
			self.U = _pinv(self.C) @ torch.as_tensor(A) @ _pinv(self.R)
		else:                                      # :49
			self.U = _pinv(intersect)
		if approx_preference == "cols":            # :60-62
			self.latent_rows = self.C @ self.U
			self.latent_cols = self.R
		elif approx_preference == "rows":          # :63-65
			self.latent_rows = self.C
			self.latent_cols = self.U @ self.R
		else:                                      # :67
			raise NotImplementedError(f"approx_preference = {approx_preference} not supported")

	def get_rows(self, row_idxs):                  # :71-75
		return self.latent_rows[row_idxs, :] @ self.latent_cols

	def get_cols(self, col_idxs):                  # :77-80
		return self.latent_rows @ self.latent_cols[:, col_idxs]

	def get(self, row_idxs, col_idxs):             # :82-86
		return self.latent_rows[row_idxs, :] @ self.latent_cols[:, col_idxs]

	def get_complete_col(self, sparse_cols):       # :88-98
		if self.approx_preference != "cols":
			raise NotImplementedError("build index w/ approx_preference = cols")
		return self.latent_rows @ sparse_cols

	def topk_in_col(self, sparse_cols, k):         # :100-106
		return torch.topk(self.get_complete_col(sparse_cols), k, dim=1)

	def get_complete_row(self, sparse_rows):       # :109-119
		if self.approx_preference != "rows":
			raise NotImplementedError("build index w/ approx_preference = rows")
		return sparse_rows @ self.latent_cols

	def topk_in_row(self, sparse_rows, k):         # :121-126
		return torch.topk(self.get_complete_row(sparse_rows), k, dim=1)


# --------------------------------------------------------------------------- a10
def _overlap_one(indices1, indices2):
	"""eval/eval_utils.py:141-150."""
	n_common = len(set(indices1).intersection(set(indices2)))
	assert len(indices1) == len(indices2), f"Len of both indices is not same => {len(indices1)} != {len(indices2)}"
	n = len(indices1)
	return {"common": n_common, "diff": n - n_common, "total": n,
			"common_frac": n_common / n, "diff_frac": (n - n_common) / n}


def compute_overlap(indices_list1, indices_list2):
	"""eval/eval_utils.py:115-138: mean / population std / median, as 4-d.p. strings."""
	per_pair = [_overlap_one(a, b) for a, b in zip(indices_list1, indices_list2)]
	metrics = ["common", "diff", "total", "common_frac", "diff_frac"]
	if len(per_pair) == 0:                         # :129-130
		return {m: ("mean 0.0", "std 0.0", "p50 0.0") for m in metrics}
	out = {}
	for m in metrics:
		vals = [r[m] for r in per_pair]
		out[m] = ("mean {:.4f}".format(np.mean(vals)), "std {:.4f}".format(np.std(vals)),
				  "p50 {:.4f}".format(np.percentile(vals, 50)))
	return out


def overlap_to_flat(overlap, prefix="exact_vs_reranked_approx_retvr"):
	"""The string->float re-parse every caller does:
	eval/run_retrieval_eval_wrt_exact_crossenc.py:130-143, ..._splits.py:114-128."""
	flat = {}
	for metric, (mean_s, std_s, p50_s) in overlap.items():
		flat[f"{prefix}~{metric}_mean"] = float(mean_s[5:])
		flat[f"{prefix}~{metric}_std"] = float(std_s[4:])
		flat[f"{prefix}~{metric}_p50"] = float(p50_s[4:])
	return flat


# --------------------------------------------------------------------------- a8
def per_query_loop(exact, approx, top_k, top_k_retvr):
	"""The reference's per-query hot loop, verbatim semantics:
	eval/run_retrieval_eval_wrt_exact_crossenc.py:97-117 == ..._splits.py:80-100.

	Returns three (indices[Q,k], scores[Q,k]) numpy pairs: exact top-k, approx
	top-k_retvr, exact re-rank of the approx-retrieved.
	"""
	ex_i, ex_s, ap_i, ap_s, rr_i, rr_s = [], [], [], [], [], []
	for q in range(exact.shape[0]):
		row = exact[q]
		s, i = row.topk(top_k)
		a_s, a_i = approx[q].topk(top_k_retvr)
		temp = torch.zeros(row.shape) + NEG_FILL
		temp[a_i] = row[a_i]
		r_s, r_i = temp.topk(top_k)
		ex_i.append(i.unsqueeze(0)); ex_s.append(s.unsqueeze(0))
		ap_i.append(a_i.unsqueeze(0)); ap_s.append(a_s.unsqueeze(0))
		rr_i.append(r_i.unsqueeze(0)); rr_s.append(r_s.unsqueeze(0))
	cat = lambda xs: torch.cat(xs).numpy()          # a9: ...crossenc.py:35-44
	return (cat(ex_i), cat(ex_s)), (cat(ap_i), cat(ap_s)), (cat(rr_i), cat(rr_s))


def _topk_stable(x, k):
	"""top-k with a DEFINED tie order: score descending, then smaller index (one of the orders torch.topk may return)."""
	order = torch.sort(x, descending=True, stable=True).indices[:k]
	return x[order], order


def per_query_loop_stable(exact, approx, top_k, top_k_retvr):
	"""per_query_loop with every topk replaced by the tie-stable one.  Identical to per_query_loop on tie-free
	inputs (the golden tests run both); on tie-heavy inputs (bf16 scores) torch.topk's arbitrary tie order makes the
	reference's own overlap numbers vary, and this variant is the well-defined statement the HIP path implements."""
	ex_i, ap_i, rr_i = [], [], []
	for q in range(exact.shape[0]):
		row = exact[q]
		_, i = _topk_stable(row, top_k)
		_, a_i = _topk_stable(approx[q], top_k_retvr)
		temp = torch.zeros(row.shape) + NEG_FILL
		temp[a_i] = row[a_i]
		_, r_i = _topk_stable(temp, top_k)
		ex_i.append(i.unsqueeze(0)); ap_i.append(a_i.unsqueeze(0)); rr_i.append(r_i.unsqueeze(0))
	cat = lambda xs: torch.cat(xs).numpy()
	return cat(ex_i), cat(ap_i), cat(rr_i)


def eval_all_topk_stable(exact, approx, arg_top_k_vals, top_k_retvr):
	top_k_vals = [k for k in arg_top_k_vals if k <= top_k_retvr]
	if not top_k_vals:
		return {}
	ex_i, _, rr_i = per_query_loop_stable(exact, approx, max(top_k_vals), top_k_retvr)
	return {k: overlap_to_flat(compute_overlap(ex_i[:, :k], rr_i[:, :k])) for k in top_k_vals}


def eval_approx_score_mat_for_all_topk(exact, approx, arg_top_k_vals, top_k_retvr):
	"""..._w_fixed_train_test_splits.py:51-135."""
	top_k_vals = [k for k in arg_top_k_vals if k <= top_k_retvr]      # :70
	if len(top_k_vals) == 0:                                          # :71-72
		return {}
	max_topk = max(top_k_vals)
	(ex_i, _), _, (rr_i, _) = per_query_loop(exact, approx, max_topk, top_k_retvr)
	res = {}
	for k in top_k_vals:                                              # :108-131
		res[k] = overlap_to_flat(compute_overlap(ex_i[:, :k], rr_i[:, :k]))
	return res


def eval_approx_score_mat(exact, approx, top_k, top_k_retvr):
	"""..._w_fixed_train_test_splits.py:138-206."""
	(ex_i, _), _, (rr_i, _) = per_query_loop(exact, approx, top_k, top_k_retvr)
	return overlap_to_flat(compute_overlap(ex_i, rr_i))


# --------------------------------------------------------------------------- entry point A
def run_approx_eval_w_seed(approx_method, A, n_ment_anchors, n_ent_anchors, top_k, top_k_retvr, seed,
						   precomp_approx=None, stable=False):
	"""eval/run_retrieval_eval_wrt_exact_crossenc.py:47-158.  stable=True: the per-query loop with the tie-stable top-k (per_query_loop_stable) --
	the well-defined statement of the same lines for tie-heavy (bf16) score matrices."""
	n_ments, n_ents = A.shape
	rng = np.random.default_rng(seed=seed)                                    # :65
	row_idxs = select_anchors(rng, n_ments, n_ment_anchors)                   # :67
	col_idxs = select_anchors(rng, n_ents, n_ent_anchors)                     # :68
	rows = A[row_idxs, :]                                                     # :73
	cols = A[:, col_idxs]                                                     # :74
	non_anchor_rows = sorted(set(range(n_ments)) - set(row_idxs))             # :76
	if approx_method == "cur":                                                # :81-84
		S = CURApproxOracle(rows, cols, row_idxs, col_idxs, "rows").get(list(range(n_ments)), list(range(n_ents)))
	elif approx_method == "cur_oracle":                                       # :85-88
		S = CURApproxOracle(rows, cols, row_idxs, col_idxs, "rows", A=A).get(list(range(n_ments)), list(range(n_ents)))
	elif precomp_approx is not None:                                          # :79-80
		S = precomp_approx
	else:
		raise NotImplementedError(f"approx_method = {approx_method} not supported")
	if stable:
		ex_i, _, rr_i = per_query_loop_stable(A, S, top_k, top_k_retvr)
	else:
		(ex_i, _), _, (rr_i, _) = per_query_loop(A, S, top_k, top_k_retvr)    # :97-122

	def score(idxs):                                                          # :124-148
		res = overlap_to_flat(compute_overlap(ex_i[idxs], rr_i[idxs]))
		res["approx_error"] = torch.norm((S - A)[idxs, :]).data.numpy()
		res["approx_error_relative"] = res["approx_error"] / torch.norm(A[idxs, :]).data.numpy()
		return res

	return {"anchor": score(row_idxs), "non_anchor": score(non_anchor_rows), "all": score(list(range(n_ments)))}


def run_approx_eval(approx_method, A, n_ment_anchors, n_ent_anchors, top_k, top_k_retvr, n_seeds, precomp_approx=None):
	"""eval/run_retrieval_eval_wrt_exact_crossenc.py:162-200 (mean over seeds)."""
	acc = defaultdict(lambda: defaultdict(list))
	for seed in range(n_seeds):
		res = run_approx_eval_w_seed(approx_method, A, n_ment_anchors, n_ent_anchors, top_k, top_k_retvr, seed,
									 precomp_approx)
		for ment_type, d in res.items():
			for metric, val in d.items():
				acc[ment_type][metric].append(float(val))
	return {t: {m: float(np.mean(v)) for m, v in d.items()} for t, d in acc.items()}


# --------------------------------------------------------------------------- entry point B
def splits_grids(n_ent, base_top_k_retr=(1, 10, 50, 100, 200, 500, 1000), base_n_anchor=(10, 50, 100, 200, 500, 1000, 2000)):
	"""The hard-coded sweep grids of ..._w_fixed_train_test_splits.py:238-251 (cur method)."""
	base = list(base_top_k_retr)
	cur = base + [int(k * frac) for k in base for frac in np.arange(0.1, 1.0, 0.1)]     # :241
	top_k_retr_vals = sorted(set(cur))                                                 # :243-247
	n_anc = [v for v in base_n_anchor if v < n_ent] + [n_ent]                          # :250
	n_anc = sorted(set(n_anc + cur))                                                   # :251
	return top_k_retr_vals, n_anc


def run_eval_method_cur(A_test, A_train, seed, top_k_vals, top_k_retr_vals, n_ent_anchors_vals, eval_only=None):
	"""..._w_fixed_train_test_splits.py:286-303 (index + approximation) and :399-429 (sweep),
	for eval_method == "cur", with the grids passed in.  eval_only (test convenience): anchor counts to evaluate -- the anchor
	stream is still drawn for EVERY count of the grid, in order, as the reference does."""
	n_train, n_ent = A_train.shape
	rng = np.random.default_rng(seed=seed)                                             # :289
	approx = {}
	for n_anc in n_ent_anchors_vals:                                                   # :293
		anc = select_anchors(rng, n_ent, n_anc)                                        # :295
		if eval_only is not None and n_anc not in eval_only:
			continue
		cols = A_train[:, anc]                                                         # :297
		cur = CURApproxOracle(rows=A_train, cols=cols, row_idxs=np.arange(n_train), col_idxs=anc,
							  approx_preference="rows")                               # :298
		approx[n_anc] = cur.get_complete_row(A_test[:, anc])                           # :300-303
	res = defaultdict(lambda: defaultdict(dict))
	for k_retvr, n_anc in itertools.product(top_k_retr_vals, n_ent_anchors_vals):      # :402-403
		if k_retvr < 0 or k_retvr > n_ent or n_anc not in approx:                      # :407-409
			continue
		per_k = eval_approx_score_mat_for_all_topk(A_test, approx[n_anc], top_k_vals, k_retvr)   # :420
		for k in top_k_vals:
			if k > k_retvr:                                                            # :427
				continue
			res[f"top_k={k}"][f"k_retvr={k_retvr}"][f"anc_n_m={n_train}_anc_n_e={n_anc}"] = per_k[k]  # :429
	return {a: {b: dict(c) for b, c in d.items()} for a, d in res.items()}


# --------------------------------------------------------------------------- a14 (flat branch)
def flat_ip_search(embeds: np.ndarray, queries: np.ndarray, k: int):
	"""Exact inner-product search == faiss.IndexFlatIP(d).add(embeds); .search(queries, k)
	as used at models/nearest_nbr.py:36-38, utils/data_process.py:351,397.

	PARITY UNPINNED: FAISS is absent (not vendored, no pinned version); restated from
	its public contract: D float32 [nq,k] descending, I int64 [nq,k].
	"""
	scores = torch.from_numpy(np.ascontiguousarray(queries, dtype=np.float32)) @ \
		torch.from_numpy(np.ascontiguousarray(embeds, dtype=np.float32)).T
	D, I = torch.topk(scores, k, dim=1)
	return D.numpy(), I.numpy().astype(np.int64)


# --------------------------------------------------------------------------- synthetic protocol-B data
def synth_protocol_b(n_train, n_test, n_items, rank=64, noise=0.05, seed=0, dtype=torch.float32):
	"""SURVEY.md section 8(d) synthetic inputs: shared item factors Z, low-rank + noise.
	(The build's own generator, not the reference's; mirrored by anncur_amd.synth.)"""
	g = torch.Generator().manual_seed(seed)
	Z = torch.randn(rank, n_items, generator=g)
	def make(n):
		return (torch.randn(n, rank, generator=g) @ Z) / (rank ** 0.5) + noise * torch.randn(n, n_items, generator=g)
	A_train, A_test = make(n_train), make(n_test)
	if dtype != torch.float32:   # the CPU reference sees the rounded values, up-cast to fp32
		A_train, A_test = A_train.to(dtype).float(), A_test.to(dtype).float()
	return A_train, A_test
