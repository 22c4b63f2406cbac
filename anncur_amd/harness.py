"""Evaluation harness of the two reference entry points, on the GPU.

Entry point A  (eval/run_retrieval_eval_wrt_exact_crossenc.py:47-200): anchors from ONE matrix, methods cur / cur_oracle,
                metrics split into anchor / non_anchor / all query rows, mean over seeds.
Entry point B  (eval/run_retrieval_eval_wrt_exact_crossenc_w_fixed_train_test_splits.py:209-443): index from the TRAIN
                matrix, test queries contribute their anchor-item scores, full (k_retvr x n_anchor) sweep.
The per-query Python loop of the reference (3 x topk + scatter per query) is replaced by one exact scan, one fused retrieval
at the largest k_retvr and one overlap kernel per sweep; results keep the reference's names, nesting and 4-decimal rounding.
"""
import itertools
import logging
import pickle
from collections import defaultdict

import numpy as np
import torch

from . import ops
from .cur import CURApprox, CURRowIndex
from .eval_utils import flatten_overlap, overlap_stats_from_counts

LOGGER = logging.getLogger(__name__)


def load_score_pickle(path):
	"""The reference's input schema (producer: eval/run_cross_encoder_for_ment_ent_matrix_zeshel.py:230-240)."""
	with open(path, "rb") as f:
		d = pickle.load(f)
	if "ment_to_ent_scores" not in d:
		raise KeyError(f"{path}: not a mention x entity score dump (missing 'ment_to_ent_scores')")
	return d


def to_device_matrix(scores, device, dtype="fp32"):
	t = scores if torch.is_tensor(scores) else torch.as_tensor(np.asarray(scores))
	t = t.to(device=device, dtype=torch.float32)
	return ops.convert(t, torch.bfloat16) if dtype == "bf16" else t.contiguous()


def _select(rng, n, size):
	"""sorted(rng.choice(n, size, replace=False)) -- the reference's anchor selection (crossenc.py:67-68, splits.py:295)."""
	return sorted(rng.choice(n, size=size, replace=False))


def _subset_metrics(counts_row, rows, k):
	return flatten_overlap(overlap_stats_from_counts(np.asarray(counts_row)[np.asarray(rows, dtype=np.int64)], k))


# ------------------------------------------------------------------------------------------------ entry point A
def run_approx_eval_w_seed(approx_method, A_dev, n_ment_anchors, n_ent_anchors, top_k, top_k_retvr, seed, exact_cache=None, pinv_backend="auto"):
	"""One seed of one grid cell -> {"anchor": {...}, "non_anchor": {...}, "all": {...}} (crossenc.py:47-158)."""
	n_ments, n_ents = A_dev.shape
	rng = np.random.default_rng(seed=seed)
	row_idxs = _select(rng, n_ments, n_ment_anchors)          # rows first, then columns, same generator
	col_idxs = _select(rng, n_ents, n_ent_anchors)
	rows = ops.gather_rows(A_dev, row_idxs)
	cols = ops.gather_cols(A_dev, col_idxs)
	non_anchor = sorted(set(range(n_ments)) - set(int(i) for i in row_idxs))
	if approx_method == "cur":
		cur = CURApprox(rows=rows, cols=cols, row_idxs=row_idxs, col_idxs=col_idxs, approx_preference="rows", pinv_backend=pinv_backend)
	elif approx_method == "cur_oracle":
		cur = CURApprox(rows=rows, cols=cols, row_idxs=row_idxs, col_idxs=col_idxs, approx_preference="rows", A=A_dev, pinv_backend=pinv_backend)
	else:
		raise NotImplementedError(f"approx_method = {approx_method} not supported")
	# approximate retrieval for EVERY query row + the per-row error terms: one sweep where the fused route takes the cell (cur.eval_rows)
	approx, err_sq, norm_sq = cur.eval_rows(cols, A_dev, top_k_retvr)
	if exact_cache is not None and exact_cache.get("k", 0) >= top_k:
		exact = exact_cache["topk"]
	else:
		exact = ops.rowwise_topk(A_dev, top_k)
		if exact_cache is not None:
			exact_cache.update(k=top_k, topk=exact)
	counts = ops.overlap_counts(exact.indices, approx.indices, [(top_k, top_k_retvr)]).cpu().numpy()[0]
	err_sq, norm_sq = err_sq.double().cpu().numpy(), norm_sq.double().cpu().numpy()

	def score(idxs):
		res = _subset_metrics(counts, idxs, top_k) if len(idxs) else flatten_overlap(overlap_stats_from_counts([], top_k))
		ii = np.asarray(idxs, dtype=np.int64)
		err = np.float32(np.sqrt(err_sq[ii].sum()))
		with np.errstate(invalid="ignore", divide="ignore"):
			res["approx_error"] = err
			res["approx_error_relative"] = err / np.float32(np.sqrt(norm_sq[ii].sum()))  # empty subset -> 0/0 = nan, like the reference
		return res

	return {"anchor": score(row_idxs), "non_anchor": score(non_anchor), "all": score(list(range(n_ments)))}


def run_approx_eval(approx_method, A_dev, n_ment_anchors, n_ent_anchors, top_k, top_k_retvr, n_seeds, exact_cache=None, pinv_backend="auto"):
	"""Mean over seeds (crossenc.py:162-200)."""
	acc = defaultdict(lambda: defaultdict(list))
	for seed in range(n_seeds):
		res = run_approx_eval_w_seed(approx_method, A_dev, n_ment_anchors, n_ent_anchors, top_k, top_k_retvr, seed, exact_cache, pinv_backend)
		for ment_type, d in res.items():
			for metric, val in d.items():
				acc[ment_type][metric].append(float(val))
	return {t: {m: float(np.mean(v)) for m, v in d.items()} for t, d in acc.items()}


def run_approx_eval_w_seed_sharded(sharded, n_ment_anchors, n_ent_anchors, top_k, top_k_retvr, seed):
	"""Method "cur" of entry point A on a ROW-SHARDED score matrix (anncur_amd.dist.ShardedScoreMatrix): every rank holds a
	contiguous block of query rows.  One all-gather assembles the anchor rows; the index is replicated; each rank retrieves,
	scans and counts for its own rows; rank 0 receives the ordered per-query counts / error terms and returns the same dict as
	run_approx_eval_w_seed (None on the other ranks).  No collective on the per-query path."""
	from .dist import gather_rows_to_rank0
	A_loc, n_ments = sharded.local, sharded.n_rows
	n_ents = A_loc.shape[1]
	rng = np.random.default_rng(seed=seed)
	row_idxs = _select(rng, n_ments, n_ment_anchors)
	col_idxs = _select(rng, n_ents, n_ent_anchors)
	R = sharded.anchor_rows(row_idxs)                                  # the one exchange: [Kq x I] on every rank
	index = CURRowIndex(R, col_idxs)
	X_loc = ops.gather_cols(A_loc, col_idxs)
	exact, approx, err_sq, norm_sq = index.eval_cell(X_loc, A_loc, top_k, top_k_retvr)   # exact scan beside ONE sweep for candidates + error sums
	counts = ops.overlap_counts(exact.indices, approx.indices, [(top_k, top_k_retvr)])[0]
	packed = torch.stack([counts.float(), err_sq, norm_sq], dim=1).contiguous()      # [n_loc x 3]
	full = gather_rows_to_rank0(packed, n_ments, sharded.group)
	if full is None:
		return None
	full = full.double().cpu().numpy()
	counts_all, err_all, norm_all = full[:, 0].round().astype(np.int64), full[:, 1], full[:, 2]
	non_anchor = sorted(set(range(n_ments)) - set(int(i) for i in row_idxs))

	def score(idxs):
		res = _subset_metrics(counts_all, idxs, top_k) if len(idxs) else flatten_overlap(overlap_stats_from_counts([], top_k))
		ii = np.asarray(idxs, dtype=np.int64)
		err = np.float32(np.sqrt(err_all[ii].sum()))
		with np.errstate(invalid="ignore", divide="ignore"):
			res["approx_error"] = err
			res["approx_error_relative"] = err / np.float32(np.sqrt(norm_all[ii].sum()))
		return res

	return {"anchor": score(row_idxs), "non_anchor": score(non_anchor), "all": score(list(range(n_ments)))}


def run_entry_A_sharded(sharded, grids, n_seeds, progress=None):
	"""run_entry_A for a row-sharded matrix (method "cur" only: "cur_oracle" needs the whole matrix on one device).
	Rank 0 returns the result dict, the other ranks None."""
	from .dist import dist
	is_root = dist.get_rank(sharded.group) == 0
	n_ment, n_ent = sharded.n_rows, sharded.local.shape[1]
	res = defaultdict(lambda: defaultdict(lambda: defaultdict(dict)))
	cells = list(itertools.product(grids["top_k_vals"], grids["top_k_retr_vals"], grids["n_ment_anchors_vals"], grids["n_ent_anchors_vals"]))
	for ctr, (top_k, kr, nm, ne) in enumerate(cells):
		if kr < top_k or kr > n_ent or nm > n_ment or ne > n_ent:
			continue
		if progress and is_root:
			progress("cur", ctr, len(cells))
		acc = defaultdict(lambda: defaultdict(list))
		for seed in range(n_seeds):
			out = run_approx_eval_w_seed_sharded(sharded, nm, ne, top_k, kr, seed)
			if is_root:
				for ment_type, d in out.items():
					for metric, val in d.items():
						acc[ment_type][metric].append(float(val))
		if is_root:
			res["cur"][f"top_k={top_k}"][f"k_retvr={kr}"][f"anc_n_m={nm}~anc_n_e={ne}"] = \
				{t: {m: float(np.mean(v)) for m, v in d.items()} for t, d in acc.items()}
	if not is_root:
		return None
	return {m: {a: {b: dict(c) for b, c in d.items()} for a, d in v.items()} for m, v in res.items()}


def default_grids_A(total_n_ment, total_n_ent):
	"""The grids hard-coded in the reference (crossenc.py:225-239)."""
	return {
		"eval_methods": ["cur", "cur_oracle"],
		"n_ment_anchors_vals": [v for v in [50, 100, 200, 500, 1000, 2000, 5000] if v <= total_n_ment],
		"n_ent_anchors_vals": [v for v in [50, 100, 200, 500, 1000, 2000] if v < total_n_ent] + [total_n_ent],
		"top_k_vals": [10],
		"top_k_retr_vals": [500],
	}


def run_entry_A(A_dev, grids, n_seeds, progress=None, pinv_backend="auto"):
	"""-> res[method]["top_k=.."]["k_retvr=.."]["anc_n_m=..~anc_n_e=.."][anchor|non_anchor|all][metric]  (crossenc.py:349-383)."""
	total_n_ment, total_n_ent = A_dev.shape
	res = defaultdict(lambda: defaultdict(lambda: defaultdict(dict)))
	exact_cache = {}
	for method in grids["eval_methods"]:
		cells = list(itertools.product(grids["top_k_vals"], grids["top_k_retr_vals"], grids["n_ment_anchors_vals"], grids["n_ent_anchors_vals"]))
		for ctr, (top_k, kr, nm, ne) in enumerate(cells):
			if kr < top_k or kr > total_n_ent:     # crossenc.py:358-359
				continue
			if nm > total_n_ment or ne > total_n_ent:
				continue
			if progress:
				progress(method, ctr, len(cells))
			res[method][f"top_k={top_k}"][f"k_retvr={kr}"][f"anc_n_m={nm}~anc_n_e={ne}"] = \
				run_approx_eval(method, A_dev, nm, ne, top_k, kr, n_seeds, exact_cache, pinv_backend)
	return {m: {a: {b: dict(c) for b, c in d.items()} for a, d in v.items()} for m, v in res.items()}


# ------------------------------------------------------------------------------------------------ entry point B
def default_grids_B(total_n_ent, method="cur"):
	"""The grids hard-coded in the reference (splits.py:238-251)."""
	base = [1, 10, 50, 100, 200, 500, 1000]
	cur = base + [int(k * frac) for k in base for frac in np.arange(0.1, 1.0, 0.1)]
	retr = cur if ("cur" in method or "fixed_anc_ent" in method) else base
	n_anc = [v for v in [10, 50, 100, 200, 500, 1000, 2000] if v < total_n_ent] + [total_n_ent]
	return {"top_k_vals": [1, 10, 50, 100], "top_k_retr_vals": sorted(set(retr)), "n_ent_anchors_vals": sorted(set(n_anc + cur))}


def _sweep_cells(A_test_dev, approx_idx, exact, top_k_vals, top_k_retr_vals, n_ent):
	"""All (top_k, k_retvr) cells of one approximation from ONE retrieval at the largest k_retvr:
	the top-k_retvr list is a prefix of the sorted top-k_max list."""
	cells = [(k, kr) for kr in top_k_retr_vals if 0 < kr <= n_ent and kr <= approx_idx.shape[1] for k in top_k_vals if k <= kr]
	if not cells:
		return {}
	counts = ops.overlap_counts(exact.indices, approx_idx, cells).cpu().numpy()
	return {cell: flatten_overlap(overlap_stats_from_counts(counts[j], cell[0])) for j, cell in enumerate(cells)}


def run_eval_method_cur(A_test_dev, A_train_dev, seed, grids, compute_dtype=None, progress=None, key_n_m=None, pinv_backend="auto"):
	"""eval_method == "cur" of entry point B for one seed (splits.py:286-303 + 399-429)."""
	n_train, n_ent = A_train_dev.shape
	top_k_vals, retr_vals, anc_vals = grids["top_k_vals"], grids["top_k_retr_vals"], grids["n_ent_anchors_vals"]
	kr_max = max([kr for kr in retr_vals if kr <= min(n_ent, ops._lib.MAX_TOPK)] or [0])
	k_max = max([k for k in top_k_vals if k <= kr_max] or [0])
	if kr_max == 0 or k_max == 0:
		return {}
	exact = ops.rowwise_topk(A_test_dev, k_max)
	rng = np.random.default_rng(seed=seed)                     # ONE stream consumed across the whole anchor-count loop
	res = defaultdict(lambda: defaultdict(dict))
	for j, n_anc in enumerate(anc_vals):
		anc = _select(rng, n_ent, n_anc)
		if progress:
			progress(j, len(anc_vals))
		if n_anc == 0:
			# the reference's own grid holds n_ent_anchors = int(1 * 0.1) = 0 (splits.py:241,250): C is [n x 0], U = pinv of an
			# empty block, S_hat = 0 everywhere.  Every item ties; under this build's tie order (score desc, index asc) the
			# retrieved list is 0..k_retvr-1 for every query (torch.topk leaves the order of an all-equal row unspecified).
			approx_idx = torch.arange(kr_max, dtype=torch.int32, device=A_test_dev.device).expand(A_test_dev.shape[0], kr_max).contiguous()
			for (k, kr), metrics in _sweep_cells(A_test_dev, approx_idx, exact, top_k_vals, retr_vals, n_ent).items():
				res[f"top_k={k}"][f"k_retvr={kr}"][f"anc_n_m={n_train if key_n_m is None else key_n_m}_anc_n_e={n_anc}"] = metrics
			continue
		cur = CURApprox(rows=A_train_dev, cols=ops.gather_cols(A_train_dev, anc), row_idxs=np.arange(n_train), col_idxs=anc,
						approx_preference="rows", compute_dtype=compute_dtype, pinv_backend=pinv_backend)
		approx = cur.topk_in_row_device(ops.gather_cols(A_test_dev, anc), kr_max)
		for (k, kr), metrics in _sweep_cells(A_test_dev, approx.indices, exact, top_k_vals, retr_vals, n_ent).items():
			res[f"top_k={k}"][f"k_retvr={kr}"][f"anc_n_m={n_train if key_n_m is None else key_n_m}_anc_n_e={n_anc}"] = metrics
		del cur, approx
	return {a: {b: dict(c) for b, c in d.items()} for a, d in res.items()}


def run_eval_method_embeds(A_test_dev, mention_embeds, label_embeds, n_train, grids):
	"""bienc / tfidf / fixed_anc_ent given PRECOMPUTED embeddings: scores = mention_embeds @ label_embeds.T (splits.py:283,324,383);
	one result is repeated for every anchor count, as the reference does (splits.py:412-418)."""
	n_ent = label_embeds.shape[0]
	top_k_vals, retr_vals, anc_vals = grids["top_k_vals"], grids["top_k_retr_vals"], grids["n_ent_anchors_vals"]
	kr_max = max([kr for kr in retr_vals if kr <= min(n_ent, ops._lib.MAX_TOPK)] or [0])
	k_max = max([k for k in top_k_vals if k <= kr_max] or [0])
	if kr_max == 0:
		return {}
	exact = ops.rowwise_topk(A_test_dev, k_max)
	approx = ops.score_topk_dense(mention_embeds, label_embeds, kr_max)
	res = defaultdict(lambda: defaultdict(dict))
	for (k, kr), metrics in _sweep_cells(A_test_dev, approx.indices, exact, top_k_vals, retr_vals, n_ent).items():
		for n_anc in anc_vals:
			res[f"top_k={k}"][f"k_retvr={kr}"][f"anc_n_m={n_train}_anc_n_e={n_anc}"] = metrics
	return {a: {b: dict(c) for b, c in d.items()} for a, d in res.items()}


def run_eval_method_fixed_anc_ent_cur(A_test_dev, e2e_scores_dev, n_fixed_anc_ent, grids, key_n_m=None):
	"""fixed_anc_ent_cur (splits.py:327-358): R = ent_to_ent_scores[:, :n_fixed].T, anchors from rng(0), U = pinv(R[:, anc])."""
	R = e2e_scores_dev[:, :n_fixed_anc_ent].t().contiguous()     # n_fixed x n_ents
	return run_eval_method_cur(A_test_dev, R, seed=0, grids=grids, key_n_m=key_n_m)
