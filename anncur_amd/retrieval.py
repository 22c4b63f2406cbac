"""Device-side replacement of the reference's per-query evaluation loop
(eval/run_retrieval_eval_wrt_exact_crossenc.py:97-154 and
 eval/run_retrieval_eval_wrt_exact_crossenc_w_fixed_train_test_splits.py:51-206):
exact top-k scan, approximate top-k_retvr, exact re-rank, overlap statistics.
One exact scan and one retrieval at the largest k serve every (top_k, k_retvr) cell of a sweep."""
import numpy as np
import torch

from . import ops
from .eval_utils import flatten_overlap, overlap_stats_batch, overlap_stats_from_counts


def exact_topk(A_dev, k):
	return ops.rowwise_topk(A_dev, k)


def overlap_cells(exact_idx, approx_idx, cells, A_dev=None, literal_rerank=False):
	"""common counts [n_cells, Q] for cells = [(top_k, k_retvr), ...].

	Closed form (default): |exact[:top_k] & rerank_{k_retvr}[:top_k]| == |exact[:top_k] & approx[:k_retvr]|
	because both rankings break ties the same way (score desc, index asc).  literal_rerank=True runs the
	re-rank kernel per k_retvr exactly like the reference's scatter + topk (tests check both agree)."""
	if not literal_rerank:
		return ops.overlap_counts(exact_idx, approx_idx, cells)
	out = torch.empty((len(cells), exact_idx.shape[0]), dtype=torch.int32, device=exact_idx.device)
	for kr in sorted({c[1] for c in cells}):
		sel = [j for j, c in enumerate(cells) if c[1] == kr]
		kmax = max(cells[j][0] for j in sel)
		rr = ops.rerank(A_dev, approx_idx, kr, kmax)
		cnt = ops.overlap_counts(exact_idx, rr.indices, [(cells[j][0], cells[j][0]) for j in sel])
		for r, j in enumerate(sel):
			out[j] = cnt[r]
	return out


def eval_topk_recall(A_dev, approx_idx, top_k_vals, k_retvr_vals, exact=None, row_subsets=None, literal_rerank=False):
	"""-> {(top_k, k_retvr): {"exact_vs_reranked_approx_retvr~common_mean": ..., ...}} in the reference's
	metric names and 4-decimal rounding.  approx_idx [Q, >= max k_retvr] sorted by approximate score.
	row_subsets: optional {name: index array}; then the value is {name: metrics} (entry point A's
	anchor / non_anchor / all split)."""
	cells = [(k, kr) for kr in k_retvr_vals for k in top_k_vals if k <= kr]
	if not cells:
		return {}
	kmax = max(k for k, _ in cells)
	if exact is None:
		exact = exact_topk(A_dev, kmax)
	counts = overlap_cells(exact.indices, approx_idx, cells, A_dev, literal_rerank).cpu().numpy()
	res = {}
	for j, (k, kr) in enumerate(cells):
		if row_subsets is None:
			res[(k, kr)] = flatten_overlap(overlap_stats_from_counts(counts[j], k))
		else:
			res[(k, kr)] = {name: flatten_overlap(overlap_stats_from_counts(counts[j][np.asarray(rows, dtype=np.int64)], k))
							for name, rows in row_subsets.items()}
	return res
