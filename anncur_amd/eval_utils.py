"""compute_overlap with the reference's exact output format (eval/eval_utils.py:115-150):
per-pair set overlap, then mean / population std / median formatted to 4 decimals."""
import numpy as np

_METRICS = ["common", "diff", "total", "common_frac", "diff_frac"]


def _median_pair(lo, hi):
	"""np.percentile([lo, hi], 50) with the default linear method, without its ~100 us call overhead:
	numpy's _lerp evaluates b - (b - a) * (1 - t) for t >= 0.5."""
	return hi - (hi - lo) * 0.5


def _fmt(mean, std, p50):
	return "mean {:.4f}".format(mean), "std {:.4f}".format(std), "p50 {:.4f}".format(p50)


def overlap_stats_from_counts(common, n):
	"""`common`: per-query |set1 & set2| (array-like), `n`: the common list length.  Returns the
	reference's dict {metric: ("mean x", "std x", "p50 x")} (np.mean / population np.std / np.percentile 50)."""
	return overlap_stats_batch(np.asarray(common, dtype=np.int64)[None, :], [n])[0]


def overlap_stats_batch(counts, ns):
	"""Statistics for several (top_k, k_retvr) cells at once: counts [n_cells, Q] ints, ns[j] = list length of cell j.

	All five metrics are affine in `common` (diff = n - c, total = n, *_frac = ./n), so one mean / std / median of `common`
	per cell gives every number: mean and median transform exactly; std is invariant under c -> n - c and scales by 1/n.
	(In floating point the derived values can differ from a direct np.std on the transformed array in the last ulp; after the
	reference's 4-decimal formatting that is invisible -- tests/test_cpu_host.py checks equality with the reference-faithful
	oracle on random inputs.)"""
	counts = np.asarray(counts, dtype=np.int64)
	n_cells, Q = counts.shape
	if Q == 0:
		return [{m: ("mean 0.0", "std 0.0", "p50 0.0") for m in _METRICS} for _ in range(n_cells)]
	out = []
	for j in range(n_cells):
		n = float(ns[j])
		# counts are small integers (0..n): everything follows from their histogram -- exact integer sums for the mean, the
		# two middle order statistics for numpy's linearly interpolated median, sum h[v] (v - mean)^2 for the population std
		h = np.bincount(counts[j], minlength=int(ns[j]) + 1)
		v = np.arange(h.size, dtype=np.float64)
		m = float((h * v).sum() / Q)
		sd = float(np.sqrt((h * (v - m) ** 2).sum() / Q))
		cum = np.cumsum(h)
		hi_rank = Q // 2                          # 0-based ranks of the two middle order statistics
		lo_rank = hi_rank - 1 if Q % 2 == 0 else hi_rank
		l = float(np.searchsorted(cum, lo_rank + 1))
		hh = float(np.searchsorted(cum, hi_rank + 1))
		out.append({
			"common": _fmt(m, sd, _median_pair(l, hh)),
			"diff": _fmt(n - m, sd, _median_pair(n - hh, n - l)),
			"total": _fmt(n, 0.0, n),
			"common_frac": _fmt(m / n, sd / n, _median_pair(l / n, hh / n)),
			"diff_frac": _fmt((n - m) / n, sd / n, _median_pair((n - hh) / n, (n - l) / n)),
		})
	return out


def compute_overlap(indices_list1, indices_list2):
	"""Overlap metrics between corresponding pairs of index lists (any sequences of sequences)."""
	commons = []
	n = None
	for a, b in zip(indices_list1, indices_list2):
		assert len(a) == len(b), f"Len of both indices is not same => {len(a)} != {len(b)}"
		commons.append((len(set(np.asarray(a).tolist()).intersection(np.asarray(b).tolist())), len(a)))
	if len(commons) == 0:
		return overlap_stats_from_counts([], 1)
	lens = {l for _, l in commons}
	if len(lens) == 1:
		return overlap_stats_from_counts([c for c, _ in commons], lens.pop())
	# ragged lengths: per-pair fractions (the reference handles this the same way, pair by pair)
	c = np.array([x for x, _ in commons]); l = np.array([x for _, x in commons])
	per = {"common": c, "diff": l - c, "total": l, "common_frac": c / l, "diff_frac": (l - c) / l}
	return {m: ("mean {:.4f}".format(np.mean(v)), "std {:.4f}".format(np.std(v)), "p50 {:.4f}".format(np.percentile(v, 50)))
			for m, v in per.items()}


def flatten_overlap(overlap, prefix="exact_vs_reranked_approx_retvr"):
	"""The string -> float re-parse of every reference caller
	(eval/run_retrieval_eval_wrt_exact_crossenc.py:130-143): values are rounded to 1e-4."""
	flat = {}
	for metric, (mean_s, std_s, p50_s) in overlap.items():
		flat[f"{prefix}~{metric}_mean"] = float(mean_s[5:])
		flat[f"{prefix}~{metric}_std"] = float(std_s[4:])
		flat[f"{prefix}~{metric}_p50"] = float(p50_s[4:])
	return flat
