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
	reference's dict {metric: ("mean x", "std x", "p50 x")} (np.mean / population np.std / np.percentile 50).

	diff = n - common and total = n are affine in `common`, so their statistics follow exactly (integer sums are
	exact in float64); the two *_frac metrics are computed on common / n directly, as the reference does."""
	common = np.asarray(common, dtype=np.int64)
	if common.size == 0:
		return {m: ("mean 0.0", "std 0.0", "p50 0.0") for m in _METRICS}
	c = common.astype(np.float64)
	mean_c, std_c = float(np.mean(c)), float(np.std(c))
	half = c.size // 2                       # median = the reference's np.percentile(..., 50) (linear interpolation)
	part = np.partition(c, [half - 1, half] if c.size > 1 else [0])
	lo, hi = (part[half - 1], part[half]) if c.size % 2 == 0 else (part[half], part[half])
	lo, hi = float(lo), float(hi)
	p50_c = _median_pair(lo, hi)
	cf, df = c / n, (n - c) / n
	nc = n - c
	p50_cf = _median_pair(lo / n, hi / n)
	p50_df = _median_pair((n - hi) / n, (n - lo) / n)
	return {
		"common": _fmt(mean_c, std_c, p50_c),
		"diff": _fmt(float(np.mean(nc)), float(np.std(nc)), _median_pair(n - hi, n - lo)),
		"total": _fmt(float(n), 0.0, float(n)),
		"common_frac": _fmt(float(np.mean(cf)), float(np.std(cf)), p50_cf),
		"diff_frac": _fmt(float(np.mean(df)), float(np.std(df)), p50_df),
	}


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
