"""anncur_eval_fused at cfg2 size with and without the first-threshold hint (anncur_eval_fused_ex: the prepass samples the leading tiles of the
norm-ordered copy of E^T that CURApprox keeps for the fused top-k; the sweep stays in item order).  Round robin, warm, one process."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
from anncur_amd.cur import CURApprox   # noqa: E402
import bench   # noqa: E402

def main():
	dev = torch.device("cuda", 0)
	cfg = bench.CONFIGS["cfg2"]
	A_train, A = bench.synth_device(cfg, dev, 0, row_seed=None)
	rng = np.random.default_rng(0)
	anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
	anc_dev = ops.as_index(anc, dev)
	cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc, approx_preference="rows", compute_dtype="bf16")
	Xq = ops.gather_cols(A, anc_dev)
	I, kr = cfg["I"], cfg["k_retvr"]
	ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
	def timed(fn, n=20):
		for _ in range(30): fn()
		ev[0].record()
		for _ in range(n): fn()
		ev[1].record(); torch.cuda.synchronize()
		return ev[0].elapsed_time(ev[1]) / n
	a = ops.eval_fused(Xq, cur._Etp, A, I, kr)
	b = ops.eval_fused(Xq, cur._Etp, A, I, kr, hint=cur._Etp_sorted)
	torch.cuda.synchronize()
	print("values equal:", bool(torch.equal(a[0].values, b[0].values)), " index sets equal:", bool(torch.equal(torch.sort(a[0].indices, 1).values, torch.sort(b[0].indices, 1).values)),
		  " err max rel diff: %.2e" % float(((a[1] - b[1]).abs() / a[1].abs().clamp_min(1e-30)).max()), flush=True)
	res = {"no hint": [], "hint": []}
	for rep in range(3):
		res["no hint"].append(timed(lambda: ops.eval_fused(Xq, cur._Etp, A, I, kr)))
		res["hint"].append(timed(lambda: ops.eval_fused(Xq, cur._Etp, A, I, kr, hint=cur._Etp_sorted)))
	for name, v in res.items():
		print(f"eval_fused, {name:8s} " + " ".join(f"{x:.3f}" for x in v) + " ms", flush=True)
	print(f"fused top-k, default (norm order, ladder): {timed(lambda: ops.score_topk_fused(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids)):.3f} ms")
	print(f"error kernel alone: {timed(lambda: ops.approx_error_packed(Xq, cur._Etp, A, I)):.3f} ms")

if __name__ == "__main__":
	main()
