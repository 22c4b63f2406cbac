"""anncur_eval_fused under the experiment modes of evalf_kernel (experiments build; ANNCUR_DEBUG_RING_STAGGER = mode, ANNCUR_DEBUG_TAU_BIAS):
what bounds the one-pass entry-A cell at cfg2 size.  Per-stage times from the library's own events are not available for this entry point:
whole-call times, with the retrieval chain's fixed launches (prepass, threshold, refinement, select) timed beside them."""
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
	def timed(fn, n=10):
		for _ in range(3): fn()
		ev[0].record()
		for _ in range(n): fn()
		ev[1].record(); torch.cuda.synchronize()
		return ev[0].elapsed_time(ev[1]) / n
	modes = (("as shipped", {}), ("exact tile L2-hot", {"ANNCUR_DEBUG_RING_STAGGER": "1"}), ("no error sums", {"ANNCUR_DEBUG_RING_STAGGER": "2"}))
	res = {}
	for rep in range(3):   # round robin, 30 untimed calls before each timing (a process's first seconds run slow)
		for name, env in modes:
			for k_ in ("ANNCUR_DEBUG_RING_STAGGER", "ANNCUR_DEBUG_TAU_BIAS"): os.environ.pop(k_, None)
			os.environ.update(env)
			for _ in range(30): ops.eval_fused(Xq, cur._Etp, A, I, kr)
			res.setdefault(name, []).append(timed(lambda: ops.eval_fused(Xq, cur._Etp, A, I, kr), n=20))
	for name, v in res.items():
		print(f"eval_fused, {name:24s} " + " ".join(f"{x:.3f}" for x in v) + " ms", flush=True)
	for k_ in ("ANNCUR_DEBUG_RING_STAGGER", "ANNCUR_DEBUG_TAU_BIAS"): os.environ.pop(k_, None)
	print(f"fused top-k on the item-ordered operand (same launches without the exact tile / sums): {timed(lambda: ops.score_topk_fused(Xq, cur._Etp, I, kr, mfma32=True)):.3f} ms")
	print(f"fused top-k, default (norm order, 16x16x32 body): {timed(lambda: ops.score_topk_fused(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids)):.3f} ms")
	print(f"error kernel alone: {timed(lambda: ops.approx_error_packed(Xq, cur._Etp, A, I)):.3f} ms")

if __name__ == "__main__":
	main()
