"""Soak (round 5): the timing-dependent paths repeated many times on one box, every result compared bit for bit with the first --
the fused top-k with the threshold ladder (counts arrive in any order), the one-pass entry-A cell with its hint, the one-call IVF search
(pairs land in their lists in the atomics' order), the ragged and short-row scans.  usage: python scripts/r5/soak.py [seconds per path]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
from anncur_amd.cur import CURApprox   # noqa: E402
from anncur_amd.nearest_nbr import IVFFlatIPIndex   # noqa: E402
import bench   # noqa: E402

def soak(name, fn, same, seconds):
	ref = fn(); torch.cuda.synchronize()
	n, bad, t0 = 0, 0, time.time()
	while time.time() - t0 < seconds:
		for _ in range(20):
			out = fn()
			n += 1
			if not same(ref, out): bad += 1
		torch.cuda.synchronize()
	print(f"{name:58s} {n:6d} calls  {bad} differ", flush=True)
	return bad

def main():
	seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
	dev = torch.device("cuda", 0)
	cfg = bench.CONFIGS["cfg2"]
	A_train, A = bench.synth_device(cfg, dev, 0, row_seed=None)
	rng = np.random.default_rng(0)
	anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
	anc_dev = ops.as_index(anc, dev)
	cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc, approx_preference="rows", compute_dtype="bf16")
	Xq = ops.gather_cols(A, anc_dev)
	I, kr = cfg["I"], cfg["k_retvr"]
	tk_same = lambda a, b: torch.equal(a.values, b.values) and torch.equal(a.indices, b.indices)
	bad = 0
	bad += soak("fused top-k, ladder (cfg2 size, k = 100)", lambda: ops.score_topk_fused(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids), tk_same, seconds)
	bad += soak("fused top-k, ladder, k = 500", lambda: ops.score_topk_fused(Xq, cur._Etp_sorted, I, 500, leading_sample=True, item_ids=cur._item_ids), tk_same, seconds)
	ef_same = lambda a, b: tk_same(a[0], b[0])   # (the error sums are atomics of floats: compared within round-off by the tests, not here)
	bad += soak("eval_fused with the norm-ordered hint", lambda: ops.eval_fused(Xq, cur._Etp, A, I, kr, hint=cur._Etp_sorted), ef_same, seconds)
	bad += soak("exact scan (k = 100)", lambda: ops.rowwise_topk(A, 100), tk_same, seconds)
	g = np.random.default_rng(5)
	n, d, nq, k = 100000, 768, 10000, 64
	C = g.standard_normal((200, d)).astype(np.float32)
	X = (C[g.integers(0, 200, n)] + 0.7 * g.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
	q = torch.tensor((C[g.integers(0, 200, nq)] + 0.7 * g.standard_normal((nq, d)).astype(np.float32)).astype(np.float32), device=dev)
	for dtype in ("bf16", "fp32"):
		index = IVFFlatIPIndex(d, 316, dtype=dtype); index.train(X); index.add(X); index.nprobe = 17
		bad += soak(f"IVF one-call search, {dtype} lists (10 000 queries, k = 64)", lambda: index.search_device(q, k), tk_same, seconds)
	print("soak:", "clean" if bad == 0 else f"{bad} differing results")
	return 1 if bad else 0

if __name__ == "__main__":
	raise SystemExit(main())
