"""k_retvr = 500 (entry A's default) and 1000 at cfg2 size: the default sweep body (32x32x16 above k = 128) against the 16x16x32 body (mfma16=True),
one process, warm, round robin."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
from anncur_amd.cur import CURApprox   # noqa: E402
import bench   # noqa: E402
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["cfg2"]
A_train, A = bench.synth_device(cfg, dev, 0, row_seed=None)
rng = np.random.default_rng(0)
anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
anc_dev = ops.as_index(anc, dev)
cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc, approx_preference="rows", compute_dtype="bf16")
Xq = ops.gather_cols(A, anc_dev)
I = cfg["I"]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for kr in (int(x) for x in os.environ.get("KRS", "200,500,1000").split(",")):
	res = {}
	ref = None
	for rep in range(3):
		for name, kw in (("default", {}), ("mfma16", {"mfma16": True})):
			call = lambda: ops.score_topk_fused(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids, **kw)
			for _ in range(10): out = call()
			ev[0].record()
			for _ in range(15): out = call()
			ev[1].record(); torch.cuda.synchronize()
			res.setdefault(name, []).append(ev[0].elapsed_time(ev[1]) / 15)
			if ref is None: ref = out.indices.clone()
			else: assert torch.equal(out.indices, ref), (kr, name)
	plan = {n: ops.fused_plan(cfg["Q"], I, 256, kr, leading_sample=True, **kw) for n, kw in (("default", {}), ("mfma16", {"mfma16": True}))}
	for name, v in res.items():
		print(f"k_retvr {kr:5d} {name:8s} " + " ".join(f"{x:.4f}" for x in v) + f" ms   lg {plan[name]['lg']} stages {plan[name]['stage_end']} body {plan[name]['stage_pred']}", flush=True)
