"""Whole-call time of the fused top-k at cfg2 size (norm-ordered operand, the bench's call) under the plan knobs of the experiments build:
first-stage share (ANNCUR_DEBUG_STAGES) and prepass sample size (ANNCUR_DEBUG_SAMPLE_GROUPS).  One process, warm, round robin."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
from anncur_amd.cur import CURApprox   # noqa: E402
import bench   # noqa: E402
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[os.environ.get("PLAN_CFG", "cfg2")]
A_train, A = bench.synth_device(cfg, dev, 0, row_seed=None)
rng = np.random.default_rng(0)
anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
anc_dev = ops.as_index(anc, dev)
cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc, approx_preference="rows", compute_dtype="bf16")
Xq = ops.gather_cols(A, anc_dev)
I, kr = cfg["I"], cfg["k_retvr"]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
KN = ("ANNCUR_DEBUG_STAGES", "ANNCUR_DEBUG_SAMPLE_GROUPS", "ANNCUR_DEBUG_CHUNK", "ANNCUR_DEBUG_CAPG", "ANNCUR_DEBUG_FLUSH_TILES", "ANNCUR_DEBUG_F1_SHRINK")
configs = [("default", {})]
if os.environ.get("PLAN_KNOBS") == "schedule":   # ticket chunk, segment capacity, flush period, first-stage shrink
	for c in ("4", "8", "16", "32", "64"): configs.append((f"chunk {c}", {"ANNCUR_DEBUG_CHUNK": c}))
	for c in ("512", "2048"): configs.append((f"capg {c}", {"ANNCUR_DEBUG_CAPG": c}))
	for c in ("1", "2", "8"): configs.append((f"flush tiles {c}", {"ANNCUR_DEBUG_FLUSH_TILES": c}))
	for c in ("0.5", "0.8", "1.0"): configs.append((f"f1 shrink {c}", {"ANNCUR_DEBUG_F1_SHRINK": c}))
elif os.environ.get("PLAN_CFG", "cfg2") == "cfg2":
	for f1 in ("0.10", "0.15", "0.18", "0.26", "0.32"): configs.append((f"stage1 {f1}", {"ANNCUR_DEBUG_STAGES": f1}))
	for f in ("0.08,0.30", "0.10,0.40"): configs.append((f"3 stages {f}", {"ANNCUR_DEBUG_STAGES": f}))
	for g in ("256", "384", "768", "1024"): configs.append((f"sample groups {g}", {"ANNCUR_DEBUG_SAMPLE_GROUPS": g}))
	for g, f1 in (("768", "0.18"), ("1024", "0.15"), ("384", "0.26")): configs.append((f"groups {g} + stage1 {f1}", {"ANNCUR_DEBUG_SAMPLE_GROUPS": g, "ANNCUR_DEBUG_STAGES": f1}))
else:   # the cfg4 per-GPU shape: three stages by default (0.06, 0.26)
	for f in ("0.04,0.20", "0.04,0.26", "0.06,0.20", "0.06,0.35", "0.09,0.30", "0.03,0.15", "0.10"):
		configs.append((f"stages {f}", {"ANNCUR_DEBUG_STAGES": f}))
	for g in ("384", "768", "1024", "2048"): configs.append((f"sample groups {g}", {"ANNCUR_DEBUG_SAMPLE_GROUPS": g}))
	for g, f in (("1024", "0.04,0.20"), ("2048", "0.04,0.20"), ("1024", "0.03,0.15")): configs.append((f"groups {g} + stages {f}", {"ANNCUR_DEBUG_SAMPLE_GROUPS": g, "ANNCUR_DEBUG_STAGES": f}))
def call(): return ops.score_topk_fused(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids)
ref = call().indices.clone()
res = {}
for rep in range(3):
	for name, env in configs:
		for k_ in KN: os.environ.pop(k_, None)
		os.environ.update(env)
		for _ in range(6): out = call()
		ev[0].record()
		for _ in range(8): out = call()
		ev[1].record(); torch.cuda.synchronize()
		res.setdefault(name, []).append(ev[0].elapsed_time(ev[1]) / 8)
		if rep == 0: assert torch.equal(torch.sort(out.indices, 1).values, torch.sort(ref, 1).values), name
for name, v in res.items():
	print(f"{name:32s} " + " ".join(f"{x:.4f}" for x in v) + " ms", flush=True)
