"""Dev tool: how many queries of a fused call end in the exact repair path (ring wrapped inside a window / segment overflow), per data seed, at
the per-GPU shape of cfg4 (the multi-rank bench draws other queries on every rank: a rank whose data trips a repair pays ~2 ms per step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.cur import CURApprox
from anncur_amd.synth import protocol_b
cfg = dict(Q=6250, I=1000000, Ki=512, Kq=1024, k=100) if len(sys.argv) < 2 or sys.argv[1] == "cfg4" else dict(Q=10000, I=100000, Ki=256, Kq=512, k=100)
dev = torch.device("cuda")
anc = sorted(np.random.default_rng(0).choice(cfg["I"], size=cfg["Ki"], replace=False))
for row_seed in (None, 1, 2, 3, 4, 5, 6, 7, 8):
	A_train, A_test = protocol_b(cfg["Kq"], cfg["Q"], cfg["I"], dev, seed=0, row_seed=row_seed)
	cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc, approx_preference="rows", compute_dtype="bf16")
	Xq = ops.gather_cols(A_test, anc)
	Kp = cur._Etp.shape[1]
	if Xq.shape[1] != Kp: Xq = ops.pack_bf16(Xq, Kp)
	(v, i), nfb = ops.score_topk_fused(Xq, cur._Etp_sorted, cfg["I"], cfg["k"], return_fallbacks=True, leading_sample=True, item_ids=cur._item_ids)
	torch.cuda.synchronize()
	_, ms = ops.score_topk_fused_timed(Xq, cur._Etp_sorted, cfg["I"], cfg["k"], leading_sample=True, item_ids=cur._item_ids)
	print(f"row_seed {row_seed}: repaired queries {int(nfb.item())}, select {ms[3]:.3f} ms, sweep {ms[4]:.3f} ms", flush=True)
	del A_train, A_test, cur, Xq
