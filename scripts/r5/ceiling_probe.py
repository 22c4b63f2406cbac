"""Round 5 (GPU box): the cfg2 sweep on SURVIVOR-FREE operands, for `rocprofv3 --kernel-trace --stats -- python3 scripts/r5/ceiling_probe.py`:
the product library's fused call on the bench queries against an index copy whose item rows past the 2048th are scaled by 2^-7 (bench.py's
roofline.ceiling).  --normal: the same calls on the real index (the profiler's figure for the kernel as it ships, same process shape)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--normal", action="store_true")
	ap.add_argument("--calls", type=int, default=40)
	args = ap.parse_args()
	import bench
	from anncur_amd import ops
	from anncur_amd.cur import CURApprox
	dev = torch.device("cuda:0")
	cfg = bench.CONFIGS["cfg2"]
	A_train, A_test = bench.synth_device(cfg, dev, 0)
	rng = np.random.default_rng(0)
	anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
	anc_dev = ops.as_index(anc, dev)
	cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc,
					approx_preference="rows", compute_dtype="bf16")
	Q, I, k = cfg["Q"], cfg["I"], cfg["k_retvr"]
	Xq = ops.gather_cols(A_test, anc_dev)
	Et = cur._Etp_sorted
	if not args.normal:
		Et = Et.clone()
		Et[2048:] *= 0.0078125
	for _ in range(args.calls):
		ops.score_topk_fused(Xq, Et, I, k, leading_sample=True, item_ids=cur._item_ids)
	torch.cuda.synchronize()
	ws = ops._Workspace.get(1 << 20, dev)
	print("survivors per query:", ops.fused_survivors(ws, Q, I, cur._Etp.shape[1], k, leading_sample=True))


if __name__ == "__main__":
	main()
