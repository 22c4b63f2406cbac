"""Round 5 dev tool (GPU box): the sweep's threshold ladder (csrc/score16.hpp) against the staged sweep of rounds 1-4 (ANNCUR_TOPK_STAGED)
on the bench matrices of cfg2, ONE process, alternating: per-stage times from HIP events, candidates kept per query, and parity -- both
against each other (values bit for bit, index sets) and against the dense route on a slice.

  python scripts/r5/ladder_probe.py [--k 100] [--rounds 3]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--k", type=int, default=100)
	ap.add_argument("--rounds", type=int, default=3)
	ap.add_argument("--config", default="cfg2")
	ap.add_argument("--out", default=None)
	args = ap.parse_args()
	import bench
	from anncur_amd import ops, _lib
	from anncur_amd.cur import CURApprox
	dev = torch.device("cuda:0")
	cfg = bench.CONFIGS[args.config]
	A_train, A_test = bench.synth_device(cfg, dev, 0)
	rng = np.random.default_rng(0)
	anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
	anc_dev = ops.as_index(anc, dev)
	cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc,
					approx_preference="rows", compute_dtype="bf16")
	Q, I, k = cfg["Q"], cfg["I"], args.k
	Kp = cur._Etp.shape[1]
	Xq = ops.gather_cols(A_test, anc_dev)
	if Xq.shape[1] != Kp:
		Xq = ops.pack_bf16(Xq, Kp)
	out = {"config": args.config, "k": k, "plans": {}, "runs": []}
	variants = (("ladder", {}), ("staged", {"staged": True}))
	for tag, kw in variants:
		out["plans"][tag] = ops.fused_plan(Q, I, Kp, k, leading_sample=True, **kw)
	print(json.dumps(out["plans"]), flush=True)
	res = {}
	for rep in range(args.rounds):
		for tag, kw in variants:
			acc = np.zeros(9)
			n = 10
			for i in range(n + 2):
				(v, idx), ms = ops.score_topk_fused_timed(Xq, cur._Etp_sorted, I, k, leading_sample=True, item_ids=cur._item_ids, **kw)
				if i >= 2: acc += np.array(ms)
			acc /= n
			ws = ops._Workspace.get(_lib.load().anncur_score_topk_workspace_bytes(Q, I, Kp, k), dev)
			surv = ops.fused_survivors(ws, Q, I, Kp, k, leading_sample=True, **kw)
			fb = int(ws[:4].view(torch.int32).item())
			row = {"variant": tag, "rep": rep, "prepass": acc[0], "threshold": acc[1], "sweep_incl_refine": acc[2], "select": acc[3], "sweep_kernels": acc[4],
				   "launches": acc[5], "stage_ms": list(acc[6:9]), "total": float(acc[:4].sum()), "survivors_per_query": surv, "fallbacks": fb,
				   "sweep_tflops": 2.0 * Q * Kp * I / (acc[4] * 1e-3) / 1e12}
			out["runs"].append(row)
			print(json.dumps(row), flush=True)
			res[tag] = (v.clone(), idx.clone())
	a, b = res["ladder"], res["staged"]
	out["values_bit_equal"] = bool(torch.equal(a[0], b[0]))
	out["indices_equal"] = bool(torch.equal(a[1], b[1]))
	# dense route on a slice (fp32 MFMA GEMM + exact scan)
	nq = 512
	dv, di = ops.score_topk_dense(Xq[:nq], cur._Etp_sorted[:I], k)
	ids = cur._item_ids.to(torch.int64)
	di_items = ids[di.to(torch.int64)]
	same = (torch.sort(di_items, 1).values == torch.sort(a[1][:nq].to(torch.int64), 1).values).all(dim=1).float().mean().item()
	out["rows_with_identical_index_set_vs_dense_slice"] = same
	out["max_rel_value_err_vs_dense_slice"] = ((dv - a[0][:nq]).abs().max() / dv.abs().max()).item()
	print(json.dumps({kk: out[kk] for kk in ("values_bit_equal", "indices_equal", "rows_with_identical_index_set_vs_dense_slice", "max_rel_value_err_vs_dense_slice")}), flush=True)
	if args.out:
		with open(args.out, "w") as f:
			json.dump(out, f, indent=1)


if __name__ == "__main__":
	main()
