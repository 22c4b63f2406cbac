"""Round 5 analysis: how many candidates per query does the sweep have to keep at cfg2, as a function of
where its threshold comes from?  Pure torch on the GPU box (dense S_hat of the bench matrices in the index's norm order); analysis
only -- nothing here is on the product path.

  python scripts/r5/survivor_model.py [--q 2048]

Prints, for the first --q queries of the bench workload:
  * tau from the k-th largest of: group maxima (today's prepass: groups of 16 over the leading 256 tiles), top-2 / top-4 per group,
    every sampled element;
  * survivors per query of a two-stage plan (stage end 689 tiles) under each, of three- and four-stage plans, and of a threshold
    refreshed every T tiles (the limit an in-sweep refinement could reach);
  * hit rates of the 64-lane compares (4 items x 16 queries of one accumulator register) per stage.
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
	ap.add_argument("--q", type=int, default=2048)
	ap.add_argument("--seed", type=int, default=0)
	ap.add_argument("--out", default=None)
	args = ap.parse_args()
	import bench
	from anncur_amd import ops
	from anncur_amd.cur import CURApprox
	dev = torch.device("cuda:0")
	cfg = bench.CONFIGS["cfg2"]
	A_train, A_test = bench.synth_device(cfg, dev, args.seed)
	rng = np.random.default_rng(args.seed)
	anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
	anc_dev = ops.as_index(anc, dev)
	cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc,
					approx_preference="rows", compute_dtype="bf16")
	Q, I, k = args.q, cfg["I"], cfg["k_retvr"]
	Xq = ops.gather_cols(A_test[:Q], anc_dev)
	Et = cur._Etp_sorted[:I].float()                      # [I, Kp] in norm order
	S = (Xq.float() @ Et.t()).contiguous()                 # [Q, I] fp32, item axis in sweep order
	TI = 32
	n_tiles = (I + TI - 1) // TI
	out = {"Q": Q, "I": I, "k": k, "n_tiles": n_tiles}

	def kth(vals, kk):   # k-th largest along dim 1
		return torch.topk(vals, kk, dim=1).values[:, kk - 1]

	# --- today's prepass: leading 256 tiles, groups of 16 items (two per tile)
	n_st = 256
	samp = S[:, :n_st * TI]
	g16 = samp.view(Q, -1, 16)
	srt = torch.sort(g16, dim=2, descending=True).values
	taus = {
		"group_max": kth(srt[:, :, 0], k),
		"top2_per_group": kth(srt[:, :, :2].reshape(Q, -1), k),
		"top4_per_group": kth(srt[:, :, :4].reshape(Q, -1), k),
		"all_sampled": kth(samp, k),
	}
	for n_st2 in (384, 512):
		taus[f"group_max_{n_st2}_tiles"] = kth(S[:, :n_st2 * TI].view(Q, -1, 16).max(dim=2).values, k)

	def survivors(lo, hi, tau):   # elements >= tau in tiles [lo, hi)
		return (S[:, lo * TI:min(hi * TI, I)] >= tau[:, None]).sum(dim=1).float()

	def exact_tau(hi, tau_floor):   # k-th best of tiles [0, hi)
		return torch.maximum(kth(S[:, :hi * TI], k), tau_floor)

	def plan(tau0, ends, skip_sample=False):
		tot = torch.zeros(Q, device=dev)
		per = []
		lo, tau = (n_st if skip_sample else 0), tau0
		for e in ends:
			s = survivors(lo, e, tau)
			per.append(s.mean().item())
			tot += s
			tau = exact_tau(e, tau)
			lo = e
		return tot.mean().item(), per

	res = {}
	for name, t0 in taus.items():
		tot, per = plan(t0, [689, n_tiles])
		res[name] = {"two_stage_689": {"total": tot, "per_stage": per}}
		# how many sampled elements pass it (the "111th element" reading of the group-maximum bound)
		res[name]["sample_elements_passing"] = (samp >= t0[:, None]).sum(dim=1).float().mean().item()
	t0 = taus["group_max"]
	for ends in ([400, n_tiles], [500, n_tiles], [900, n_tiles], [300, 1000, n_tiles], [256, 689, n_tiles], [256, 600, 1400, n_tiles], [128, 256, 512, 1024, 2048, n_tiles]):
		tot, per = plan(t0, ends)
		res[f"group_max_stages_{'_'.join(map(str, ends))}"] = {"total": tot, "per_stage": per}
	# candidates emitted by the prepass itself (its region never swept again)
	for ends in ([689, n_tiles], [600, 1400, n_tiles]):
		tot, per = plan(t0, ends, skip_sample=True)
		res[f"skip_sample_stages_{'_'.join(map(str, ends))}"] = {"total_after_sample": tot, "per_stage": per,
															   "sample_elements_passing": res["group_max"]["sample_elements_passing"]}
	# threshold refreshed every T tiles to the exact k-th best so far
	for T in (16, 64, 128, 256):
		tot = torch.zeros(Q, device=dev)
		tau = t0.clone()
		for lo in range(0, n_tiles, T):
			hi = min(lo + T, n_tiles)
			tot += survivors(lo, hi, tau)
			if hi * TI >= k:
				tau = exact_tau(hi, tau)
		res[f"refresh_every_{T}_tiles"] = {"total": tot.mean().item()}
	# --- in-sweep level counters (round 5 design): per query a few fixed LEVELS above tau0 (order statistics of the prepass' group maxima); every
	# kept candidate is counted at its level; once a level has seen k candidates the query's threshold moves up to it.  Waves re-read the counters
	# once per chunk of 4 tiles, and a row block's workgroups run ~13 chunks abreast: the counts a chunk sees lag by `lag` chunks.
	gm_sorted = torch.sort(srt[:, :, 0], dim=1, descending=True).values        # [Q, 512] group maxima, descending
	def level_plan(ranks, lag, chunk=4):
		levels = torch.stack([gm_sorted[:, m - 1] for m in ranks], dim=1)      # [Q, L] ascending in value as ranks descend
		tot = torch.zeros(Q, device=dev)
		n_chunks = (n_tiles + chunk - 1) // chunk
		counts = torch.zeros(Q, len(ranks), device=dev)
		hist = []                                                              # counts after each chunk
		tau = t0.clone()
		dense_tiles = 0
		for c in range(n_chunks):
			if c - lag - 1 >= 0:
				seen = hist[c - lag - 1]
				ok = seen >= k
				cand = torch.where(ok, levels, torch.full_like(levels, -float("inf"))).max(dim=1).values
				tau = torch.maximum(tau, cand)
			lo, hi = c * chunk * TI, min((c + 1) * chunk * TI, I)
			blk = S[:, lo:hi]
			keep = blk >= tau[:, None]
			tot += keep.sum(dim=1).float()
			# a kept candidate counts at every level it reaches
			for j in range(len(ranks)):
				counts[:, j] += (keep & (blk >= levels[:, j:j + 1])).sum(dim=1).float()
			hist.append(counts.clone())
			if keep.float().mean().item() > 0.005: dense_tiles += chunk
		return {"total": tot.mean().item(), "tiles_with_pass_rate_above_0.5pct": dense_tiles,
				"final_tau_is_exact_kth_frac": (tau >= exact_tau(n_tiles, t0) - 0).float().mean().item()}
	for ranks in ([80, 60, 45, 33], [85, 70, 58, 48, 40, 33, 25, 18], [90, 80, 70, 60, 52, 45, 38, 32, 27, 22, 18, 14], [88, 76, 66, 57, 49, 42, 36, 31, 26, 22, 18, 15, 12, 10, 8, 6]):
		for lag in (0, 16, 32):
			res[f"levels_{len(ranks)}_lag{lag}"] = dict(level_plan(ranks, lag), ranks=ranks)
	out["plans"] = res

	# --- hit rate of a 64-lane compare: 16 queries (consecutive) x 4 items (consecutive rows 4 g4 + r of a 16-item half)
	def block_rate(lo, hi, tau):
		sub = S[:, lo * TI:hi * TI] >= tau[:, None]                     # [Q, items]
		b = sub.view(Q // 16, 16, -1, 4).any(dim=3).any(dim=1)          # [Q/16, items/4]: (16 queries x 4 items) -- one register of one lane group
		# a compare covers 4 lane groups (g4) of DIFFERENT item quads of the same half: 16 items x 16 queries
		c = sub.view(Q // 16, 16, -1, 16).any(dim=3).any(dim=1)
		return b.float().mean().item(), c.float().mean().item()
	tau1 = exact_tau(689, t0)
	out["hit_rates"] = {
		"stage1_tiles_0_256": dict(zip(("quad_x_16q", "compare_16items_x_16q"), block_rate(0, 256, t0))),
		"stage1_tiles_256_689": dict(zip(("quad_x_16q", "compare_16items_x_16q"), block_rate(256, 689, t0))),
		"stage2": dict(zip(("quad_x_16q", "compare_16items_x_16q"), block_rate(689, 3125, tau1))),
	}
	print(json.dumps(out, indent=1))
	if args.out:
		with open(args.out, "w") as f:
			json.dump(out, f, indent=1)


if __name__ == "__main__":
	main()
