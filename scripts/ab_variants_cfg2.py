"""Dev tool: A/B of the sweep variants at the cfg2 shape in ONE process on ONE device (product library, flags of
anncur_score_topk_ex): default (32x32x16, two sub-tiles staggered inside a wave), mfma16 (16x16x32), qt1 (one sub-tile per
wave, cross-tile pipeline, 3 workgroups per CU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.cur import _norm_sorted_pack
dev = torch.device("cuda")
Q, I, K, k = 10000, 100000, int(os.environ.get("AB_K", "256")), int(os.environ.get("AB_TOPK", "100"))
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, I, generator=g, device=dev)
X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
Kp = ops.padded_k(K)
Etp, ids = _norm_sorted_pack(E.t().contiguous().float(), Kp)
Xp = ops.pack_bf16(X, Kp)
def run(tag, **kw):
	acc = np.zeros(9)
	for i in range(12):
		(v, idx), ms = ops.score_topk_fused_timed(Xp, Etp, I, k, leading_sample=True, item_ids=ids, **kw)
		if i >= 2: acc += np.array(ms)
	acc /= 10
	print("%-8s" % tag, [round(float(x), 4) for x in acc], "sweep TFLOP/s %.1f  total %.4f ms" % (2.0 * Q * Kp * I / (acc[4] * 1e-3) / 1e12, acc[:4].sum()), flush=True)
	return v, idx
ref = None
for rep in range(2):
	for tag, kw in (("default", {}), ("mfma16", {"mfma16": True}), ("qt1", {"qt1": True})):
		v, idx = run(tag, **kw)
		if ref is None: ref = (v, idx)
		else: print("   same values:", torch.equal(v, ref[0]), " same index sets: %.5f" % (torch.sort(idx, 1).values == torch.sort(ref[1], 1).values).float().mean().item())
