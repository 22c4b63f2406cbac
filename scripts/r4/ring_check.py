"""round-4 dev check: the ring body (ring=True) against the default 4-wave barrier body -- values bit for bit, index sets equal, no fallbacks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from anncur_amd import ops
from anncur_amd.cur import _norm_sorted_pack
dev = torch.device("cuda")
def case(Q, I, K, k, seed=0, norm=True):
	g = torch.Generator(device=dev).manual_seed(seed)
	Z = torch.randn(64, I, generator=g, device=dev)
	X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
	E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
	Kp = ops.padded_k(K)
	Etf = E.t().contiguous().float()
	if norm: Etp, ids = _norm_sorted_pack(Etf, Kp)
	else: Etp, ids = ops.pack_bf16(Etf, Kp, row_multiple=32), None
	Xp = ops.pack_bf16(X, Kp)
	plan = ops.fused_plan(Q, I, Kp, k, leading_sample=norm, ring=True)
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, leading_sample=norm, item_ids=ids, ring=True)
	(v0, i0), nfb0 = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, leading_sample=norm, item_ids=ids)
	torch.cuda.synchronize()
	same_v = torch.equal(v, v0)
	same_i = torch.equal(i.sort(dim=1).values, i0.sort(dim=1).values)
	print(f"Q={Q} I={I} K={K} k={k} norm={norm}: body {plan['stage_pred']} splits {plan['splits']} nfb {int(nfb[0])}/{int(nfb0[0])} values_equal {same_v} index_sets_equal {same_i}", flush=True)
	return same_v and same_i and int(nfb[0]) == 0
ok = True
for args in [(1000, 40000, 256, 10), (10000, 100000, 256, 100), (777, 50001, 200, 64), (513, 30000, 128, 32), (10000, 100000, 128, 100), (5, 20000, 256, 7), (2049, 123457, 256, 128, 3, False)]:
	ok &= case(*args)
print("RING CHECK", "OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
