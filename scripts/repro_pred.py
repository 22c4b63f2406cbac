import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anncur_amd import ops
def case(Q, I, K, k, rank, noise, seed, **kw):
	g = torch.Generator().manual_seed(seed)
	X = torch.randn(Q, K, generator=g).bfloat16()
	E = (torch.randn(K, rank, generator=g) @ torch.randn(rank, I, generator=g) / rank ** 0.5 + noise * torch.randn(K, I, generator=g)).bfloat16()
	Kp = ops.padded_k(K)
	Xp = ops.pack_bf16(X.cuda(), Kp); Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, **kw)
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	bad = (v.cpu().double() - rv).abs().max().item()
	miss = [(q, sorted(set(ri[q].tolist()) - set(i[q].cpu().tolist()))) for q in range(Q)]
	miss = [(q, m, [(x // 32, x % 32) for x in m]) for q, m in miss if m]
	print(kw, "plan", ops.fused_plan(Q, I, Kp, k), "max |dv| %.4g" % bad, "fallbacks", int(nfb.item()), "missing (query, items, (tile,row)):", miss[:6], flush=True)
for env in ({}, {"ANNCUR_DEBUG_NO_PRED": "1"}, {"ANNCUR_DEBUG_ALL_PRED": "1"}):
	for k_ in list(os.environ):
		if k_.startswith("ANNCUR_DEBUG_"): del os.environ[k_]
	os.environ.update(env)
	print("== env", env)
	case(4, 12479, 241, 18, 2, 0.0, 0)
	case(300, 40000, 200, 60, 8, 0.05, 1)
	case(64, 70000, 256, 100, 16, 0.1, 2)
