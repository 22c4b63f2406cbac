"""Dev tool: time the four launches of the fused score+top-k on the headline shape (HIP events on the launch stream)."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--Q", type=int, default=10000); ap.add_argument("--I", type=int, default=100000)
ap.add_argument("--K", type=int, default=256); ap.add_argument("--k", type=int, default=100); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--scan", type=int, default=1); ap.add_argument("--by-norm", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, a.I, generator=g, device=dev)
X = (torch.randn(a.Q, 64, generator=g, device=dev) @ torch.randn(64, a.K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(a.K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(a.K, a.I, generator=g, device=dev)).bfloat16()
Kp = ops.padded_k(a.K)
Xp = ops.pack_bf16(X, Kp); Etp = ops.pack_bf16(E.t().contiguous(), Kp, row_multiple=32)
ids = None
if a.by_norm:
	from anncur_amd.cur import _norm_sorted_pack
	Etp, ids = _norm_sorted_pack(E.t().contiguous().float(), Kp)
print("plan", ops.fused_plan(a.Q, a.I, Kp, a.k))
acc = np.zeros(9)
for i in range(a.iters + 2):
	(v, idx), ms = ops.score_topk_fused_timed(Xp, Etp, a.I, a.k, leading_sample=bool(a.by_norm), item_ids=ids)
	if i >= 2: acc += np.array(ms)
acc /= a.iters
print("stage_ms prepass/threshold/sweep-stage/select/sweep-kernels/n:", [round(float(x), 4) for x in acc], "sweep kernels TFLOP/s: %.1f" % (2.0 * a.Q * Kp * a.I / (acc[4] * 1e-3) / 1e12))
print("fallbacks:", int(ops._Workspace._bufs[("cuda", 0)][(-ops._Workspace._bufs[("cuda", 0)].data_ptr()) % 256:][:4].view(torch.int32).item()))
if a.scan:
	A = torch.randn(a.Q, a.I, generator=g, device=dev).bfloat16()
	ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	ops.rowwise_topk(A, a.k); torch.cuda.synchronize()
	ev0.record()
	for _ in range(a.iters): ops.rowwise_topk(A, a.k)
	ev1.record(); torch.cuda.synchronize()
	ms = ev0.elapsed_time(ev1) / a.iters
	print("exact scan ms: %.4f  GB/s: %.0f" % (ms, a.Q * a.I * 2 / ms / 1e6))
