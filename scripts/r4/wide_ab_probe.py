"""wide_kernel (Kp = 1024) at 10 000 x 100 000: eight waves of 128 x 64 against the four-wave 128 x 128 variant (ANNCUR_DEBUG_WIDE4, experiments
build), one process, warm, round robin; results compared."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
dev = torch.device("cuda", 0)
Q, I, K, k = 10000, 100000, 1024, 100
g = torch.Generator(device="cuda").manual_seed(1)
U = torch.randn(Q, 64, device=dev, generator=g); V = torch.randn(64, K, device=dev, generator=g)
X = ((U @ V) / 8 + 0.3 * torch.randn(Q, K, device=dev, generator=g)).to(torch.bfloat16)
W = torch.randn(I, 64, device=dev, generator=g)
Et = ((W @ V) / 8 + 0.3 * torch.randn(I, K, device=dev, generator=g)).mul_(1.0 / K ** 0.5).to(torch.bfloat16)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
res = {}
ref = None
for rep in range(3):
	for name, env in (("8 waves (128 x 64)", None), ("4 waves (128 x 128)", "1")):
		os.environ.pop("ANNCUR_DEBUG_WIDE4", None)
		if env: os.environ["ANNCUR_DEBUG_WIDE4"] = env
		for _ in range(5): out = ops.score_topk_fused(X, Et, I, k)
		ev[0].record()
		for _ in range(10): out = ops.score_topk_fused(X, Et, I, k)
		ev[1].record(); torch.cuda.synchronize()
		res.setdefault(name, []).append(ev[0].elapsed_time(ev[1]) / 10)
		if ref is None: ref = (out.values.clone(), out.indices.clone())
		else: assert torch.equal(out.values, ref[0]) and torch.equal(out.indices, ref[1]), name
for name, v in res.items():
	print(f"{name:22s} " + " ".join(f"{x:.3f}" for x in v) + f" ms  ({2e-9 * Q * I * K / v[-1]:.0f} TFLOP/s whole call)", flush=True)
