"""Round 5 (GPU box): the K-general fused kernel (wide_kernel, Kp > 512) at 10 000 x 100 000 x Kp: per-stage times from HIP events
(anncur_score_topk_timed), sweep TFLOP/s, and a parity check of the result against the dense route on a slice.
  python scripts/r5/wide_probe.py [--kp 1024] [--rounds 3]"""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--kp", type=int, default=1024)
	ap.add_argument("--rounds", type=int, default=3)
	ap.add_argument("--q", type=int, default=10000)
	ap.add_argument("--i", type=int, default=100000)
	ap.add_argument("--ab", action="store_true", help="(unused since the two-workgroup body was dropped)")
	ap.add_argument("--no-parity", action="store_true", help="ablation builds: the result is garbage by design")
	args = ap.parse_args()
	dev = torch.device("cuda", 0)
	Q, I, K, k = args.q, args.i, args.kp, 100
	g = torch.Generator(device="cuda").manual_seed(1)
	U = torch.randn(Q, 64, device=dev, generator=g); V = torch.randn(64, K, device=dev, generator=g)
	X = ((U @ V) / 8 + 0.3 * torch.randn(Q, K, device=dev, generator=g)).to(torch.bfloat16)
	W = torch.randn(I, 64, device=dev, generator=g)
	Et = ((W @ V) / 8 + 0.3 * torch.randn(I, K, device=dev, generator=g)).mul_(1.0 / K ** 0.5).to(torch.bfloat16)
	ref = None
	for rep in range(args.rounds):
		for tag, kw in (("", {}),):
			acc = np.zeros(9)
			for i in range(7):
				out, ms = ops.score_topk_fused_timed(X, Et, I, k, **kw)
				if i >= 2: acc += np.array(ms)
			acc /= 5
			n = max(1, int(round(acc[5])))
			if args.ab and not args.no_parity:
				if ref is None: ref = (out.values.clone(), out.indices.clone())
				else: assert torch.equal(out.values, ref[0]) and torch.equal(out.indices, ref[1]), "v1 and v2 disagree"
			print(tag + " rep %d: prepass %.4f threshold %.4f sweep(+refine) %.4f select %.4f | sweep kernels %.4f ms in %d launches = %.0f TFLOP/s (%.3f of 2500) | total %.4f ms" % (
				rep, acc[0], acc[1], acc[2], acc[3], acc[4], n, 2e-9 * Q * I * K / acc[4], 2e-9 * Q * I * K / acc[4] / 2500, acc[:4].sum()), flush=True)
	if args.no_parity:
		return
	nq = 256
	dv, di = ops.score_topk_dense(X[:nq], Et, k)
	same = (torch.sort(di.long(), 1).values == torch.sort(out.indices[:nq].long(), 1).values).all(dim=1).float().mean().item()
	print("rows with identical index set vs the dense route (first %d): %.4f; max rel value err %.2e" % (nq, same, ((dv - out.values[:nq]).abs().max() / dv.abs().max()).item()))


if __name__ == "__main__":
	main()
