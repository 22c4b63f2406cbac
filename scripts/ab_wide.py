"""Dev tool (experiments build): A/B of wide_kernel vs the ping-pong wide2_kernel, one process, one device.
ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so python scripts/ab_wide.py [K] [Q] [I] [k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import ops
dev = torch.device("cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
I = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, I, generator=g, device=dev)
X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
Kp = ops.padded_k(K)
Xp = ops.pack_bf16(X, Kp); Etp = ops.pack_bf16(E.t().contiguous(), Kp, row_multiple=32)
def run(tag):
	acc = np.zeros(9)
	for i in range(8):
		(v, idx), ms = ops.score_topk_fused_timed(Xp, Etp, I, k)
		if i >= 2: acc += np.array(ms)
	acc /= 6
	print("%-9s" % tag, [round(float(x), 4) for x in acc], "sweep TFLOP/s %.1f" % (2.0 * Q * Kp * I / (acc[4] * 1e-3) / 1e12), flush=True)
	return v, idx
ref = None
for rep in range(2):
	for tag, env in (("wide", None), ("pingpong", "1")):
		if env: os.environ["ANNCUR_DEBUG_WIDE2"] = env
		else: os.environ.pop("ANNCUR_DEBUG_WIDE2", None)
		v, idx = run(tag)
		if ref is None: ref = (v, idx)
		else: print("   same values:", torch.equal(v, ref[0]), " same index sets: %.5f" % (torch.sort(idx, 1).values == torch.sort(ref[1], 1).values).float().mean().item())
