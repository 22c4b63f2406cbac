"""Dev tool (experiments build): A/B of the Kp = 512 sweep with and without the cross-tile software pipeline (stagger1_tile) at the
per-GPU shape of cfg4, one process, one device.  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so python scripts/ab_stagger512.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from anncur_amd import ops
dev = torch.device("cuda")
Q, I, K, k = 6250, 1000000, 512, 100
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, I, generator=g, device=dev)
X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
from anncur_amd.cur import _norm_sorted_pack
Etp, ids = _norm_sorted_pack(E.t().contiguous().float(), 512)
Xp = ops.pack_bf16(X, 512)
del Z, E
def run(tag):
	acc = np.zeros(9)
	for i in range(7):
		(v, idx), ms = ops.score_topk_fused_timed(Xp, Etp, I, k, leading_sample=True, item_ids=ids)
		if i >= 2: acc += np.array(ms)
	acc /= 5
	print(tag, [round(float(x), 4) for x in acc], "TFLOP/s %.1f" % (2.0 * Q * 512 * I / (acc[4] * 1e-3) / 1e12), flush=True)
	return v, idx
os.environ["ANNCUR_DEBUG_PLAIN512"] = "1"
v0, i0 = run("plain    ")
del os.environ["ANNCUR_DEBUG_PLAIN512"]
v1, i1 = run("stagger1 ")
os.environ["ANNCUR_DEBUG_PLAIN512"] = "1"
v2, i2 = run("plain    ")
del os.environ["ANNCUR_DEBUG_PLAIN512"]
v3, i3 = run("stagger1 ")
print("values equal:", torch.equal(v0, v1), "indices equal:", torch.equal(i0, i1), torch.equal(v0, v2))
