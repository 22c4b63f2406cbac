"""Dev tool: per-step time of the bench pipeline with / without the D2H + host part, interleaved rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import ops
dev = torch.device("cuda")
Q, I, K, k = 10000, 100000, 256, 100
A = torch.randn(Q, I, device=dev).bfloat16()
anc = ops.as_index(sorted(np.random.default_rng(0).choice(I, K, replace=False)), dev)
Et = ops.pack_bf16(torch.randn(I, K, device=dev), K, 32)
cells = [(1, 100), (10, 100), (50, 100), (100, 100)]
pin = [torch.empty((4, Q), dtype=torch.int32, pin_memory=True) for _ in range(2)]
ev = [torch.cuda.Event() for _ in range(2)]
def gpu_step():
	Xq = ops.gather_cols(A, anc)
	ap = ops.score_topk_fused(Xq, Et, I, k)
	ex = ops.rowwise_topk(A, k)
	return ops.overlap_counts(ex.indices, ap.indices, cells)
def loop(n, mode):
	torch.cuda.synchronize(); t0 = time.perf_counter()
	for i in range(n):
		c = gpu_step()
		if mode >= 1: pin[i & 1].copy_(c, non_blocking=True)
		if mode >= 2: ev[i & 1].record()
		if mode >= 3 and i > 0: ev[(i - 1) & 1].synchronize()
	torch.cuda.synchronize()
	return (time.perf_counter() - t0) / n * 1e3
for r in range(4):
	print("round", r, " ".join("mode%d %.3f ms" % (m, loop(30, m)) for m in (0, 1, 2, 3, 0)), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(3):
	e0.record()
	for i in range(30): gpu_step()
	e1.record(); torch.cuda.synchronize()
	print("device-timed pure GPU loop: %.3f ms/step" % (e0.elapsed_time(e1) / 30))
