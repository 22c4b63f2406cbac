"""Dev tool: GPU-side duration and spacing of each replayed step (timing events around every graph replay)."""
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
pin = torch.empty((4, Q), dtype=torch.int32, pin_memory=True)
def gpu_step():
	Xq = ops.gather_cols(A, anc); ap = ops.score_topk_fused(Xq, Et, I, k); ex = ops.rowwise_topk(A, k)
	return ops.overlap_counts(ex.indices, ap.indices, cells)
ops.copy_to_mapped_host(gpu_step(), pin); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
	ops.copy_to_mapped_host(gpu_step(), pin)
N = 60
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(N):
	ev[i][0].record(); g.replay(); ev[i][1].record()
	if i > 0: ev[i - 1][1].synchronize()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / N * 1e3
dur = np.array([a.elapsed_time(b) for a, b in ev]); gap = np.array([ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(N - 1)])
print("wall/step %.3f ms | GPU step duration min %.3f med %.3f max %.3f | gap between steps med %.3f max %.3f" % (wall, dur.min(), np.median(dur), dur.max(), np.median(gap), gap.max()))
