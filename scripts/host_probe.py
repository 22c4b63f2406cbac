"""Dev tool: where does the host time of one bench step go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.eval_utils import flatten_overlap, overlap_stats_from_counts
dev = torch.device("cuda")
Q, I, K, k = 10000, 100000, 256, 100
A = torch.randn(Q, I, device=dev).bfloat16()
anc = ops.as_index(sorted(np.random.default_rng(0).choice(I, K, replace=False)), dev)
Et = ops.pack_bf16(torch.randn(I, K, device=dev), K, 32)
cells = [(1, 100), (10, 100), (50, 100), (100, 100)]
def T(name, fn, n=20):
	fn(); torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(n): r = fn()
	t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
	print("%-22s host %.3f ms/call   (+sync %.3f ms total)" % (name, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3), flush=True)
	return r
Xq = T("gather_cols", lambda: ops.gather_cols(A, anc))
ap = T("score_topk_fused", lambda: ops.score_topk_fused(Xq, Et, I, k))
ex = T("rowwise_topk", lambda: ops.rowwise_topk(A, k))
cn = T("overlap_counts", lambda: ops.overlap_counts(ex.indices, ap.indices, cells))
pin = torch.empty((4, Q), dtype=torch.int32, pin_memory=True)
T("pinned copy_", lambda: pin.copy_(cn, non_blocking=True))
c = cn.cpu().numpy()
t0 = time.perf_counter()
for _ in range(20): r = {t: flatten_overlap(overlap_stats_from_counts(c[j], t)) for j, (t, _) in enumerate(cells)}
print("host stats %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count())
