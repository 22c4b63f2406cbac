"""Dev tool: does running consecutive steps on two streams (two batches in flight, private workspaces) raise the throughput?"""
import os, sys, time, faulthandler; faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.cur import CURApprox
from anncur_amd.synth import protocol_b
dev = torch.device("cuda")
Q, I, Ki, Kq, k = 10000, 100000, 256, 512, 100
A_train, A_test = protocol_b(Kq, Q, I, dev, seed=0, rank=64, noise=0.05, dtype=torch.bfloat16)
anc = sorted(np.random.default_rng(0).choice(I, size=Ki, replace=False)); anc_dev = ops.as_index(anc, dev)
cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(Kq), col_idxs=anc, approx_preference="rows", compute_dtype="bf16", pinv_backend="device")
Kp = cur._Etp.shape[1]
cells = [(t, k) for t in (1, 10, 50, 100)]
def step(ws, side):
	main = torch.cuda.current_stream()
	side.wait_stream(main)
	with torch.cuda.stream(side):
		exact = ops.rowwise_topk(A_test, k)
	Xq = ops.gather_cols(A_test, anc_dev)
	approx = ops.score_topk_fused(Xq, cur._Etp_sorted, I, k, workspace=ws, leading_sample=True, item_ids=cur._item_ids)
	main.wait_stream(side)
	return ops.overlap_counts(exact.indices, approx.indices, cells)
def build(n_streams):
	streams = [torch.cuda.Stream() for _ in range(n_streams)]
	graphs = []
	for s in range(2):
		ws = ops.fused_workspace(Q, I, Kp, k, dev); side = torch.cuda.Stream()
		with torch.cuda.stream(streams[s % n_streams]):
			step(ws, side); torch.cuda.synchronize()
			g = torch.cuda.CUDAGraph()
			with torch.cuda.graph(g, stream=streams[s % n_streams], capture_error_mode="thread_local"):
				out = step(ws, side)
		graphs.append((g, streams[s % n_streams], out, ws, side))
	return graphs
for n_streams in (1, 2):
	graphs = build(n_streams)
	def run(n):
		for i in range(n):
			g, st = graphs[i & 1][:2]
			with torch.cuda.stream(st):
				g.replay()
	run(6); torch.cuda.synchronize()
	t0 = time.perf_counter(); run(60); torch.cuda.synchronize(); el = time.perf_counter() - t0
	print("%d replay stream(s): %.4f ms per step, %.2f M q/s" % (n_streams, 1e3 * el / 60, Q * 60 / el / 1e6), flush=True)
	ref = graphs[0][2].cpu(); assert torch.equal(ref, graphs[1][2].cpu())
