"""Dev tool: can the HBM-bound exact scan of batch i+1 fill the bubbles of batch i's retrieval chain (gather, threshold, refinement,
select: latency-bound kernels that leave most of the chip idle)?  Two streams, the scan on the LOW priority one; eager launches.
usage: python scripts/overlap_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from anncur_amd import ops
from anncur_amd.cur import CURApprox
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda")
cfg = bench.CONFIGS["cfg2"]
A_train, A_test = bench.synth_device(cfg, dev, 0, row_seed=None)
anc = sorted(np.random.default_rng(0).choice(cfg["I"], size=cfg["Ki"], replace=False)); anc_dev = ops.as_index(anc, dev)
cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc, approx_preference="rows", compute_dtype="bf16")
Kp = cur._Etp.shape[1]; Q, I, k, kr = cfg["Q"], cfg["I"], cfg["k"], cfg["k_retvr"]
cells = [(t, kr) for t in (1, 10, 50, 100)]

def chain():
	Xq = ops.gather_cols(A_test, anc_dev)
	return ops.score_topk_fused(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids)

def run(mode):
	lo_p, hi_p = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
	H = torch.cuda.Stream(priority=-1) if mode != "serial" else torch.cuda.current_stream()
	L = torch.cuda.Stream(priority=0) if mode != "serial" else H
	if mode == "twostream_noprio": H, L = torch.cuda.Stream(), torch.cuda.Stream()
	exacts = [None, None]; evs = [torch.cuda.Event(), torch.cuda.Event()]
	def step(i):
		with torch.cuda.stream(L):
			exacts[i & 1] = ops.rowwise_topk(A_test, k); evs[i & 1].record(L)
		with torch.cuda.stream(H):
			approx = chain()
			H.wait_event(evs[i & 1])
			return ops.overlap_counts(exacts[i & 1].indices, approx.indices, cells)
	for i in range(10): step(i)
	torch.cuda.synchronize(); t0 = time.perf_counter()
	for i in range(steps): out = step(i)
	torch.cuda.synchronize(); dt = time.perf_counter() - t0
	print(f"{mode}: {1e3 * dt / steps:.4f} ms per step, {Q * steps / dt / 1e6:.2f} M queries/s")
for m in ("serial", "twostream_noprio", "prio"): run(m)

# the same with the bench's host protocol: counts copied into mapped pinned memory by a kernel, the host waits for step i-1's event
# (spin) before it launches step i+1
pinned = [torch.empty((len(cells), Q), dtype=torch.int32, pin_memory=True) for _ in range(2)]
def run_sync(mode, copy):
	H = torch.cuda.current_stream()
	L = torch.cuda.Stream() if mode != "serial" else H
	exacts = [None, None]; evs = [torch.cuda.Event(), torch.cuda.Event()]; done = [torch.cuda.Event(), torch.cuda.Event()]
	def step(i):
		s = i & 1
		if L is not H:
			with torch.cuda.stream(L):
				exacts[s] = ops.rowwise_topk(A_test, k); evs[s].record(L)
		else:
			exacts[s] = ops.rowwise_topk(A_test, k)
		approx = chain()
		if L is not H: H.wait_event(evs[s])
		c = ops.overlap_counts(exacts[s].indices, approx.indices, cells)
		if copy: ops.copy_to_mapped_host(c, pinned[s])
		done[s].record()
	for i in range(10): step(i)
	torch.cuda.synchronize(); t0 = time.perf_counter()
	for i in range(steps):
		step(i)
		if i >= 1:
			while not done[(i - 1) & 1].query(): pass
	torch.cuda.synchronize(); dt = time.perf_counter() - t0
	print(f"host-synced {mode} copy={copy}: {1e3 * dt / steps:.4f} ms per step")
for m in ("serial", "two"):
	for c in (False, True): run_sync(m, c)

# ... and with the host statistics of the bench between the wait and the next launch
from anncur_amd.eval_utils import overlap_stats_batch
def run_stats(mode):
	H = torch.cuda.current_stream()
	L = torch.cuda.Stream() if mode != "serial" else H
	exacts = [None, None]; evs = [torch.cuda.Event(), torch.cuda.Event()]; done = [torch.cuda.Event(), torch.cuda.Event()]
	def step(i):
		s = i & 1
		if L is not H:
			with torch.cuda.stream(L):
				exacts[s] = ops.rowwise_topk(A_test, k); evs[s].record(L)
		else:
			exacts[s] = ops.rowwise_topk(A_test, k)
		approx = chain()
		if L is not H: H.wait_event(evs[s])
		ops.copy_to_mapped_host(ops.overlap_counts(exacts[s].indices, approx.indices, cells), pinned[s])
		done[s].record()
	for i in range(10): step(i)
	torch.cuda.synchronize(); t0 = time.perf_counter()
	for i in range(steps):
		step(i)
		if i >= 1:
			while not done[(i - 1) & 1].query(): pass
			overlap_stats_batch(np.array(pinned[(i - 1) & 1].numpy()), [t for t, _ in cells])
	torch.cuda.synchronize(); dt = time.perf_counter() - t0
	print(f"host-synced + statistics {mode}: {1e3 * dt / steps:.4f} ms per step")
for m in ("serial", "two", "serial", "two"): run_stats(m)

