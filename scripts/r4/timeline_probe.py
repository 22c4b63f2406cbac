"""Timeline of bench.py's partition placement at cfg2, eager launches, HIP events at every stage boundary of every step against ONE base
event: which launches of the two retrieval chains and the scan actually overlap.  Experiments build:
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so python scripts/r4/timeline_probe.py [variant ...]
variants: cur (scan masked to 96 CUs, two unmasked retrieval streams), one (one retrieval stream), disj (retrieval streams masked to the other 160)"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import _lib, ops   # noqa: E402
from anncur_amd.cur import CURApprox   # noqa: E402
import bench   # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]


def new_event():
	h = ctypes.c_void_p()
	assert hip.hipEventCreate(ctypes.byref(h)) == 0
	return h


def rec(ev, stream):
	assert hip.hipEventRecord(ev, ctypes.c_void_p(stream.cuda_stream)) == 0


def since(base, ev):
	ms = ctypes.c_float()
	rc = hip.hipEventElapsedTime(ctypes.byref(ms), base, ev)
	return float(ms.value) if rc == 0 else float("nan")


def masked_stream(lo, hi, n_cu=256):
	"""A stream on the CUs whose mask bits are lo .. hi - 1 (ops.cu_partition_streams' call, uncached)."""
	fn = hip.hipExtStreamCreateWithCUMask
	fn.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
	words = (n_cu + 31) // 32
	mask = (ctypes.c_uint32 * words)(*[sum(1 << b for b in range(32) if lo <= 32 * w + b < hi) for w in range(words)])
	h = ctypes.c_void_p()
	rc = fn(ctypes.byref(h), words, mask)
	assert rc == 0 and h.value, rc
	return torch.cuda.ExternalStream(h.value, device=torch.device("cuda", 0))


def main():
	variants = sys.argv[1:] or ["cur", "one", "disj"]
	device = torch.device("cuda", 0)
	cfg = bench.CONFIGS["cfg2"]
	raw = ctypes.CDLL(os.environ["ANNCUR_LIB"])
	fn = raw.anncur_score_topk_events
	fn.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
				   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
	fn.restype = ctypes.c_int
	work = torch.cuda.Stream(device=device)
	with torch.cuda.stream(work):
		A_train, A_test = bench.synth_device(cfg, device, 0, row_seed=None)
		rng = np.random.default_rng(0)
		anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
		anc_dev = ops.as_index(anc, device)
		cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc, approx_preference="rows", compute_dtype="bf16")
		Q, I, k, kr = cfg["Q"], cfg["I"], cfg["k"], cfg["k_retvr"]
		Kp = cur._Etp.shape[1]
		cells = [(t, kr) for t in (1, 10, 50, 100)]
		lib = _lib.load()
		nbytes = lib.anncur_score_topk_workspace_bytes(Q, I, Kp, kr)
		wss = [ops.fused_workspace(Q, I, Kp, kr, device) for _ in range(2)]
		pinned = [torch.empty((len(cells), Q), dtype=torch.int32).pin_memory() for _ in range(2)]
		val = [torch.empty((Q, kr), dtype=torch.float32, device=device) for _ in range(2)]
		idx = [torch.empty((Q, kr), dtype=torch.int32, device=device) for _ in range(2)]
		flags = _lib.TOPK_LEADING_SAMPLE
		torch.cuda.synchronize()
	out = {}
	for variant in variants:
		n_scan = int(os.environ.get("PROBE_SCAN_CUS", "96"))
		s_st = masked_stream(n_scan, 256) if variant == "disj2" else masked_stream(0, n_scan)
		if variant in ("disj", "disj2"):
			# the retrieval chains on two streams masked to the other CUs
			mains = [masked_stream(n_scan, 256), masked_stream(n_scan, 256)] if variant == "disj" else [masked_stream(0, n_scan), masked_stream(0, n_scan)]
		elif variant.startswith("ovl"):
			# the retrieval chains masked to CUs lo .. 255 with lo < n_scan: the scan keeps lo CUs to itself and shares the rest of its own
			lo = int(variant[3:])
			mains = [masked_stream(lo, 256), masked_stream(lo, 256)]
		elif variant.startswith("after"):
			m = torch.cuda.Stream(device=device); mains = [m, m]
			s_st = torch.cuda.Stream(device=device) if variant == "after" else masked_stream(0, int(variant[5:]))   # afterN: the scan on N CUs
		elif variant == "one":
			m = torch.cuda.Stream(device=device); mains = [m, m]
		else:
			mains = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
		n_steps, skip = 14, 4
		tail_done = [torch.cuda.Event() for _ in range(2)]
		scan_done = [torch.cuda.Event() for _ in range(2)]
		done = [torch.cuda.Event() for _ in range(2)]
		exact = [None, None]
		for slot in range(2):
			tail_done[slot].record(mains[slot])
		base = new_event()
		rec(base, mains[0])
		marks = []
		def step_after(i):
			"""scan(i + 1) on the side stream behind the last sweep stage of step i: it runs beside the latency-bound launches between two
			steps' sweeps (select, overlap, copy of step i; gather, prepass, threshold of step i + 1), the sweeps run alone"""
			slot = i & 1
			main = mains[0]
			m = {"scan": [new_event(), new_event()], "gather": new_event(), "fused": [new_event() for _ in range(11)], "end": new_event()}
			if i == 0:   # the first step's own scan
				m0 = [new_event(), new_event()]
				with torch.cuda.stream(s_st):
					exact[0] = ops.rowwise_topk(A_test, k)
				scan_done[0].record(s_st)
			rec(m["gather"], main)
			with torch.cuda.stream(main):
				Xq = ops.gather_cols(A_test, anc_dev)
				arr = (ctypes.c_void_p * 11)(*[e.value for e in m["fused"]])
				rc = fn(Xq.data_ptr(), Xq.stride(0), cur._Etp_sorted.data_ptr(), Kp, Q, I, Kp, kr, val[slot].data_ptr(), idx[slot].data_ptr(),
						wss[slot].data_ptr(), nbytes, flags, cur._item_ids.data_ptr(), ctypes.c_void_p(main.cuda_stream), arr)
				assert rc == 0, rc
			# the next step's scan: behind this step's last sweep stage (event 3 of the call)
			assert hip.hipStreamWaitEvent(ctypes.c_void_p(s_st.cuda_stream), m["fused"][3], 0) == 0
			rec(m["scan"][0], s_st)
			with torch.cuda.stream(s_st):
				exact[slot ^ 1] = ops.rowwise_topk(A_test, k)
			rec(m["scan"][1], s_st)
			scan_done[slot ^ 1].record(s_st)
			with torch.cuda.stream(main):
				main.wait_event(scan_done[slot])
				ops.copy_to_mapped_host(ops.overlap_counts(exact[slot].indices, idx[slot], cells), pinned[slot])
			rec(m["end"], main)
			done[slot].record(main)
			marks.append(m)
		def step(i):
			if variant.startswith("after"):
				return step_after(i)
			slot = i & 1
			main = mains[slot]
			m = {"scan": [new_event(), new_event()], "gather": new_event(), "fused": [new_event() for _ in range(11)], "end": new_event()}
			s_st.wait_event(tail_done[slot])
			rec(m["scan"][0], s_st)
			with torch.cuda.stream(s_st):
				exact[slot] = ops.rowwise_topk(A_test, k)
			rec(m["scan"][1], s_st)
			scan_done[slot].record(s_st)
			rec(m["gather"], main)
			with torch.cuda.stream(main):
				Xq = ops.gather_cols(A_test, anc_dev)
				arr = (ctypes.c_void_p * 11)(*[e.value for e in m["fused"]])
				rc = fn(Xq.data_ptr(), Xq.stride(0), cur._Etp_sorted.data_ptr(), Kp, Q, I, Kp, kr, val[slot].data_ptr(), idx[slot].data_ptr(),
						wss[slot].data_ptr(), nbytes, flags, cur._item_ids.data_ptr(), ctypes.c_void_p(main.cuda_stream), arr)
				assert rc == 0, rc
				main.wait_event(scan_done[slot])
				ops.copy_to_mapped_host(ops.overlap_counts(exact[slot].indices, idx[slot], cells), pinned[slot])
			rec(m["end"], main)
			tail_done[slot].record(main)
			done[slot].record(main)
			marks.append(m)
		for i in range(n_steps):
			step(i)
			if i >= 1:
				done[(i - 1) & 1].synchronize()   # at most two steps in flight, as bench.py's run_steps
		torch.cuda.synchronize()
		rows = []
		for i, m in enumerate(marks):
			f = [since(base, e) for e in m["fused"]]
			rows.append({"step": i, "slot": i & 1, "scan": [since(base, m["scan"][0]), since(base, m["scan"][1])], "gather_start": since(base, m["gather"]),
						 "prepass": [f[0], f[1]], "threshold_end": f[2], "stage1": [f[5], f[6]], "stage2": [f[7], f[8]], "select": [f[3], f[4]], "end": since(base, m["end"])})
		steady = rows[skip:]
		per_step = (steady[-1]["end"] - steady[0]["end"]) / (len(steady) - 1)
		out[variant] = {"ms_per_step": per_step, "rows": rows}
		print(f"== {variant}: {per_step:.3f} ms per step (eager launches, steps {skip}..{n_steps - 1})")
		t0 = steady[0]["gather_start"]
		for r in steady[:6]:
			g = lambda x: f"{x - t0:7.3f}"
			print(f" step {r['step']:2d} slot {r['slot']}  scan {g(r['scan'][0])}-{g(r['scan'][1])} | gather {g(r['gather_start'])} prepass {g(r['prepass'][0])}-{g(r['prepass'][1])} thr-{g(r['threshold_end'])}"
				  f" s1 {g(r['stage1'][0])}-{g(r['stage1'][1])} s2 {g(r['stage2'][0])}-{g(r['stage2'][1])} select {g(r['select'][0])}-{g(r['select'][1])} end {g(r['end'])}")
		d = lambda key: float(np.mean([r[key][1] - r[key][0] for r in steady]))
		print(f"   mean durations: scan {d('scan'):.3f}  prepass {d('prepass'):.3f}  stage1 {d('stage1'):.3f}  stage2 {d('stage2'):.3f}  select {d('select'):.3f}"
			  f"  chain (gather start -> end) {float(np.mean([r['end'] - r['gather_start'] for r in steady])):.3f}")
	os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
	json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r4_timeline.json"), "w"))


if __name__ == "__main__":
	main()
