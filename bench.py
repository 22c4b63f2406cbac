#!/usr/bin/env python
"""Headline benchmark: CUR-256 retrieval + exact-rerank evaluation on a synthetic dense Q x I
score matrix (BASELINE.json configs[1]: Q=10k, I=100k bf16, 256 anchor items, k=k_retvr=100).

One "step" = one pass of the online query path over one batch of Q queries whose exact score
rows are already resident in HBM (index E = U.R built once, outside the timed region):
    C_q = A_test[:, anchors]  ->  fused S_hat = C_q.E + top-k_retvr  ->  exact top-k scan of A_test
    ->  overlap counts for top_k in {1,10,50,100}  ->  reference-format recall statistics on the host.
value = queries / second over all ranks (weak scaling: every rank evaluates its own Q queries).

    python bench.py --gpus N --steps K --warmup W
prints ONE JSON line on rank 0 (metric, value, roofline, cpu_baseline, ...).

N > 1: one process per GPU.  Under torchrun (RANK / WORLD_SIZE in the environment) this process is one of the ranks; started
plainly with --gpus N > 1 it launches the N ranks itself (torch.distributed.run, before any GPU call) and relays their result.
ONE workload per JSON key at every N: the top-level value / config is the headline workload (cfg2's shape on every rank: weak scaling,
the anchor rows assembled by ONE RCCL all-gather, timed as allgather_ms), so value(N) / value(1) is a scaling curve of one workload and
the N = 1 point is the single-GPU line.  At N > 1 the line also carries "cfg4": BASELINE.json configs[3] (50k x 1M bf16, 512 anchors)
at its per-GPU shape -- 6 250 query rows per rank -- measured in the same job with its own value, roofline, allgather_ms, solo_rank0.
"""
import argparse
import glob
import json
import os
import sys
import time

# The GPU box gives this process a CPU quota of ~16 cores on a 256-core host: BLAS / OpenMP pools sized for 256 threads
# (numpy.linalg.pinv in the index build, torch CPU ops) spin inside that quota and get the launching thread throttled.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
	os.environ.setdefault(_v, "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)


# ------------------------------------------------------------------ supervisor: the measurement runs in a CHILD of the process the launcher started
# The default placement of the exact scan (a CU-masked stream beside the retrieval, --scan-mode partition) once ended in a GPU memory access
# fault when RCCL was in the process (round 4: profiles/r04_rccl_single_rank_fault.txt).  A process that has touched the GPU can neither
# recover from such a fault nor re-exec itself on this pool -- so the process the launcher starts (plain `python bench.py`, or a torchrun rank)
# never touches the GPU: it starts the measuring process as a child ("worker", same arguments, same environment), relays its exit code, and if
# a partition-mode attempt dies of a GPU fault (SIGABRT / SIGSEGV / SIGBUS, or a fault message on stderr) it starts ONE more attempt as FRESH
# children with --scan-mode side; the line then says so (scan_mode.fallback_reason).  With N > 1 the supervisors of the ranks agree through
# a key-value store (the launcher's rendezvous store when torchrun provides one): a rank whose worker died flags the attempt, the others end
# their workers (which would otherwise wait in a collective for ever) and all of them start the fallback attempt together on a fresh port.
# --direct (or ANNCUR_BENCH_WORKER=1) runs the measurement in this process: what the profiling scripts put behind `rocprofv3 --`.
_FAULT_MARKS = ("Memory access fault", "HSA_STATUS_ERROR", "hipErrorIllegalAddress", "hipErrorLaunchFailure", "HW Exception", "GPU core dump")
_FAULT_RCS = (134, 139, 135, -6, -11, -7)


def _scan_mode_arg(argv):
	for i, a in enumerate(argv):
		if a == "--scan-mode" and i + 1 < len(argv): return argv[i + 1]
		if a.startswith("--scan-mode="): return a.split("=", 1)[1]
	return "serial" if "--no-overlap" in argv else None


def _supervise(argv):
	import signal
	import socket
	import subprocess
	import threading
	rank_env = os.environ.get("RANK")
	world = int(os.environ.get("WORLD_SIZE", "1")) if rank_env is not None else 1
	rank = int(rank_env) if rank_env is not None else 0
	store = None
	if world > 1:
		from datetime import timedelta
		from torch.distributed import TCPStore   # (imports torch; nothing here initialises a GPU)
		host, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
		agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
		store = TCPStore(host, port, world, is_master=(rank == 0 and not agent), timeout=timedelta(seconds=600), wait_for_workers=False)
	tag = f"anncur_bench/{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}/{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}"
	child = [None]

	def _on_signal(signum, _frame):   # the launcher ends its ranks with SIGTERM: take the worker along
		if child[0] is not None and child[0].poll() is None:
			child[0].kill()
		raise SystemExit(128 + signum)
	for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
		signal.signal(sg, _on_signal)

	def attempt(a, extra, reason):
		env = dict(os.environ, ANNCUR_BENCH_WORKER="1")
		env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
		if reason:
			env["ANNCUR_BENCH_FALLBACK_REASON"] = reason
		if store is not None:
			# the workers rendezvous on a port of their own (rank 0's worker hosts that store): a second attempt must not meet the keys of the first
			key = f"{tag}/a{a}/port"
			if rank == 0:
				with socket.socket() as sk:
					sk.bind(("127.0.0.1", 0)); wport = sk.getsockname()[1]
				store.set(key, str(wport))
			env["MASTER_PORT"] = store.get(key).decode()
			env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
		proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
		child[0] = proc
		tails = {"out": [], "err": bytearray()}

		def pump_err():
			for chunk in iter(lambda: proc.stderr.read1(65536), b""):
				sys.stderr.buffer.write(chunk); sys.stderr.buffer.flush()
				tails["err"] += chunk
				del tails["err"][:-16384]
		def pump_out():
			for line in proc.stdout:
				tails["out"].append(line.decode(errors="replace"))
		threads = [threading.Thread(target=pump_err, daemon=True), threading.Thread(target=pump_out, daemon=True)]
		for t in threads: t.start()
		fkey, okkey = f"{tag}/a{a}/failed", f"{tag}/a{a}/ok"
		reported, peer_failed = False, None
		while True:
			rc = proc.poll()
			if rc is not None and not reported:
				reported = True
				if store is None:
					break
				if rc == 0: store.add(okkey, 1)
				else: store.set(fkey, f"rank {rank}: worker exit code {rc}"); break
			if store is not None:
				if store.check([fkey]):
					peer_failed = store.get(fkey).decode()
					if proc.poll() is None: proc.kill()
					break
				if reported and store.add(okkey, 0) >= world:
					break
			time.sleep(0.2)
		rc = proc.wait()
		for t in threads: t.join(timeout=5)
		lines = [l for l in tails["out"] if l.startswith("{") and '"metric"' in l]
		err = tails["err"].decode(errors="replace")
		own_fault = rc != 0 and (rc in _FAULT_RCS or any(m in err for m in _FAULT_MARKS))
		return rc, lines, own_fault, peer_failed, err

	mode0 = _scan_mode_arg(argv)
	rc, lines, own_fault, peer_failed, err = attempt(0, [], None)
	failed = (rc != 0 and not lines) or peer_failed is not None
	if failed and mode0 in (None, "partition"):
		# was it a GPU fault somewhere in the job?  every rank has to reach the same verdict: one more store round
		fault = own_fault
		if store is not None:
			store.set(f"{tag}/verdict/{rank}", "fault" if own_fault else ("peer" if rc == 0 or peer_failed is not None and not own_fault else "error"))
			verdicts = [store.get(f"{tag}/verdict/{r}").decode() for r in range(world)]
			fault = "fault" in verdicts
		if fault:
			mark = next((m for m in _FAULT_MARKS if m in err), None)
			reason = (f"the --scan-mode partition attempt died (rank {rank}: exit code {rc}" + (f", '{mark}' on stderr" if mark else "") + "); "
					  "fresh worker processes re-ran the measurement with --scan-mode side") if own_fault else \
					 f"the --scan-mode partition attempt died on another rank ({peer_failed}); fresh worker processes re-ran the measurement with --scan-mode side"
			print(f"[bench] {reason}", file=sys.stderr, flush=True)
			extra = ["--scan-mode", "side"]
			rc, lines, own_fault, peer_failed, err = attempt(1, extra, reason)
			failed = (rc != 0 and not lines) or peer_failed is not None
	if lines and rank == 0:
		sys.stdout.write(lines[-1] if lines[-1].endswith("\n") else lines[-1] + "\n"); sys.stdout.flush()
	if failed:
		return rc if rc not in (0, None) else 1
	return 0


def _gpus_arg(argv):
	for i, a in enumerate(argv):
		if a == "--gpus" and i + 1 < len(argv) and argv[i + 1].isdigit(): return int(argv[i + 1])
		if a.startswith("--gpus=") and a[7:].isdigit(): return int(a[7:])
	return None


def _wants_supervisor(argv):
	if os.environ.get("ANNCUR_BENCH_WORKER") or "--direct" in argv or "-h" in argv or "--help" in argv:
		return False
	# `--gpus N` (N > 1) outside a launcher: this process only starts torch.distributed.run (self_launch); its RANKS become the supervisors
	return not (os.environ.get("RANK") is None and (_gpus_arg(argv) or 1) > 1)


if __name__ == "__main__" and _wants_supervisor(sys.argv[1:]):
	raise SystemExit(_supervise(sys.argv[1:]))

import numpy as np
import torch

CONFIGS = {
	# name: Q per GPU, I, n anchor items (Ki), n anchor queries (Kq = 2 Ki), k, k_retvr, storage dtype
	"cfg2": dict(Q=10000, I=100000, Ki=256, Kq=512, k=100, k_retvr=100, dtype="bf16"),
	"cfg4_per_gpu": dict(Q=6250, I=1000000, Ki=512, Kq=1024, k=100, k_retvr=100, dtype="bf16"),
	"small": dict(Q=2000, I=40000, Ki=64, Kq=128, k=10, k_retvr=100, dtype="bf16"),
}
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def synth_device(cfg, device, seed, row_seed=None):
	"""Protocol-B synthetic matrices on the device: shared item factors, low rank + noise, bf16 storage."""
	from anncur_amd.synth import protocol_b
	return protocol_b(cfg["Kq"], cfg["Q"], cfg["I"], device, seed=seed, row_seed=row_seed)


def self_launch(args):
	"""--gpus N > 1 without a torchrun environment: start the N ranks as children of this process -- which has not touched the GPU
	and never will -- relay rank 0's JSON line and exit with the launcher's code.  (A process that has initialised the GPU must
	not exec or re-launch itself on this pool.)"""
	import socket
	import subprocess
	with socket.socket() as sk:
		sk.bind(("127.0.0.1", 0))
		port = sk.getsockname()[1]
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
		   "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
	env = dict(os.environ)
	env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL / cross-process device memory on this host
	proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
	out = proc.stdout.decode(errors="replace")
	lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
	if lines:
		print(lines[-1], flush=True)
	else:
		sys.stderr.write(out)
	raise SystemExit(proc.returncode if proc.returncode != 0 or lines else 1)


def ivf_sideline(device, seed, n=100000, d=768, nq=10000, k=64):
	"""models/nearest_nbr.py:40-52 at the size of the reference's hard-negative mining (utils/data_process.py:343-365: every mention queries
	the entity index): n clustered vectors, nlist = floor(sqrt(n)), nprobe = floor(sqrt(nlist)); batched list-grouped search on the matrix
	cores.  Figures per index dtype: the whole search() call with host numpy in / out like FAISS, search_device() on device-resident queries (probe
	GEMM + top-nprobe, then anncur_ivf_search_grouped: pair grouping, tile GEMM, ragged scan of the packed score rows, id map), and the tile
	kernel alone against the matrix peak of the operand type (fp32: 157 TFLOP/s; bf16 lists: 2500).  Algorithmic flops = 2 x vectors scanned x d;
	the tile launch also multiplies the padding of its 128 x 128 (bf16) / 64 x 64 (fp32) tiles (reported as tile_flops_ratio)."""
	from anncur_amd import ops
	from anncur_amd.nearest_nbr import build_flat_or_ivff_index
	g = torch.Generator(device=device).manual_seed(seed + 99)
	C = torch.randn(200, d, generator=g, device=device)
	X = (C[torch.randint(0, 200, (n,), generator=g, device=device)] + 0.7 * torch.randn(n, d, generator=g, device=device)).cpu().numpy()
	Qv = (C[torch.randint(0, 200, (nq,), generator=g, device=device)] + 0.7 * torch.randn(nq, d, generator=g, device=device)).cpu().numpy()
	out = {"n": n, "d": d, "nq": nq, "k": k, "data": "synthetic clustered vectors (200 centres), host numpy in / out like FAISS",
		   "parity": "unpinned (FAISS absent): recall-judged in tests/"}
	for dtype, peak in (("fp32", 157.3), ("bf16", PEAK_BF16_TFLOPS)):
		t0 = time.perf_counter(); index = build_flat_or_ivff_index(X, force_exact_search=False, dtype=dtype); torch.cuda.synchronize(); build_s = time.perf_counter() - t0
		for _ in range(2): index.search(Qv, k)
		torch.cuda.synchronize()
		times = []
		for _ in range(5):
			t0 = time.perf_counter(); D, I = index.search(Qv, k); torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
		search_s = float(np.median(times))
		sizes = index._sizes
		q_dev = torch.as_tensor(Qv).to(device)
		probe = ops.score_topk_dense(q_dev, index.centroids, index.nprobe).indices.cpu().numpy()
		scanned = float(sizes[probe].sum())
		flops = 2.0 * scanned * d
		gemm_ms, scan_ms, tile_flops = [], [], 0.0
		k_eff = min(k, ops._lib.MAX_TOPK)
		grouped = index.grouped_call and ops.ivf_search_grouped_ok(k_eff, index.nlist)
		if grouped:
			# kernels alone, the one-call search (round 5): the call timed WITH and WITHOUT its tile launch on the same device-resident operands (HIP
			# events on the launch stream; without the launch the scan reads the scores the previous call left) -- the difference is the tile kernel,
			# the remainder pair grouping + ragged scan + id map.  The profiler's per-kernel figures are in profiles/r05_ivf_*.
			lists = index._Xs16 if index._Xs16 is not None else index._Xs
			qd = torch.zeros((nq, index._dp), dtype=torch.float32, device=device); qd[:, :d] = q_dev
			qd = ops.convert(qd, torch.bfloat16) if index._Xs16 is not None else qd
			pr = torch.as_tensor(probe).to(device)
			def timed(skip):
				for _ in range(2): ops.ivf_search_grouped(lists, index._offsets, index._ids, sizes, qd, pr, k_eff, _skip_gemm=skip)
				ms = []
				for _ in range(9):
					e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
					e0.record(); ops.ivf_search_grouped(lists, index._offsets, index._ids, sizes, qd, pr, k_eff, _skip_gemm=skip); e1.record(); torch.cuda.synchronize()
					ms.append(e0.elapsed_time(e1))
				return float(np.median(ms))
			full, rest = timed(False), timed(True)
			gemm_ms, scan_ms = [max(full - rest, 1e-6)], [rest]
			T = int(ops._lib.load().anncur_ivf_search_tile(ops._dt(lists), index._dp, ops._ld(lists), ops._ld(qd), nq))
			pairs = np.bincount(probe.reshape(-1), minlength=index.nlist)
			tile_flops = 2.0 * float((-(-pairs // T) * -(-sizes // T)).sum()) * T * T * index._dp
		else:
			for _ in range(5):   # kernels alone: device-resident queries and results, events on the launch stream
				prof = {}
				index.search_device(q_dev, k, profile=prof)
				torch.cuda.synchronize()
				gemm_ms.append(sum(e[0].elapsed_time(e[1]) for e in prof["events"])); scan_ms.append(sum(e[1].elapsed_time(e[2]) for e in prof["events"]))
				tile_flops = 2.0 * sum(int(t[-1].item()) for t in prof["tile_starts"]) * 64 * 64 * index._dp   # (the device-built worklist's tile count)
		gm, sm = float(np.median(gemm_ms)), float(np.median(scan_ms))
		# the whole search on DEVICE-RESIDENT queries and results (probe GEMM + top-nprobe, pair sort, tile worklist, per-list GEMMs, scan, id map):
		# what a caller that already holds its embeddings on the GPU pays (VERDICT r4 item 8 asks for this figure)
		dev_ms = []
		for _ in range(5):
			e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			e0.record(); index.search_device(q_dev, k); e1.record(); torch.cuda.synchronize()
			dev_ms.append(e0.elapsed_time(e1))
		dm = float(np.median(dev_ms))
		row = {"nlist": index.nlist, "nprobe": index.nprobe, "build_s": build_s, "search_ms": 1e3 * search_s, "queries_per_s": nq / search_s,
			   "vectors_scanned_per_query": scanned / nq,
			   "search_device_ms": dm, "queries_per_s_device_resident": nq / (dm * 1e-3),
			   "kernels": {"group_gemm_ms": gm, "scan_and_id_map_ms": sm, "queries_per_s_kernels_only": nq / ((gm + sm) * 1e-3), "tile_flops_ratio": tile_flops / flops,
						   "how": ("anncur_ivf_search_grouped timed with and without its tile launch: group_gemm_ms = the difference, scan_and_id_map_ms = the rest "
								   "(pair grouping, ragged scan, id map)") if grouped else "HIP events around the launches of the round-4 call sequence",
						   "roofline": {"bound": "mfma", "kernel": (("ivf_tile128_kernel (bf16 MFMA, 128 x 128 tiles)" if index._dp % 128 == 0 else "ivf_tile64_packed_kernel<bf16>") if dtype == "bf16" else "ivf_tile64_packed_kernel<float> (fp32 MFMA)") if grouped else ("ivf_group_scores_bf16_kernel (bf16 MFMA)" if dtype == "bf16" else "ivf_group_scores_kernel (fp32 MFMA)"),
										"achieved": flops / (gm * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": flops / (gm * 1e-3) / 1e12 / peak,
										"what": "the per-list GEMM launch alone (HIP events), algorithmic flops = 2 x vectors scanned x d"}},
			   "whole_call": {"achieved_tflops": flops / search_s / 1e12, "what": "search() incl. host numpy in / out (30 MB of queries over PCIe), probe, pair sort, exact scan of the scores"}}
		if dtype == "fp32":
			ref_I = I
		else:
			row["recall_vs_fp32_index"] = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(I.tolist(), ref_I.tolist())]))
		out[dtype] = row
	# (kept at the top level for readers of earlier rounds' lines: the fp32 index's whole-call figures)
	out.update({kk: out["fp32"][kk] for kk in ("nlist", "nprobe", "build_s", "search_ms", "queries_per_s", "vectors_scanned_per_query")})
	out["roofline"] = out["fp32"]["kernels"]["roofline"]
	return out


def _mark(msg):
	"""Progress marker on stderr (ANNCUR_BENCH_DEBUG): where a run was when it died -- the GPU is synchronised first, so a fault is
	attributed to the phase before the last marker printed."""
	if os.environ.get("ANNCUR_BENCH_DEBUG"):
		torch.cuda.synchronize()
		print(f"[bench mark] {msg}", file=sys.stderr, flush=True)


def run_config(args, cfg_name, ctx, light=False):
	"""The whole measurement of ONE workload (CONFIGS[cfg_name]) on every rank: data + index build, the timed K steps, the per-kernel
	side-lines.  Returns the result dict on rank 0 (None elsewhere).  light: the timed steps, solo_rank0 and the roofline figures only
	(the cfg4 sub-object of a multi-rank run)."""
	rank, world, device, use_dist, ranks_seen = ctx["rank"], ctx["world"], ctx["device"], ctx["use_dist"], ctx["ranks_seen"]
	cfg = CONFIGS[cfg_name]
	scan_mode = args.scan_mode
	from anncur_amd import _lib, ops
	from anncur_amd.cur import CURApprox
	from anncur_amd.eval_utils import flatten_overlap, overlap_stats_batch, overlap_stats_from_counts
	from anncur_amd.dist import allgather_anchor_rows
	_lib.load()

	# ------------------------------------------------------------------ data + index (outside the timed region)
	# Every rank owns Q queries; the index matrix (anchor queries' rows) is the same on every rank.  With N > 1 it is
	# assembled the way a row-sharded score matrix delivers it: each rank contributes Kq/N anchor rows, one all-gather.
	# one score model for the whole job (item factors from --seed); every rank draws its own queries and its own share of the anchor rows
	A_train, A_test = synth_device(cfg, device, args.seed, row_seed=None if world == 1 else args.seed * 1000 + rank + 1)
	_mark(f"{cfg_name}: data synthesised")
	allgather_ms = None
	if use_dist:
		# the path's ONE collective: every rank owns Kq / N of the anchor rows, an all-gather assembles R [Kq x I] everywhere
		allgather_anchor_rows(A_train, cfg["Kq"], rank, world)   # warm-up (communicator set-up, buffers)
		_mark("all-gather warm-up done")
		torch.distributed.barrier(); torch.cuda.synchronize()
		t0 = time.perf_counter()
		A_train = allgather_anchor_rows(A_train, cfg["Kq"], rank, world)
		torch.cuda.synchronize()
		t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
		torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
		allgather_ms = 1e3 * t.item()
		_mark("timed all-gather + all-reduce done")
	rng = np.random.default_rng(args.seed)
	anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
	anc_dev = ops.as_index(anc, device)
	def build_index():
		return CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc,
						 approx_preference="rows", compute_dtype="bf16")
	build_index()   # first call: code objects, allocator
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	cur = build_index()
	torch.cuda.synchronize()
	index_build_s = time.perf_counter() - t0
	_mark("index built")
	Kp = cur._Etp.shape[1]
	Q, I, k, kr = cfg["Q"], cfg["I"], cfg["k"], cfg["k_retvr"]
	assert ops.fused_supported(Q, I, Kp, kr), "headline config must run on the fused path"
	top_k_vals = [t for t in (1, 10, 50, 100) if t <= min(k, kr)]
	cells = [(t, kr) for t in top_k_vals]

	# Host statistics of step i (the reference's mean/std/median formatting over 4 x Q counts) run while the GPU already
	# executes step i+1: counts go D2H into one of two pinned buffers, an event marks their arrival.
	prof = {"launch": 0.0, "finish": 0.0}
	pinned = [torch.empty((len(cells), Q), dtype=torch.int32, pin_memory=True) for _ in range(2)]
	events = [torch.cuda.Event() for _ in range(2)]

	if args.no_overlap:
		scan_mode = "serial"
	if scan_mode is None:
		# Measured on MI355X, one box (round 3): cfg2 (Kp = 256) 0.939 ms per step with the partition, 1.007 with "side"; the cfg4 per-GPU
		# shape (Kp = 512: static tile shares, and a scan a third of the step) 9.00 vs 7.42 -- a workgroup that shares its CU with the scan
		# for the whole launch holds a static share back, the dynamic schedule just hands it fewer tiles.
		# Late round 4, one box, alternating runs: with the tiles of the Kp = 512 body drawn as XCD-sliced tickets the partition wins at the cfg4
		# per-GPU shape too -- 7.01 ms per step with "side", 6.69-6.76 with the scan on 96 CUs, 6.49-6.51 on 64 (7.8 on 32; 6.71 with one
		# retrieval stream): 12.5 GB of scan on 64 CUs and 6.4 TFLOP of sweep on the rest end together (gpurun_out/r4_cfg4_scan_modes.txt).
		scan_mode = "partition"
	# CUs of the scan's stream in the partition: the split that lets the scan and the sweeps end together -- 96 of 256 where the scan is a third
	# of the step's CU-time (cfg2: 64 CUs 1.041 ms, 96 0.939, 128 0.955), 64 where it is a quarter (cfg4 per-GPU shape, above)
	# Round 5: with ONE sweep launch per retrieval (threshold ladder, csrc/score16.hpp: no stage boundary, every workgroup draws tiles for the whole
	# launch) the retrieval absorbs whatever share of the chip the scan leaves it, and the optimum moved to a larger scan partition -- same box, steps
	# after the sustained loop / sustained: 96 CUs 0.900 / 0.909 ms, 128 0.878-0.885 / 0.887-0.889, 160 0.852-0.865 / 0.869-0.886, 192 0.872 / 0.877,
	# 224 0.986 / 0.994 (one retrieval stream: 0.867 / 0.870 at 128, 0.878 / 0.883 at 160, 0.953 at 192); --scan-mode side 0.959 / 0.955.
	# Late round 5: the scan's selector got cheaper (one histogram pass instead of two per compaction, no reductions: csrc/wave_select.hpp) -- alone
	# it streams 7-9 % faster on a part of the chip (96 CUs 0.580 -> 0.538 ms, 128 0.470 -> 0.437) -- and the split moved back: one box, sustained,
	# three runs each: 96 CUs 0.904-0.908 ms, 128 0.880-0.883, 160 0.908-0.915 (the scan before the change at 160: 0.891-0.893).
	scan_cus = args.scan_cus if args.scan_cus else (128 if Kp <= 256 else 64)
	rounds_rows = int(os.environ.get("ANNCUR_BENCH_ROUND_ROWS", "4096"))

	def retrieve(workspace=None):
		Xq = ops.gather_cols(A_test, anc_dev)                          # a2: C_q
		if Xq.shape[1] != Kp:
			Xq = ops.pack_bf16(Xq, Kp)
		return ops.score_topk_fused(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged, workspace=workspace)   # a6 + a7 fused (item rows in the index's norm order)

	def make_launcher(mode):
		"""launch(slot) for one way of placing the exact scan (a8's HBM-bound half) beside the MFMA-bound retrieval:
		  side      the scan starts on a second stream with the step, the hardware interleaves its workgroups with the retrieval's;
		  partition the chip's CUs are split between the two (streams with CU masks, ops.cu_partition_streams): the scan streams on
		            --scan-cus CUs while the retrieval's dynamic tile schedule runs on the rest -- no time-slicing of a CU between a
		            register-filling MFMA kernel and a latency-bound stream;
		  tail / chunks / serial   earlier schedules, kept for A/B."""
		side = ops.aux_stream(device) if mode in ("side", "tail") else None

		def gpu_step_tail():
			# The exact scan runs one wave per row and keeps `rounds_rows` rows in flight; a launch costs whole rounds: a FULL round is
			# bandwidth-bound (4096 rows of 200 KB: 145 us = 5.7 TB/s), a partial one is latency-bound (92 us however few rows).  So: the
			# full rounds first, alone on the chip; the partial round on the second stream beside the retrieval's latency-bound head
			# (gather, prepass, threshold: 145 us that leave HBM and most CUs idle); the MFMA-bound sweeps then run undisturbed.
			main = torch.cuda.current_stream()
			ev = torch.empty((Q, k), dtype=torch.float32, device=device); ei = torch.empty((Q, k), dtype=torch.int32, device=device)
			full = (Q // rounds_rows) * rounds_rows
			if full > 0:
				ops.rowwise_topk(A_test[:full], k, out=(ev[:full], ei[:full]))
			if full < Q:
				side.wait_stream(main)
				with torch.cuda.stream(side):
					ops.rowwise_topk(A_test[full:], k, out=(ev[full:], ei[full:]))
			approx = retrieve()
			main.wait_stream(side)
			return ops.overlap_counts(ei, approx.indices, cells)

		def gpu_step():
			if mode == "tail":
				return gpu_step_tail()
			main = torch.cuda.current_stream()
			if mode == "side":
				# the exact scan (HBM-bound, a8) is independent of the retrieval until the overlap count: it starts on a second stream with the
				# step and the hardware interleaves its workgroups with the retrieval's (measured, one box, alternating: 1.097 ms per step against
				# 1.134 on one stream; cutting the scan into row chunks beside the retrieval's latency-bound launches -- --scan-mode chunks,
				# anncur_eval_topk -- lost to both at this size: a chunk of one round of rows streams at 3.7 TB/s, the whole scan at 5.6)
				side.wait_stream(main)
				with torch.cuda.stream(side):
					exact = ops.rowwise_topk(A_test, k)
				approx = retrieve()
				main.wait_stream(side)
			else:
				Xq = ops.gather_cols(A_test, anc_dev)
				if Xq.shape[1] != Kp:
					Xq = ops.pack_bf16(Xq, Kp)
				exact, approx = ops.eval_topk(A_test, k, Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged, serial=mode == "serial")
			return ops.overlap_counts(exact.indices, approx.indices, cells)   # a8 rerank (closed form) + a10

		if mode == "partition":
			# The scan on a stream whose CU mask leaves it --scan-cus CUs (ops.cu_partition_streams), issued FIRST; the retrieval on the
			# unmasked stream: its workgroups take the other CUs outright and share the scan's (the dynamic tile schedule gives the slower
			# workgroups fewer tiles).  Three pieces, each a HIP graph of its own replayed ON its stream (a forked branch inside one graph
			# would run on a stream of the runtime's choosing, without the mask): scan | gather + retrieval | overlap counts + copy to the
			# mapped host buffer.  The scan of step i+1 may start while step i's retrieval runs: it waits only for the overlap count that
			# last read its result buffers (two steps back).
			# (neither piece on the default stream: hipExtStreamCreateWithCUMask makes a BLOCKING stream, which the NULL stream synchronises
			#  with implicitly -- with the retrieval there the two ran strictly one after the other: 1.41 ms per step)
			_, s_st = ops.cu_partition_streams(device, scan_cus)
			# --retr-streams 2 (default): the retrievals of consecutive steps on two streams with a workspace each -- the latency-bound tail of
			# one chain (refinement, select, overlap count: most CUs idle) overlaps the head of the next (gather, prepass)
			n_rs = 2 if args.retr_streams == 2 else 1
			mains = [torch.cuda.Stream(device=device) for _ in range(n_rs)]
			mains = [mains[s % n_rs] for s in range(2)]
			wss = [ops.fused_workspace(Q, I, Kp, kr, device) for _ in range(2)] if n_rs == 2 else [None, None]
			for m in mains: m.wait_stream(torch.cuda.current_stream())
			s_st.wait_stream(torch.cuda.current_stream())
			state = [{} for _ in range(2)]
			tail_done = [torch.cuda.Event() for _ in range(2)]
			# --fold-gather: C_q = A[:, anchors] out of the scan's own pass over A (anncur_rowwise_topk_gather: a2 folded into a8's first pass)
			# instead of the separate gather, which re-reads one 64-byte sector per anchor and query (164 MB at cfg2, 0.065 ms).  The
			# retrieval of step i then waits for scan i; scan i+1 is issued one step ahead, beside it (two result slots).  Measured (round 3,
			# MI355X): the fused kernel takes 0.50 ms on the whole chip against 0.357 (scan) + 0.049 (gather) -- 256 anchors per row are 1.3
			# per 64-vector step, so 72 % of the steps take the extraction branch (a divergent per-lane loop of 2-byte stores) and the table
			# word per vector doubles the loads in flight; the step goes 0.944 -> 1.26 ms.  Off by default; kept as the measured answer.
			fold = args.fold_gather and ops.rowwise_topk_gather_ok(A_test, k) and Kp == len(anc)
			tabs = ops.gather_tables(anc_dev, I, A_test.dtype) if fold else None
			def piece_scan(slot):
				if fold: state[slot]["exact"], state[slot]["Xq"] = ops.rowwise_topk_gather(A_test, k, tabs)
				else: state[slot]["exact"] = ops.rowwise_topk(A_test, k)
			def piece_retr(slot):
				state[slot]["approx"] = (ops.score_topk_fused(state[slot]["Xq"], cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged, workspace=wss[slot])
										 if fold else retrieve(wss[slot]))
			# (the counts go straight into the mapped pinned buffer: the copy launch that followed the overlap kernel until late round 5 was 5.5 us of every step)
			def piece_tail(slot): ops.overlap_counts(state[slot]["exact"].indices, state[slot]["approx"].indices, cells, mapped_host_out=pinned[slot])
			fns = (piece_scan, piece_retr, piece_tail)
			def stream_of(j, slot): return s_st if j == 0 else mains[slot]
			for slot in range(2):   # workspaces / code objects loaded outside capture
				for j, fn in enumerate(fns):
					with torch.cuda.stream(stream_of(j, slot)): fn(slot)
					torch.cuda.synchronize()
			pgraphs = None
			if not args.no_graph:
				try:
					pgraphs = []
					for slot in range(2):
						gs = []
						for j, fn in enumerate(fns):
							g = torch.cuda.CUDAGraph()
							with torch.cuda.graph(g, stream=stream_of(j, slot), capture_error_mode="thread_local"):
								fn(slot)
							gs.append(g)
						pgraphs.append(gs)
				except Exception as exc:
					print(f"[bench] HIP graph capture failed ({type(exc).__name__}: {exc}); launching eagerly", file=sys.stderr)
					pgraphs = None
					torch.cuda.synchronize()
			for slot in range(2): tail_done[slot].record(mains[slot])
			scan_done = [torch.cuda.Event() for _ in range(2)]
			scan_ready = [False, False]   # a scan whose results slot s holds has been issued and not yet consumed by a retrieval
			def run_piece(j, slot):
				with torch.cuda.stream(stream_of(j, slot)):
					if pgraphs is not None: pgraphs[slot][j].replay()
					else: fns[j](slot)
			def issue_scan(slot):
				s_st.wait_event(tail_done[slot])       # the overlap count that last read this slot's scan results
				run_piece(0, slot)
				scan_done[slot].record(s_st)
				scan_ready[slot] = True
			def launch_partition(slot):
				main = mains[slot]
				if not fold:   # scan i beside retrieval i, joined before the overlap count
					issue_scan(slot)
					run_piece(1, slot)
					main.wait_event(scan_done[slot])
					run_piece(2, slot)
				else:          # retrieval i needs C_q from scan i: the scans run ONE STEP AHEAD -- scan i+1 is issued first, beside retrieval i
					if not scan_ready[slot]:
						issue_scan(slot)
					issue_scan(slot ^ 1)   # (run_steps alternates the slots: this is the next step's)
					main.wait_event(scan_done[slot])
					scan_ready[slot] = False
					run_piece(1, slot)
					run_piece(2, slot)
				tail_done[slot].record(main)
				events[slot].record(main)
				return slot
			return launch_partition, pgraphs is not None

		# The ten launches of a step are captured once into a HIP graph and replayed: per-dispatch latency on a busy host
		# otherwise dominates (the kernels of a step total ~1.6 ms).  --no-graph keeps the eager launches.
		# The counts reach the host through a copy kernel that writes into mapped pinned memory (inside the graph): a step is ONE
		# graph replay, with no copy-engine hop whose cross-queue dependency a busy host would have to resolve.
		graphs = None
		if not args.no_graph:
			ops.copy_to_mapped_host(gpu_step(), pinned[0]); torch.cuda.synchronize()   # workspace / code objects loaded outside capture
			try:
				graphs = []
				for slot in range(2):
					g = torch.cuda.CUDAGraph()
					# thread_local: other threads of the process (the RCCL watchdog of a multi-rank run) may touch the runtime meanwhile
					with torch.cuda.graph(g, capture_error_mode="thread_local"):
						ops.copy_to_mapped_host(gpu_step(), pinned[slot])
					graphs.append(g)
			except Exception as exc:  # never lose the measurement to a capture problem: fall back to eager launches
				print(f"[bench] HIP graph capture failed ({type(exc).__name__}: {exc}); launching eagerly", file=sys.stderr)
				graphs = None
				torch.cuda.synchronize()

		def launch_one(slot):
			if graphs is not None:
				graphs[slot].replay()
			else:
				ops.copy_to_mapped_host(gpu_step(), pinned[slot])
			events[slot].record()
			return slot
		return launch_one, graphs is not None

	try:
		launchers = {scan_mode: make_launcher(scan_mode)}
	except Exception as exc:   # measurement plumbing only: a runtime without CU-masked streams falls back to the second-stream placement
		if scan_mode != "partition":
			raise
		print(f"[bench] --scan-mode partition unavailable ({type(exc).__name__}: {exc}); using --scan-mode side", file=sys.stderr)
		torch.cuda.synchronize()
		scan_mode = "side"
		launchers = {"side": make_launcher("side")}
	launch, graphed = None, False

	def finish(slot):
		t_a = time.perf_counter()
		while not events[slot].query():  # spin: a blocking wait can add milliseconds of wake-up latency on a busy host
			pass
		t_b = time.perf_counter()
		prof["wait"] = prof.get("wait", 0.0) + t_b - t_a
		c = np.array(pinned[slot].numpy())  # one memcpy out of the pinned (uncached-for-the-CPU) buffer, then the statistics
		prof["memcpy"] = prof.get("memcpy", 0.0) + time.perf_counter() - t_b
		stats = overlap_stats_batch(c, [t for t, _ in cells])
		return {t: flatten_overlap(stats[j]) for j, (t, _) in enumerate(cells)}

	step_no = [0]   # the result slots alternate across ALL calls of run_steps (the partition launcher issues the next slot's scan ahead)
	step_trace = [] if os.environ.get("ANNCUR_BENCH_STEP_TRACE") else None   # (debug) host time at the end of every loop iteration
	def run_steps(n):
		res, pending = None, None
		for i in range(n):
			t_a = time.perf_counter()
			cur_slot = launch(step_no[0] & 1)
			step_no[0] += 1
			t_b = time.perf_counter()
			if pending is not None:
				res = finish(pending)
			prof["launch"] += t_b - t_a
			prof["finish"] += time.perf_counter() - t_b
			if step_trace is not None: step_trace.append((time.perf_counter(), t_b - t_a, prof.get("wait", 0.0)))
			pending = cur_slot
		if pending is not None:
			res = finish(pending)
		return res

	def barrier():
		if use_dist:
			torch.distributed.barrier()
		torch.cuda.synchronize()

	_mark(f"launcher built ({scan_mode})")
	if scan_mode == "partition" and os.environ.get("ANNCUR_BENCH_FAIL_PARTITION_RANK") == str(rank):
		# fault injection (tests/test_gpu_bench_multirank.py): die the way a GPU memory fault ends a process -- SIGABRT, no Python teardown --
		# once the partition's streams and graphs exist; the supervisor must answer with fresh workers and --scan-mode side
		torch.cuda.synchronize()
		print("[bench] injected abort in partition mode (ANNCUR_BENCH_FAIL_PARTITION_RANK)", file=sys.stderr, flush=True)
		os.abort()
	launch, graphed = launchers[scan_mode]
	scan_mode_used = scan_mode
	# sustained: the same step looped for >= 10 s, BEFORE the W warm-up and K timed steps (round 5; it ran after them before).  Two reasons:
	# under a long MFMA load the chip lowers its clock (MI355X_MICROARCH.md 'DVFS give-back') and a 5 s utilisation sampler sees the GPU busy; and
	# the timed burst then starts on a chip that is out of its idle power state -- measured on one box, K = 20 steps after W = 5: 0.995 ms per step
	# straight after the (host-heavy, GPU-idle) graph capture, 0.919 after 200 warm-up steps, 0.918 after 1000, 0.904 for K = 200: the first tens
	# of milliseconds after idle run ~8 % slow, which is the box's clock ramp, not the step.  (--sustained-seconds 0: no such loop, the old order.)
	sustained = None
	sus_target = args.sustained_seconds if not light else min(args.sustained_seconds, 1.5)   # (the cfg4 sub-object of a multi-rank run: a short loop, same purpose, same field)
	if sus_target > 0:
		barrier()
		t0 = time.perf_counter(); n_sus = 0
		while True:
			run_steps(20); n_sus += 20
			torch.cuda.synchronize()
			if time.perf_counter() - t0 >= sus_target:
				break
		sus_s = time.perf_counter() - t0
		if use_dist:
			t = torch.tensor([sus_s / n_sus], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
			torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
			sus_step = t.item()
		else:
			sus_step = sus_s / n_sus
		sustained = {"value": world * Q / sus_step, "unit": "queries/s", "ms_per_step": 1e3 * sus_step, "seconds": sus_s, "steps": n_sus,
					 "order": "runs BEFORE the W warm-up and K timed steps: the timed burst starts at the chip's settled clock, not in its ramp out of idle"}
	_mark("sustained section done")
	res = run_steps(args.warmup)
	barrier()
	_mark("warm-up steps done")
	t0 = time.perf_counter()
	res = run_steps(args.steps)      # every one of the K steps is launched AND its statistics finished inside the timed region
	barrier()
	elapsed = time.perf_counter() - t0
	_mark("timed steps done")
	if step_trace is not None and len(step_trace) > 1:
		tr = np.array(step_trace)
		print("[bench step trace] ms between loop iterations (warm-up + timed): " + " ".join(f"{x:.3f}" for x in np.diff(tr[:, 0]) * 1e3), file=sys.stderr)
		print("[bench step trace] host ms in launch(): " + " ".join(f"{x:.3f}" for x in tr[1:, 1] * 1e3), file=sys.stderr)
		print("[bench step trace] host ms spinning on the previous step's event: " + " ".join(f"{x:.3f}" for x in np.diff(tr[:, 2]) * 1e3), file=sys.stderr)
	if os.environ.get("ANNCUR_BENCH_DEBUG"):
		print(f"[bench debug] per step: launch {1e3 * prof['launch'] / (args.steps + args.warmup):.3f} ms, "
			  f"finish {1e3 * prof['finish'] / (args.steps + args.warmup):.3f} ms (of which event wait {1e3 * prof.get('wait', 0) / (args.steps + args.warmup):.3f} ms, pinned memcpy {1e3 * prof.get('memcpy', 0) / (args.steps + args.warmup):.3f} ms)", file=sys.stderr)
	if use_dist:
		t = torch.tensor([elapsed], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
		torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
		elapsed = t.item()
		_mark("max-over-ranks of the step time done")
	ms_per_step = 1e3 * elapsed / args.steps
	value = world * Q * args.steps / elapsed
	if os.environ.get("ANNCUR_BENCH_NULL_PROBE"):
		# the repro without RCCL: the NULL-stream work of one device-side reduction (8-byte H2D, D2H, a fill) between two batches of steps
		with torch.cuda.stream(torch.cuda.default_stream(device)):
			t = torch.tensor([elapsed], device=device, dtype=torch.float64); t.item()
			torch.zeros(1, device=device)
		torch.cuda.synchronize()
		_mark("null-stream probe: copies done")
		run_steps(args.steps); torch.cuda.synchronize()
		_mark("null-stream probe: steps done")

	# the same K steps on rank 0 ALONE (the other ranks wait): the single-GPU rate of this very workload inside the same job, so that
	# weak-scaling efficiency = value / (N x solo) can be read off one line
	solo = None
	if use_dist:
		barrier()
		_mark("solo: barrier passed")
		if rank == 0:
			t0 = time.perf_counter(); run_steps(args.steps); torch.cuda.synchronize()
			solo = Q * args.steps / (time.perf_counter() - t0)
		_mark("solo: steps done")
		barrier()
	_mark("solo section done")
	# ------------------------------------------------------------------ per-kernel durations (HIP events on the launch stream)
	def preheat(seconds=0.4):
		"""The per-kernel figures below come from short event-timed calls with a host synchronisation each: on a chip that has just idled (the
		barrier / solo sections above) they read 15-20 % long -- the box's clock ramp, not the kernels (one run measured the sweep at 0.548 ms
		there and 0.449 after the sustained loop).  A short loop of the real step in front of each block keeps them comparable."""
		t0_ = time.perf_counter()
		while time.perf_counter() - t0_ < seconds:
			run_steps(20)
		torch.cuda.synchronize()
	if args.sustained_seconds > 0: preheat()
	stage = np.zeros(9)
	Xq = ops.gather_cols(A_test, anc_dev)
	if Xq.shape[1] != Kp:
		Xq = ops.pack_bf16(Xq, Kp)
	n_prof = max(3, min(args.steps, 10))
	# Every event-timed call below is queued BEHIND two untimed calls of the same op, without a host synchronisation in between: its kernels then start
	# where the timed steps' kernels do -- on a busy chip at its loaded clock --, not after the idle gap of the previous call's read-back (one box read
	# the sweep at 0.558 ms with a synchronisation before every timed call, 0.449 with the queue kept full; the steps of that same run took 0.859 ms).
	for _ in range(n_prof):
		for _ in range(2): ops.score_topk_fused(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged)
		_, ms = ops.score_topk_fused_timed(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged)
		stage += np.array(ms)
	stage /= n_prof
	ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
	scan_ms = gath_ms = 0.0
	for _ in range(n_prof):
		for _ in range(2): ops.gather_cols(A_test, anc_dev)
		ev[0].record(); ops.gather_cols(A_test, anc_dev); ev[1].record()
		ops.rowwise_topk(A_test, k)
		ev[2].record(); ops.rowwise_topk(A_test, k); ev[3].record()
		torch.cuda.synchronize()
		gath_ms += ev[0].elapsed_time(ev[1]) / n_prof
		scan_ms += ev[2].elapsed_time(ev[3]) / n_prof
	# retrieve-only (the deployable path: gather C_q + fused S_hat/top-k, no exact scan / overlap), eager launches
	n_ro = max(5, min(args.steps, 20))
	ev[0].record()
	for _ in range(n_ro):
		Xr = ops.gather_cols(A_test, anc_dev)
		if Xr.shape[1] != Kp:
			Xr = ops.pack_bf16(Xr, Kp)
		ops.score_topk_fused(Xr, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged)
	ev[1].record(); torch.cuda.synchronize()
	retrieve_ms = ev[0].elapsed_time(ev[1]) / n_ro
	def survivors(kk):   # candidates per query the sweep of the last call kept (the default workspace those calls ran on)
		ws_ = ops._Workspace.get(_lib.load().anncur_score_topk_workspace_bytes(Q, I, Kp, kk), device)
		return ops.fused_survivors(ws_, Q, I, Kp, kk, leading_sample=True, staged=args.sweep_staged)
	survivors_k = survivors(kr)
	survivors_k500 = None
	# retrieve-only at k_retvr = 500, the reference's default for entry A (crossenc.py:238): more survivors, wave-level select with
	# 8 keys per lane, predicated sweep stages
	retrieve500_ms = None
	if not args.no_k500 and not light and ops.fused_supported(Q, I, Kp, 500):
		for _ in range(2):
			ops.score_topk_fused(Xr, cur._Etp_sorted, I, 500, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged)
		ev[0].record()
		for _ in range(n_ro):
			Xr = ops.gather_cols(A_test, anc_dev)
			if Xr.shape[1] != Kp:
				Xr = ops.pack_bf16(Xr, Kp)
			ops.score_topk_fused(Xr, cur._Etp_sorted, I, 500, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged)
		ev[1].record(); torch.cuda.synchronize()
		retrieve500_ms = ev[0].elapsed_time(ev[1]) / n_ro
		survivors_k500 = survivors(500)
	# the same index build with the reference's own pseudo-inverse call (numpy.linalg.pinv on the host: U bit-identical to the
	# reference) instead of the default "auto" route (fp64 Newton-Schulz on the GPU while the block is well conditioned)
	torch.cuda.synchronize(); t0 = time.perf_counter()
	CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc,
			  approx_preference="rows", compute_dtype="bf16", pinv_backend="numpy")
	torch.cuda.synchronize(); index_build_numpy_s = time.perf_counter() - t0
	# one grid cell of entry point A (crossenc.py:84-147: retrieval + per-row error terms): anncur_eval_fused (ONE sweep, round 4) against the
	# two-kernel route (fused top-k + error kernel); HIP events, eager launches; the exact scan of the cell is exact_scan above
	entry_a = None
	if not light and ops.eval_fused_ok(Kp, A_test, Q, I, kr):
		def _timed(fn, n=max(5, min(args.steps, 20))):
			for _ in range(2): fn()
			ev[0].record()
			for _ in range(n): fn()
			ev[1].record(); torch.cuda.synchronize()
			return ev[0].elapsed_time(ev[1]) / n
		one_ms = _timed(lambda: ops.eval_fused(Xq, cur._Etp, A_test, I, kr))
		two_ms = _timed(lambda: (ops.score_topk_fused(Xq, cur._Etp_sorted, I, kr, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged), ops.approx_error_packed(Xq, cur._Etp, A_test, I)))
		err_ms = _timed(lambda: ops.approx_error_packed(Xq, cur._Etp, A_test, I))
		entry_a = {"eval_fused_ms": one_ms, "two_kernel_route_ms": two_ms, "error_kernel_alone_ms": err_ms, "exact_scan_ms": scan_ms,
				   "cell_kernels_ms": one_ms + scan_ms, "cell_kernels_two_kernel_route_ms": two_ms + scan_ms,
				   "what": "retrieval (prepass, threshold, sweep stages, refinement, select) + per-row sum (S_hat - A)^2, sum A^2 of one entry-A grid cell at this size; "
						   "eval_fused = one S_hat GEMM per sweep stage (csrc/score_evalf.hpp), two-kernel route = the fused top-k and error_lds_kernel (two S_hat GEMMs)"}
	# ------------------------------------------------------------------ roofline.ceiling: the same sweep with (almost) nothing to keep
	# "a bare-loop ceiling measurement (same tile schedule, survivors suppressed)".  On the PRODUCT library: the same queries against
	# a copy of the index whose item rows from the 2049th on are scaled by 2^-7 (an exponent shift: the bf16 mantissas, hence the operands' bit
	# activity, stay what they are) -- every query's top-k then lies among the first 2048 items, the prepass threshold clears everything after the
	# first 64 tiles, and the sweep kernel runs its plan (tickets, tile DMA, MFMA chain, filter compares, barrier) with ~no hit block taken.
	ceiling = None
	if not light and not args.no_ceiling and I > 8192:
		if args.sustained_seconds > 0: preheat()
		Et_bare = cur._Etp_sorted.clone()
		Et_bare[2048:] *= 0.0078125
		bare = np.zeros(9)
		for i_ in range(n_prof + 1):
			_, ms = ops.score_topk_fused_timed(Xq, Et_bare, I, kr, leading_sample=True, item_ids=cur._item_ids, staged=args.sweep_staged)
			if i_ > 0: bare += np.array(ms)
		bare /= n_prof
		ws_ = ops._Workspace.get(_lib.load().anncur_score_topk_workspace_bytes(Q, I, Kp, kr), device)
		bare_surv = ops.fused_survivors(ws_, Q, I, Kp, kr, leading_sample=True, staged=args.sweep_staged)
		ceiling = {"sweep_kernels_ms": float(bare[4]), "achieved": 2.0 * Q * Kp * I / (float(bare[4]) * 1e-3) / 1e12, "unit": "TFLOP/s",
				   "frac": 2.0 * Q * Kp * I / (float(bare[4]) * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "survivors_per_query": bare_surv,
				   "what": "the same sweep kernel(s), plan and queries against an index copy whose item rows past the 2048th are scaled by 2^-7: the top-k lies in the first 64 tiles and "
						   "the rest of the sweep takes no hit block -- what this tile loop does on this box with the candidate path idle (HIP events, this run)"}
		del Et_bare
	_mark("per-kernel timings done")
	plan_now = ops.fused_plan(Q, I, Kp, kr, leading_sample=True, staged=args.sweep_staged)
	n_sweep = max(1, int(round(stage[5])))                 # the sweep runs as n_sweep launches of the same kernel (threshold refined in between)
	sweep_flops = 2.0 * Q * Kp * I / n_sweep               # algorithmic flops per launch (average over the stages)
	sweep_ms = stage[4] / n_sweep                          # average launch duration of score_kernel<Kp,sweep>
	sweep_tflops = sweep_flops / (sweep_ms * 1e-3) / 1e12
	scan_bytes = Q * I * 2 + Q * k * 8
	# HBM bytes per launch of the sweep from the PMC passes of THIS config (scripts/profile_round.sh -> profiles/); null if this
	# config has not been profiled -- never another shape's number
	traffic = traffic_src = None
	for tfile in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_{cfg_name}.json")), reverse=True):
		try:
			d = json.load(open(tfile))
			if d.get("config") == cfg_name and d.get("Q") == Q and d.get("I") == I and d.get("Kp") == Kp:
				traffic, traffic_src = d.get("score_kernel_sweep_hbm_bytes_per_launch"), os.path.basename(tfile)
				break
		except Exception:
			pass

	# the same kernel's duration in the last COMMITTED rocprofv3 summary of this config (profiles/rNN_kernel_stats_<config>.csv, written by
	# scripts/profile_round.sh): one number per kernel and source on the line -- the un-profiled HIP events above and the profiler's average
	# (which runs a few per cent longer: the profiled chip holds a lower clock, MI355X_MICROARCH.md 'DVFS give-back' (2))
	rocprof_ms = rocprof_src = rocprof_stages = None
	for sfile in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_kernel_stats_{cfg_name}.csv")), reverse=True):
		try:
			import csv
			tot_ns = calls = 0
			for r_ in csv.DictReader(open(sfile)):
				if any(nm in r_["Name"] for nm in (f"score16_kernel<{Kp}>", f"score16_kernel<{Kp},", f"score16r_kernel<{Kp}>", f"scoreq16_kernel<{Kp}>", f"scoreq1_kernel<{Kp}>", f"score_kernel<{Kp}, 1, ")):
					tot_ns += float(r_["TotalDurationNs"]); calls += int(r_["Calls"])
			if calls:
				rocprof_ms, rocprof_src = tot_ns / calls / 1e6, os.path.basename(sfile)
				pj = sfile.replace("_kernel_stats_", "_pmc_summary_").replace(".csv", ".json")
				if os.path.exists(pj):
					rocprof_stages = json.load(open(pj)).get("sweep_stages_rocprof")
				break
		except Exception:
			pass

	out = None
	if rank == 0:
		recall = {f"recall@{t}": res[t]["exact_vs_reranked_approx_retvr~common_frac_mean"] for t in top_k_vals}
		out = {
			"metric": "queries/sec + Top-k-Recall@100 vs exact, CUR-256 on QxI score matrix",
			"value": value, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
			"dtype": "bf16", "data": "synthetic",
			"config": {"workload": f"{cfg_name}: Q={Q}/GPU x I={I} bf16 score matrix, {cfg['Ki']} anchor items, {cfg['Kq']} anchor queries, "
								   f"k={k}, k_retvr={kr}; step = gather C_q + fused S_hat/top-k + exact top-k scan + overlap/recall",
					   "Q_per_gpu": Q, "I": I, "anchors": cfg["Ki"], "anchor_queries": cfg["Kq"], "k": k, "k_retvr": kr,
					   "parallelism": f"row-sharded x{world}, index replicated (one RCCL all-gather of anchor rows at build time)"},
			"recall": recall,
			"roofline": {"bound": "mfma", "kernel": "sweep stages (fused S_hat GEMM + threshold filter): " + " + ".join(
							 ({2: f"score16_kernel<{Kp}>", 3: f"scoreq1_kernel<{Kp}>", 4: f"scoreq16_kernel<{Kp}>", 5: f"score16r_kernel<{Kp}>"}.get(b, f"score_kernel<{Kp},sweep>")) for b in plan_now["stage_pred"]),
						 "achieved": sweep_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": sweep_tflops / PEAK_BF16_TFLOPS,
						 "traffic": traffic, "flops_per_launch": sweep_flops, "avg_launch_ms": float(sweep_ms), "launches_per_step": n_sweep,
						 "frac_rocprof": (sweep_flops / (rocprof_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS if rocprof_ms else None), "avg_launch_ms_rocprof": rocprof_ms,
						 "rocprof_source": rocprof_src, "stages_rocprof": rocprof_stages, "ceiling": ceiling,
						 "survivors_per_query": survivors_k,
						 "what": "achieved / frac / avg_launch_ms: HIP events around every sweep launch of this (un-profiled) run; *_rocprof: the same kernel in the last committed rocprofv3 summary under profiles/"},
			"roofline_scan": {"bound": "hbm", "kernel": "rowwise_topk_wave_kernel<bf16> (exact top-k scan, one wave per row)",
							  "achieved": scan_bytes / (scan_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
							  "frac": scan_bytes / (scan_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_launch": scan_bytes, "avg_launch_ms": scan_ms},
			"stage_ms": {"gather_cols": gath_ms, "prepass": float(stage[0]), "threshold": float(stage[1]), "sweep": float(stage[2]), "sweep_kernels_only": float(stage[4]),
						 "select": float(stage[3]), "exact_scan": scan_ms,
						 # the retrieval chain's launches one after the other vs the step as timed: what the scan's placement and the pipelining of
						 # consecutive steps actually hide (chain + scan - step; the overlap count and the D2H ride in the step only)
						 "chain_sum": float(gath_ms + stage[0] + stage[1] + stage[2] + stage[3]), "chain_plus_scan": float(gath_ms + stage[0] + stage[1] + stage[2] + stage[3] + scan_ms),
						 "hidden_by_overlap": float(gath_ms + stage[0] + stage[1] + stage[2] + stage[3] + scan_ms - ms_per_step)},
			"entry_A_cell": entry_a,
			# the sweep launch by launch: tiles swept, duration and MFMA rate of each stage (the first one runs against the prepass threshold)
			"sweep_stages": [{"tiles": int(t1 - t0), "ms": float(stage[6 + g]), "tflops": 2.0 * Q * Kp * 32.0 * (t1 - t0) / (max(float(stage[6 + g]), 1e-9) * 1e-3) / 1e12}
							 for g, (t0, t1) in enumerate(zip([0] + plan_now["stage_end"][:-1], plan_now["stage_end"]))],
			"retrieve_only": {"value": world * Q / (retrieve_ms * 1e-3), "unit": "queries/s", "ms_per_step": retrieve_ms,
							  "survivors_per_query": survivors_k,
							  "what": "gather C_q + fused S_hat/top-k_retvr only (no exact scan, no overlap), eager launches, this rank x world; survivors = candidates the sweep kept per query (k ln(I/k) is what a sequential threshold can reach)"},
			"retrieve_only_k500": ({"value": world * Q / (retrieve500_ms * 1e-3), "unit": "queries/s", "ms_per_step": retrieve500_ms,
									"survivors_per_query": survivors_k500,
									"what": "as retrieve_only with k_retvr = 500 (the reference's default for entry A)"} if retrieve500_ms else None),
			"index_build_s": index_build_s, "index_build_numpy_pinv_s": index_build_numpy_s,
			"index_build_what": "gather anchor columns + U = pinv(W) + E = U.R + bf16 packs; pinv 'auto' = fp64 Newton-Schulz on the GPU (host LAPACK only for ill-conditioned blocks); numpy = the reference's host call",
			"value_with_index_build": world * Q / (ms_per_step * 1e-3 + index_build_s),
			"fused_plan": plan_now,
			"launch_mode": "eager" if not graphed else "hipGraph replay (the step's launches captured once per result slot)",
			"scan_mode": {"used": scan_mode_used, "requested": "partition" if os.environ.get("ANNCUR_BENCH_FALLBACK_REASON") else (args.scan_mode or "partition"),
						  "fallback_reason": os.environ.get("ANNCUR_BENCH_FALLBACK_REASON"),   # set by the supervisor when a partition attempt died of a GPU fault and fresh workers re-ran with "side"
						  "scan_cus": scan_cus if scan_mode_used == "partition" else None,
						  "retrieval_streams": args.retr_streams if scan_mode_used == "partition" else 1, "gather_folded_into_scan": bool(scan_mode_used == "partition" and args.fold_gather and ops.rowwise_topk_gather_ok(A_test, k) and Kp == len(anc))},
			"sustained": sustained, "ranks_seen": ranks_seen, "allgather_ms": allgather_ms, "backend": (args.backend if use_dist else None),
			"solo_rank0": ({"value": solo, "unit": "queries/s", "what": "the same K steps on rank 0 alone, other ranks idle: N x this is the ideal weak-scaling value"} if solo else None),
		}

	# ------------------------------------------------------------------ side-line: the IVF-flat branch at the hard-negative-mining size
	if rank == 0 and world == 1 and not args.no_ivf and not light:
		out["ivf_search"] = ivf_sideline(device, args.seed)

	# ------------------------------------------------------------------ CPU baseline: the oracle (reference-faithful loop) on a bounded sample
	if rank == 0 and world == 1 and args.cpu_sample_queries > 0 and not light:
		from oracle import cur_oracle as O
		# bounded sample (~10-30 s of CPU work): the oracle's per-query cost grows with I, so the sample shrinks with it (cfg2: 4096
		# queries, the per-GPU shape of cfg4 with I = 10^6: 409); a fixed 4096 at I = 10^6 ran for minutes without a line of output
		n = min(max(64, args.cpu_sample_queries * 100000 // max(I, 100000)), Q)
		cores = max(1, min(args.cpu_threads, os.cpu_count() or 1))
		torch.set_num_threads(cores)
		At = A_train.float().cpu()
		Aq = A_test[:n].float().cpu()
		ref = O.CURApproxOracle(rows=At, cols=At[:, anc], row_idxs=np.arange(cfg["Kq"]), col_idxs=anc, approx_preference="rows")
		t0 = time.perf_counter()
		S_hat = ref.get_complete_row(Aq[:, anc])
		want = O.eval_approx_score_mat_for_all_topk(Aq, S_hat, top_k_vals, kr)
		cpu_s = time.perf_counter() - t0
		want_stable = O.eval_all_topk_stable(Aq, S_hat, top_k_vals, kr)
		# "vectorised CPU": the same work without the per-query Python loop (batched topk + sorted-membership overlap)
		t0 = time.perf_counter()
		S2 = Aq[:, anc] @ ref.latent_cols
		ai = torch.topk(S2, kr, dim=1).indices
		ei = torch.topk(Aq, k, dim=1).indices
		srt = torch.sort(ai, dim=1).values
		vec_recall = {}
		for t in top_k_vals:
			pos = torch.searchsorted(srt, ei[:, :t].contiguous()).clamp_(max=kr - 1)
			vec_recall[f"recall@{t}"] = round(float((torch.gather(srt, 1, pos) == ei[:, :t]).sum(dim=1).double().mean() / t), 4)
		cpu_vec_s = time.perf_counter() - t0
		# ... and the same vectorised work with torch's thread pool at EVERY host core the process sees (BASELINE.md 3 says all cores).  On
		# the GPU boxes of this pool os.cpu_count() reports the host's 256 cores while the container's CPU quota is ~16, so this line mostly
		# measures oversubscription; it is printed for completeness, on a quarter of the sample.
		all_cores = os.cpu_count() or cores
		vec_all = None
		if all_cores > cores:
			n4 = max(64, n // 4)
			torch.set_num_threads(all_cores)
			t0 = time.perf_counter()
			S4 = Aq[:n4, anc] @ ref.latent_cols
			ai4 = torch.topk(S4, kr, dim=1).indices
			ei4 = torch.topk(Aq[:n4], k, dim=1).indices
			srt4 = torch.sort(ai4, dim=1).values
			for t in top_k_vals:
				pos = torch.searchsorted(srt4, ei4[:, :t].contiguous()).clamp_(max=kr - 1)
				(torch.gather(srt4, 1, pos) == ei4[:, :t]).sum(dim=1).double().mean()
			vec_all = {"value": n4 / (time.perf_counter() - t0), "unit": "queries/s", "threads": all_cores, "sample_queries": n4}
			torch.set_num_threads(cores)
		# the same n queries through the GPU path, for the recall comparison on identical inputs
		approx = cur.topk_in_row_device(Xq[:n].contiguous(), kr)
		exact = ops.rowwise_topk(A_test[:n], k)
		c = ops.overlap_counts(exact.indices, approx.indices, cells).cpu().numpy()
		got = {t: flatten_overlap(overlap_stats_from_counts(c[j], t)) for j, (t, _) in enumerate(cells)}
		key = "exact_vs_reranked_approx_retvr~common_frac_mean"
		out["cpu_baseline"] = {"value": n / cpu_s, "unit": "queries/s", "cores": cores, "kind": "port",
							   "sample": f"first {n} of the {Q} queries of the same workload: fp32 S_hat GEMM + the reference's per-query loop "
										 f"(3x topk + scatter + overlap) via oracle/cur_oracle.py, torch {torch.__version__} CPU, {cpu_s:.1f} s",
							   "host_cores_total": os.cpu_count(),
							   "vectorised": {"value": n / cpu_vec_s, "unit": "queries/s", "threads": cores, "what": "same sample, batched torch.topk + sorted-membership overlap instead of the reference's per-query loop", "recall": vec_recall,
											  "all_host_cores": vec_all},
							   "recall_cpu_fp32": {f"recall@{t}": want[t][key] for t in top_k_vals},
							   "recall_cpu_fp32_tie_stable": {f"recall@{t}": want_stable[t][key] for t in top_k_vals},
							   "recall_gpu_same_queries": {f"recall@{t}": got[t][key] for t in top_k_vals}}
		out["speedup_vs_cpu"] = value / out["cpu_baseline"]["value"]
	return out


def _fake_worker(args):
	"""Test scaffolding for the supervisor (tests/test_cpu_host.py, no GPU): ANNCUR_BENCH_FAKE_WORKER=fault_in_partition makes the worker of
	rank ANNCUR_BENCH_FAKE_RANK die like a GPU fault does (message on stderr, SIGABRT) while the scan placement is the partition, the other
	ranks wait as they would in a collective; any other attempt prints a result line that names its placement and the fallback reason."""
	rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
	mode = args.scan_mode or "partition"
	if os.environ["ANNCUR_BENCH_FAKE_WORKER"] == "fault_in_partition" and mode == "partition":
		if rank == int(os.environ.get("ANNCUR_BENCH_FAKE_RANK", "0")):
			print("Memory access fault by GPU node-1 (fake worker)", file=sys.stderr, flush=True)
			os.abort()
		time.sleep(600)   # (a rank waiting in a collective for the one that died: its supervisor ends it)
	if os.environ["ANNCUR_BENCH_FAKE_WORKER"] == "plain_error":
		raise SystemExit(3)
	if rank == 0:
		print(json.dumps({"metric": "fake", "value": 1.0, "n_gpus": world, "scan_mode": {"used": mode, "requested": "partition" if os.environ.get("ANNCUR_BENCH_FALLBACK_REASON") else mode,
																				   "fallback_reason": os.environ.get("ANNCUR_BENCH_FALLBACK_REASON")}}), flush=True)
	return 0


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=None, help="default: WORLD_SIZE under torchrun, else 1")
	ap.add_argument("--steps", type=int, default=30)
	ap.add_argument("--warmup", type=int, default=5)
	ap.add_argument("--config", default=None, choices=sorted(CONFIGS), help="default: cfg2 (the headline workload) at every N, plus -- at N > 1 -- cfg4_per_gpu as the \"cfg4\" sub-object; given: that workload alone")
	ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the multi-rank path with ranks sharing GPUs")
	ap.add_argument("--share-gpu", action="store_true", help="rehearsal: rank r uses GPU r %% device_count (RCCL cannot; use --backend gloo)")
	ap.add_argument("--no-k500", action="store_true", help="skip the retrieve_only_k500 side-line (profiling runs: its launches share the sweep kernel's name)")
	ap.add_argument("--sustained-seconds", type=float, default=10.0, help="also loop the same step for this long and report it (DVFS-settled rate); 0 = skip")
	ap.add_argument("--cpu-sample-queries", type=int, default=4096, help="queries timed through the CPU oracle (0 = skip)")
	ap.add_argument("--cpu-threads", type=int, default=8, help="torch CPU threads for the baseline (the per-query loop gets SLOWER with more)")
	ap.add_argument("--no-ivf", action="store_true", help="skip the ivf_search side-line (the IVF-flat branch of build_flat_or_ivff_index at the hard-negative-mining size)")
	ap.add_argument("--seed", type=int, default=0)
	ap.add_argument("--no-overlap", action="store_true", help="exact scan and retrieval one after the other on one stream (= --scan-mode serial)")
	ap.add_argument("--scan-cus", type=int, default=None, help="CUs the exact scan streams on in --scan-mode partition (a multiple of 32: four per XCD); default 128 for Kp <= 256, 64 above")
	ap.add_argument("--scan-mode", default=None, choices=["side", "partition", "tail", "chunks", "serial"],
					help="how the exact scan is scheduled against the retrieval: partition = on a stream whose CU mask leaves it --scan-cus CUs, beside the "
						 "retrieval on all of them (the default at every Kp: 128 scan CUs for Kp <= 256, 64 above); side = on a second stream "
						 "from the start of the step, joined before the overlap count (the fallback when CU-masked streams are unavailable or a partition run died); chunks = anncur_eval_topk (row chunks forked "
						 "beside the retrieval's latency-bound launches); serial = one stream")
	ap.add_argument("--retr-streams", type=int, default=2, choices=[1, 2], help="--scan-mode partition: retrieval chains of consecutive steps on one stream or on two (a workspace each)")
	ap.add_argument("--fold-gather", action="store_true", help="--scan-mode partition: C_q out of the scan's own pass over A (anncur_rowwise_topk_gather) instead of "
					"the gather kernel -- built for SURVEY a2's 'fold into the first pass', measured slower (see the comment at its use): off by default")
	ap.add_argument("--no-graph", action="store_true", help="launch the step's kernels eagerly instead of replaying a captured HIP graph")
	ap.add_argument("--sweep-staged", action="store_true", help="A/B: the retrieval's sweep in stages with a refinement launch between them (ANNCUR_TOPK_STAGED, rounds 1-4) instead of one launch with the in-flight threshold ladder")
	ap.add_argument("--no-ceiling", action="store_true", help="skip roofline.ceiling (the sweep on survivor-free operands); profiling runs: its launches share the sweep kernel's name")
	ap.add_argument("--direct", action="store_true", help="measure in THIS process (no supervisor / worker split, no fallback attempt): what profiling scripts put behind `rocprofv3 --`")
	args = ap.parse_args()
	world = int(os.environ.get("WORLD_SIZE", "1"))
	if args.gpus is None:
		args.gpus = world if os.environ.get("RANK") is not None else 1
	if args.gpus > 1 and os.environ.get("RANK") is None:
		self_launch(args)
	if os.environ.get("ANNCUR_BENCH_FAKE_WORKER"):
		return _fake_worker(args)
	if args.gpus != world:   # before any process group exists: nothing to tear down, no rank left waiting in a collective
		raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
	# stdout carries exactly ONE line (the result JSON): libraries that print banners to fd 1 (RCCL at communicator creation
	# does) are sent to stderr for the whole run, the JSON goes to the saved descriptor at the end
	sys.stdout.flush()
	result_fd = os.dup(1)
	os.dup2(2, 1)

	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	if not torch.cuda.is_available():
		raise SystemExit("bench.py needs an MI355X: anncur_amd has no CPU path")
	torch.set_num_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
	if args.share_gpu:
		local_rank %= torch.cuda.device_count()
	torch.cuda.set_device(local_rank)
	device = torch.device("cuda", local_rank)
	use_dist = world > 1 or (os.environ.get("RANK") is not None and os.environ.get("ANNCUR_BENCH_FORCE_DIST"))
	if use_dist:
		import torch.distributed as dist
		if args.backend == "nccl":
			dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
		else:
			dist.init_process_group("gloo")
	ranks_seen = torch.distributed.get_world_size() if use_dist else 1

	if os.environ.get("ANNCUR_BENCH_FAIL_RANK") == str(rank):   # fault injection for tests/test_gpu_bench_multirank.py
		raise RuntimeError(f"[bench] injected failure on rank {rank} (ANNCUR_BENCH_FAIL_RANK)")

	ctx = dict(rank=rank, world=world, device=device, use_dist=use_dist, ranks_seen=ranks_seen)
	# ONE workload per JSON key at every N: the top-level value / config is the headline workload (cfg2: 10 000 queries x 100 000 items per
	# rank, anchor rows assembled by the one all-gather) whatever N is -- value(N) / value(1) is then a weak-scaling curve of one workload,
	# and the N = 1 point is the single-GPU bench line.  BASELINE cfg4's per-GPU shape (6 250 x 10^6, 512 anchors) rides along at N > 1 as
	# the "cfg4" sub-object with its own value, roofline, allgather_ms and solo_rank0.  (--config X: that workload alone, at any N.)
	# The whole measurement runs with a NON-default stream current.  The scan's CU-masked stream is a blocking stream (the only kind
	# hipExtStreamCreateWithCUMask makes): anything issued on the NULL stream -- the tiny H2D / D2H copies and the barrier tensor of the
	# RCCL max-over-ranks reductions between two batches of steps -- synchronises with it implicitly, and graph replays on the masked stream
	# right after such NULL-stream work ended in a GPU memory access fault (single-rank RCCL run, round 4: 5 of 5 runs; never with the
	# second-stream placement, never with gloo whose reductions stay on the host).  With a pool stream current, the collectives, their
	# copies and the per-kernel timings never touch the NULL stream.  ANNCUR_BENCH_DEFAULT_STREAM=1: the old behaviour (for the repro).
	work_stream = torch.cuda.default_stream(device) if os.environ.get("ANNCUR_BENCH_DEFAULT_STREAM") else torch.cuda.Stream(device=device)
	torch.cuda.synchronize()
	with torch.cuda.stream(work_stream):
		out = _run_all(args, ctx, world, rank)
	torch.cuda.synchronize()
	if rank == 0:
		sys.stdout.flush()
		os.write(result_fd, (json.dumps(out) + "\n").encode())
	_mark("result line written")
	if use_dist:
		torch.distributed.destroy_process_group()
	_mark("process group destroyed")


def _run_all(args, ctx, world, rank):
	out = run_config(args, args.config or "cfg2", ctx)
	if world > 1 and args.config is None:
		import gc
		gc.collect(); torch.cuda.empty_cache()
		sub = run_config(args, "cfg4_per_gpu", ctx, light=True)
		if rank == 0:
			out["cfg4"] = {kk: sub.get(kk) for kk in ("value", "unit", "ms_per_step", "n_gpus", "steps", "warmup", "scaling", "config", "recall", "roofline", "roofline_scan",
														"stage_ms", "sweep_stages", "allgather_ms", "solo_rank0", "scan_mode", "launch_mode", "index_build_s", "sustained")}
			out["cfg4"]["what"] = "BASELINE configs[3] (50k x 1M bf16, 512 anchors, 8 GPUs) at its per-GPU shape on every rank, same job, same ranks; weak scaling like the top level"
	return out


if __name__ == "__main__":
	raise SystemExit(main())
