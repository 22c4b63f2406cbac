"""The multi-rank path of bench.py (self-launch -> torch.distributed.run -> N ranks -> anchor-row all-gather -> timed steps -> rank 0's
JSON line) under test, so that the driver's one multi-GPU run cannot die on plumbing.  Two ranks share the box's single GPU over the
gloo backend (RCCL needs one GPU per rank: RCCL with N > 1 stays unmeasured on the builder's side)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--gpus", "2", "--backend", "gloo", "--share-gpu", "--config", "small", "--steps", "3", "--warmup", "1", "--sustained-seconds", "0",
		"--cpu-sample-queries", "0", "--no-k500"]


def _run(extra_env=None, args=ARGS, timeout=600):
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	env = dict(os.environ)
	for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
		env.pop(v, None)
	env.update(extra_env or {})
	return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def test_bench_two_ranks_self_launch_prints_one_json_line():
	p = _run()
	assert p.returncode == 0, p.stderr[-3000:]
	lines = [l for l in p.stdout.splitlines() if l.strip()]
	assert len(lines) == 1, p.stdout[-2000:]          # stdout carries exactly one line: the result
	out = json.loads(lines[0])
	assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3 and out["warmup"] == 1
	assert out["scaling"] == "weak" and out["unit"] == "queries/s" and out["backend"] == "gloo"
	assert out["allgather_ms"] is not None and out["allgather_ms"] > 0
	assert out["solo_rank0"] is not None and out["solo_rank0"]["value"] > 0
	assert out["value"] == pytest.approx(2 * 2000 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-6)   # whole-job rate: both ranks' queries / max time
	assert 0.0 <= out["recall"]["recall@10"] <= 1.0 and out["roofline"]["achieved"] > 0
	assert "cpu_baseline" not in out                 # the CPU leg runs at N = 1 only


_N1_DEFAULT = {}


def _n1_default_line(mode=None):
	"""The N = 1 line with the driver's default config (cached: two tests read it)."""
	if mode not in _N1_DEFAULT:
		short = ["--steps", "3", "--warmup", "1", "--sustained-seconds", "0", "--cpu-sample-queries", "0", "--no-k500", "--no-ivf"]
		p = _run(args=short + (["--scan-mode", mode] if mode else []))
		assert p.returncode == 0, p.stderr[-3000:]
		assert "Memory access fault" not in p.stderr
		lines = [l for l in p.stdout.splitlines() if l.strip()]
		assert len(lines) == 1, p.stdout[-2000:]
		_N1_DEFAULT[mode] = json.loads(lines[0])
	return _N1_DEFAULT[mode]


def test_bench_default_config_is_one_workload_at_every_n():
	"""What the driver runs: plain `bench.py --gpus N` for N = 1, 2, 4, 8.  The top-level value / config must name the SAME workload at
	every N (the headline cfg2 shape per rank), so that value(N) / value(1) is a scaling curve; BASELINE cfg4's per-GPU shape rides along
	at N > 1 as the "cfg4" sub-object with its own value, roofline, allgather_ms and solo_rank0."""
	one = _n1_default_line()
	p = _run(args=["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1", "--sustained-seconds", "0", "--no-k500", "--no-ivf"], timeout=1200)
	assert p.returncode == 0, p.stderr[-3000:]
	lines = [l for l in p.stdout.splitlines() if l.strip()]
	assert len(lines) == 1, p.stdout[-2000:]
	two = json.loads(lines[0])
	assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and one["n_gpus"] == 1
	assert two["config"]["workload"] == one["config"]["workload"] and two["config"]["workload"].startswith("cfg2:")
	assert two["metric"] == one["metric"] and two["unit"] == one["unit"] and two["config"]["Q_per_gpu"] == one["config"]["Q_per_gpu"] == 10000
	assert two["value"] == pytest.approx(2 * 10000 * 2 / (two["ms_per_step"] * 2e-3), rel=1e-6)
	assert two["allgather_ms"] > 0 and two["solo_rank0"]["value"] > 0 and two["recall"]["recall@1"] >= 0.999
	assert "cfg4" not in one
	c4 = two["cfg4"]
	assert c4["config"]["workload"].startswith("cfg4_per_gpu:") and c4["config"]["Q_per_gpu"] == 6250 and c4["config"]["I"] == 1000000
	assert c4["n_gpus"] == 2 and c4["value"] == pytest.approx(2 * 6250 * 2 / (c4["ms_per_step"] * 2e-3), rel=1e-6)
	assert c4["allgather_ms"] > 0 and c4["solo_rank0"]["value"] > 0 and c4["roofline"]["bound"] == "mfma" and c4["roofline"]["achieved"] > 0
	assert 0.5 < c4["recall"]["recall@100"] <= 1.0
	# the scan's placement: the CU partition at both shapes, 128 scan CUs at cfg2's shape (one sweep launch per retrieval and the cheaper selector of round 5), 64 where it is a quarter
	assert two["scan_mode"]["used"] == "partition" and two["scan_mode"]["scan_cus"] == 128
	assert c4["scan_mode"]["used"] == "partition" and c4["scan_mode"]["scan_cus"] == 64


def test_bench_rccl_code_path_with_one_rank():
	"""RCCL itself (backend "nccl") on the one GPU a builder's box has: a torchrun-style environment with WORLD_SIZE = 1 and
	ANNCUR_BENCH_FORCE_DIST takes bench.py through init_process_group("nccl", device_id=...), the anchor-row all-gather on device tensors, the
	barriers, the MAX all-reduce of the step time and destroy_process_group -- every collective call the N > 1 line makes, with one rank.
	(N > 1 over xGMI stays unmeasured on the builder's side.)"""
	import socket
	with socket.socket() as sk:
		sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
	env = {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "ANNCUR_BENCH_FORCE_DIST": "1",
		   "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
	p = _run(env, args=["--gpus", "1", "--backend", "nccl", "--config", "small", "--steps", "3", "--warmup", "1", "--sustained-seconds", "0", "--cpu-sample-queries", "0",
						"--no-k500", "--no-ivf"])
	assert p.returncode == 0, p.stderr[-3000:]
	lines = [l for l in p.stdout.splitlines() if l.strip()]
	assert len(lines) == 1, p.stdout[-2000:]
	out = json.loads(lines[0])
	assert out["backend"] == "nccl" and out["ranks_seen"] == 1 and out["n_gpus"] == 1 and out["allgather_ms"] > 0 and out["solo_rank0"]["value"] > 0


def test_bench_fails_loudly_when_a_rank_raises():
	p = _run({"ANNCUR_BENCH_FAIL_RANK": "1"})
	assert p.returncode != 0
	assert not [l for l in p.stdout.splitlines() if l.startswith("{")]   # no result line from a broken job


def test_bench_rejects_a_world_size_mismatch_before_any_collective():
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
	p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
	assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_bench_single_gpu_default_line_and_scan_placements():
	"""The N = 1 line of the driver: default flags except a short run.  The exact scan's placement is the CU partition (masked stream +
	three graphs per step) at cfg2's Kp = 256; the same steps with the second-stream placement must report the same recall."""
	outs = {mode: _n1_default_line(mode) for mode in (None, "side")}
	d = outs[None]
	assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "queries/s" and d["vs_baseline"] is None and d["dtype"] == "bf16"
	assert d["scan_mode"]["used"] == "partition" and d["scan_mode"]["scan_cus"] == 128 and outs["side"]["scan_mode"]["used"] == "side"
	assert d["value"] == pytest.approx(10000 * 3 / (d["ms_per_step"] * 3e-3), rel=1e-6)
	assert d["roofline"]["bound"] == "mfma" and 0.2 < d["roofline"]["frac"] < 1.0 and d["roofline_scan"]["bound"] == "hbm"
	assert all(b == 2 for b in d["fused_plan"]["stage_pred"])          # the 16x16x32 body
	assert d["recall"] == outs["side"]["recall"] and d["recall"]["recall@1"] == 1.0


def test_bench_partition_abort_falls_back_to_side_with_fresh_workers():
	"""the N > 1 line must not be lost to a GPU fault of the CU-partition placement.  Rank 1's WORKER aborts (SIGABRT, the way a
	GPU memory fault ends a process) once its partition streams and graphs exist; the supervisors -- the processes the launcher started, which
	never touch the GPU -- end the surviving worker and start ONE fresh attempt with --scan-mode side; rank 0 prints one line that says so."""
	p = _run({"ANNCUR_BENCH_FAIL_PARTITION_RANK": "1"}, timeout=900)
	assert p.returncode == 0, p.stderr[-3000:]
	lines = [l for l in p.stdout.splitlines() if l.strip()]
	assert len(lines) == 1, p.stdout[-2000:]
	out = json.loads(lines[0])
	assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3
	assert out["scan_mode"]["used"] == "side" and out["scan_mode"]["requested"] == "partition"
	assert "partition attempt died" in out["scan_mode"]["fallback_reason"] and "rank 1" in out["scan_mode"]["fallback_reason"]
	assert out["value"] > 0 and 0.0 <= out["recall"]["recall@10"] <= 1.0
	assert "injected abort in partition mode" in p.stderr


def test_bench_partition_abort_single_rank_and_no_fallback_for_an_explicit_mode():
	short = ["--config", "small", "--steps", "3", "--warmup", "1", "--sustained-seconds", "0", "--cpu-sample-queries", "0", "--no-k500", "--no-ivf"]
	p = _run({"ANNCUR_BENCH_FAIL_PARTITION_RANK": "0"}, args=short)
	assert p.returncode == 0, p.stderr[-3000:]
	out = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
	assert out["scan_mode"]["used"] == "side" and "partition attempt died" in out["scan_mode"]["fallback_reason"]
	p = _run({"ANNCUR_BENCH_FAIL_PARTITION_RANK": "0"}, args=short + ["--scan-mode", "partition"])   # an explicit placement is not overridden... but a GPU fault of it still is answered
	assert p.returncode == 0 and json.loads([l for l in p.stdout.splitlines() if l.strip()][0])["scan_mode"]["used"] == "side"
	p = _run({"ANNCUR_BENCH_FAIL_PARTITION_RANK": "0"}, args=short + ["--direct"])                   # --direct: no supervisor, the abort comes through
	assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_cu_partition_streams_refuses_the_null_stream():
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from anncur_amd import _lib, ops
	dev = torch.device("cuda:0")
	with pytest.raises(_lib.AnncurHipError, match="NULL stream"):
		ops.cu_partition_streams(dev, 96)
	with torch.cuda.stream(torch.cuda.Stream(device=dev)):
		retr, scan = ops.cu_partition_streams(dev, 96)
	assert scan.cuda_stream != 0
