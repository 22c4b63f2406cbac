"""CPU-only checks: the C-ABI library exports exactly what include/anncur_hip.h declares, host logic against the oracle,
row-sharding over gloo (world_size 2), error behaviour without a GPU."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
	src = open(os.path.join(ROOT, "include", "anncur_hip.h")).read()
	src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
	return sorted(set(re.findall(r"\b(anncur_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
	from anncur_amd import _lib
	lib = _lib.load()                       # loads without a GPU
	declared = _header_functions()
	assert len(declared) >= 15
	for name in declared:
		assert hasattr(lib, name), f"{name} declared in include/anncur_hip.h but not exported"
	assert sorted(_lib.SIGNATURES) == declared, "ctypes SIGNATURES and the header disagree"
	out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
	exported = sorted(set(re.findall(r" T (anncur_[a-z0-9_]+)", out)))
	assert set(declared) <= set(exported)
	assert lib.anncur_version() >= 1 and lib.anncur_last_error() is not None


def test_no_compute_without_gpu_and_plan_queries_work():
	from anncur_amd import _lib, ops
	lib = _lib.load()
	assert lib.anncur_score_topk_supported(10000, 100000, 256, 100) == 1
	assert lib.anncur_score_topk_supported(1000, 5000, 64, 10) == 0          # too small for a sampled threshold
	assert lib.anncur_score_topk_supported(10000, 100000, 300, 100) == 0     # Kp must be 64/128/256/512
	assert lib.anncur_score_topk_workspace_bytes(10000, 100000, 256, 100) > 0
	with pytest.raises(_lib.AnncurHipError):
		ops.rowwise_topk(torch.zeros(4, 10), 2)                               # CPU tensor: there is no CPU fallback
	with pytest.raises(_lib.AnncurHipError):
		ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
	assert ops.padded_k(200) == 256 and ops.padded_k(513) == 640 and ops.padded_k(768) == 768 and ops.padded_k(4097) is None
	assert lib.anncur_score_topk_supported(10000, 100000, 1024, 100) == 1    # wide inner dimension: the LDS-tiled K-general kernel
	assert lib.anncur_score_topk_supported(10000, 100000, 1000, 100) == 0    # ... takes multiples of 128 (the caller zero-pads)
	assert lib.anncur_score_topk_workspace_bytes(1200, 15603, 1024, 100) > 0


def test_sweep_variant_flags_reach_the_plan():
	"""ops.score_topk_fused accepted mfma16= / qt1= and dropped them.  The plan query takes the same flags word the launch
	does (ops._topk_flags), so a dropped flag shows here without a GPU."""
	from anncur_amd import _lib, ops
	assert ops._topk_flags() == 0 and ops._topk_flags(True, True, True, True) == (_lib.TOPK_LEADING_SAMPLE | _lib.TOPK_MFMA16 | _lib.TOPK_QT1 | _lib.TOPK_MFMA32)
	base = ops.fused_plan(10000, 100000, 256, 100)
	# round 5: the default 16x16x32 body sweeps in ONE launch and raises its thresholds up a ladder of levels in flight (csrc/score16.hpp)
	assert (base["lg"], base["QT"]) == (1, 2) and base["n_stages"] == len(base["stage_end"]) == 1 and base["stage_end"][-1] == base["n_tiles"]
	assert base["ladder"] and base["ladder_top_rank"] == 10 and ops.fused_plan(10000, 100000, 256, 100, leading_sample=True)["ladder_top_rank"] == 16
	assert all(b == 2 for b in base["stage_pred"])                              # the 16x16x32 body (k <= 1024 with the ladder)
	staged = ops.fused_plan(10000, 100000, 256, 100, staged=True)               # ANNCUR_TOPK_STAGED: rounds 1-4's staged sweep of the same body (A/B, parity reference)
	assert not staged["ladder"] and staged["n_stages"] == len(staged["stage_end"]) == 2 and all(b == 2 for b in staged["stage_pred"]) and staged["lg"] == 1
	assert ops._topk_flags(staged=True) == _lib.TOPK_STAGED and not ops.fused_plan(10000, 100000, 256, 100, mfma32=True)["ladder"]
	assert not ops.fused_plan(6250, 1000000, 512, 100)["ladder"]
	m32 = ops.fused_plan(10000, 100000, 256, 100, mfma32=True)
	assert m32["lg"] == 2 and all(b in (0, 1) for b in m32["stage_pred"])
	p500 = ops.fused_plan(10000, 100000, 256, 500)                                # round 5: with the ladder the 16x16x32 body is the default up to k = 1024
	assert p500["lg"] == 1 and p500["ladder"] and p500["n_stages"] == 1 and ops.fused_plan(10000, 100000, 256, 1000)["ladder"]
	assert ops.fused_plan(10000, 100000, 256, 500, staged=True)["lg"] == 2      # without it (ANNCUR_TOPK_STAGED) round 4's limit stands: 32x32x16 above k = 384
	assert ops.fused_plan(10000, 100000, 256, 300, staged=True)["lg"] == 1 and ops.fused_plan(10000, 100000, 256, 1100)["lg"] == 2   # (k > 1024: > 4096 group maxima, no ladder)
	assert all(b == 2 for b in ops.fused_plan(10000, 100000, 256, 100, mfma16=True)["stage_pred"])
	assert ops.fused_plan(10000, 100000, 256, 100, mfma16=True)["lg"] == 1
	assert ops.fused_plan(300, 40000, 64, 10, mfma16=True)["lg"] == 1
	assert ops.fused_plan(10000, 100000, 256, 100, qt1=True)["QT"] == 1 and ops.fused_plan(10000, 100000, 128, 100, qt1=True)["QT"] == 1
	assert ops.fused_plan(10000, 100000, 64, 100, qt1=True)["QT"] == 2       # Kp = 64 has no QT = 1 body
	p512 = ops.fused_plan(6250, 1000000, 512, 100)                                    # Kp = 512: default = the wave-level queue + tickets
	assert p512["lg"] == 1 and p512["QT"] == 1 and all(b == 4 for b in p512["stage_pred"])            # ... on 16x16x32 MFMAs
	assert ops.fused_plan(6250, 1000000, 512, 100, mfma32=True)["lg"] == 2           # the per-lane-ring body on request
	assert ops.fused_plan(64, (1 << 26) + 64, 64, 10)["lg"] == 2                      # queue entries carry the query beside the item: I < 2^26
	# ANNCUR_TOPK_RING (round 4: the 16x16x32 body in 8-wave workgroups with a flag-synchronised tile ring) measured slower than the default and lives
	# in the experiments library only since round 5: the product refuses the flag instead of silently running something else
	if _lib.IS_EXPERIMENTS_LIB:
		ring = ops.fused_plan(10000, 100000, 256, 100, ring=True)
		assert all(b == 5 for b in ring["stage_pred"]) and ring["lg"] == 1 and ring["splits"] == 13 and not ring["ladder"]
	else:
		with pytest.raises(_lib.AnncurHipError, match="experiments library"):
			ops.fused_plan(10000, 100000, 256, 100, ring=True)
	assert ops._topk_flags(ring=True) == _lib.TOPK_RING
	import inspect
	src = inspect.getsource(ops.score_topk_fused.__wrapped__) + inspect.getsource(ops.score_topk_fused_timed.__wrapped__)
	assert src.count("_topk_flags(leading_sample, mfma16, qt1, mfma32, ring, staged)") == 2 and "mfma16=mfma16, qt1=qt1, mfma32=mfma32, ring=ring, staged=staged" in src


def test_product_never_imports_the_oracle():
	pkg = os.path.join(ROOT, "anncur_amd")
	offenders = []
	for dirpath, _, files in os.walk(pkg):
		for f in files:
			if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dirpath, f)).read(), flags=re.M):
				offenders.append(f)
	for f in ("eval/run_retrieval_eval_wrt_exact_crossenc.py", "eval/run_retrieval_eval_wrt_exact_crossenc_w_fixed_train_test_splits.py",
			  "eval/matrix_approx_zeshel.py", "models/nearest_nbr.py"):
		if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(ROOT, f)).read(), flags=re.M):
			offenders.append(f)
	assert not offenders, offenders


def test_overlap_statistics_match_reference_format():
	from anncur_amd.eval_utils import compute_overlap, flatten_overlap, overlap_stats_from_counts
	from oracle import cur_oracle as O
	rng = np.random.default_rng(0)
	for n in (1, 7, 10, 100):
		for Q in (1, 2, 3, 10, 1001):
			c = rng.integers(0, n + 1, size=Q)
			a = [list(range(n)) for _ in range(Q)]
			b = [list(range(n - ci, 2 * n - ci)) for ci in c]
			want = {k: tuple(v) for k, v in O.compute_overlap(a, b).items()}
			assert overlap_stats_from_counts(c, n) == want
			assert compute_overlap(a, b) == want
			assert flatten_overlap(want) == O.overlap_to_flat(want)
	assert compute_overlap([], []) == {k: tuple(v) for k, v in O.compute_overlap([], []).items()}
	with pytest.raises(AssertionError):
		compute_overlap([[1, 2]], [[1]])


def test_zeshel_constants_and_filenames():
	from utils import zeshel_utils as z
	assert z.N_ENTS_ZESHEL["yugioh"] == 10031 and z.N_MENTS_ZESHEL["yugioh"] == 3374 and z.N_ENTS_ZESHEL["military"] == 104520
	assert sum(z.N_MENTS_ZESHEL[w] for s, w in z.get_zeshel_world_info() if s == "test") == 10000
	assert len(z.get_zeshel_world_info()) == 16
	d = z.get_dataset_info("D", "R", z.get_zeshel_world_info(), n_ment=100)
	assert d["lego"]["crossenc_ment_to_ent_scores"] == "R/lego/ment_to_ent_scores_n_m_100_n_e_10076_all_layers_False.pkl"
	assert d["lego"]["ent_tokens_file"] == "D/tokenized_entities/lego_128_bert_base_uncased.npy"


def test_cli_flags_match_the_reference():
	for script, flags in (("eval/run_retrieval_eval_wrt_exact_crossenc.py",
						   ["--data_name", "--bi_model_file", "--res_dir", "--n_seeds", "--plot_only", "--n_ment", "--batch_size", "--misc", "--disable_wandb"]),
						  ("eval/run_retrieval_eval_wrt_exact_crossenc_w_fixed_train_test_splits.py",
						   ["--data_name", "--eval_method", "--res_dir", "--test_data_file", "--train_data_file", "--n_seeds", "--bi_model_file", "--batch_size",
							"--e2e_fname", "--n_fixed_anc_ent", "--mention_file", "--entity_file", "--mode", "--misc", "--use_wandb"])):
		out = subprocess.run([sys.executable, os.path.join(ROOT, script), "--help"], capture_output=True, text=True, check=True).stdout
		for f in flags:
			assert f in out, (script, f)


# ------------------------------------------------------------------ row sharding over gloo, world_size 2
def _shard_worker(rank, world, port, n_rows, row_idxs, q):
	import torch.distributed as dist
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
	dist.init_process_group("gloo", rank=rank, world_size=world)
	from anncur_amd.dist import ShardedScoreMatrix, gather_rows_to_rank0, shard_bounds
	g = torch.Generator().manual_seed(0)
	A = torch.randn(n_rows, 40, generator=g)
	s, e = shard_bounds(n_rows, rank, world)
	sm = ShardedScoreMatrix(A[s:e].clone(), n_rows, pack=lambda local, idx: local[torch.as_tensor(np.asarray(idx), dtype=torch.long)])
	R = sm.anchor_rows(row_idxs)                                   # the single collective of the path
	ok = torch.equal(R, A[row_idxs])
	local_result = A[s:e, :3] * 2                                   # stands for per-query results of this rank's rows
	full = gather_rows_to_rank0(local_result.contiguous(), n_rows)
	if rank == 0:
		ok = ok and torch.equal(full, A[:, :3] * 2)
	# bf16 rows (the storage type of cfg2 / cfg4): gloo has no bf16, the rows travel as bytes
	Ab = A.bfloat16()
	smb = ShardedScoreMatrix(Ab[s:e].clone(), n_rows, pack=lambda local, idx: local[torch.as_tensor(np.asarray(idx), dtype=torch.long)])
	ok = ok and torch.equal(smb.anchor_rows(row_idxs), Ab[row_idxs])
	fb = gather_rows_to_rank0(Ab[s:e, :5].contiguous(), n_rows)
	if rank == 0:
		ok = ok and torch.equal(fb, Ab[:, :5])
	q.put((rank, bool(ok)))
	dist.destroy_process_group()


@pytest.mark.parametrize("n_rows,row_idxs", [(11, [0, 3, 4, 9, 10]), (8, [5, 6, 7]), (9, [])])
def test_row_sharding_allgather_world2(n_rows, row_idxs):
	import torch.multiprocessing as mp
	ctx = mp.get_context("spawn")
	q = ctx.Queue()
	port = 29500 + (os.getpid() % 500) + n_rows
	procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, n_rows, row_idxs, q)) for r in range(2)]
	for p in procs: p.start()
	res = [q.get(timeout=120) for _ in procs]
	for p in procs: p.join(timeout=60)
	assert sorted(res) == [(0, True), (1, True)]


def _anchor_rows_worker(rank, world, port, Kq, n_cols, dtype_name, q):
	import torch.distributed as dist
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
	dist.init_process_group("gloo", rank=rank, world_size=world)
	from anncur_amd.dist import allgather_anchor_rows, shard_bounds
	g = torch.Generator().manual_seed(7)
	full = torch.randn(Kq, n_cols, generator=g).to(getattr(torch, dtype_name))
	# every rank holds a [Kq x I] buffer of which only ITS share [s, e) is meaningful (bench.py: the rank's own draws); poison the rest
	mine = torch.full_like(full, float("nan"))
	s, e = shard_bounds(Kq, rank, world)
	mine[s:e] = full[s:e]
	got = allgather_anchor_rows(mine, Kq, rank, world)
	q.put((rank, tuple(got.shape) == (Kq, n_cols) and got.dtype == full.dtype and torch.equal(got, full)))
	dist.destroy_process_group()


@pytest.mark.parametrize("world,Kq,dtype_name", [(2, 5, "float32"), (2, 7, "bfloat16"), (3, 8, "bfloat16"), (3, 2, "float32")])
def test_allgather_anchor_rows_uneven_shares(world, Kq, dtype_name):
	"""bench.py's one collective (anncur_amd/dist.py::allgather_anchor_rows) when Kq is not a multiple of the world size: padded
	all-gather, ranks in order, shares of different length (one of them empty for Kq < world), bf16 rows travelling as bytes over gloo."""
	import torch.multiprocessing as mp
	ctx = mp.get_context("spawn")
	q = ctx.Queue()
	port = 29100 + (os.getpid() % 300) + 7 * world + Kq
	procs = [ctx.Process(target=_anchor_rows_worker, args=(r, world, port, Kq, 33, dtype_name, q)) for r in range(world)]
	for p in procs: p.start()
	res = [q.get(timeout=120) for _ in procs]
	for p in procs: p.join(timeout=60)
	assert sorted(res) == [(r, True) for r in range(world)]


def test_shard_bounds_and_split():
	from anncur_amd.dist import shard_bounds, split_sorted_indices
	assert [shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
	parts = split_sorted_indices([0, 2, 3, 7, 9], 10, 4)
	assert [p.tolist() for p in parts] == [[0, 2], [0], [1], [1]]


def test_split_producer_matches_reference_kat(tmp_path, golden_dir):
	import json, pickle
	from utils import split_zeshel_ment2ent_for_cur_exps as sp
	n_m, n_e = 130, 40
	dump = {"ment_to_ent_scores": torch.arange(n_m * n_e, dtype=torch.float32).reshape(n_m, n_e), "test_data": [{"id": i} for i in range(n_m)],
			"mention_tokens_list": [[i, i + 1] for i in range(n_m)], "entity_id_list": [], "entity_tokens_list": [], "arg_dict": {"a": 1}}
	with open(tmp_path / "m2e.pkl", "wb") as f:
		pickle.dump(dump, f)
	sp.run("yugioh", str(tmp_path / "m2e.pkl"), [50, 100, 200], 2, 7, 0.1, str(tmp_path / "out"))
	kat = json.load(open(os.path.join(golden_dir, "split_kat.json")))["splits"]
	for key, want in kat.items():
		nm, si, name = key.split("/")
		with open(tmp_path / "out" / f"nm_train={nm}" / f"split_idx={si}" / f"{name}.pkl", "rb") as f:
			d = pickle.load(f)
		assert [int(x) for x in d["ment_idxs"]] == want["ment_idxs"], key
		assert float(d["ment_to_ent_scores"][0, 0]) == want["first_score"] and sorted(d.keys()) == want["keys"]
	assert not (tmp_path / "out" / "nm_train=200").exists()     # more train mentions than there are: skipped


# ------------------------------------------------------------------ chunk ingestion: host logic on the CPU
def _write_chunks(tmp_path, world, n_ent, sizes, seed=0, bad_entities_in=None):
	import pickle
	from anncur_amd import ingest
	g = torch.Generator().manual_seed(seed)
	files, mats, start = [], [], 0
	for ci, n in enumerate(sizes):
		m = torch.randn(n, n_ent, generator=g)
		ids = np.arange(n_ent) if bad_entities_in != ci else np.arange(n_ent)[::-1].copy()
		path = ingest.chunk_filename(str(tmp_path), world, n, n_ent, mstart=start)
		os.makedirs(os.path.dirname(path), exist_ok=True)
		with open(path, "wb") as f:
			pickle.dump({"ment_to_ent_scores": m, "ment_to_ent_scores.shape": m.shape, "test_data": [{"mention_id": start + j} for j in range(n)],
						 "mention_tokens_list": [[start + j] * 4 for j in range(n)], "entity_id_list": ids, "entity_tokens_list": None,
						 "arg_dict": {"chunk": ci}}, f)
		files.append(path); mats.append(m); start += n
	return files, torch.cat(mats)


def test_chunk_ingestion_names_shards_and_schema(tmp_path):
	from anncur_amd import ingest
	sizes = [50, 50, 27, 10]
	files, full = _write_chunks(tmp_path, "lego", 300, sizes)
	# the producer's naming convention round-trips
	assert files[2].endswith("lego/ment_to_ent_scores_n_m_27_n_e_300_all_layers_Falsemstart_100.pkl")
	assert ingest.parse_chunk_filename(files[2]) == (27, 300, 100)
	assert ingest.parse_chunk_filename("ment_to_ent_scores_n_m_2500_n_e_34430_all_layers_False.pkl") == (2500, 34430, None)
	assert ingest.parse_chunk_filename("something_else.pkl") is None
	assert ingest.chunk_row_counts(files) == sizes
	up = lambda rows, device, dtype: rows.clone()     # CPU stand-in for the HIP upload/convert
	whole = ingest.load_score_chunks(files, "cpu", "fp32", upload=up)
	assert whole["n_rows"] == 137 and whole["n_ent"] == 300 and whole["row_range"] == (0, 137)
	assert torch.equal(whole["A_local"], full)
	assert [d["mention_id"] for d in whole["test_data"]] == list(range(137)) and whole["arg_dict"] == {"chunk": 3}
	# row-sharded: the blocks of the ranks tile the matrix in order, whatever the chunk boundaries
	for world in (2, 3, 5):
		blocks = [ingest.load_score_chunks(files, "cpu", "fp32", rank=r, world=world, upload=up) for r in range(world)]
		assert torch.equal(torch.cat([b["A_local"] for b in blocks]), full)
		assert [b["row_range"] for b in blocks] == [__import__("anncur_amd.dist", fromlist=["x"]).shard_bounds(137, r, world) for r in range(world)]
	# the combined file has the reference's schema and content
	out = ingest.combine_score_chunks(files, str(tmp_path / "comb.pkl"))
	import pickle
	d = pickle.load(open(out, "rb"))
	assert set(d) == {"ment_to_ent_scores", "ment_to_ent_scores.shape", "test_data", "mention_tokens_list", "entity_id_list", "entity_tokens_list", "arg_dict"}
	assert torch.equal(d["ment_to_ent_scores"], full) and tuple(d["ment_to_ent_scores.shape"]) == (137, 300) and len(d["mention_tokens_list"]) == 137
	with pytest.raises(FileExistsError):
		ingest.combine_score_chunks(files, out)
	# the reference's consistency checks
	bad, _ = _write_chunks(tmp_path / "bad", "lego", 300, [20, 20], bad_entities_in=1)
	with pytest.raises(ValueError, match="entity_id_list"):
		ingest.load_score_chunks(bad, "cpu", "fp32", upload=up)
	with pytest.raises(ValueError, match="empty"):
		ingest.load_score_chunks([], "cpu")


# ------------------------------------------------------------------ index handling (torch indexing semantics on the host side)
def test_as_index_wraps_negatives_and_rejects_out_of_range():
	from anncur_amd import ops
	cpu = torch.device("cpu")
	assert ops.as_index([0, 2, -1], cpu, n=5).tolist() == [0, 2, 4]          # negative indices wrap like torch indexing
	assert ops.as_index(np.array([3, 1]), cpu, n=4).dtype == torch.int32
	assert ops.as_index([], cpu, n=4).numel() == 0
	for bad in ([0, 5], [-6], [2 ** 31]):
		with pytest.raises(IndexError):
			ops.as_index(bad, cpu, n=5)


def test_full_range_shortcut_is_the_identity_only():
	from anncur_amd.cur import _is_full_range
	assert _is_full_range([0, 1, 2, 3], 4) and _is_full_range(np.arange(7), 7) and _is_full_range(torch.arange(3), 3)
	assert not _is_full_range([0, 2, 1, 3], 4)      # a permutation with the right ends
	assert not _is_full_range([0, 1, 1, 3], 4)      # a repeat with the right ends
	assert not _is_full_range([0, 1, 2], 4) and not _is_full_range([], 0)


def test_built_library_passes_the_static_hazard_checks():
	"""The sweep kernels mix MFMA builtins with inline-asm LDS reads and counted waits; hipcc neither pads nor orders what it cannot
	see, and where it under-pads the result depends on code placement (DESIGN.md 4.1 'A latent hazard').  The shipped library is
	therefore disassembled and walked: (a) no instruction may touch an MFMA's result registers earlier than hipcc's own floor for that
	MFMA unless it is the next MFMA of the accumulate chain, (b) no instruction may touch the destination of an in-flight ds_read."""
	sys.path.insert(0, os.path.join(ROOT, "scripts"))
	import check_lds_hazards, check_mfma_hazards
	lib = os.path.join(ROOT, "anncur_amd", "lib", "libanncur_hip.so")
	f, nk, nm = check_mfma_hazards.check(lib)
	assert nk >= 30 and nm >= 1000, (nk, nm)   # the walk really saw the kernels
	assert f == [], "\n".join(f[:20])
	f, nk, nr = check_lds_hazards.check(lib)
	assert nk >= 20 and nr >= 1000, (nk, nr)
	assert f == [], "\n".join(f[:20])


def _listing(lines, base=0x1000):
	"""[(address, instruction, objdump tail)] from instruction strings; `-> N` after a branch = its target's line number."""
	body = []
	for i, ins in enumerate(lines):
		tail = f" {base + 4 * i:08X}: 00000000"
		if "->" in ins:
			ins, tgt = ins.split("->")
			tail += f" <kernel+0x{4 * int(tgt):x}>"
		body.append((base + 4 * i, ins.strip(), tail))
	return body


def test_static_check_of_returning_atomics_finds_a_planted_early_use():
	"""The ticket draw (inline-asm returning atomic, result consumed a tile later): the walk must flag a use before the vmcnt(0) drain and
	accept the waited form; the shipped sweep kernels contain such draws and pass."""
	sys.path.insert(0, os.path.join(ROOT, "scripts"))
	import check_lds_hazards
	bad = _listing(["global_atomic_add v5, v[2:3], v6, off sc0", "v_add_u32_e32 v7, 1, v8", "ds_write_b32 v9, v5", "s_waitcnt vmcnt(0)", "s_endpgm"])
	f = check_lds_hazards.check_vm_body("kernel", bad)
	assert len(f) == 1 and "ds_write_b32 v9, v5" in f[0]
	good = _listing(["global_atomic_add v5, v[2:3], v6, off sc0", "v_add_u32_e32 v7, 1, v8", "s_waitcnt vmcnt(0)", "ds_write_b32 v9, v5", "s_endpgm"])
	assert check_lds_hazards.check_vm_body("kernel", good) == []
	partial = _listing(["global_atomic_add v5, v[2:3], v6, off sc0", "s_waitcnt vmcnt(1)", "v_mov_b32_e32 v1, v5", "s_waitcnt vmcnt(0)", "s_endpgm"])
	assert len(check_lds_hazards.check_vm_body("kernel", partial)) == 1          # a counted wait is not taken to retire it


def test_static_hazard_checks_find_planted_hazards():
	"""The two ISA walks on synthetic listings: they must report what they exist to report (and nothing on the padded / waited forms)."""
	sys.path.insert(0, os.path.join(ROOT, "scripts"))
	import check_lds_hazards, check_mfma_hazards
	mfma = "v_mfma_f32_32x32x16_bf16 v[16:31], v[0:3], v[4:7], v[16:31]"
	# (a) the round-2 bug: the accumulator copied 8 states after the chain's last MFMA, on the fall-through path of a branch
	bad = _listing([mfma, "s_cbranch_vccnz -> 6", "v_lshl_or_b32 v40, s0, 5, v41", "s_nop 5", "v_mov_b32_e32 v14, v30", "s_endpgm",
					"v_add_u32_e32 v50, 1, v50", "s_branch -> 2"])
	f = check_mfma_hazards.check_body("kernel", bad)
	assert len(f) == 1 and "v_mov_b32_e32 v14, v30" in f[0] and "8 states" in f[0]
	good = _listing([mfma, "s_cbranch_vccnz -> 6", "v_lshl_or_b32 v40, s0, 5, v41", "s_nop 8", "v_mov_b32_e32 v14, v30", "s_endpgm",
					 "v_add_u32_e32 v50, 1, v50", "s_branch -> 2"])
	assert check_mfma_hazards.check_body("kernel", good) == []
	chain = _listing([mfma, mfma, "s_nop 7", "s_nop 2", "v_cmp_ge_f32_e32 vcc, v16, v60", "s_endpgm"])   # the accumulate chain itself needs no states
	assert check_mfma_hazards.check_body("kernel", chain) == []
	# (b) a register of an in-flight ds_read used as a temporary before the counted wait that retires the read
	bad = _listing(["ds_read_b128 v[20:23], v9", "ds_read_b128 v[24:27], v10", "v_and_or_b32 v21, v5, s2, v6", "s_waitcnt lgkmcnt(1)",
					"v_mfma_f32_32x32x16_bf16 v[32:47], v[20:23], v[48:51], v[32:47]", "s_endpgm"])
	f = check_lds_hazards.check_body("kernel", bad)
	assert len(f) == 1 and "v_and_or_b32 v21" in f[0]
	good = _listing(["ds_read_b128 v[20:23], v9", "ds_read_b128 v[24:27], v10", "s_waitcnt lgkmcnt(1)", "v_and_or_b32 v28, v5, s2, v6",
					 "v_mfma_f32_32x32x16_bf16 v[32:47], v[20:23], v[48:51], v[32:47]", "s_waitcnt lgkmcnt(0)",
					 "v_mfma_f32_32x32x16_bf16 v[32:47], v[24:27], v[52:55], v[32:47]", "s_endpgm"])
	assert check_lds_hazards.check_body("kernel", good) == []
	early = _listing(["ds_read_b128 v[20:23], v9", "ds_read_b128 v[24:27], v10", "s_waitcnt lgkmcnt(1)",
					  "v_mfma_f32_32x32x16_bf16 v[32:47], v[24:27], v[52:55], v[32:47]", "s_endpgm"])   # consumes the YOUNGER read: still in flight
	assert len(check_lds_hazards.check_body("kernel", early)) == 1


# ------------------------------------------------------------------ bench.py's supervisor: the N > 1 line must survive a GPU fault of the partition placement
def _bench_with_fake_workers(fake, extra_args, extra_env=None, timeout=240):
	env = dict(os.environ, ANNCUR_BENCH_FAKE_WORKER=fake)
	for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "ANNCUR_BENCH_WORKER", "ANNCUR_BENCH_FALLBACK_REASON"):
		env.pop(v, None)
	env.update(extra_env or {})
	return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def test_bench_supervisor_falls_back_to_side_after_a_gpu_fault_single_rank():
	"""The process the driver starts never touches the GPU: it runs the measurement in a child.  A partition-mode child that dies like a GPU
	fault (message on stderr + SIGABRT; faked here, no GPU) is answered with ONE fresh child in --scan-mode side, and the line says so."""
	p = _bench_with_fake_workers("fault_in_partition", ["--steps", "2"])
	assert p.returncode == 0, p.stderr[-2000:]
	lines = [l for l in p.stdout.splitlines() if l.strip()]
	assert len(lines) == 1, p.stdout
	out = json.loads(lines[0])
	assert out["scan_mode"]["used"] == "side" and out["scan_mode"]["requested"] == "partition"
	assert "partition attempt died" in out["scan_mode"]["fallback_reason"] and "Memory access fault" in out["scan_mode"]["fallback_reason"]


def test_bench_supervisor_relays_a_clean_run_and_a_plain_error_unchanged():
	p = _bench_with_fake_workers("ok", ["--steps", "2"])
	assert p.returncode == 0 and json.loads(p.stdout.strip())["scan_mode"] == {"used": "partition", "requested": "partition", "fallback_reason": None}
	p = _bench_with_fake_workers("plain_error", ["--steps", "2"])       # not a GPU fault: no second attempt, the exit code comes through
	assert p.returncode == 3 and not p.stdout.strip() and "partition attempt died" not in p.stderr
	p = _bench_with_fake_workers("fault_in_partition", ["--steps", "2", "--scan-mode", "side"])   # an explicit placement is never overridden (and the fake only faults in partition mode)
	assert p.returncode == 0 and json.loads(p.stdout.strip())["scan_mode"]["fallback_reason"] is None


@pytest.mark.parametrize("bad_rank", [0, 1])
def test_bench_supervisor_falls_back_on_every_rank_when_one_rank_faults(bad_rank):
	"""--gpus 2 through the self-launch (torch.distributed.run -> two supervisors -> two workers): the worker of ONE rank dies of a (faked) GPU
	fault while the other waits in a collective; its supervisor flags the attempt in the launcher's store, the other supervisor ends its own
	worker, both start fresh workers with --scan-mode side on a fresh rendezvous port, and rank 0 prints ONE line that carries the reason."""
	p = _bench_with_fake_workers("fault_in_partition", ["--gpus", "2", "--steps", "2"], {"ANNCUR_BENCH_FAKE_RANK": str(bad_rank)}, timeout=400)
	assert p.returncode == 0, p.stderr[-3000:]
	lines = [l for l in p.stdout.splitlines() if l.strip()]
	assert len(lines) == 1, p.stdout
	out = json.loads(lines[0])
	assert out["n_gpus"] == 2 and out["scan_mode"]["used"] == "side" and f"rank {bad_rank}" in out["scan_mode"]["fallback_reason"]
