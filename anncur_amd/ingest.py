"""Ingestion of cross-encoder score CHUNKS.

The reference computes a mention x entity score matrix in row chunks
(eval/run_cross_encoder_for_ment_ent_matrix_zeshel.py:107-119 writes one pickle per chunk with the schema of :230-240,
file name ...n_m_{chunk}_n_e_{N}_all_layers_False{misc}.pkl with misc = "mstart_{first mention}"), concatenates them on the host
(eval/combine_chunked_computations.py:193-243: torch.cat of all chunks + one more pickle of the whole matrix) and only then
loads that pickle again for the evaluation: pickle -> RAM -> pickle -> RAM -> device.

Here the chunks go straight into device memory, one at a time, and -- under torchrun -- every rank keeps only the rows of
its own contiguous block (anncur_amd.dist.shard_bounds), so no process ever holds the whole matrix on the host.  The checks of
the reference's combiner are kept (same entity id list in every chunk :212-213, row counts match the mention lists :222-223).
`combine_score_chunks` writes the combined pickle with the reference's schema for tools that still want the file.
"""
import os
import pickle
import re

import numpy as np
import torch

from .dist import shard_bounds

_CHUNK_RE = re.compile(r"ment_to_ent_scores_n_m_(\d+)_n_e_(\d+)_all_layers_False(?:mstart_(\d+))?\.pkl$")


def chunk_filename(res_dir, world, n_ment_chunk, n_ent, mstart=None):
	"""The producer's naming (run_cross_encoder_for_ment_ent_matrix_zeshel.py:230): misc = f"mstart_{mstart}" for chunks."""
	misc = "" if mstart is None else f"mstart_{mstart}"
	return f"{res_dir}/{world}/ment_to_ent_scores_n_m_{n_ment_chunk}_n_e_{n_ent}_all_layers_False{misc}.pkl"


def parse_chunk_filename(path):
	"""(rows in the chunk, n_ent of the NAME, first mention or None) from a producer-style file name, None if it is not one."""
	m = _CHUNK_RE.search(os.path.basename(path))
	if not m:
		return None
	return int(m.group(1)), int(m.group(2)), (int(m.group(3)) if m.group(3) is not None else None)


def _load_chunk(path):
	with open(path, "rb") as f:
		d = pickle.load(f)
	for key in ("ment_to_ent_scores", "test_data", "mention_tokens_list", "entity_id_list"):
		if key not in d:
			raise KeyError(f"{path}: not a score chunk (missing {key!r})")
	scores = d["ment_to_ent_scores"]
	if not torch.is_tensor(scores):
		scores = torch.as_tensor(np.asarray(scores))
	if scores.dim() != 2:
		raise ValueError(f"{path}: ment_to_ent_scores must be 2-d, got shape {tuple(scores.shape)}")
	if len(d["test_data"]) != scores.shape[0] or len(d["mention_tokens_list"]) != scores.shape[0]:
		raise ValueError(f"{path}: {scores.shape[0]} score rows but {len(d['test_data'])} mentions / {len(d['mention_tokens_list'])} token lists")
	return d, scores


def _same_entities(ref_ids, ref_tokens, d, path):
	ids = np.asarray(d["entity_id_list"])
	if ids.shape != ref_ids.shape or not (ids == ref_ids).all():
		raise ValueError(f"{path}: entity_id_list differs from the first chunk's (combine_chunked_computations.py:213)")
	tok = d.get("entity_tokens_list")
	if ref_tokens is not None and tok is not None and not np.array_equal(np.asarray(ref_tokens), np.asarray(tok)):
		raise ValueError(f"{path}: entity_tokens_list differs from the first chunk's (combine_chunked_computations.py:212)")


def chunk_row_counts(file_list):
	"""Rows per chunk without unpickling when the names follow the producer's convention, else by loading each file once."""
	counts = []
	for path in file_list:
		parsed = parse_chunk_filename(path)
		if parsed is not None:
			counts.append(parsed[0])
		else:
			counts.append(int(_load_chunk(path)[1].shape[0]))
	return counts


def _default_upload(rows_fp32_cpu, device, dtype):
	from . import ops  # HIP convert; tests on CPU inject their own uploader
	t = rows_fp32_cpu.to(device=device, dtype=torch.float32, non_blocking=False)
	return ops.convert(t, torch.bfloat16) if dtype == "bf16" else t


def load_score_chunks(file_list, device, dtype="fp32", rank=0, world=1, upload=None):
	"""Chunks (in mention order) -> this rank's contiguous row block on `device`.

	Returns a dict: "A_local" [rows_local x n_ent] (fp32 or bf16), "row_range" (start, end) in global mention numbering,
	"n_rows", "n_ent", and the host-side metadata of ALL mentions in order ("test_data", "mention_tokens_list"), plus
	"entity_id_list", "entity_tokens_list", "arg_dict" (of the last chunk, as the reference's combiner keeps it)."""
	if not file_list:
		raise ValueError("load_score_chunks: empty file list")
	upload = upload or _default_upload
	counts = chunk_row_counts(file_list)
	n_rows = int(sum(counts))
	start, end = shard_bounds(n_rows, rank, world)
	pieces, test_data, tokens = [], [], []
	ref_ids = ref_tokens = None
	n_ent = None
	arg_dict = None
	off = 0
	for path, cnt in zip(file_list, counts):
		lo, hi = max(start, off), min(end, off + cnt)
		d, scores = _load_chunk(path)
		if scores.shape[0] != cnt:
			raise ValueError(f"{path}: file name says {cnt} mentions, the tensor has {scores.shape[0]}")
		if ref_ids is None:
			ref_ids, ref_tokens, n_ent = np.asarray(d["entity_id_list"]), d.get("entity_tokens_list"), int(scores.shape[1])
		else:
			_same_entities(ref_ids, ref_tokens, d, path)
			if scores.shape[1] != n_ent:
				raise ValueError(f"{path}: {scores.shape[1]} entities, the first chunk has {n_ent}")
		test_data += list(d["test_data"])
		tokens += list(d["mention_tokens_list"])
		arg_dict = d.get("arg_dict")
		if lo < hi:  # rows of this chunk that belong to this rank
			pieces.append(upload(scores[lo - off: hi - off].float().contiguous(), device, dtype))
		off += cnt
	if pieces:
		A_local = pieces[0] if len(pieces) == 1 else torch.cat(pieces, dim=0)
	else:
		A_local = torch.empty((0, n_ent), device=device, dtype=torch.bfloat16 if dtype == "bf16" else torch.float32)
	return {"A_local": A_local, "row_range": (start, end), "n_rows": n_rows, "n_ent": n_ent, "test_data": test_data,
			"mention_tokens_list": tokens, "entity_id_list": ref_ids, "entity_tokens_list": ref_tokens, "arg_dict": arg_dict}


def combine_score_chunks(file_list, out_file, overwrite=False):
	"""The reference's combiner (combine_chunked_computations.py:193-243) without the prompt: one pickle, reference schema."""
	if os.path.exists(out_file) and not overwrite:
		raise FileExistsError(f"{out_file} exists (pass overwrite=True)")
	mats, test_data, tokens = [], [], []
	ref_ids = ref_tokens = None
	arg_dict = None
	for path in file_list:
		d, scores = _load_chunk(path)
		if ref_ids is None:
			ref_ids, ref_tokens = np.asarray(d["entity_id_list"]), d.get("entity_tokens_list")
		else:
			_same_entities(ref_ids, ref_tokens, d, path)
		mats.append(scores)
		test_data += list(d["test_data"])
		tokens += list(d["mention_tokens_list"])
		arg_dict = d.get("arg_dict")
	comb = torch.cat(mats)
	assert comb.shape[0] == len(test_data) == len(tokens)
	os.makedirs(os.path.dirname(os.path.abspath(out_file)), exist_ok=True)
	with open(out_file, "wb") as f:
		pickle.dump({"ment_to_ent_scores": comb, "ment_to_ent_scores.shape": comb.shape, "test_data": test_data,
					 "mention_tokens_list": tokens, "entity_id_list": ref_ids, "entity_tokens_list": ref_tokens, "arg_dict": arg_dict}, f)
	return out_file
