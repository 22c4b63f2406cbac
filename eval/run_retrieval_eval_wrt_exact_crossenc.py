#!/usr/bin/env python
"""Entry point A -- drop-in for the reference's eval/run_retrieval_eval_wrt_exact_crossenc.py (CUR / CUR-oracle branches).

Same flags, same input pickle ({res_dir}/{data_name}/ment_to_ent_scores_n_m_{n_ment}_n_e_{N_ENTS}_all_layers_False.pkl),
same output ({res_dir}/{data_name}/Retrieval_wrt_Exact_CrossEnc/nm=.._ne=.._s=..{_misc}/retrieval_wrt_exact_crossenc.json with
res[method]["top_k=.."]["k_retvr=.."]["anc_n_m=..~anc_n_e=.."][anchor|non_anchor|all][metric]).  The reference hard-codes its
sweep grids; the extra --*_vals flags override them (defaults reproduce the reference).  Compute runs on the MI355X.
"""
import argparse
import json
import logging
import os
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

import numpy as np
import torch

from utils.zeshel_utils import get_dataset_info, get_zeshel_world_info

logging.basicConfig(stream=sys.stderr, format="%(asctime)s - %(levelname)s - %(name)s - %(message)s ",
					datefmt="%d/%m/%Y %H:%M:%S", level=logging.INFO)
LOGGER = logging.getLogger(__name__)


def _ints(s):
	return [int(x) for x in s.split(",") if x != ""]


def plot(res_dir, method_vals):
	"""Heat maps of the non-anchor-query metrics over the (anchor mentions x anchor entities) grid, one PDF per
	(method, top_k, k_retvr, metric) under {res_dir}/plots_non_anchor/ (reference: crossenc.py:404-510)."""
	from eval.matrix_approx_zeshel import plot_heat_map
	with open(f"{res_dir}/retrieval_wrt_exact_crossenc.json") as f:
		res = json.load(f)
	grids = res["other_args"]
	rows, cols = grids["n_ment_anchors_vals"], grids["n_ent_anchors_vals"]
	made = []
	for method in method_vals:
		for tk_key, by_kr in res.get(method, {}).items():
			for kr_key, cells in by_kr.items():
				for metric in ("exact_vs_reranked_approx_retvr~common_frac_mean", "approx_error_relative"):
					M = np.full((len(rows), len(cols)), np.nan)
					for i, nm in enumerate(rows):
						for j, ne in enumerate(cols):
							cell = cells.get(f"anc_n_m={nm}~anc_n_e={ne}")
							if cell is not None:
								M[i, j] = cell["non_anchor"][metric]
					made.append(plot_heat_map(M, rows, cols, metric, tk_key, f"{res_dir}/plots_non_anchor/{method}",
											  title=f"{method} {tk_key} {kr_key} {metric}", fname=f"{tk_key}_{kr_key}_{metric.replace('~', '_')}"))
	return made


def run(base_res_dir, data_info, n_seeds, plot_only, misc, arg_dict, grid_overrides, dtype, device, pinv_backend="auto", score_chunks=None):
	from anncur_amd import harness
	data_name, data_fname = data_info
	world = int(os.environ.get("WORLD_SIZE", "1"))
	chunked = None
	if score_chunks:
		# the producer's row chunks go straight to the device (each rank keeps its own row block): no combined pickle
		from anncur_amd import ingest
		LOGGER.info(f"Ingesting {len(score_chunks)} score chunks")
		if world > 1:
			device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
			torch.cuda.set_device(device)
		chunked = ingest.load_score_chunks(score_chunks, device, dtype, rank=int(os.environ.get("RANK", "0")), world=world)
		total_n_ment, total_n_ent = chunked["n_rows"], chunked["n_ent"]
		scores = None
	else:
		LOGGER.info("Loading precomputed ment_to_ent scores")
		dump = harness.load_score_pickle(data_fname["crossenc_ment_to_ent_scores"])
		scores = dump["ment_to_ent_scores"]
		total_n_ment, total_n_ent = scores.shape
	grids = harness.default_grids_A(total_n_ment, total_n_ent)
	for key, val in grid_overrides.items():
		if val is not None:
			grids[key] = val
	res_dir = f"{base_res_dir}/nm={total_n_ment}_ne={total_n_ent}_s={n_seeds}{misc}"
	Path(res_dir).mkdir(exist_ok=True, parents=True)
	other_args = {"arg_dict": arg_dict, "top_k_vals": grids["top_k_vals"], "top_k_retr_vals": grids["top_k_retr_vals"],
				  "n_ent_anchors_vals": grids["n_ent_anchors_vals"], "n_ment_anchors_vals": grids["n_ment_anchors_vals"]}
	if not plot_only:
		def progress(method, ctr, n):
			if ctr % max(1, n // 10) == 0:
				LOGGER.info(f"method={method}: cell {ctr}/{n}")
		if world > 1:
			# launched with torchrun: one process per GPU, the matrix row-sharded, one RCCL all-gather of the anchor rows per index
			import torch.distributed as dist
			from anncur_amd.dist import ShardedScoreMatrix, shard_bounds
			local_rank = int(os.environ.get("LOCAL_RANK", "0"))
			device = torch.device("cuda", local_rank)
			torch.cuda.set_device(device)
			if not dist.is_initialized():
				dist.init_process_group(os.environ.get("ANNCUR_DIST_BACKEND", "nccl"))
			s, e = shard_bounds(total_n_ment, dist.get_rank(), world)
			local = chunked["A_local"] if chunked is not None else harness.to_device_matrix(scores[s:e], device, dtype)
			sharded = ShardedScoreMatrix(local, total_n_ment)
			if "cur_oracle" in grids["eval_methods"]:
				LOGGER.info("row-sharded run: only method=cur is evaluated (cur_oracle needs the whole matrix on one device)")
			grids["eval_methods"] = ["cur"]
			eval_res = harness.run_entry_A_sharded(sharded, grids, n_seeds, progress)
			if eval_res is None:          # ranks > 0 are done
				return res_dir
		else:
			if torch.device(device).type == "cuda":
				torch.cuda.set_device(device)   # the launch stream and torch's allocations follow --device
			A_dev = chunked["A_local"] if chunked is not None else harness.to_device_matrix(scores, device, dtype)
			eval_res = harness.run_entry_A(A_dev, grids, n_seeds, progress, pinv_backend)
		eval_res["other_args"] = other_args
		with open(f"{res_dir}/retrieval_wrt_exact_crossenc.json", "w") as fout:
			json.dump(obj=eval_res, fp=fout, indent=4)
		LOGGER.info(f"Wrote {res_dir}/retrieval_wrt_exact_crossenc.json")
	try:
		plot(res_dir=res_dir, method_vals=grids["eval_methods"])
	except ImportError:
		LOGGER.info("matplotlib not available: skipping heat maps")
	return res_dir


def main(argv=None):
	data_dir = "../../data/zeshel"
	worlds = get_zeshel_world_info()
	parser = argparse.ArgumentParser(description="Run eval for various retrieval methods wrt exact crossencoder scores. "
												 "This evaluation does not use ground-truth entity information into account")
	parser.add_argument("--data_name", type=str, choices=[w for _, w in worlds], help="Dataset name")
	parser.add_argument("--bi_model_file", type=str, default="", help="File for biencoder ckpt (bienc baseline: not part of this build)")
	parser.add_argument("--res_dir", type=str, required=True, help="Res dir with score matrices, and to save results")
	parser.add_argument("--n_seeds", type=int, default=10, help="Number of seeds to run")
	parser.add_argument("--plot_only", type=int, default=0, choices=[0, 1], help="1 to only plot results, 0 to run exp and then plot results")
	parser.add_argument("--n_ment", type=int, default=100, help="Number of mentions in precomputed mention-entity score matrix")
	parser.add_argument("--batch_size", type=int, default=50, help="Batch size to use with biencoder")
	parser.add_argument("--misc", type=str, default="", help="Misc suffix")
	parser.add_argument("--disable_wandb", type=int, default=0, choices=[0, 1], help="1 to disable wandb and 0 to use it (wandb is optional here)")
	# overrides of the grids the reference hard-codes (crossenc.py:225-239); defaults reproduce the reference
	parser.add_argument("--eval_methods", type=lambda s: s.split(","), default=None, help="comma list out of cur,cur_oracle")
	parser.add_argument("--n_ment_anchors_vals", type=_ints, default=None)
	parser.add_argument("--n_ent_anchors_vals", type=_ints, default=None)
	parser.add_argument("--top_k_vals", type=_ints, default=None)
	parser.add_argument("--top_k_retr_vals", type=_ints, default=None)
	parser.add_argument("--data_dir", type=str, default=data_dir)
	parser.add_argument("--dtype", type=str, default="fp32", choices=["fp32", "bf16"], help="storage/compute type of the score matrix on the GPU")
	parser.add_argument("--device", type=str, default="cuda:0")
	parser.add_argument("--pinv", type=str, default="auto", choices=["numpy", "device", "auto", "device32"],
						help="pseudo-inverse: numpy = the reference's numpy.linalg.pinv on the host (bit-identical U); device = fp64 Newton-Schulz on the GPU "
							 "(exact pseudo-inverse of the fp32 block, rounded once); auto = device while the block is well conditioned, else numpy")
	parser.add_argument("--score_chunks", type=str, nargs="+", default=None,
						help="the producer's per-chunk score pickles (mention order) instead of the combined file: ingested chunk by chunk, row-sharded under torchrun")
	args = parser.parse_args(argv)
	if args.bi_model_file != "":
		raise SystemExit("--bi_model_file: the bi-encoder baseline needs the reference's BERT models and is out of scope of this build")
	misc = "_" + args.misc if args.misc != "" else ""
	datasets = get_dataset_info(data_dir=args.data_dir, res_dir=args.res_dir, worlds=worlds, n_ment=args.n_ment)
	LOGGER.info(f"Running inference for world = {args.data_name}")
	return run(base_res_dir=f"{args.res_dir}/{args.data_name}/Retrieval_wrt_Exact_CrossEnc", data_info=(args.data_name, datasets[args.data_name]),
			   n_seeds=args.n_seeds, plot_only=bool(args.plot_only), misc=misc, arg_dict=dict(args.__dict__),
			   grid_overrides={"eval_methods": args.eval_methods, "n_ment_anchors_vals": args.n_ment_anchors_vals,
							   "n_ent_anchors_vals": args.n_ent_anchors_vals, "top_k_vals": args.top_k_vals, "top_k_retr_vals": args.top_k_retr_vals},
			   dtype=args.dtype, device=torch.device(args.device), pinv_backend=args.pinv, score_chunks=args.score_chunks)


if __name__ == "__main__":
	main()
