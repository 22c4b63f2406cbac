#!/usr/bin/env python
"""Entry point B -- drop-in for the reference's eval/run_retrieval_eval_wrt_exact_crossenc_w_fixed_train_test_splits.py.

Same flags; reads the train/test split pickles (keys ment_to_ent_scores, mention_tokens_list, ment_idxs); writes
{res_dir}/method={eval_method}_{misc}.json with res["seed=s"]["top_k=.."]["k_retvr=.."]["anc_n_m=.._anc_n_e=.."][metric] and
res["other_args"]["retriever_params"], the layout eval/compile_emnlp_retrieval_eval_wrt_exact_crossenc.py:334 consumes.
Methods: cur and fixed_anc_ent_cur run fully on the GPU; bienc / tfidf / fixed_anc_ent run from PRECOMPUTED embeddings
(--mention_embeds_file / --entity_embeds_file, or the e2e pickle): the encoders themselves are out of scope.
"""
import argparse
import json
import logging
import os
import pickle
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


def run_eval_method(curr_method, test_data_file, train_data_file, args, seed, device):
	from anncur_amd import harness, ops
	LOGGER.info("Loading precomputed ment_to_ent scores")
	test = harness.load_score_pickle(test_data_file)
	A_test = test["ment_to_ent_scores"]
	test_ment_idxs = test["ment_idxs"]  # required by the reference as well (splits.py:225)
	n_test, n_ent = A_test.shape
	train = harness.load_score_pickle(train_data_file)
	A_train = train["ment_to_ent_scores"]
	n_train, n_ent_train = A_train.shape
	assert n_ent_train == n_ent, "Train and test entities differ! Use entity_id_list from data dump to resolve this"
	grids = harness.default_grids_B(n_ent, curr_method)
	for key in ("top_k_vals", "top_k_retr_vals", "n_ent_anchors_vals"):
		if getattr(args, key) is not None:
			grids[key] = sorted(set(getattr(args, key)))
	A_test_dev = harness.to_device_matrix(A_test, device, args.dtype)
	LOGGER.info(f"Computing approximate test mention-to-entity scores using method={curr_method}")
	if curr_method == "cur":
		A_train_dev = harness.to_device_matrix(A_train, device, args.dtype)
		res = harness.run_eval_method_cur(A_test_dev, A_train_dev, seed, grids,
										  progress=lambda j, n: LOGGER.info(f"anchor count {j + 1}/{n}"), pinv_backend=args.pinv)
	elif curr_method in ("bienc", "tfidf"):
		if not (args.mention_embeds_file and args.entity_embeds_file):
			raise SystemExit(f"eval_method={curr_method}: pass --mention_embeds_file and --entity_embeds_file (.npy); "
							 "computing them needs the reference's encoders, which are out of scope of this build")
		ment = np.load(args.mention_embeds_file)
		if curr_method == "tfidf" and ment.shape[0] != n_test:
			ment = ment[np.asarray(test_ment_idxs)]        # splits.py:378
		res = harness.run_eval_method_embeds(A_test_dev, torch.as_tensor(ment, dtype=torch.float32).to(device),
											 torch.as_tensor(np.load(args.entity_embeds_file), dtype=torch.float32).to(device), n_train, grids)
	elif curr_method in ("fixed_anc_ent", "fixed_anc_ent_cur"):
		if not os.path.isfile(args.e2e_fname):
			raise SystemExit(f"File {args.e2e_fname} not found")
		with open(args.e2e_fname, "rb") as fin:
			e2e = pickle.load(fin)
		full = torch.as_tensor(np.asarray(e2e["ent_to_ent_scores"]), dtype=torch.float32).to(device)   # n_ents x n_anchors
		if curr_method == "fixed_anc_ent":
			anc = [int(x) for x in np.asarray(e2e["topk_ents"][0])[:args.n_fixed_anc_ent]]
			ment = ops.gather_cols(A_test_dev, anc, out_dtype=torch.float32)
			res = harness.run_eval_method_embeds(A_test_dev, ment, full[:, :args.n_fixed_anc_ent].contiguous(), n_train, grids)
		else:
			res = harness.run_eval_method_fixed_anc_ent_cur(A_test_dev, full, args.n_fixed_anc_ent, grids, key_n_m=n_train)
	else:
		raise NotImplementedError(f"Method = {curr_method} not supported")
	params = {"top_k_retr_vals": grids["top_k_retr_vals"], "top_k_vals": grids["top_k_vals"], "n_ent_anchors_vals": grids["n_ent_anchors_vals"]}
	return res, params


def run(args, device):
	eval_method, n_seeds = args.eval_method, args.n_seeds
	if device.type == "cuda":
		torch.cuda.set_device(device)   # the launch stream and torch's allocations follow --device
	assert eval_method == "cur" or n_seeds == 1, f"n_seed = {n_seeds} only allowed for eval_method = cur "
	if args.use_wandb:
		LOGGER.info("--use_wandb: wandb logging is optional and not configured in this build; continuing without it")
	eval_res, retvr_params = {}, {}
	for seed in range(n_seeds):
		curr_res, retvr_params = run_eval_method(eval_method, args.test_data_file, args.train_data_file, args, seed, device)
		eval_res[f"seed={seed}"] = curr_res
	arg_dict = dict(args.__dict__)
	eval_res["other_args"] = arg_dict
	eval_res["other_args"]["retriever_params"] = retvr_params
	res_file = f"{args.res_dir}/method={eval_method}_{args.misc}.json"
	Path(os.path.dirname(res_file)).mkdir(exist_ok=True, parents=True)
	with open(res_file, "w") as fout:
		json.dump(eval_res, fout, indent=4)
	LOGGER.info(f"Wrote {res_file}")
	return res_file


def main(argv=None):
	worlds = get_zeshel_world_info()
	parser = argparse.ArgumentParser(description="Run eval for various retrieval methods wrt exact crossencoder scores using a fixed train/test "
												 "split. This evaluation does not use ground-truth entity information into account")
	parser.add_argument("--data_name", type=str, choices=[w for _, w in worlds], help="Dataset name")
	parser.add_argument("--eval_method", type=str, choices=["cur", "bienc", "fixed_anc_ent", "fixed_anc_ent_cur", "tfidf"], help="Eval method")
	parser.add_argument("--res_dir", type=str, required=True, help="Result directory")
	parser.add_argument("--test_data_file", type=str, required=True, help="Test data file")
	parser.add_argument("--train_data_file", type=str, default="", help="Training data file. Used for method=cur")
	parser.add_argument("--n_seeds", type=int, default=1, help="Number of seeds to run")
	parser.add_argument("--bi_model_file", type=str, default="", help="File for biencoder ckpt (not used: pass precomputed embeddings instead)")
	parser.add_argument("--batch_size", type=int, default=50, help="Batch size to use with biencoder (unused)")
	parser.add_argument("--e2e_fname", type=str, default="", help="File w/ entity2entity scores. Used for method=fixed_anc_ent(_cur)")
	parser.add_argument("--n_fixed_anc_ent", type=int, default=0, help="Number of fixed anchor entities to use")
	parser.add_argument("--mention_file", type=str, default="", help="Raw mention data (tfidf; unused: pass precomputed embeddings)")
	parser.add_argument("--entity_file", type=str, default="", help="Raw entity data (tfidf; unused: pass precomputed embeddings)")
	parser.add_argument("--mode", type=str, choices=["eval", "plot", "eval_n_plot"], default="eval", help="To run in eval mode or just plotting or both")
	parser.add_argument("--misc", type=str, default="", help="Misc suffix")
	parser.add_argument("--use_wandb", type=int, default=0, choices=[0, 1], help="1 to enable wandb and 0 to disable it ")
	# additions: grid overrides (defaults = the reference's hard-coded grids, splits.py:238-251), precomputed embeddings, device/dtype
	parser.add_argument("--top_k_vals", type=_ints, default=None)
	parser.add_argument("--top_k_retr_vals", type=_ints, default=None)
	parser.add_argument("--n_ent_anchors_vals", type=_ints, default=None)
	parser.add_argument("--mention_embeds_file", type=str, default="")
	parser.add_argument("--entity_embeds_file", type=str, default="")
	parser.add_argument("--dtype", type=str, default="fp32", choices=["fp32", "bf16"])
	parser.add_argument("--device", type=str, default="cuda:0")
	parser.add_argument("--pinv", type=str, default="auto", choices=["numpy", "device", "auto", "device32"],
						help="pseudo-inverse: numpy = the reference's numpy.linalg.pinv on the host (bit-identical U); device = fp64 Newton-Schulz on the GPU "
							 "(exact pseudo-inverse of the fp32 block, rounded once); auto = device while the block is well conditioned, else numpy")
	args = parser.parse_args(argv)
	_ = get_dataset_info(data_dir="../../data/zeshel", res_dir=args.res_dir, worlds=worlds)  # kept for parity with the reference's main()
	LOGGER.info(f"Running inference for world = {args.data_name}")
	return run(args, torch.device(args.device))


if __name__ == "__main__":
	main()
