#!/usr/bin/env python
"""Train/test mention splits of one mention x entity score dump -- the producer of entry point B's inputs
(drop-in for the reference's utils/split_zeshel_ment2ent_for_cur_exps.py: same flags, same generator consumption order,
same output files and pickle keys).

For every (nm_train, split_idx): train = sorted(rng.choice(n_ments, nm_train)), test = complement,
train_dev = sorted(rng.choice(train, int(nm_train * dev_frac))), train_train = train minus train_dev; written to
{out}/m2e_splits/nm_train={nm_train}/split_idx={i}/{train,train_train,train_dev,test}.pkl with the extra key "ment_idxs".
Host-side data preparation (row gathers of a pickle): no GPU involved.
"""
import argparse
import itertools
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

from utils.zeshel_utils import N_ENTS_ZESHEL

logging.basicConfig(stream=sys.stderr, format="%(asctime)s - %(levelname)s - %(name)s - %(message)s ",
					datefmt="%d/%m/%Y %H:%M:%S", level=logging.INFO)
LOGGER = logging.getLogger(__name__)

NUM_TRAIN_MENT_VALS = [50, 100, 200, 500, 1000, 2000]


def plan_splits(n_ments, num_train_ment_vals, num_splits, seed, dev_frac):
	"""Yields (nm_train, split_idx, {split name: sorted mention indices}) in the reference's generator order."""
	assert 0 <= dev_frac < 1
	rng = np.random.default_rng(seed=seed)
	everything = set(range(n_ments))
	for nm_train, split_idx in itertools.product(num_train_ment_vals, range(num_splits)):
		if nm_train > n_ments:
			LOGGER.info(f"Number of train ments = {nm_train} > n_ments = {n_ments}")
			continue
		train = sorted(rng.choice(n_ments, size=nm_train, replace=False))
		dev = sorted(rng.choice(a=train, size=int(nm_train * dev_frac), replace=False))
		yield nm_train, split_idx, {
			"train_dev": dev,
			"train_train": sorted(set(train) - set(dev)),
			"train": train,
			"test": sorted(everything - set(train)),
		}


def write_split(out_dir, name, ment_idxs, dump):
	if len(ment_idxs) == 0:
		LOGGER.info(f"Empty list of indices for split = {name}")
		return None
	sub = {
		"ment_to_ent_scores": dump["ment_to_ent_scores"][ment_idxs, :],
		"test_data": [dump["test_data"][i] for i in ment_idxs],       # key name kept from the entity-linking dumps
		"mention_tokens_list": [dump["mention_tokens_list"][i] for i in ment_idxs],
		"ment_idxs": ment_idxs,
		"entity_id_list": [],
		"entity_tokens_list": [],
		"arg_dict": dump["arg_dict"],
	}
	Path(out_dir).mkdir(exist_ok=True, parents=True)
	path = f"{out_dir}/{name}.pkl"
	with open(path, "wb") as f:
		pickle.dump(sub, f)
	return path


def run(data_name, m2e_file, num_train_ment_vals, num_splits, seed, dev_frac, base_out_dir):
	with open(m2e_file, "rb") as f:
		dump = pickle.load(f)
	ids = dump["entity_id_list"]
	assert len(ids) == 0 or (np.asarray(ids) == np.arange(N_ENTS_ZESHEL[data_name])).all(), \
		"entity_id_list is not stored per split: it must be empty or arange(n_ents)"
	n_ments = dump["ment_to_ent_scores"].shape[0]
	assert n_ments == len(dump["test_data"]) and n_ments == len(dump["mention_tokens_list"])
	written = []
	for nm_train, split_idx, parts in plan_splits(n_ments, num_train_ment_vals, num_splits, seed, dev_frac):
		for name, idxs in parts.items():
			written.append(write_split(f"{base_out_dir}/nm_train={nm_train}/split_idx={split_idx}", name, idxs, dump))
	return [w for w in written if w]


def main(argv=None):
	parser = argparse.ArgumentParser(description="Split zeshel mention-entity score matrices into train/test mentions")
	parser.add_argument("--data_name", type=str, required=True, help="Data/domain name")
	parser.add_argument("--m2e_file", type=str, required=True, help="Mention-Entity score file")
	parser.add_argument("--out_dir", type=str, default="", help="Output dir")
	parser.add_argument("--seed", type=int, default=0, help="Random seed")
	parser.add_argument("--dev_frac", type=float, default=0.1, help="Fraction of the train mentions held out as train_dev")
	parser.add_argument("--num_splits", type=int, default=5, help="Number of random splits")
	args = parser.parse_args(argv)
	base = os.path.dirname(args.m2e_file) if args.out_dir == "" else args.out_dir
	out_dir = f"{base}/m2e_splits"
	files = run(args.data_name, args.m2e_file, NUM_TRAIN_MENT_VALS, args.num_splits, args.seed, args.dev_frac, out_dir)
	with open(f"{out_dir}/split_args.json", "w") as f:
		json.dump(args.__dict__, f)
	return files


if __name__ == "__main__":
	main()
