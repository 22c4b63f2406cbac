"""TEST INFRASTRUCTURE ONLY (dev tool): mention-index KAT of the reference's split producer
(utils/split_zeshel_ment2ent_for_cur_exps.py:54-129), run unmodified from /root/reference on a synthetic dump.
Run:  cd /tmp && PYTHONDONTWRITEBYTECODE=1 python -O /root/repo/oracle/make_golden_split.py      -> tests/golden/split_kat.json"""
import json, os, pickle, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import _StubFinder, REF, OUT  # noqa: E402


def main():
	sys.dont_write_bytecode = True
	sys.meta_path.insert(0, _StubFinder())
	sys.path.insert(0, REF)
	import numpy as np, torch
	import utils.split_zeshel_ment2ent_for_cur_exps as ref
	tmp = tempfile.mkdtemp()
	n_m, n_e = 130, 40
	dump = {"ment_to_ent_scores": torch.arange(n_m * n_e, dtype=torch.float32).reshape(n_m, n_e), "test_data": [{"id": i} for i in range(n_m)],
			"mention_tokens_list": [[i, i + 1] for i in range(n_m)], "entity_id_list": [], "entity_tokens_list": [], "arg_dict": {"a": 1}}
	with open(f"{tmp}/m2e.pkl", "wb") as f:
		pickle.dump(dump, f)
	ref.run(data_name="yugioh", m2e_file=f"{tmp}/m2e.pkl", num_train_ment_vals=[50, 100, 200], num_splits=2, seed=7, dev_frac=0.1, base_out_dir=f"{tmp}/out")
	kat = {}
	for nm in (50, 100):
		for si in (0, 1):
			for name in ("train", "train_train", "train_dev", "test"):
				with open(f"{tmp}/out/nm_train={nm}/split_idx={si}/{name}.pkl", "rb") as f:
					d = pickle.load(f)
				kat[f"{nm}/{si}/{name}"] = {"ment_idxs": [int(x) for x in d["ment_idxs"]], "first_score": float(d["ment_to_ent_scores"][0, 0]),
											"keys": sorted(d.keys())}
	assert not os.path.exists(f"{tmp}/out/nm_train=200")
	with open(os.path.join(OUT, "split_kat.json"), "w") as f:
		json.dump({"input": "130x40 arange scores, seed 7, dev_frac 0.1, nm_train in (50,100,200), 2 splits", "splits": kat}, f, indent=0)
	print("wrote split_kat.json", len(kat))


if __name__ == "__main__":
	main()
