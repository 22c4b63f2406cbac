"""BASELINE config 5 stand-in: the four ZeShEL test domains' SHAPES (no dataset here: synthetic protocol-B matrices, labelled as
such), 1024 anchor items, 2048 anchor queries, k = k_retvr = 100; fp32 vs bf16 storage/compute, macro and micro recall."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import harness, ops
from anncur_amd.synth import protocol_b
DOMAINS = {"forgotten_realms": (1200, 15603), "lego": (1199, 10076), "star_trek": (4227, 34430), "yugioh": (3374, 10031)}
dev = torch.device("cuda")
key = "exact_vs_reranked_approx_retvr~common_frac_mean"
out = {"data": "synthetic stand-in with the ZeShEL test domains' shapes (rank 64 + 0.05 noise)", "anchors": 1024, "anchor_queries": 2048, "k": 100, "k_retvr": 100, "domains": {}}
grids = {"top_k_vals": [1, 10, 50, 100], "top_k_retr_vals": [100], "n_ent_anchors_vals": [1024]}
for d_i, (name, (n_m, n_e)) in enumerate(DOMAINS.items()):
	A_train, A_test = protocol_b(2048, n_m, n_e, dev, seed=10 + d_i, dtype=torch.float32)
	row = {"n_ment": n_m, "n_ent": n_e}
	for dtype, pinv in (("fp32", "numpy"), ("bf16", "numpy"), ("fp32", "auto"), ("bf16", "auto")):
		At = A_train if dtype == "fp32" else ops.convert(A_train, torch.bfloat16)
		Aq = A_test if dtype == "fp32" else ops.convert(A_test, torch.bfloat16)
		harness.run_eval_method_cur(Aq, At, 0, grids, pinv_backend=pinv); torch.cuda.synchronize()   # warm-up
		t0 = time.perf_counter()
		res = harness.run_eval_method_cur(Aq, At, 0, grids, pinv_backend=pinv)
		torch.cuda.synchronize()
		tag = dtype if pinv == "numpy" else dtype + "_auto_pinv"   # (auto = fp64 Newton-Schulz on the GPU for these well-conditioned blocks)
		row[tag] = {f"recall@{k}": res[f"top_k={k}"]["k_retvr=100"][f"anc_n_m=2048_anc_n_e=1024"][key] for k in (1, 10, 50, 100)}
		row[tag]["seconds_incl_index_build"] = round(time.perf_counter() - t0, 4)
	out["domains"][name] = row
tot = sum(v[0] for v in DOMAINS.values())
for dtype in ("fp32", "bf16", "fp32_auto_pinv", "bf16_auto_pinv"):
	out[f"macro_{dtype}"] = {f"recall@{k}": round(float(np.mean([out["domains"][n][dtype][f"recall@{k}"] for n in DOMAINS])), 4) for k in (1, 10, 50, 100)}
	out[f"micro_{dtype}"] = {f"recall@{k}": round(float(sum(out["domains"][n][dtype][f"recall@{k}"] * DOMAINS[n][0] for n in DOMAINS) / tot), 4) for k in (1, 10, 50, 100)}
print(json.dumps(out, indent=1))
