"""Dev tool: wall time of entry point B's FULL default sweep (the reference's hard-coded grids, splits.py:238-251: ~70 anchor
counts x ~60 k_retvr values x 4 top_k) on a ZeShEL-domain-shaped synthetic problem, per anchor count."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import harness
from oracle import cur_oracle as O
ap = argparse.ArgumentParser()
ap.add_argument("--n_train", type=int, default=1000); ap.add_argument("--n_test", type=int, default=2374); ap.add_argument("--n_ent", type=int, default=10031)
ap.add_argument("--dtype", default="fp32"); ap.add_argument("--pinv", default="auto")
a = ap.parse_args()
A_train, A_test = O.synth_protocol_b(a.n_train, a.n_test, a.n_ent, rank=64, noise=0.05, seed=3)
At, Aq = harness.to_device_matrix(A_train, "cuda", a.dtype), harness.to_device_matrix(A_test, "cuda", a.dtype)
grids = harness.default_grids_B(a.n_ent, "cur")
print("anchor counts:", len(grids["n_ent_anchors_vals"]), "k_retvr values:", len(grids["top_k_retr_vals"]), flush=True)
marks = []
def progress(j, n):
	torch.cuda.synchronize(); marks.append((grids["n_ent_anchors_vals"][j], time.perf_counter()))
harness.run_eval_method_cur(Aq, At, 0, {"top_k_vals": [1], "top_k_retr_vals": [10], "n_ent_anchors_vals": [64]}, pinv_backend=a.pinv)  # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
res = harness.run_eval_method_cur(Aq, At, 0, grids, progress=progress, pinv_backend=a.pinv)
torch.cuda.synchronize(); t1 = time.perf_counter()
marks.append((None, t1))
per = [(marks[i][0], 1e3 * (marks[i + 1][1] - marks[i][1])) for i in range(len(marks) - 1)]
print("total %.2f s for %d cells" % (t1 - t0, sum(len(v2) for v in res.values() for v2 in v.values())))
print("slowest anchor counts (n_anc, ms):", sorted(per, key=lambda x: -x[1])[:8])
print("all:", [(n, round(ms, 1)) for n, ms in per])
if os.environ.get("ENTRYB_CPU"):   # calibrate the CPU oracle (reference-faithful loops) on a slice of the grid and scale linearly
	torch.set_num_threads(8)
	sub_anc, sub_retr = [100, 500], [10, 100, 1000]
	t0 = time.perf_counter()
	O.run_eval_method_cur(A_test, A_train, seed=0, top_k_vals=[1, 10, 50, 100], top_k_retr_vals=sub_retr, n_ent_anchors_vals=sub_anc)
	dt = time.perf_counter() - t0
	n_cells_full = len([r for r in grids["top_k_retr_vals"] if r > 0]) * len([n for n in grids["n_ent_anchors_vals"] if n > 0])
	print("CPU oracle: %.1f s for %d (k_retvr, n_anc) pairs -> ~%.0f s for the %d pairs of the full grid (8 threads)" % (dt, len(sub_anc) * len(sub_retr), dt / (len(sub_anc) * len(sub_retr)) * n_cells_full, n_cells_full))
