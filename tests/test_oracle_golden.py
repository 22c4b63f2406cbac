"""Pins oracle/cur_oracle.py (the CPU restatement) against vectors produced by the
reference itself (oracle/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import cur_oracle as O


def _close(a, b, rtol=2e-5, atol=2e-5):
	np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def test_anchor_selection_kat(golden_meta):
	kat = golden_meta["anchor_kat"]
	rng = np.random.default_rng(seed=kat["seed"])
	assert [int(x) for x in O.select_anchors(rng, 1000, 8)] == kat["rows_1000_8"] == [16, 40, 75, 268, 306, 508, 633, 844]
	assert [int(x) for x in O.select_anchors(rng, 5000, 8)] == kat["cols_5000_8"] == [1386, 2715, 2797, 3157, 3354, 3642, 4078, 4672]


def test_worked_4x5(golden_dir):
	g = np.load(os.path.join(golden_dir, "worked_4x5.npz"))
	A = torch.tensor(g["A"])
	ri, ci = g["row_idxs"].tolist(), g["col_idxs"].tolist()
	cur = O.CURApproxOracle(A[ri, :], A[:, ci], ri, ci, "rows")
	_close(cur.U, g["U"]); _close(cur.latent_cols, g["latent_cols"])
	_close(cur.get(list(range(4)), list(range(5))), g["S"])
	_close(cur.get_rows([1, 3]), g["get_rows"]); _close(cur.get_cols([0, 4]), g["get_cols"])
	tv, ti = cur.topk_in_row(A[:, ci], 2)
	_close(tv, g["topk_val"]); assert (ti.numpy() == g["topk_idx"]).all()
	# the survey's printed values
	_close(cur.U, [[-0.422951, 0.167213], [0.432787, -0.124590], [-0.065574, 0.049180]], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("tag", ["lr_64x200", "lr_200x1000"])
def test_lowrank_cases(golden_dir, tag):
	g = np.load(os.path.join(golden_dir, f"{tag}.npz"))
	A = torch.tensor(g["A"]); ri, ci, k = g["row_idxs"].tolist(), g["col_idxs"].tolist(), int(g["k"])
	n, m = A.shape
	for method, extra in (("cur", {}), ("cur_oracle", {"A": A})):
		cur = O.CURApproxOracle(A[ri, :], A[:, ci], ri, ci, "rows", **extra)
		_close(cur.U, g[f"{method}_U"], rtol=1e-4, atol=1e-4)
		_close(cur.latent_cols, g[f"{method}_E"], rtol=1e-4, atol=1e-4)
		_close(cur.get(list(range(n)), list(range(m))), g[f"{method}_S"], rtol=1e-4, atol=1e-4)
		tv, ti = cur.topk_in_row(A[:, ci], k)
		assert (ti.numpy() == g[f"{method}_topk_idx"]).all()
	curc = O.CURApproxOracle(A[ri, :], A[:, ci], ri, ci, "cols")
	_close(curc.latent_rows, g["cols_latent_rows"], rtol=1e-4, atol=1e-4)
	_close(curc.get_complete_col(A[ri, :][:, :7]), g["cols_complete_col"], rtol=1e-4, atol=1e-4)
	with pytest.raises(NotImplementedError):
		curc.get_complete_row(A[:, ci])
	with pytest.raises(NotImplementedError):
		O.CURApproxOracle(A[ri, :], A[:, ci], ri, ci, "rows").get_complete_col(A[ri, :])
	with pytest.raises(NotImplementedError):
		O.CURApproxOracle(A[ri, :], A[:, ci], ri, ci, "bogus")
	with pytest.raises(AssertionError):
		O.CURApproxOracle(A[ri, :], A[:, ci], ri[::-1], ci, "rows")


def test_compute_overlap_kat(golden_meta):
	kat = golden_meta["overlap_kat"]
	out = O.compute_overlap(kat["l1"], kat["l2"])
	assert {k: list(v) for k, v in out.items()} == kat["out"]
	assert out["common"] == ("mean 2.0000", "std 1.6330", "p50 2.0000")
	assert out["common_frac"] == ("mean 0.5000", "std 0.4082", "p50 0.5000")
	assert {k: list(v) for k, v in O.compute_overlap([], []).items()} == kat["empty"]
	with pytest.raises(AssertionError):
		O.compute_overlap([[1, 2]], [[1]])


def test_entry_point_A_results(golden_meta):
	torch.manual_seed(0)
	A = torch.randn(1000, 32) @ torch.randn(32, 5000) / (32 ** 0.5) + 0.1 * torch.randn(1000, 5000)
	gold = golden_meta["entryA"]["results"]
	res = O.run_approx_eval_w_seed("cur", A, 64, 64, 10, 100, 0)
	for t in ("anchor", "non_anchor", "all"):
		for m, v in gold["cur_seed0"][t].items():
			assert float(res[t][m]) == pytest.approx(v, rel=1e-4, abs=1e-4), (t, m)
	assert gold["cur_seed0"]["all"]["exact_vs_reranked_approx_retvr~common_frac_mean"] == pytest.approx(0.5237)
	res = O.run_approx_eval("cur_oracle", A, 64, 64, 10, 100, 2)
	for t in ("anchor", "non_anchor", "all"):
		for m, v in gold["cur_oracle_2seeds"][t].items():
			assert res[t][m] == pytest.approx(v, rel=1e-4, abs=1e-4), (t, m)
	res = O.run_approx_eval("cur", A, 128, 64, 10, 100, 2)
	for t in ("anchor", "non_anchor", "all"):
		for m, v in gold["cur_kq128_ki64_2seeds"][t].items():
			assert res[t][m] == pytest.approx(v, rel=1e-4, abs=1e-4), (t, m)


def test_entry_point_B_results(golden_dir, golden_meta):
	A_train, A_test = O.synth_protocol_b(500, 2000, 20000, rank=64, noise=0.05, seed=0)
	g = np.load(os.path.join(golden_dir, "protoB_2000x20000.npz"))
	rng = np.random.default_rng(seed=0)
	anc = O.select_anchors(rng, 20000, 256)
	assert (np.array(anc) == g["anc"]).all()
	cur = O.CURApproxOracle(rows=A_train, cols=A_train[:, anc], row_idxs=np.arange(500), col_idxs=anc, approx_preference="rows")
	S = cur.get_complete_row(A_test[:, anc])
	_close(cur.U[:8, :8], g["U_sample"], rtol=1e-4, atol=1e-5)
	_close(S[:16, :256], g["S_sample"], rtol=1e-4, atol=1e-4)
	_, ti = torch.topk(S, 100, dim=1)
	assert (np.sort(ti.numpy(), 1) == np.sort(g["approx_topk_idx"], 1)).mean() > 0.9999
	_, ei = torch.topk(A_test, 100, dim=1)
	assert (ei.numpy() == g["exact_topk_idx"]).all()
	res = O.eval_approx_score_mat_for_all_topk(A_test, S, [1, 10, 50, 100], 100)
	gold = golden_meta["entryB"]["all_topk_kretvr100"]
	for k in (1, 10, 50, 100):
		for m, v in gold[str(k)].items():
			assert res[k][m] == pytest.approx(v, abs=2e-4), (k, m)
	# the tie-stable variant agrees with the reference-faithful loop on this (tie-free up to a handful of fp32 coincidences) input
	res_st = O.eval_all_topk_stable(A_test[:300], S[:300], [1, 10, 50, 100], 100)
	res_rf = O.eval_approx_score_mat_for_all_topk(A_test[:300], S[:300], [1, 10, 50, 100], 100)
	for k in (1, 10, 50, 100):
		for m in res_rf[k]:
			assert res_st[k][m] == pytest.approx(res_rf[k][m], abs=1e-3), (k, m)
	res1 = O.eval_approx_score_mat(A_test, S, 10, 64)
	for m, v in golden_meta["entryB"]["single_k10_kretvr64"].items():
		assert res1[m] == pytest.approx(v, abs=2e-4), m


def test_entry_point_B_sweep(golden_meta):
	g = torch.Generator().manual_seed(3)
	Z = torch.randn(16, 600, generator=g)
	A_train = torch.randn(60, 16, generator=g) @ Z / 4 + 0.05 * torch.randn(60, 600, generator=g)
	A_test = torch.randn(40, 16, generator=g) @ Z / 4 + 0.05 * torch.randn(40, 600, generator=g)
	res = O.run_eval_method_cur(A_test, A_train, seed=5, top_k_vals=[1, 10, 50, 100], top_k_retr_vals=[5, 10, 50],
								n_ent_anchors_vals=[10, 20, 30])
	gold = golden_meta["entryB_sweep"]["results"]
	seen = 0
	for key, v in gold.items():
		tk, kr, na = key.split("|")
		got = res[tk][kr][f"anc_n_m=60_{na}"]
		assert got == pytest.approx(v, abs=1e-4), key
		seen += 1
	assert seen == len(gold) and seen > 0
	# cells with top_k > k_retvr are absent, as in the reference
	assert "k_retvr=5" not in res.get("top_k=10", {})


def test_grids_match_reference_defaults():
	kr, na = O.splits_grids(10031)
	assert kr[0] == 0 and 1000 in kr and 900 in kr and 45 in kr
	assert 10031 in na and 2000 in na and 10 in na


def test_oracle_loop_reproduces_the_fixed_anc_ent_goldens(golden_dir):
	"""splits.py:305-358 through the oracle's statement of the per-query loop, against what the reference's own run_eval_method returned on the same
	inputs (oracle/make_golden.py (viii)): fixed_anc_ent on a few k_retvr values, fixed_anc_ent_cur on the first anchor counts of the reference's
	sequential rng(0) stream."""
	import torch
	from oracle import cur_oracle as O
	gold = np.load(os.path.join(golden_dir, "fixed_anc_ent_1100.npz"))
	g = torch.Generator().manual_seed(13)
	n_ent, n_train, n_test, r, n_fixed = 1100, 50, 150, 12, 40
	Z = torch.randn(r, n_ent, generator=g)
	A_train = torch.randn(n_train, r, generator=g) @ Z / r ** 0.5 + 0.05 * torch.randn(n_train, n_ent, generator=g)
	A_test = torch.randn(n_test, r, generator=g) @ Z / r ** 0.5 + 0.05 * torch.randn(n_test, n_ent, generator=g)
	topk_ents = torch.randperm(n_ent, generator=g)[:60]
	e2e = (Z.t() @ Z[:, topk_ents]) / r + 0.02 * torch.randn(n_ent, 60, generator=g)
	key = "exact_vs_reranked_approx_retvr~common_frac_mean"

	def cells(method):
		return {tuple(k): v for k, v in zip(gold[f"{method}_keys"].tolist(), gold[f"{method}_common_frac_mean"].tolist())}
	want = cells("fixed_anc_ent")
	S_fix = A_test[:, topk_ents[:n_fixed]] @ e2e[:, :n_fixed].t()
	anc_vals = sorted({k[2] for k in want})
	for kr in (10, 50, 200):
		for k, m in O.eval_approx_score_mat_for_all_topk(A_test, S_fix, [1, 10, 50, 100], kr).items():
			for na in (anc_vals[0], anc_vals[7], anc_vals[-1]):   # the same numbers under every anchor count (splits.py:399-407)
				assert m[key] == pytest.approx(want[(k, kr, na)], abs=1.01e-4), (k, kr, na)
	want = cells("fixed_anc_ent_cur")
	R = e2e[:, :n_fixed].t()
	rng = np.random.default_rng(seed=0)
	for n_anc in anc_vals[:6]:                                     # (0, 1, 2, ...: consumed in grid order from ONE generator)
		anc = sorted(rng.choice(n_ent, size=n_anc, replace=False))
		if n_anc == 0:
			continue
		U = torch.tensor(np.linalg.pinv(R[:, anc].numpy()))
		S = A_test[:, anc] @ (U @ R)
		for kr in (10, 100):
			for k, m in O.eval_approx_score_mat_for_all_topk(A_test, S, [1, 10, 50, 100], kr).items():
				assert m[key] == pytest.approx(want[(k, kr, n_anc)], abs=1.01e-4), (k, kr, n_anc)
