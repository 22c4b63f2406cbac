"""Size-independent properties at BASELINE.json's full single-GPU size (cfg2: 10 000 x 100 000 bf16, 256 anchors, k = 100) and a
ZeShEL-shaped (cfg3 stand-in: yugioh 3374 x 10031, 256 anchors, k = 64) parity run against the oracle.  Needs an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg2():
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from anncur_amd import ops
	from anncur_amd.cur import CURRowIndex
	from anncur_amd.synth import protocol_b
	dev = torch.device("cuda")
	A_train, A_test = protocol_b(512, 10000, 100000, dev, seed=0)
	anc = sorted(np.random.default_rng(0).choice(100000, 256, replace=False))
	index = CURRowIndex(A_train, anc)
	X = ops.gather_cols(A_test, anc)
	(av, ai), nfb = ops.score_topk_fused(X, index._Etp, 100000, 100, return_fallbacks=True)
	ev, ei = ops.rowwise_topk(A_test, 100)
	torch.cuda.synchronize()
	return dict(ops=ops, A=A_test, A_train=A_train, anc=anc, X=X, index=index, av=av, ai=ai, ev=ev, ei=ei, nfb=int(nfb.item()))


def test_cfg2_fused_topk_properties(cfg2):
	ops, av, ai = cfg2["ops"], cfg2["av"], cfg2["ai"]
	assert cfg2["nfb"] <= 2     # the sampled thresholds held (a wrapped LDS ring, p ~ 1e-10 per lane and window, is repaired exactly)
	assert (av[:, :-1] >= av[:, 1:]).all()                                # sorted by score
	tie = av[:, :-1] == av[:, 1:]
	assert (~tie | (ai[:, :-1] < ai[:, 1:])).all()                        # ties by index
	assert (ai >= 0).all() and (ai < 100000).all()
	assert (torch.sort(ai, dim=1).values[:, 1:] != torch.sort(ai, dim=1).values[:, :-1]).all()   # distinct per query
	rows = torch.arange(0, 10000, 157, device=av.device)                 # 64 sampled queries against the unfused route
	dv, di = ops.score_topk_dense(cfg2["X"][rows], cfg2["index"]._Etp[:100000], 100)
	torch.testing.assert_close(av[rows], dv, rtol=1e-5, atol=1e-5)
	assert (torch.sort(ai[rows], 1).values == torch.sort(di, 1).values).float().mean() > 0.999
	# checksum of checksums: the top-k score mass agrees to fp32 round-off
	assert abs(av[rows].double().sum().item() - dv.double().sum().item()) < 1e-3 * abs(dv.double().sum().item())
	# nothing outside the result beats the k-th score (dense row of S_hat for 8 queries)
	S = ops.gemm(cfg2["X"][rows[:8]], cfg2["index"]._Etp[:100000].t())
	S.scatter_(1, ai[rows[:8]].long(), -float("inf"))
	assert (S.max(dim=1).values <= av[rows[:8], -1] + 1e-6).all()
	# idempotence: the same call again gives the same bits
	av2, ai2 = ops.score_topk_fused(cfg2["X"], cfg2["index"]._Etp, 100000, 100)
	assert torch.equal(av2, av) and torch.equal(ai2, ai)


def test_cfg2_exact_scan_properties(cfg2):
	ops, A, ev, ei = cfg2["ops"], cfg2["A"], cfg2["ev"], cfg2["ei"]
	assert (ev[:, :-1] >= ev[:, 1:]).all() and (ei >= 0).all() and (ei < 100000).all()
	assert torch.equal(torch.gather(A, 1, ei.long()).float(), ev)          # indices point at the reported scores
	rows = torch.arange(3, 10000, 313, device=A.device)
	tv, _ = torch.topk(A[rows].float(), 100, dim=1)
	assert torch.equal(ev[rows], tv)                                        # values bit-exact against torch on a sample
	# top-k of the already selected elements is the selection itself
	sub = torch.gather(A, 1, ei.long())
	sv, si = ops.rowwise_topk(sub.contiguous(), 100)
	assert torch.equal(sv, ev)
	# a k=10 scan is the prefix of the k=100 scan
	v10, i10 = ops.rowwise_topk(A, 10)
	assert torch.equal(v10, ev[:, :10]) and torch.equal(i10, ei[:, :10])


def test_cfg2_recall_properties(cfg2):
	ops, A, ai, ei = cfg2["ops"], cfg2["A"], cfg2["ai"], cfg2["ei"]
	from anncur_amd.retrieval import eval_topk_recall
	cells = [(k, kr) for kr in (10, 50, 100) for k in (1, 10, 50, 100) if k <= kr]
	c = ops.overlap_counts(ei, ai, cells).cpu().numpy()
	by = {cell: c[j] for j, cell in enumerate(cells)}
	assert (by[(10, 50)] <= by[(10, 100)]).all() and (by[(1, 10)] <= by[(1, 100)]).all()      # retrieving more never loses a hit
	assert (by[(10, 100)] <= 10).all() and (by[(100, 100)] <= 100).all()
	self_c = ops.overlap_counts(ei, ei, [(100, 100), (10, 100)]).cpu().numpy()
	assert (self_c[0] == 100).all() and (self_c[1] == 10).all()                                 # exact vs exact: recall 1
	lit = eval_topk_recall(A[:512], ai[:512], [1, 10, 100], [100], literal_rerank=True)
	fast = eval_topk_recall(A[:512], ai[:512], [1, 10, 100], [100], literal_rerank=False)
	assert lit == fast                                                                          # closed form == the reference's scatter + topk re-rank
	assert 0.85 < fast[(100, 100)]["exact_vs_reranked_approx_retvr~common_frac_mean"] < 0.92


def test_cfg2_recall_matches_oracle_on_a_slice(cfg2):
	"""The headline config against the ORACLE, not only against its own unfused route: the oracle's recall (tie-stable statement of the
	reference loop -- bf16 scores are tie-heavy --, fp32 CUR of the same bf16-rounded matrices) on 512 of the 10 000 queries, as
	tests/test_gpu_cfg45.py does for cfg4.  This is bench.py's `recall_gpu_same_queries` vs `recall_cpu_fp32_tie_stable` as a test."""
	from anncur_amd.retrieval import eval_topk_recall
	from oracle import cur_oracle as O
	KEY = "exact_vs_reranked_approx_retvr~common_frac_mean"
	n, anc = 512, cfg2["anc"]
	At = cfg2["A_train"].float().cpu()
	Aq = cfg2["A"][:n].float().cpu()
	ref = O.CURApproxOracle(rows=At, cols=At[:, anc], row_idxs=np.arange(At.shape[0]), col_idxs=anc, approx_preference="rows")
	S_hat = ref.get_complete_row(Aq[:, anc])
	want = O.eval_all_topk_stable(Aq, S_hat, [1, 10, 50, 100], 100)
	got = eval_topk_recall(cfg2["A"][:n], cfg2["ai"][:n], [1, 10, 50, 100], [100], exact=None)
	for k in (1, 10, 50, 100):
		g, w = got[(k, 100)][KEY], want[k][KEY]
		assert g == pytest.approx(w, abs=5e-3), (k, g, w)                   # bf16 item embeddings vs the fp32 oracle
	# the index-builder's route (norm-ordered rows, leading sample, id map) returns the same lists
	ops, index = cfg2["ops"], cfg2["index"]
	av2, ai2 = ops.score_topk_fused(cfg2["X"][:n].contiguous(), index._Etp_sorted, 100000, 100, leading_sample=True, item_ids=index._item_ids)
	assert torch.equal(av2, cfg2["av"][:n])
	assert (torch.sort(ai2, 1).values == torch.sort(cfg2["ai"][:n], 1).values).float().mean() > 0.9999


def test_cfg3_shape_matches_oracle_fp32_and_bf16():
	"""One ZeShEL test domain's shape (yugioh: 3374 mentions x 10031 entities; no dataset here: synthetic stand-in, labelled),
	256 anchor items, 500 train / 2874 test mentions, k = k_retvr = 64."""
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from anncur_amd import harness
	from oracle import cur_oracle as O
	A_train, A_test = O.synth_protocol_b(500, 2874, 10031, rank=64, noise=0.05, seed=3)
	grids = {"top_k_vals": [1, 10, 64], "top_k_retr_vals": [64], "n_ent_anchors_vals": [256]}
	want = O.run_eval_method_cur(A_test, A_train, seed=0, top_k_vals=[1, 10, 64], top_k_retr_vals=[64], n_ent_anchors_vals=[256])
	key = "exact_vs_reranked_approx_retvr~common_frac_mean"
	for dtype, tol in (("fp32", 2e-4), ("bf16", 5e-3)):
		got = harness.run_eval_method_cur(harness.to_device_matrix(A_test, "cuda", dtype), harness.to_device_matrix(A_train, "cuda", dtype), 0, grids)
		for k in (1, 10, 64):
			g = got[f"top_k={k}"]["k_retvr=64"]["anc_n_m=500_anc_n_e=256"][key]
			w = want[f"top_k={k}"]["k_retvr=64"]["anc_n_m=500_anc_n_e=256"][key]
			assert g == pytest.approx(w, abs=tol), (dtype, k, g, w)
