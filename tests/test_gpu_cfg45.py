"""BASELINE.json configs 4 and 5 under parity (needs an MI355X):
  cfg4  the per-GPU shape of the 8-GPU config: 6 250 query rows x 1 000 000 items bf16, 512 anchor items (Kp = 512, one 32-query
        sub-tile per wave, three-stage sweep, 4 096-entry segments), k = k_retvr = 100: size-independent properties at full size
        and the oracle's recall on a 512-query slice;
  cfg5  one ZeShEL test-domain shape (forgotten_realms: 1 200 mentions x 15 603 entities; no dataset here: synthetic stand-in,
        labelled), 1 024 anchor items / 2 048 train mentions, k = k_retvr = 100, fp32 and bf16 against oracle.run_eval_method_cur
        (reference: eval/run_retrieval_eval_wrt_exact_crossenc_w_fixed_train_test_splits.py:286-303 + 399-429);
  plus entry point B with the reference's DEFAULT grids (which contain n_ent_anchors = 0) on a small matrix.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEY = "exact_vs_reranked_approx_retvr~common_frac_mean"


@pytest.fixture(scope="module")
def cfg4():
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from anncur_amd import ops
	from anncur_amd.cur import CURRowIndex
	from anncur_amd.synth import protocol_b
	dev = torch.device("cuda")
	Q, I, Ki, Kq, k = 6250, 1000000, 512, 1024, 100
	A_train, A_test = protocol_b(Kq, Q, I, dev, seed=0)
	anc = sorted(np.random.default_rng(0).choice(I, Ki, replace=False))
	index = CURRowIndex(A_train, anc)
	X = ops.gather_cols(A_test, anc)
	assert index._Etp_sorted.shape[1] == 512 and ops.fused_supported(Q, I, 512, k)
	(av, ai), nfb = ops.score_topk_fused(X, index._Etp_sorted, I, k, return_fallbacks=True, leading_sample=True, item_ids=index._item_ids)
	ev, ei = ops.rowwise_topk(A_test, k)
	torch.cuda.synchronize()
	yield dict(ops=ops, A=A_test, A_train=A_train, X=X, index=index, anc=anc, av=av, ai=ai, ev=ev, ei=ei, nfb=int(nfb.item()), Q=Q, I=I, k=k)
	del A_train, A_test
	torch.cuda.empty_cache()


def test_cfg4_fused_topk_properties(cfg4):
	ops, av, ai, I = cfg4["ops"], cfg4["av"], cfg4["ai"], cfg4["I"]
	assert cfg4["nfb"] <= 2     # (a wrapped LDS ring is rare, p ~ 1e-10 per lane and window, and repaired exactly: never an error)
	assert (av[:, :-1] >= av[:, 1:]).all()                                # sorted by score
	assert (ai >= 0).all() and (ai < I).all()
	srt = torch.sort(ai, dim=1).values
	assert (srt[:, 1:] != srt[:, :-1]).all()                              # distinct per query
	# sampled rows vs the unfused route (dense fp32-accumulated S_hat of the same bf16 operands + exact scan), item order
	rows = torch.arange(0, cfg4["Q"], 199, device=av.device)             # 32 queries
	dv, di = ops.score_topk_dense(cfg4["X"][rows], cfg4["index"]._Etp[:I], 100)
	torch.testing.assert_close(av[rows], dv, rtol=1e-5, atol=1e-5)
	assert (srt[rows] == torch.sort(di, 1).values).float().mean() > 0.999
	# ... and vs fp64 on the host for 4 of them: values within fp32 summation error of the true bf16 x bf16 products
	Xh = cfg4["X"][rows[:4]].double().cpu()
	Eh = cfg4["index"]._Etp[:I].double().cpu()
	S64 = Xh @ Eh.t()
	v64 = torch.topk(S64, 100, dim=1).values
	torch.testing.assert_close(av[rows[:4]].double().cpu(), v64, rtol=2e-5, atol=2e-5)
	# nothing outside the result beats the k-th score
	S = S64.clone()
	S.scatter_(1, ai[rows[:4]].long().cpu(), -float("inf"))
	assert (S.max(dim=1).values <= av[rows[:4], -1].double().cpu() + 1e-5).all()
	# the norm-ordered index and the item-ordered one give the same top-k (values bit for bit; sets up to exact ties)
	av2, ai2 = ops.score_topk_fused(cfg4["X"], cfg4["index"]._Etp, I, 100)
	assert torch.equal(av2, av)
	assert (torch.sort(ai2, 1).values == srt).float().mean() > 0.9999
	# idempotence
	av3, ai3 = ops.score_topk_fused(cfg4["X"], cfg4["index"]._Etp_sorted, I, 100, leading_sample=True, item_ids=cfg4["index"]._item_ids)
	assert torch.equal(av3, av) and torch.equal(ai3, ai)


def test_cfg4_exact_scan_properties(cfg4):
	ops, A, ev, ei, I = cfg4["ops"], cfg4["A"], cfg4["ev"], cfg4["ei"], cfg4["I"]
	assert (ev[:, :-1] >= ev[:, 1:]).all() and (ei >= 0).all() and (ei < I).all()
	assert torch.equal(torch.gather(A, 1, ei.long()).float(), ev)          # indices point at the reported scores
	rows = torch.arange(5, cfg4["Q"], 97, device=A.device)                 # 65 rows of 2 MB each
	tv, _ = torch.topk(A[rows].float(), 100, dim=1)
	assert torch.equal(ev[rows], tv)                                        # bit-exact against torch on a sample
	tie = ev[:, :-1] == ev[:, 1:]
	assert (~tie | (ei[:, :-1] < ei[:, 1:])).all()                          # ties by index


def test_cfg4_recall_matches_oracle_on_a_slice(cfg4):
	"""The oracle's recall (tie-stable statement of the reference loop: bf16 scores are tie-heavy) on 512 of the 6 250 queries."""
	from anncur_amd.retrieval import eval_topk_recall
	from oracle import cur_oracle as O
	n, anc = 512, cfg4["anc"]
	At = cfg4["A_train"].float().cpu()
	Aq = cfg4["A"][:n].float().cpu()
	ref = O.CURApproxOracle(rows=At, cols=At[:, anc], row_idxs=np.arange(At.shape[0]), col_idxs=anc, approx_preference="rows")
	S_hat = ref.get_complete_row(Aq[:, anc])
	want = O.eval_all_topk_stable(Aq, S_hat, [1, 10, 50, 100], 100)
	got = eval_topk_recall(cfg4["A"][:n], cfg4["ai"][:n], [1, 10, 50, 100], [100], exact=None)
	for k in (1, 10, 50, 100):
		g, w = got[(k, 100)][KEY], want[k][KEY]
		assert g == pytest.approx(w, abs=5e-3), (k, g, w)                   # bf16 item embeddings vs the fp32 oracle
	assert 0.8 < got[(100, 100)][KEY] < 0.95


# ------------------------------------------------------------------ cfg5
@pytest.fixture(scope="module")
def cfg5_data():
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from oracle import cur_oracle as O
	A_train, A_test = O.synth_protocol_b(2048, 1200, 15603, rank=64, noise=0.05, seed=5)
	want = O.run_eval_method_cur(A_test, A_train, seed=0, top_k_vals=[1, 10, 50, 100], top_k_retr_vals=[100], n_ent_anchors_vals=[1024])
	return A_train, A_test, want


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-4), ("bf16", 5e-3)])
def test_cfg5_domain_shape_matches_oracle(cfg5_data, dtype, tol):
	from anncur_amd import harness
	A_train, A_test, want = cfg5_data
	grids = {"top_k_vals": [1, 10, 50, 100], "top_k_retr_vals": [100], "n_ent_anchors_vals": [1024]}
	got = harness.run_eval_method_cur(harness.to_device_matrix(A_test, "cuda", dtype), harness.to_device_matrix(A_train, "cuda", dtype), 0, grids)
	for k in (1, 10, 50, 100):
		g = got[f"top_k={k}"]["k_retvr=100"]["anc_n_m=2048_anc_n_e=1024"][KEY]
		w = want[f"top_k={k}"]["k_retvr=100"]["anc_n_m=2048_anc_n_e=1024"][KEY]
		assert g == pytest.approx(w, abs=tol), (dtype, k, g, w)


@pytest.mark.parametrize("Q,I,K,k", [(1200, 15603, 1024, 100), (333, 40000, 768, 64), (2100, 9000, 2048, 10), (64, 70000, 640, 500), (100, 50000, 1024, 1000),
									  (700, 20000, 4096, 100)])
def test_wide_inner_dimension_route_values_and_sets(Q, I, K, k):
	"""Whatever serves K > 512 (cfg5's 1 024 anchors, d = 768 bi-encoder embeddings): values and index sets against fp64 on the
	same bf16 operands.  S_hat must not be materialised by the library GEMM: the route is the K-general fused kernel."""
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from anncur_amd import ops
	g = torch.Generator().manual_seed(Q + K)
	Z = torch.randn(48, I, generator=g)
	X = (torch.randn(Q, 48, generator=g) @ torch.randn(48, K, generator=g) / 7).bfloat16()
	Et = (torch.randn(K, 48, generator=g) @ Z / 7 + 0.05 * torch.randn(K, I, generator=g)).t().contiguous().bfloat16()
	S = X.double() @ Et.double().t()
	wv, wi = torch.topk(S, k, dim=1)
	Xd, Ed = X.cuda(), Et.cuda()
	kp = ops.padded_k(K)
	assert kp is not None and kp >= K and ops.fused_supported(Q, I, kp, k), "K > 512 must be served by a fused kernel"
	(gv, gi), nfb = ops.score_topk_fused(ops.pack_bf16(Xd, kp), ops.pack_bf16(Ed, kp, row_multiple=32), I, k, return_fallbacks=True)
	torch.cuda.synchronize()
	scale = S.abs().max().item()
	assert (gv.double().cpu() - wv).abs().max().item() <= 2e-5 * scale + 1e-6
	assert (gv[:, :-1] >= gv[:, 1:]).all()
	same = (torch.sort(gi.cpu().long(), 1).values == torch.sort(wi, 1).values).float().mean().item()
	assert same > 0.999, same                                              # (fp32 vs fp64 sums may swap a boundary near-tie)
	np.testing.assert_allclose(torch.gather(S, 1, gi.cpu().long()).numpy(), gv.double().cpu().numpy(), rtol=0, atol=2e-5 * scale + 1e-6)


# ------------------------------------------------------------------ entry point B with the reference's default grids
def test_entry_B_default_grids_including_zero_anchors():
	"""splits.py:238-251: the default n_ent_anchors grid starts at int(1 * 0.1) = 0.  The reference then builds an empty index and
	S_hat = 0; here that cell reports the overlap of the exact top-k with items 0..k_retvr-1 (this build's tie order)."""
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from anncur_amd import harness
	from oracle import cur_oracle as O
	A_train, A_test = O.synth_protocol_b(40, 1000, 1200, rank=16, noise=0.05, seed=11)   # 1000 queries: a wrong anchor stream cannot hide in the tolerance
	grids = harness.default_grids_B(1200, "cur")
	assert grids["n_ent_anchors_vals"][0] == 0 and grids["top_k_retr_vals"][0] == 0
	got = harness.run_eval_method_cur(A_test.cuda(), A_train.cuda(), 0, grids)
	cell0 = got["top_k=10"]["k_retvr=100"]["anc_n_m=40_anc_n_e=0"]
	ex = torch.topk(A_test, 10, dim=1).indices
	want0 = float(np.mean([(row < 100).sum().item() for row in ex]))
	assert cell0["exact_vs_reranked_approx_retvr~common_mean"] == pytest.approx(round(want0, 4), abs=1e-4)
	# the oracle over the same anchor-count sequence (one rng stream) for a few cells with at least as many train rows as anchors
	anc_vals = grids["n_ent_anchors_vals"]
	want = O.run_eval_method_cur(A_test, A_train, seed=0, top_k_vals=[1, 10], top_k_retr_vals=[100], n_ent_anchors_vals=[a for a in anc_vals if a > 0],
								 eval_only=(10, 20, 30))
	for n_anc in (10, 20, 30):
		for k in (1, 10):
			g = got[f"top_k={k}"]["k_retvr=100"][f"anc_n_m=40_anc_n_e={n_anc}"][KEY]
			w = want[f"top_k={k}"]["k_retvr=100"][f"anc_n_m=40_anc_n_e={n_anc}"][KEY]
			assert g == pytest.approx(w, abs=5e-3), (n_anc, k, g, w)          # 1000 queries: a swapped boundary near-tie moves the mean by 1e-3 at most
	assert "k_retvr=0" not in got.get("top_k=1", {})                          # k_retvr = 0 < top_k: skipped like the reference


def test_other_device_index_is_validated_and_full_range_is_identity_only():
	"""out-of-range / negative anchor indices follow torch indexing; a permutation is not the identity."""
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from anncur_amd import ops
	from anncur_amd.cur import CURApprox
	A = torch.randn(20, 30).cuda()
	with pytest.raises(IndexError):
		ops.gather_cols(A, [0, 30])
	with pytest.raises(IndexError):
		ops.gather_rows(A, [-21])
	assert torch.equal(ops.gather_cols(A, [-1, 0]), A[:, [-1, 0]]) and torch.equal(ops.gather_rows(A, [-2]), A[[-2]])
	assert ops.gather_cols(A, []).shape == (20, 0) and ops.gather_rows(A, []).shape == (0, 30)
	rows, cols = [1, 4, 7, 9], [0, 3, 5]
	cur = CURApprox(rows=A[rows], cols=A[:, cols], row_idxs=rows, col_idxs=cols, approx_preference="rows")
	perm = [0, 2, 1] + list(range(3, 20))
	full = cur.get_rows(list(range(20)))
	torch.testing.assert_close(cur.get_rows(perm), full[perm])
	dup = list(range(29)) + [0]
	torch.testing.assert_close(cur.get_cols(dup), cur.get_cols(list(range(30)))[:, dup])
	# reference attribute names stay readable, on the device of the caller's tensors
	cpu = CURApprox(rows=A[rows].cpu(), cols=A[:, cols].cpu(), row_idxs=rows, col_idxs=cols, approx_preference="rows")
	assert cpu.U.device.type == "cpu" and cpu.C.device.type == "cpu" and cpu.latent_cols.shape == (3, 30)
	np.testing.assert_allclose(cpu.U.numpy(), np.linalg.pinv(A[rows][:, cols].cpu().numpy()), rtol=1e-5, atol=1e-6)
