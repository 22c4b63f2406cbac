"""Parity of every C-ABI kernel against torch-CPU / the oracle on seeded inputs (needs an MI355X)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from anncur_amd import ops as _ops
	return _ops


def _g(seed):
	return torch.Generator().manual_seed(seed)


def _sets_equal(a, b):
	return (np.sort(np.asarray(a), axis=1) == np.sort(np.asarray(b), axis=1)).all()


# ------------------------------------------------------------------ gemm
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (37, 53, 19), (128, 128, 16), (130, 257, 100), (512, 300, 256)])
@pytest.mark.parametrize("layout", ["nn", "nt", "tn", "tt"])
def test_gemm_fp32_layouts(ops, M, N, K, layout):
	A = torch.randn(M, K, generator=_g(1)); B = torch.randn(K, N, generator=_g(2))
	ref = (A.double() @ B.double()).float()
	a = A.cuda() if layout[0] == "n" else A.t().contiguous().cuda().t()
	b = B.cuda() if layout[1] == "n" else B.t().contiguous().cuda().t()
	out = ops.gemm(a, b).cpu()
	# tolerance: fp32 fmaf chain over K terms
	torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5 * (K ** 0.5))


def test_gemm_bf16_inputs_and_transposed_output(ops):
	A = torch.randn(200, 96, generator=_g(3)).bfloat16(); B = torch.randn(96, 333, generator=_g(4)).bfloat16()
	ref = A.double() @ B.double()
	out = ops.gemm(A.cuda(), B.cuda())
	torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-4)
	outT = torch.empty(333, 200, device="cuda")
	ops.gemm(A.cuda(), B.cuda(), out=outT.t())
	torch.testing.assert_close(outT.t().cpu().double(), ref, rtol=1e-5, atol=1e-4)
	ob = ops.gemm(A.cuda(), B.cuda(), out_dtype=torch.bfloat16).cpu()
	torch.testing.assert_close(ob.float(), ref.float().bfloat16().float(), rtol=1e-2, atol=1e-2)


def test_gemm_matches_cpu_sgemm_within_1e4(ops):
	# the north-star tolerance: reconstructed scores within 1e-4 fp32 rel-tol of the CPU path
	A = torch.randn(300, 256, generator=_g(5)); B = torch.randn(256, 1000, generator=_g(6))
	ref = A @ B
	out = ops.gemm(A.cuda(), B.cuda()).cpu()
	assert ((out - ref).norm() / ref.norm()).item() < 1e-6
	torch.testing.assert_close(out, ref, rtol=1e-4, atol=1e-4)


# ------------------------------------------------------------------ gather / convert
def test_gather_and_convert(ops):
	A = torch.randn(77, 501, generator=_g(7))
	cols = sorted(np.random.default_rng(0).choice(501, 33, replace=False).tolist())
	rows = sorted(np.random.default_rng(1).choice(77, 9, replace=False).tolist())
	assert torch.equal(ops.gather_cols(A.cuda(), cols).cpu(), A[:, cols])
	assert torch.equal(ops.gather_rows(A.cuda(), rows).cpu(), A[rows, :])
	Ab = A.bfloat16()
	assert torch.equal(ops.gather_cols(Ab.cuda(), cols).cpu(), Ab[:, cols])
	assert torch.equal(ops.gather_cols(Ab.cuda(), cols, out_dtype=torch.float32).cpu(), Ab[:, cols].float())
	assert torch.equal(ops.convert(A.cuda(), torch.bfloat16).cpu(), Ab)  # round-to-nearest-even like torch
	assert torch.equal(ops.convert(Ab.cuda(), torch.float32).cpu(), Ab.float())
	p = ops.pack_bf16(A.cuda(), 512, row_multiple=32).cpu()
	assert p.shape == (96, 512) and torch.equal(p[:77, :501], Ab) and p[77:].abs().sum() == 0 and p[:, 501:].abs().sum() == 0


# ------------------------------------------------------------------ exact scan
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Q,I,k", [(3, 1, 1), (5, 7, 7), (9, 100, 10), (17, 5000, 100), (4, 10031, 64), (6, 40000, 1000),
								  (2, 70001, 2048), (33, 4097, 129), (3, 20000, 513),
								  (1001, 316, 17), (7, 1024, 128), (7, 1025, 128), (5, 1000, 129), (130, 64, 64), (9, 65, 1)])   # (short rows: rowwise_topk_short_kernel up to 1024 x 128)
def test_rowwise_topk_matches_torch(ops, dtype, Q, I, k):
	A = torch.randn(Q, I, generator=_g(Q * 1000 + I)).to(dtype)
	# torch ties on bf16 are arbitrary: compare values exactly, and index SETS above the k-th value
	ref_v, ref_i = torch.topk(A.float(), k, dim=1)
	v, i = ops.rowwise_topk(A.cuda(), k)
	v, i = v.cpu(), i.cpu().long()
	assert torch.equal(v, ref_v)
	assert torch.equal(torch.gather(A.float(), 1, i), v)          # indices point at the reported values
	assert all(len(set(r.tolist())) == k for r in i)               # distinct
	if dtype == torch.float32:
		assert _sets_equal(i, ref_i)
	# descending, ties -> smaller index first
	assert ((v[:, :-1] > v[:, 1:]) | ((v[:, :-1] == v[:, 1:]) & (i[:, :-1] < i[:, 1:]))).all()


def test_rowwise_topk_adversarial_and_strided(ops):
	I = 30000
	asc = torch.arange(I, dtype=torch.float32).repeat(3, 1)          # ascending: every element beats the threshold
	v, i = ops.rowwise_topk(asc.cuda(), 100)
	assert torch.equal(i.cpu().long(), torch.arange(I - 1, I - 101, -1).repeat(3, 1))
	const = torch.zeros(2, 5000)                                      # all ties: smallest indices win
	v, i = ops.rowwise_topk(const.cuda(), 64)
	assert torch.equal(i.cpu().long(), torch.arange(64).repeat(2, 1))
	big = torch.randn(8, 9000, generator=_g(11)).cuda()
	view = big[:, 3:8003]                                             # misaligned rows, ld > I
	v, i = ops.rowwise_topk(view, 50)
	rv, ri = torch.topk(view.cpu(), 50, dim=1)
	assert torch.equal(v.cpu(), rv) and torch.equal(i.cpu().long(), ri)
	ninf = torch.full((1, 300), -float("inf")); ninf[0, 5] = 1.0
	v, i = ops.rowwise_topk(ninf.cuda(), 3)
	assert i[0, 0].item() == 5 and v[0, 1].item() == -float("inf")


@pytest.mark.parametrize("I", [(1 << 30) + 4099, (1 << 30) - 4093])
def test_rowwise_topk_rows_of_2gb_and_more_take_the_pointer_loads(ops, I):
	"""The stream loop issues raw buffer loads with 32-bit offsets for rows under 2 GB; longer rows keep the 64-bit pointer form
	(csrc/topk.hip: `buf`).  One bf16 row just past 2 GB and one just under it (the buffer form at its largest offsets), built on the
	device: small noise with 24 distinct planted values, the last of them in the row's final, partial block."""
	A = torch.empty((1, I), dtype=torch.bfloat16, device="cuda")
	A.normal_(0.0, 0.01, generator=torch.Generator(device="cuda").manual_seed(5))
	pos = torch.tensor([7, 4096 * 3 + 1, 1 << 20, (1 << 29) + 13, I - 4000, I - 513, I - 2] + [97 * 1000003 * (j + 1) % I for j in range(17)], dtype=torch.long)
	assert len(set(pos.tolist())) == 24
	vals = torch.arange(24, dtype=torch.float32).mul(0.5).add(10.0).to(torch.bfloat16)    # 10.0, 10.5, ...: distinct in bf16
	A[0, pos.cuda()] = vals.cuda()
	v, i = ops.rowwise_topk(A, 24)
	order = torch.argsort(vals.float(), descending=True)
	assert torch.equal(i.cpu().long()[0], pos[order]) and torch.equal(v.cpu()[0], vals.float()[order])
	v10, i10 = ops.rowwise_topk(A, 10)
	assert torch.equal(i10.cpu().long()[0], pos[order][:10])
	del A


# ------------------------------------------------------------------ rerank + overlap vs the oracle loop
def test_rerank_and_overlap_match_reference_loop(ops):
	from oracle import cur_oracle as O
	g = _g(21)
	A = torch.randn(40, 3000, generator=g)
	S = A + 0.5 * torch.randn(40, 3000, generator=g)
	(ex_i, ex_s), (ap_i, ap_s), (rr_i, rr_s) = O.per_query_loop(A, S, 20, 100)
	ex = ops.rowwise_topk(A.cuda(), 20)
	ap = ops.rowwise_topk(S.cuda(), 100)
	assert np.array_equal(ex.indices.cpu().numpy(), ex_i) and np.array_equal(ap.indices.cpu().numpy(), ap_i)
	rr = ops.rerank(A.cuda(), ap.indices, 100, 20)
	assert np.array_equal(rr.indices.cpu().numpy(), rr_i) and np.array_equal(rr.values.cpu().numpy(), rr_s)
	# prefix re-rank in place (k_retvr = 37 of the 100 retrieved)
	(_, _), (_, _), (rr37_i, _) = O.per_query_loop(A, S, 20, 37)
	assert np.array_equal(ops.rerank(A.cuda(), ap.indices, 37, 20).indices.cpu().numpy(), rr37_i)
	pairs = [(1, 1), (10, 10), (20, 20), (5, 20)]
	cnt = ops.overlap_counts(ex.indices, rr.indices, pairs).cpu().numpy()
	for p, (ka, kb) in enumerate(pairs):
		want = [len(set(ex_i[q, :ka]) & set(rr_i[q, :kb])) for q in range(40)]
		assert cnt[p].tolist() == want
	# closed form used by the sweep: exact[:k] & rerank[:k] == exact[:k] & approx[:k_retvr]
	cf = ops.overlap_counts(ex.indices, ap.indices, [(10, 100), (20, 100)]).cpu().numpy()
	lit = ops.overlap_counts(ex.indices, rr.indices, [(10, 10), (20, 20)]).cpu().numpy()
	assert np.array_equal(cf, lit)


# ------------------------------------------------------------------ fused score + top-k
def _fused_case(ops, Q, I, K, k, seed, noise=0.05, rank=32):
	g = _g(seed)
	Z = torch.randn(rank, I, generator=g)
	X = (torch.randn(Q, K, generator=g)).bfloat16()
	E = (torch.randn(K, rank, generator=g) @ Z / rank ** 0.5 + noise * torch.randn(K, I, generator=g)).bfloat16()
	Kp = ops.padded_k(K)
	Xp = ops.pack_bf16(X.cuda(), Kp)
	Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	return X, E, Xp, Etp


@pytest.mark.parametrize("Q,I,K,k", [(300, 40000, 64, 10), (257, 65536, 128, 100), (100, 50007, 256, 100), (64, 70000, 256, 1),
									  (130, 33000, 512, 64), (50, 131072, 200, 500), (20, 300000, 100, 1000)])
def test_fused_score_topk_matches_dense_and_cpu(ops, Q, I, K, k):
	X, E, Xp, Etp = _fused_case(ops, Q, I, K, k, seed=Q + I + K + k)
	Kp = Xp.shape[1]
	assert ops.fused_supported(Q, I, Kp, k)
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True)
	torch.cuda.synchronize()
	assert nfb.item() == 0
	# (a) against the unfused device route (same bf16 operands, fp32 fmaf sums)
	dv, di = ops.score_topk_dense(Xp, Etp[:I], k)
	torch.testing.assert_close(v.cpu(), dv.cpu(), rtol=1e-4, atol=1e-4)
	# (b) against torch on the CPU in fp64: reconstructed scores within 1e-4, index sets identical where separated
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	torch.testing.assert_close(v.cpu().double(), rv, rtol=1e-4, atol=1e-4)
	got = i.cpu().long()
	assert (got >= 0).all() and (got < I).all()
	torch.testing.assert_close(torch.gather(S, 1, got), v.cpu().double(), rtol=1e-4, atol=1e-4)
	kth = rv[:, -1:]
	gap_ok = (S - kth).abs().gt(1e-3 * S.abs().max()).all(dim=1) | True
	same = [set(a.tolist()) == set(b.tolist()) for a, b in zip(got, ri)]
	# rows whose k-th / (k+1)-th scores are closer than fp32 round-off may swap one boundary element
	kp1 = torch.topk(S, min(k + 1, I), dim=1).values
	close = (kp1[:, k - 1] - kp1[:, -1]).abs() < 1e-5 * S.abs().max() if k < I else torch.zeros(Q, dtype=torch.bool)
	assert all(s or c for s, c in zip(same, close.tolist()))


@pytest.mark.parametrize("Q,I,K,k", [(300, 40000, 64, 10), (257, 65536, 128, 100), (1000, 50007, 256, 100), (64, 70001, 200, 1), (2049, 123457, 256, 128), (130, 200000, 256, 500)])
def test_eval_fused_equals_the_two_kernel_route(ops, Q, I, K, k):
	"""anncur_eval_fused: ONE sweep yields the top-k of S_hat AND the per-row sum (S_hat - A)^2, sum A^2 of entry
	point A's cell (crossenc.py:106,146-147).  Against the two-kernel route: top-k values bit for bit with the 32x32x16 sweep body
	(ANNCUR_TOPK_MFMA32: the same MFMAs in the same order), index sets identical, no repaired query; the two sums within fp32
	summation order (different partition of the tiles over workgroups, atomics) of anncur_approx_error_packed, and within 1e-4 of fp64
	on sampled rows.  Ragged Q / I (a partial last tile: candidates from the sweep, error terms from the strided kernel)."""
	X, E, Xp, Etp = _fused_case(ops, Q, I, K, k, seed=Q + I + K + k)
	Kp = Xp.shape[1]
	g = _g(Q + I)
	A = (X.float() @ E.float() + 0.3 * torch.randn(Q, I, generator=g)).bfloat16().cuda()
	if I % 8:   # rows must be 16-byte aligned: a padded pitch
		Ap = torch.zeros((Q, -(-I // 8) * 8), dtype=torch.bfloat16, device="cuda"); Ap[:, :I] = A; A = Ap[:, :I]
	assert ops.eval_fused_ok(Kp, A, Q, I, k)
	(tk, err, nrm, nfb) = ops.eval_fused(Xp, Etp, A, I, k, return_fallbacks=True)
	(v2, i2), nfb2 = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, mfma32=True)
	err2, nrm2 = ops.approx_error_packed(Xp, Etp, A, I)
	torch.cuda.synchronize()
	assert nfb.item() == 0 and nfb2.item() == 0
	assert torch.equal(tk.values, v2)
	assert torch.equal(torch.sort(tk.indices, 1).values, torch.sort(i2, 1).values)
	torch.testing.assert_close(err, err2, rtol=2e-5, atol=1e-6)
	torch.testing.assert_close(nrm, nrm2, rtol=2e-5, atol=1e-6)
	rows = torch.arange(0, Q, max(1, Q // 16))
	S64 = X[rows].double() @ E.double()
	A64 = A[rows.cuda()].double().cpu()
	torch.testing.assert_close(err[rows.cuda()].double().cpu(), ((S64 - A64) ** 2).sum(1), rtol=1e-4, atol=1e-6)
	torch.testing.assert_close(nrm[rows.cuda()].double().cpu(), (A64 ** 2).sum(1), rtol=1e-4, atol=1e-6)
	plan_codes = ops.fused_plan(Q, I, Kp, k)   # (the public plan is untouched by the new entry point)
	assert 6 not in plan_codes["stage_pred"]
	# round 5, anncur_eval_fused_ex: the first threshold sampled from the LEADING tiles of a norm-ordered copy of E^T (CURApprox's hint copy) --
	# and from the worst copy there is, norms ascending: the same results bit for bit (any subset of the items yields a valid threshold)
	order = torch.argsort(E.float().norm(dim=0), descending=True)
	for perm in (order, order.flip(0)):
		hint = torch.zeros_like(Etp); hint[:I] = Etp[:I][perm.cuda()]
		(tk3, err3, nrm3, nfb3) = ops.eval_fused(Xp, Etp, A, I, k, return_fallbacks=True, hint=hint)
		torch.cuda.synchronize()
		assert torch.equal(tk3.values, tk.values) and torch.equal(torch.sort(tk3.indices, 1).values, torch.sort(tk.indices, 1).values)
		torch.testing.assert_close(err3, err, rtol=2e-5, atol=1e-6)
		torch.testing.assert_close(nrm3, nrm, rtol=2e-5, atol=1e-6)
	with pytest.raises(ValueError):
		ops.eval_fused(Xp, Etp, A, I, k, hint=Etp[:32])
	# unsupported operands are refused loudly, not routed elsewhere
	with pytest.raises(Exception):
		ops.eval_fused(Xp, Etp, A.float(), I, k)


@pytest.mark.parametrize("kind", ["bench_like", "ascending_norms", "identical_queries", "integer_ties", "flat"])
def test_fused_threshold_ladder_equals_the_staged_sweep(ops, kind):
	"""Round 5 (csrc/score16.hpp): the default 16x16x32 body sweeps in ONE launch and moves its thresholds up a ladder of levels from device-wide
	counts of the candidates kept so far.  Any timing of the counts must give THE top-k: against the staged sweep of rounds 1-4
	(ANNCUR_TOPK_STAGED) bit for bit, and against the fp64 order.  Shapes of the data that stress it: item rows whose norms ASCEND (every later
	tile beats the running threshold: the opposite of the index's order, segments overflow into the repair path), all queries identical (every
	workgroup counts into the same few words), exactly representable integers with ties at every level, a flat ladder (all scores equal)."""
	Q, I, K, k = 700, 60000, 256, 100
	g = _g(123)
	if kind == "integer_ties":
		X = torch.randint(0, 3, (Q, K), generator=g).float().bfloat16()
		E = torch.randint(-2, 3, (K, I), generator=g).float().bfloat16()
	elif kind == "flat":
		X = torch.ones(Q, K).bfloat16(); E = torch.ones(K, I).bfloat16()
	else:
		Z = torch.randn(24, I, generator=g)
		X = torch.randn(Q, K, generator=g).bfloat16()
		E = (torch.randn(K, 24, generator=g) @ Z / 24 ** 0.5 + 0.05 * torch.randn(K, I, generator=g))
		if kind == "ascending_norms": E = E * torch.linspace(0.2, 2.0, I)[None, :]
		if kind == "identical_queries": X = X[:1].expand(Q, K).contiguous()
		E = E.bfloat16()
	Kp = ops.padded_k(K)
	Xp = ops.pack_bf16(X.cuda(), Kp); Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	plan, plan_s = ops.fused_plan(Q, I, Kp, k), ops.fused_plan(Q, I, Kp, k, staged=True)
	assert plan["ladder"] and plan["n_stages"] == 1 and not plan_s["ladder"] and plan_s["n_stages"] >= 2
	for _ in range(3):   # (a dependence on the timing of the counts would come and go)
		a = ops.score_topk_fused(Xp, Etp, I, k)
		torch.cuda.synchronize()
		b = ops.score_topk_fused(Xp, Etp, I, k, staged=True)
		torch.cuda.synchronize()
		assert torch.equal(a.values, b.values) and torch.equal(a.indices, b.indices), kind
	S = X.double() @ E.double()
	order = torch.argsort(S, dim=1, descending=True, stable=True)[:, :k]
	if kind in ("integer_ties", "flat"):    # exactly representable: THE top-k under the defined order
		assert torch.equal(a.indices.cpu().long(), order) and torch.equal(a.values.cpu().double(), torch.gather(S, 1, order))
	else:
		rv = torch.gather(S, 1, order)
		assert (a.values.cpu().double() - rv).abs().max() <= 1e-4 * float(S.abs().max())
		assert (torch.gather(S, 1, a.indices.cpu().long()) - a.values.cpu().double()).abs().max() <= 1e-4 * float(S.abs().max())
	if kind == "bench_like":   # ... and it does what it is for: fewer candidates kept than the staged sweep keeps
		from anncur_amd import _lib
		ws = ops._Workspace.get(_lib.load().anncur_score_topk_workspace_bytes(Q, I, Kp, k), Xp.device)
		ops.score_topk_fused(Xp, Etp, I, k); n_lad = ops.fused_survivors(ws, Q, I, Kp, k)
		ops.score_topk_fused(Xp, Etp, I, k, staged=True); n_stg = ops.fused_survivors(ws, Q, I, Kp, k, staged=True)
		assert k <= n_lad < n_stg, (n_lad, n_stg)


def test_error_kernels_with_a_row_pitch_beyond_the_32_bit_tile_offsets(ops):
	"""error_lds_kernel / evalf_kernel address their tile of the exact matrix as a uniform base + a 32-bit byte offset (row within the row
	block x pitch x 2); from a pitch of 8 421 504 elements on, row 255's offset wraps and the DMA would read the wrong rows -- silently.  The routes
	that use those kernels are now taken only while the offsets fit: with a pitch of 8.6 M elements eval_fused is refused (the harness then takes
	the two-kernel route) and approx_error_packed runs error_kernel's 64-bit row pointers -- sums equal to the compact matrix'."""
	Q, I, K, k = 300, 40000, 64, 10
	X, E, Xp, Etp = _fused_case(ops, Q, I, K, k, seed=77)
	Kp = Xp.shape[1]
	A = (X.float() @ E.float() + 0.3 * torch.randn(Q, I, generator=_g(5))).bfloat16().cuda()
	pitch = 8_600_000                                                  # > (2^32 - 64) / (255 * 2): 5.2 GB of bf16 for 300 rows
	buf = torch.empty((Q, pitch), dtype=torch.bfloat16, device="cuda")
	Aw = buf[:, :I]
	Aw.copy_(A)
	assert Aw.stride(0) == pitch and ops.eval_fused_ok(Kp, A, Q, I, k) and not ops.eval_fused_ok(Kp, Aw, Q, I, k)
	with pytest.raises(Exception):
		ops.eval_fused(Xp, Etp, Aw, I, k)
	e0, n0 = ops.approx_error_packed(Xp, Etp, A, I)
	e1, n1 = ops.approx_error_packed(Xp, Etp, Aw, I)
	torch.cuda.synchronize()
	torch.testing.assert_close(e1, e0, rtol=2e-5, atol=1e-6)
	torch.testing.assert_close(n1, n0, rtol=2e-5, atol=1e-6)
	rows = torch.tensor([0, 1, 130, 255, 256, 299])                     # (row 255 of a row block: the offset that wrapped)
	S64 = X[rows].double() @ E.double()
	torch.testing.assert_close(e1[rows.cuda()].double().cpu(), ((S64 - A[rows.cuda()].double().cpu()) ** 2).sum(1), rtol=1e-4, atol=1e-6)
	del buf


@pytest.mark.parametrize("Q,I,K,k", [(1000, 40000, 256, 10), (777, 50001, 200, 64), (513, 30000, 128, 32), (5, 20000, 256, 7), (2049, 123457, 256, 128), (4100, 200000, 128, 100)])
def test_fused_ring_body_equals_the_barrier_body(ops, Q, I, K, k):
	"""ANNCUR_TOPK_RING (round 4, csrc/score16r.hpp): 8-wave workgroups of 512 queries, the item tiles through a ring of four LDS slots
	synchronised by per-wave landed / done counters (no barrier in the tile loop), the tile sequence published by wave 0.  The tile
	arithmetic is the barrier body's instruction for instruction: values bit for bit, index sets identical, no repaired query, no spin
	timeout (a timeout adds 2^30 to the fallback counter).  Ragged row blocks (Q not a multiple of 512), a partial last tile, both
	item orders, one- and two-stage plans."""
	from anncur_amd import _lib
	if not _lib.IS_EXPERIMENTS_LIB:
		pytest.skip("the tile-ring body is compiled into the experiments library only since round 5 (ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so)")
	X, E, Xp, Etp = _fused_case(ops, Q, I, K, k, seed=Q + I + K + k)
	Kp = Xp.shape[1]
	plan = ops.fused_plan(Q, I, Kp, k, ring=True)
	assert all(b == 5 for b in plan["stage_pred"]) and all(b == 2 for b in ops.fused_plan(Q, I, Kp, k)["stage_pred"])
	(v0, i0), nfb0 = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True)
	for _ in range(3):   # (a protocol race would come and go)
		(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, ring=True)
		torch.cuda.synchronize()
		assert nfb.item() == 0 and nfb0.item() == 0
		assert torch.equal(v, v0)
		assert torch.equal(torch.sort(i, 1).values, torch.sort(i0, 1).values)
	# with the index builder's hints (norm-ordered rows, leading sample, id map)
	from anncur_amd.cur import _norm_sorted_pack
	Ets, ids = _norm_sorted_pack(E.t().contiguous().float().cuda(), Kp)
	a = ops.score_topk_fused(Xp, Ets, I, k, leading_sample=True, item_ids=ids, ring=True)
	b = ops.score_topk_fused(Xp, Ets, I, k, leading_sample=True, item_ids=ids)
	assert torch.equal(a.values, b.values) and torch.equal(torch.sort(a.indices, 1).values, torch.sort(b.indices, 1).values)


@pytest.mark.parametrize("Q,I,K,k", [(300, 40000, 64, 10), (257, 65536, 128, 100), (1000, 50007, 256, 100), (64, 70000, 256, 1), (50, 131072, 200, 500)])
def test_fused_mfma16_sweep_gives_the_same_topk(ops, Q, I, K, k):
	"""The 16x16x32 sweep (ANNCUR_TOPK_MFMA16, score16.hpp): other lane <-> (query, item) map, survivors through one queue per wave
	into ONE segment per query and split.  Same products, same fp32 sums per output element -> values bit for bit, sets identical."""
	X, E, Xp, Etp = _fused_case(ops, Q, I, K, k, seed=Q + I + K + k)
	Kp = Xp.shape[1]
	# the flags must reach the plan (round 2 dropped them in ops.py and this test compared the default kernel with itself)
	assert ops.fused_plan(Q, I, Kp, k, mfma32=True)["lg"] == 2 and ops.fused_plan(Q, I, Kp, k)["QT"] == 2
	assert ops.fused_plan(Q, I, Kp, k, mfma16=True)["lg"] == 1
	assert ops.fused_plan(Q, I, Kp, k, qt1=True)["QT"] == (1 if Kp >= 128 else 2)
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, mfma32=True)
	# the default plan (the 16x16x32 body for k <= 128, the 32x32x16 body above) -- same answer
	(vd, idd), nfbd = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True)
	torch.cuda.synchronize()
	assert nfbd.item() == 0
	torch.testing.assert_close(vd, v, rtol=1e-6, atol=1e-6)
	assert (torch.sort(idd, 1).values == torch.sort(i, 1).values).float().mean() > 0.9995
	(v16, i16), nfb16 = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, mfma16=True)
	torch.cuda.synchronize()
	assert nfb.item() == 0 and nfb16.item() == 0
	torch.testing.assert_close(v16, v, rtol=1e-6, atol=1e-6)     # (the MFMA shapes may associate the k-sum differently)
	assert (torch.sort(i16, 1).values == torch.sort(i, 1).values).float().mean() > 0.9995
	# ANNCUR_TOPK_QT1 (Kp = 128 / 256): one sub-tile per wave, cross-tile pipeline, three workgroups per CU -- same MFMA, same sums
	(v1, i1), nfb1 = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, qt1=True)
	torch.cuda.synchronize()
	assert nfb1.item() == 0 and torch.equal(v1, v)
	assert (torch.sort(i1, 1).values == torch.sort(i, 1).values).float().mean() > 0.9995


@pytest.mark.parametrize("Q,I,K,k", [(257, 70000, 512, 100), (1000, 60007, 400, 10), (130, 131072, 512, 500)])
def test_fused_kp512_queue_body_equals_the_ring_body(ops, Q, I, K, k):
	"""Kp = 512 has two candidate paths: one queue per wave with the dynamic tile schedule (score_q16.hpp: 16x16x32 MFMAs, the default;
	score_q1.hpp: the same on 32x32x16 MFMAs, experiments build) and per-lane rings with static shares on 32x32x16 MFMAs
	(ANNCUR_TOPK_MFMA32).  Same products, fp32 sums that may associate differently between the MFMA shapes: values to 1e-6, sets identical."""
	X, E, Xp, Etp = _fused_case(ops, Q, I, K, k, seed=Q + I + K + k)
	assert Xp.shape[1] == 512
	pq, pr = ops.fused_plan(Q, I, 512, k), ops.fused_plan(Q, I, 512, k, mfma32=True)
	assert pq["lg"] == 1 and all(b == 4 for b in pq["stage_pred"]) and pr["lg"] == 2 and all(b in (0, 1) for b in pr["stage_pred"])
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True)
	(vr, ir), nfbr = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, mfma32=True)
	torch.cuda.synchronize()
	assert nfb.item() == 0 and nfbr.item() == 0
	torch.testing.assert_close(v, vr, rtol=1e-6, atol=1e-6)
	assert (torch.sort(i, 1).values == torch.sort(ir, 1).values).float().mean() > 0.9995


def test_fused_kp512_dense_block_drains_inside_the_tile_function(ops):
	# a contiguous block of items far above the rest: nearly every compare of those tiles passes, the wave queues (448 entries) drain
	# inside the tile function; segments may overflow -> exact repair through the chunk-owner map
	Q, I, K, k = 300, 60000, 512, 100
	g = _g(512512)
	X = torch.randn(Q, K, generator=g).abs().bfloat16()
	E = (0.01 * torch.randn(K, I, generator=g))
	E[:, 20000:24000] += 1.0
	E = E.bfloat16()
	Xp = ops.pack_bf16(X.cuda(), 512); Etp = ops.pack_bf16(E.t().contiguous().cuda(), 512, row_multiple=32)
	assert ops.fused_plan(Q, I, 512, k)["lg"] == 1
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True)
	torch.cuda.synchronize()
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	torch.testing.assert_close(v.cpu().double(), rv, rtol=1e-4, atol=1e-4)
	torch.testing.assert_close(torch.gather(S, 1, i.cpu().long()), v.cpu().double(), rtol=1e-4, atol=1e-4)
	assert all(len(set(r.tolist())) == k for r in i.cpu())


def test_fused_mfma16_dense_block_and_segment_overflow_stay_exact(ops):
	# a contiguous block of items far above the rest: the wave queues drain inside the tile function, segments may overflow -> exact repair
	Q, I, K, k = 300, 80000, 128, 100
	g = _g(4242)
	X = torch.randn(Q, K, generator=g).abs().bfloat16()
	E = (0.01 * torch.randn(K, I, generator=g))
	E[:, 30000:36000] += 1.0
	E = E.bfloat16()
	Xp = ops.pack_bf16(X.cuda(), 128); Etp = ops.pack_bf16(E.t().contiguous().cuda(), 128, row_multiple=32)
	assert ops.fused_plan(Q, I, 128, k, mfma16=True)["lg"] == 1
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, mfma16=True)
	torch.cuda.synchronize()
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	torch.testing.assert_close(v.cpu().double(), rv, rtol=1e-4, atol=1e-4)
	# (until r2b this input also needed repairs; a stage that drains every tile now hands wrapped rings over as raw tiles and the
	#  segments are large enough: exact without one)
	assert nfb.item() >= 0
	torch.testing.assert_close(torch.gather(S, 1, i.cpu().long()), v.cpu().double(), rtol=1e-4, atol=1e-4)


def test_fused_regression_dense_first_stage_found_by_fuzzing(ops):
	"""Round-2 fuzz find: Q=4, I=12479, K=241, k=18, rank-2 scores (half of the first stage's compares hit).  The plan briefly sent
	such stages to the branch-free inline-asm filter, whose asm read an accumulator register the compiler had not ordered behind its
	MFMA: one survivor (query 3, item 5760) was lost.  The product now always uses the compiler-generated compare."""
	g = torch.Generator().manual_seed(0)
	Q, I, K, k, rank = 4, 12479, 241, 18, 2
	X = torch.randn(Q, K, generator=g).bfloat16()
	E = (torch.randn(K, rank, generator=g) @ torch.randn(rank, I, generator=g) / rank ** 0.5 + 0.0 * torch.randn(K, I, generator=g)).bfloat16()
	Kp = ops.padded_k(K)
	Xp = ops.pack_bf16(X.cuda(), Kp); Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	for kw in ({}, {"mfma32": True}, {"mfma16": True}, {"qt1": True}):
		plan = ops.fused_plan(Q, I, Kp, k, **kw)
		assert (plan["lg"], plan["QT"]) in {(): ((1, 2),), ("mfma32",): ((2, 2),), ("mfma16",): ((1, 2),), ("qt1",): ((2, 1),)}[tuple(kw)], (kw, plan)
		v, i = ops.score_topk_fused(Xp, Etp, I, k, **kw)
		assert (v.cpu().double() - rv).abs().max() <= 1e-4 * float(S.abs().max())
		assert all(set(a.tolist()) == set(b.tolist()) for a, b in zip(i.cpu(), ri)), kw


def test_fused_overflow_fallback_is_exact(ops):
	# ascending scores along the item axis and the threshold sampled from the LEADING tiles (the lowest scores): nearly every element
	# beats it, every candidate segment overflows, the in-call repair recomputes the queries exactly
	Q, I, K, k = 40, 60000, 64, 50
	X = torch.ones(Q, K).bfloat16()
	E = (torch.arange(I, dtype=torch.float32) / 256).floor().repeat(K, 1).bfloat16() / K
	Xp = ops.pack_bf16(X.cuda(), 64); Etp = ops.pack_bf16(E.t().contiguous().cuda(), 64, row_multiple=32)
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, leading_sample=True)
	torch.cuda.synchronize()
	S = X.double() @ E.double()
	rv, _ = torch.topk(S, k, dim=1)
	torch.testing.assert_close(v.cpu().double(), rv, rtol=1e-5, atol=1e-5)
	assert nfb.item() > 0
	# ties -> smallest indices of the top plateau
	top = S[0].max()
	first = int((S[0] == top).nonzero()[0])
	assert i[0, 0].item() == first
	# the same input with the strided sample needs no repair at all (interleaved splits share the high-scoring end of the item axis)
	(v2, i2), nfb2 = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True)
	torch.testing.assert_close(v2.cpu().double(), rv, rtol=1e-5, atol=1e-5)
	assert i2[0, 0].item() == first


def test_fused_local_overflow_is_repaired_exactly(ops):
	# STATIC tile shares (the one-sub-tile sweep, ANNCUR_TOPK_QT1, keeps them): the tiles ONE item split sweeps (tile index = 3 mod S: the
	# splits interleave the tiles) score far above the rest for every query: that split's candidate segments overflow, only it is
	# recomputed (select_candidates_kernel), the result is exact
	Q, I, K, k = 3000, 80000, 128, 100
	plan = ops.fused_plan(Q, I, 128, k, qt1=True)
	S_ = plan["splits"]
	assert S_ >= 8 and plan["QT"] == 1
	g = _g(4242)
	X = (1.0 + 0.1 * torch.randn(Q, K, generator=g)).bfloat16()
	E = 0.05 * torch.randn(K, I, generator=g)
	tile = torch.arange(I) // 32
	E[:, (tile % S_) == 3] += 0.5
	E = E.bfloat16()
	Xp = ops.pack_bf16(X.cuda(), 128); Etp = ops.pack_bf16(E.t().contiguous().cuda(), 128, row_multiple=32)
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, qt1=True)
	torch.cuda.synchronize()
	assert nfb.item() > 0
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	torch.testing.assert_close(v.cpu().double(), rv, rtol=1e-4, atol=1e-4)
	got = i.cpu().long()
	assert (((got // 32) % S_) == 3).all()                     # every result comes from the boosted tiles
	assert all(set(a.tolist()) == set(b.tolist()) for a, b in zip(got[::37], ri[::37]))


@pytest.mark.parametrize("body", ["mfma32", "default"])
def test_fused_local_overflow_under_the_dynamic_tile_schedule(ops, body):
	"""The sweep draws its tiles as tickets (chunks of 4 tiles per query row block): which workgroup sweeps which tiles is decided at run
	time, and the repair path finds a split's tiles in the chunk-owner map the sweep leaves behind.  Forced here: the first 256 items
	(the first two chunks, which ONE workgroup of every row block draws together at its start) score far above the rest for every
	query -- 256 survivors per query in that workgroup's segments (32x32x16 body: 128 per lane half against a capacity of 64, and rings
	that wrap; 16x16x32 body, the default at this k: 256 against its one segment per split): the segments overflow, the workgroup's
	chunks are recomputed from the owner map, the result is exact."""
	Q, I, K, k = 3000, 80000, 128, 10
	kw = {"mfma32": True} if body == "mfma32" else {}
	plan = ops.fused_plan(Q, I, 128, k, **kw)
	assert plan["QT"] == 2 and plan["lg"] == (2 if body == "mfma32" else 1) and plan["segment_capacity"] * plan["lg"] < 256
	g = _g(777)
	X = (1.0 + 0.1 * torch.randn(Q, K, generator=g)).bfloat16()
	E = 0.05 * torch.randn(K, I, generator=g)
	E[:, :256] += 0.5
	E = E.bfloat16()
	Xp = ops.pack_bf16(X.cuda(), 128); Etp = ops.pack_bf16(E.t().contiguous().cuda(), 128, row_multiple=32)
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, **kw)
	torch.cuda.synchronize()
	assert nfb.item() > 0                                      # the local repair ran
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	torch.testing.assert_close(v.cpu().double(), rv, rtol=1e-4, atol=1e-4)
	got = i.cpu().long()
	assert (got < 256).all()
	assert all(set(a.tolist()) == set(b.tolist()) for a, b in zip(got[::37], ri[::37]))
	# and again: the owner map of the previous call must not leak into this one (another assignment, same answer)
	(v2, i2), _ = ops.score_topk_fused(Xp, Etp, I, k, return_fallbacks=True, **kw)
	assert torch.equal(v2, v) and torch.equal(torch.sort(i2, 1).values, torch.sort(i, 1).values)


@pytest.mark.parametrize("Q,I,K,k,kr,dt", [(700, 40000, 64, 10, 100, "bf16"), (3300, 50007, 256, 100, 100, "bf16"), (7000, 33000, 128, 64, 200, "f32"),
										   (257, 70000, 512, 100, 100, "bf16"), (500, 20000, 1024, 50, 100, "bf16")])
def test_eval_topk_equals_the_two_separate_calls(ops, Q, I, K, k, kr, dt):
	"""anncur_eval_topk = anncur_rowwise_topk + anncur_score_topk_ex with the scan's row chunks forked onto a second stream between the
	retrieval's launches: the SAME kernels on the same inputs, so both results are bit for bit those of the separate calls -- with one
	chunk (Q below a round of the scan), several chunks, the wide kernel, fp32 scores, serial mode, and replayed from a captured graph."""
	g = _g(Q + I + K)
	X, E, Xp, Etp = _fused_case(ops, Q, I, K, kr, seed=Q + I + K + kr)
	A = torch.randn(Q, I, generator=g)
	A = (A.bfloat16() if dt == "bf16" else A).cuda()
	want_e = ops.rowwise_topk(A, k)
	want_a = ops.score_topk_fused(Xp, Etp, I, kr)
	torch.cuda.synchronize()
	for serial in (False, True):
		e, a = ops.eval_topk(A, k, Xp, Etp, I, kr, serial=serial)
		torch.cuda.synchronize()
		assert torch.equal(e.values, want_e.values) and torch.equal(e.indices, want_e.indices), serial
		assert torch.equal(a.values, want_a.values) and torch.equal(a.indices, want_a.indices), serial
	# captured: the fork / join events become graph edges, the auxiliary stream joins the capture and leaves it again
	ws = ops.fused_workspace(Q, I, Xp.shape[1], kr, Xp.device)
	ops.eval_topk(A, k, Xp, Etp, I, kr, workspace=ws); torch.cuda.synchronize()
	gr = torch.cuda.CUDAGraph()
	with torch.cuda.graph(gr, capture_error_mode="thread_local"):
		e, a = ops.eval_topk(A, k, Xp, Etp, I, kr, workspace=ws)
	for _ in range(2):
		e.values.zero_(); a.indices.zero_()
		gr.replay(); torch.cuda.synchronize()
		assert torch.equal(e.values, want_e.values) and torch.equal(e.indices, want_e.indices)
		assert torch.equal(a.values, want_a.values) and torch.equal(a.indices, want_a.indices)


def test_fused_unsupported_shapes_raise(ops):
	from anncur_amd._lib import AnncurHipError
	assert not ops.fused_supported(1000, 5000, 64, 10)          # too few items for a sampled threshold
	Xp = torch.zeros(8, 64, dtype=torch.bfloat16, device="cuda"); Etp = torch.zeros(5024, 64, dtype=torch.bfloat16, device="cuda")
	with pytest.raises(AnncurHipError):
		ops.score_topk_fused(Xp, Etp, 5000, 10)
	with pytest.raises(AnncurHipError):
		ops.rowwise_topk(torch.zeros(4, 10, device="cuda"), 11)  # k > I
	with pytest.raises(AnncurHipError):
		ops.rowwise_topk(torch.zeros(4, 10), 2)                  # CPU tensor: no fallback


# ------------------------------------------------------------------ approximation error
def test_approx_error(ops):
	g = _g(31)
	X = torch.randn(150, 48, generator=g); Et = torch.randn(700, 48, generator=g); A = torch.randn(150, 700, generator=g)
	err, nrm = ops.approx_error(X.cuda(), Et.cuda(), A.cuda())
	S = X.double() @ Et.double().t()
	torch.testing.assert_close(err.cpu().double(), ((S - A.double()) ** 2).sum(1), rtol=1e-4, atol=1e-3)
	torch.testing.assert_close(nrm.cpu().double(), (A.double() ** 2).sum(1), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rowwise_topk_negative_ties_and_kth_negative(ops, dtype):
	# all-negative rows with heavy ties (bf16 keys of negative scores have all-ones low bits): exercises the masked radix compare
	g = _g(77)
	A = (-(torch.rand(12, 20000, generator=g) * 4).round() / 4 - 1.0).to(dtype)
	for k in (1, 50, 128):
		v, i = ops.rowwise_topk(A.cuda(), k)
		rv, _ = torch.topk(A.float(), k, dim=1)
		assert torch.equal(v.cpu(), rv)
		i = i.cpu().long()
		assert torch.equal(torch.gather(A.float(), 1, i), rv)
		assert ((v.cpu()[:, :-1] > v.cpu()[:, 1:]) | ((v.cpu()[:, :-1] == v.cpu()[:, 1:]) & (i[:, :-1] < i[:, 1:]))).all() or k == 1
	# top-k of a short row that reaches into the negative part
	B = torch.cat([torch.full((3, 10), 2.0), -torch.arange(1, 301, dtype=torch.float32).repeat(3, 1) / 64], dim=1).to(dtype)
	v, i = ops.rowwise_topk(B.cuda(), 100)
	rv, _ = torch.topk(B.float(), 100, dim=1)
	assert torch.equal(v.cpu(), rv) and torch.equal(i.cpu()[:, :10].long(), torch.arange(10).repeat(3, 1))


def test_overlap_counts_long_lists(ops):
	g = np.random.default_rng(5)
	a = np.stack([g.permutation(5000)[:300] for _ in range(20)]).astype(np.int32)
	b = np.stack([g.permutation(5000)[:700] for _ in range(20)]).astype(np.int32)
	pairs = [(300, 700), (100, 700), (300, 10), (1, 1)]
	got = ops.overlap_counts(torch.tensor(a).cuda(), torch.tensor(b).cuda(), pairs).cpu().numpy()
	for p, (ka, kb) in enumerate(pairs):
		assert got[p].tolist() == [len(set(a[q, :ka]) & set(b[q, :kb])) for q in range(20)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rowwise_topk_threshold_crossing_zero_nan_and_sorted_rows(ops, dtype):
	"""The scan's integer-domain compares switch form with the sign of the running threshold, and its seed is a bound taken
	from the row's first 512 vectors: rows built to sit on those edges."""
	g = _g(123)
	I = 60000
	def check(A, k, exact_idx=True):
		v, i = ops.rowwise_topk(A.cuda(), k)
		v, i = v.cpu(), i.cpu().long()
		Af = A.float()
		Ar = torch.where(torch.isnan(Af), torch.full_like(Af, -float("inf")), Af)   # NaN is never selected
		rv, _ = torch.topk(Ar, k, dim=1)
		assert torch.equal(v, rv)
		assert torch.equal(torch.gather(Ar, 1, i), rv)
		assert all(len(set(r.tolist())) == k for r in i)
		if exact_idx:  # defined tie order: score descending, then smaller index
			order = torch.argsort(Ar, dim=1, descending=True, stable=True)[:, :k]
			assert torch.equal(i, order)
	# (1) ReLU-like: mostly exact zeros, top-k reaches into the zeros (seed bound = +0.0)
	relu = torch.relu(torch.randn(6, I, generator=g) - 2.2).to(dtype)
	for k in (1, 10, 100, 128): check(relu, k)
	# (2) first 10 % strongly negative (threshold starts negative), positives only later: the threshold crosses zero mid-stream
	cross = torch.randn(5, I, generator=g)
	cross[:, :6000] = -5.0 - torch.rand(5, 6000, generator=g)
	for k in (10, 100): check(cross.to(dtype), k)
	# (3) all negative (log-prob like) with ties
	neg = (-torch.rand(4, I, generator=g) * 8).to(dtype)
	check(neg, 100)
	# (4) NaN / +inf / -inf scattered, including inside the seed region
	sp = torch.randn(4, I, generator=g)
	sp[:, ::97] = float("nan"); sp[:, 5::1013] = float("inf"); sp[:, 7::511] = -float("inf")
	check(sp.to(dtype), 100)
	# (5) sorted rows: descending (the seed bound is the final threshold), ascending (every element is a candidate)
	desc = torch.sort(torch.randn(3, I, generator=g), dim=1, descending=True).values.to(dtype)
	check(desc, 100)
	asc = torch.sort(torch.randn(3, I, generator=g), dim=1).values.to(dtype)
	check(asc, 100)
	# (7) mostly -inf rows (masked scores): fewer finite values than k, also inside the seed region of a long row; -inf entries
	#     are real candidates (smallest indices first), only NaN is never selected
	masked = torch.full((5, I), -float("inf"))
	cols = torch.randperm(I, generator=g)[:37]
	masked[:, cols] = torch.randn(5, 37, generator=g)
	masked[2, :] = -float("inf")                       # nothing finite at all
	for k in (1, 37, 64, 128): check(masked.to(dtype), k)
	check(torch.full((9, 1), -float("inf")).to(dtype), 1)
	check(torch.tensor([[-float("inf"), 1.0, -float("inf"), float("inf"), -2.0]]).to(dtype), 5)
	# (6) misaligned row start (unaligned head elements come first in index order)
	base = torch.randn(4, I + 16, generator=g).to(dtype).cuda()
	for off in (1, 3, 7):
		view = base[:, off:off + I - 9]
		v, i = ops.rowwise_topk(view, 64)
		order = torch.argsort(view.float().cpu(), dim=1, descending=True, stable=True)[:, :64]
		assert torch.equal(i.cpu().long(), order)


@pytest.mark.parametrize("Q,I,n_idx,k,dt", [(300, 100000, 256, 100, torch.bfloat16), (257, 40007, 100, 10, torch.bfloat16), (64, 4096, 500, 128, torch.bfloat16),
											 (130, 20003, 64, 50, torch.float32), (1000, 9000, 1024, 1, torch.bfloat16), (5, 1031, 7, 5, torch.float32)])
def test_rowwise_topk_gather_equals_scan_plus_gather(ops, Q, I, n_idx, k, dt):
	"""a2 folded into a8's pass (anncur_rowwise_topk_gather): the same top-k as anncur_rowwise_topk bit for bit, and C_q equal to the
	separate gather -- anchors in the first / last vectors, in the row's tail (I % V != 0), adjacent anchors, padded row pitch."""
	g = _g(Q + I + n_idx)
	vec = 8 if dt == torch.bfloat16 else 4
	ld = (I + vec - 1) // vec * vec + 2 * vec                      # 16-byte aligned rows, padded
	Abuf = torch.randn(Q, ld, generator=g).to(dt).cuda()
	A = Abuf[:, :I]
	cols = torch.randperm(I, generator=g)[:n_idx]
	cols[0] = 0; cols[1] = I - 1; cols[2] = 1; cols[3] = max(2, I - 2)   # first vector, tail, neighbours
	cols = torch.unique(cols).sort().values.cuda()
	assert ops.rowwise_topk_gather_ok(A, k)
	tab = ops.gather_tables(cols, I, dt)
	(v, i), cq = ops.rowwise_topk_gather(A, k, tab)
	v0, i0 = ops.rowwise_topk(A, k)
	assert torch.equal(v, v0) and torch.equal(i, i0)
	assert torch.equal(cq, A[:, cols.long()])
	assert torch.equal(cq, ops.gather_cols(A, cols))
	assert not ops.rowwise_topk_gather_ok(Abuf[:, 1:I + 1], k)     # misaligned rows: the caller gathers separately
	with pytest.raises(ValueError):
		ops.gather_tables(cols.flip(0), I, dt)                     # not ascending


@pytest.mark.parametrize("Q,I,K,adt,pad", [(300, 4096, 64, torch.bfloat16, 8), (257, 10031, 128, torch.bfloat16, 8), (130, 7000, 256, torch.bfloat16, 8),
										   (70, 5023, 500, torch.bfloat16, 8), (129, 6400, 256, torch.float32, 8), (40, 40, 64, torch.bfloat16, 8),
										   (33, 20, 128, torch.float32, 8), (257, 10031, 128, torch.bfloat16, 4), (600, 9000, 256, torch.bfloat16, 4),
										   (1000, 33000, 256, torch.bfloat16, 8), (300, 20000, 512, torch.bfloat16, 8)])
def test_approx_error_packed_matches_strided_and_fp64(ops, Q, I, K, adt, pad):
	"""a11 on the sweep's MFMA loop (full 32-item tiles + the strided kernel for the last I % 32 columns) against the strided fp32
	GEMM reduction on the same bf16 operands and against fp64 on the CPU.  bf16 exact matrix with a row pitch that is a multiple of 8
	(pad = 8): the exact tile goes through LDS (error_lds_kernel); pitch = 4 mod 8 (pad = 4) and fp32: per-lane loads (error_kernel)."""
	g = _g(Q + I + K)
	X = torch.randn(Q, K, generator=g).bfloat16()
	E = (torch.randn(K, I, generator=g) / K ** 0.5).bfloat16()
	ld = (I + 7) // 8 * 8 + pad                                    # padded rows (row pitch a multiple of 4; of 8 iff pad = 8)
	Abuf = torch.zeros(Q, ld, dtype=adt)
	Abuf[:, :I] = (X.float() @ E.float() + 0.3 * torch.randn(Q, I, generator=g)).to(adt)
	A = Abuf.cuda()[:, :I]
	Kp = ops.padded_k(K)
	Xp = ops.pack_bf16(X.cuda(), Kp); Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	assert ops.approx_error_packed_ok(Kp, A)
	err, nrm = ops.approx_error_packed(Xp, Etp, A, I)
	err2, nrm2 = ops.approx_error(X.cuda(), E.t().contiguous().cuda(), A)
	torch.testing.assert_close(err.cpu(), err2.cpu(), rtol=2e-4, atol=1e-4)
	torch.testing.assert_close(nrm.cpu(), nrm2.cpu(), rtol=1e-5, atol=1e-5)
	S = X.double() @ E.double()
	Ad = Abuf[:, :I].double()
	torch.testing.assert_close(err.cpu().double(), ((S - Ad) ** 2).sum(1), rtol=2e-4, atol=1e-4)
	torch.testing.assert_close(nrm.cpu().double(), (Ad ** 2).sum(1), rtol=1e-5, atol=1e-5)
	assert not ops.approx_error_packed_ok(Kp, Abuf.cuda()[:, 1:I + 1])   # misaligned view: callers fall back to the strided kernel


def test_fused_index_hints_leave_the_result_unchanged(ops):
	"""anncur_score_topk_ex: item rows reordered by descending norm + leading-tile threshold sample + id map give the same top-k
	(values and item sets) as the plain call on the original order."""
	from anncur_amd.cur import _norm_sorted_pack
	Q, I, K, k = 333, 70001, 200, 100
	g = _g(99)
	X = torch.randn(Q, K, generator=g).bfloat16()
	E = (torch.randn(K, 24, generator=g) @ torch.randn(24, I, generator=g) / 5 + 0.05 * torch.randn(K, I, generator=g)) * (0.5 + torch.rand(1, I, generator=g))
	E = E.bfloat16()
	Kp = ops.padded_k(K)
	Xp = ops.pack_bf16(X.cuda(), Kp); Et = E.t().contiguous().cuda()
	v0, i0 = ops.score_topk_fused(Xp, ops.pack_bf16(Et, Kp, row_multiple=32), I, k)
	Es, ids = _norm_sorted_pack(Et.float(), Kp)
	assert ids.dtype == torch.int32 and sorted(ids.cpu().tolist()) == list(range(I))
	n = (Et.float() ** 2).sum(1)[ids.long()]                 # descending norm at the builder's bucket granularity (256 buckets between
	b = ((n.max() - n) / (n.max() - n.min()) * 256).floor()    # the largest and the smallest norm; ids ascending inside a bucket)
	assert (b[1:] - b[:-1] >= -1).all() and (b[-1] - b[0]) >= 250 and n[:100].mean() > 10 * n[-100:].mean()
	(v1, i1), nfb = ops.score_topk_fused(Xp, Es, I, k, leading_sample=True, item_ids=ids, return_fallbacks=True)
	assert nfb.item() == 0
	torch.testing.assert_close(v1.cpu(), v0.cpu(), rtol=1e-6, atol=1e-6)
	S = X.double() @ E.double()
	torch.testing.assert_close(torch.gather(S, 1, i1.cpu().long()), v1.cpu().double(), rtol=1e-4, atol=1e-4)
	same = [set(a.tolist()) == set(b.tolist()) for a, b in zip(i0.cpu(), i1.cpu())]
	assert sum(same) >= Q - 2           # (a boundary tie may resolve by row order instead of by id)
	v2, i2 = ops.score_topk_fused(Xp, Es, I, k, leading_sample=True)          # without the map: row numbers of the sorted matrix
	assert torch.equal(ids.long()[i2.long()].cpu(), i1.cpu().long())


@pytest.mark.parametrize("K,k", [(256, 500), (512, 300), (64, 400)])
def test_fused_dense_leading_tiles_need_no_repair(ops, K, k):
	"""Norm-ordered rows, large k: the prepass threshold lets about half of the leading tiles' elements through, so single lanes find
	more than 8 survivors in one tile -- more than their LDS ring holds.  Those lanes hand the tile's raw accumulator to the next
	flush (mark_wrapped_raw / RING_RAW): exact results, and NO query goes to the repair kernel (round 1..2a: five repairs per call on
	this kind of input, 0.26 ms of the k = 500 call).  All three loop shapes: staggered (Kp = 256), cross-tile (512), plain (64)."""
	from anncur_amd.cur import _norm_sorted_pack
	Q, I = 640, 60000
	g = _g(1234 + K)
	Z = torch.randn(8, I, generator=g)
	X = (torch.randn(Q, 8, generator=g) @ torch.randn(8, K, generator=g) / 8).bfloat16()
	E = ((torch.randn(K, 8, generator=g) @ Z / 8 / 12 + 0.002 * torch.randn(K, I, generator=g)) * (0.4 + 1.6 * torch.rand(1, I, generator=g) ** 6)).bfloat16()
	Kp = ops.padded_k(K)
	Xp = ops.pack_bf16(X.cuda(), Kp); Et = E.t().contiguous().cuda()
	Es, ids = _norm_sorted_pack(Et.float(), Kp)
	(v, i), nfb = ops.score_topk_fused(Xp, Es, I, k, leading_sample=True, item_ids=ids, return_fallbacks=True)
	assert nfb.item() == 0
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	torch.testing.assert_close(v.cpu().double(), rv, rtol=1e-4, atol=1e-4)
	torch.testing.assert_close(torch.gather(S, 1, i.cpu().long()), v.cpu().double(), rtol=1e-4, atol=1e-4)
	assert all(len(set(r.tolist())) == k for r in i.cpu())
	# the case must really be dense: restate the prepass threshold (k-th largest maximum over the lanes' accumulator groups in the
	# leading sample tiles) and count, per (query, tile, lane half), the survivors among the lane's 16 rows of a leading tile
	plan = ops.fused_plan(Q, I, Kp, k)
	n_st, grp = plan["n_sample_tiles"], plan["group"]
	Ss = S[:, ids.long().cpu()]                                          # scores in the kernel's row order
	lead = Ss[:, : 32 * n_st].reshape(Q, n_st, 4, 2, 4)                  # [query, tile, row octet, lane half, row]: row = 8 b + 4 h + j
	gmax = lead.amax(dim=4) if grp == 4 else lead.amax(dim=(2, 4))       # group = 4 rows of one (octet, half), or the lane's 16 rows
	tau0 = torch.topk(gmax.reshape(Q, -1), k, dim=1).values[:, -1]
	per_lane = (lead >= tau0[:, None, None, None, None]).sum(dim=(2, 4))
	assert int((per_lane > 8).sum()) >= 1, int(per_lane.max())            # (lanes that wrap their 8-slot ring exist on this input)


def test_fused_wrong_index_hint_is_still_exact(ops):
	"""ANNCUR_TOPK_LEADING_SAMPLE on rows ordered the WRONG way (ascending norm, strongly skewed norms): the threshold sampled from the
	leading tiles is far too low, segments overflow, the in-call repair keeps the result exact."""
	Q, I, K, k = 200, 60000, 64, 50
	g = _g(7)
	X = torch.randn(Q, K, generator=g).bfloat16()
	scale = torch.linspace(0.01, 3.0, I)                              # item norms grow along the row order
	E = (torch.randn(K, I, generator=g) * scale).bfloat16()
	Xp = ops.pack_bf16(X.cuda(), 64); Etp = ops.pack_bf16(E.t().contiguous().cuda(), 64, row_multiple=32)
	(v, i), nfb = ops.score_topk_fused(Xp, Etp, I, k, leading_sample=True, return_fallbacks=True)
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	torch.testing.assert_close(v.cpu().double(), rv, rtol=1e-4, atol=1e-4)
	torch.testing.assert_close(torch.gather(S, 1, i.cpu().long()), v.cpu().double(), rtol=1e-4, atol=1e-4)
	assert sum(set(a.tolist()) == set(b.tolist()) for a, b in zip(i.cpu(), ri)) >= Q - 2


def test_fused_row_chunking_when_the_workspace_would_be_huge(ops, monkeypatch):
	X, E, Xp, Etp = _fused_case(ops, 2100, 40000, 128, 50, seed=5)
	v0, i0 = ops.score_topk_fused(Xp, Etp, 40000, 50)
	full = ops._lib.load().anncur_score_topk_workspace_bytes(2100, 40000, 128, 50)
	monkeypatch.setattr(ops, "FUSED_WS_LIMIT_BYTES", full // 3)          # forces three or more row chunks
	v1, i1 = ops.score_topk_fused(Xp, Etp, 40000, 50)
	assert torch.equal(v0, v1) and torch.equal(i0, i1)
