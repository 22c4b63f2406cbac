"""Randomised shape / distribution sweeps of the two selection kernels (hypothesis, bounded): exact top-k scan and fused score + top-k.
Needs an MI355X."""
import numpy as np
import pytest
import torch
import os

from hypothesis import example, HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu
# Deterministic by default (the same examples every run); ANNCUR_FUZZ=1 draws fresh ones and ANNCUR_FUZZ_EXAMPLES=n draws more.
_FUZZ = os.environ.get("ANNCUR_FUZZ", "") not in ("", "0")
_N = int(os.environ.get("ANNCUR_FUZZ_EXAMPLES", "0"))


@pytest.fixture(scope="module")
def ops():
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	from anncur_amd import ops as _ops
	return _ops


def _row_data(kind, Q, I, g):
	if kind == "normal":
		return torch.randn(Q, I, generator=g)
	if kind == "ties":       # a handful of distinct values: every compare is a tie somewhere
		return torch.randint(-3, 4, (Q, I), generator=g).float() / 4
	if kind == "const":
		return torch.full((Q, I), -1.5)
	if kind == "negative":
		return -torch.rand(Q, I, generator=g) * 100 - 1
	if kind == "special":
		A = torch.randn(Q, I, generator=g)
		m = torch.rand(Q, I, generator=g)
		A[m < 0.02] = float("nan"); A[(m >= 0.02) & (m < 0.03)] = float("inf"); A[(m >= 0.03) & (m < 0.05)] = -float("inf")
		return A
	if kind == "masked":     # mostly -inf, a few finite values
		A = torch.full((Q, I), -float("inf"))
		m = torch.rand(Q, I, generator=g) < 0.01
		A[m] = torch.randn(int(m.sum()), generator=g)
		return A
	if kind == "ascending":
		return torch.sort(torch.randn(Q, I, generator=g), dim=1).values
	return torch.sort(torch.randn(Q, I, generator=g), dim=1, descending=True).values   # "descending"


@settings(max_examples=_N or 60, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(Q=st.integers(1, 24), I=st.integers(1, 40000), kfrac=st.floats(0.0, 1.0), bf16=st.booleans(),
	   kind=st.sampled_from(["normal", "ties", "const", "negative", "special", "masked", "ascending", "descending"]), off=st.integers(0, 7), seed=st.integers(0, 10 ** 6),
	   short=st.booleans())
def test_rowwise_topk_random(ops, Q, I, kfrac, bf16, kind, off, seed, short):
	g = torch.Generator().manual_seed(seed)
	if short: I = 1 + I % 1100   # rows of at most 1024 elements with k <= 128 take rowwise_topk_short_kernel (round 5)
	k = max(1, min(I, 1 + int(kfrac * min(I, 300))))
	buf = torch.zeros(Q, I + 16)
	buf[:, off:off + I] = _row_data(kind, Q, I, g)
	buf = buf.to(torch.bfloat16 if bf16 else torch.float32).cuda()
	A = buf[:, off:off + I]                                      # misaligned, padded rows
	v, i = ops.rowwise_topk(A, k)
	v, i = v.cpu(), i.cpu().long()
	Af = A.float().cpu()
	Ar = torch.where(torch.isnan(Af), torch.full_like(Af, -float("inf")), Af)   # NaN is never selected (ranks with -inf)
	order = torch.argsort(Ar, dim=1, descending=True, stable=True)[:, :k]
	want_v = torch.gather(Ar, 1, order)
	assert torch.equal(v, want_v)
	nan_free = not torch.isnan(Af).any()
	if nan_free:
		assert torch.equal(i, order)                              # defined tie order: score descending, then smaller index
	else:
		valid = i >= 0
		assert torch.equal(torch.gather(Ar, 1, i.clamp(min=0))[valid], v[valid])
		assert all(len(set(r[r >= 0].tolist())) == int((r >= 0).sum()) for r in i)


@settings(max_examples=_N or 40, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(Q=st.integers(1, 24), I=st.integers(1, 40000), kfrac=st.floats(0.0, 1.0), bf16=st.booleans(), nfrac=st.floats(0.0, 1.0),
	   kind=st.sampled_from(["normal", "ties", "const", "negative", "special", "masked", "ascending", "descending"]), pad=st.integers(0, 3), seed=st.integers(0, 10 ** 6))
def test_rowwise_topk_gather_random(ops, Q, I, kfrac, bf16, nfrac, kind, pad, seed):
	"""anncur_rowwise_topk_gather == anncur_rowwise_topk + A[:, columns] bit for bit: random row lengths (tails of I % V elements), anchor
	counts from 1 to min(I, 2000) (dense and sparse, first / last columns included), padded 16-byte aligned rows, special values."""
	g = torch.Generator().manual_seed(seed)
	dt = torch.bfloat16 if bf16 else torch.float32
	vec = 8 if bf16 else 4
	k = max(1, min(I, 128, 1 + int(kfrac * min(I, 128))))
	ld = (I + vec - 1) // vec * vec + pad * vec
	buf = torch.zeros(Q, ld)
	buf[:, :I] = _row_data(kind, Q, I, g)
	A = buf.to(dt).cuda()[:, :I]
	n_idx = max(1, min(I, 2000, 1 + int(nfrac * nfrac * min(I, 2000))))
	cols = torch.randperm(I, generator=g)[:n_idx]
	if n_idx >= 2: cols[0] = 0; cols[1] = I - 1
	cols = torch.unique(cols).sort().values.cuda()
	assert ops.rowwise_topk_gather_ok(A, k)
	(v, i), cq = ops.rowwise_topk_gather(A, k, ops.gather_tables(cols, I, dt))
	v0, i0 = ops.rowwise_topk(A, k)
	assert torch.equal(v, v0) and torch.equal(i, i0)
	want = A[:, cols.long()]
	assert torch.equal(cq.view(torch.int16 if bf16 else torch.int32), want.contiguous().view(torch.int16 if bf16 else torch.int32))   # bit patterns (NaN-safe)


@settings(max_examples=(_N // 4) or 25, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(Q=st.integers(1, 300), I=st.integers(2500, 90000), K=st.integers(8, 512), k=st.integers(1, 400), rank=st.integers(2, 48),
	   noise=st.floats(0.0, 0.3), seed=st.integers(0, 10 ** 6), variant=st.sampled_from(["", "", "mfma16", "qt1", "mfma32", "ring"]))
def test_fused_score_topk_random(ops, Q, I, K, k, rank, noise, seed, variant):
	g = torch.Generator().manual_seed(seed)
	X = torch.randn(Q, K, generator=g).bfloat16()
	E = (torch.randn(K, rank, generator=g) @ torch.randn(rank, I, generator=g) / rank ** 0.5 + noise * torch.randn(K, I, generator=g)).bfloat16()
	Kp = ops.padded_k(K)
	if not ops.fused_supported(Q, I, Kp, k):
		return
	Xp = ops.pack_bf16(X.cuda(), Kp); Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	from anncur_amd import _lib
	if variant == "ring" and not _lib.IS_EXPERIMENTS_LIB:
		variant = "staged"   # (round 5) the tile-ring body lives in the experiments library only; its draws exercise ANNCUR_TOPK_STAGED (the staged sweep of rounds 1-4) instead
	kw = dict(mfma16=variant == "mfma16", qt1=variant == "qt1", mfma32=variant == "mfma32", ring=variant == "ring", staged=variant == "staged")
	plan = ops.fused_plan(Q, I, Kp, k, **kw)
	if variant == "ring":   # (round 4) the tile-ring body runs where the 16x16x32 body would: Kp = 128 / 256, k <= 128
		assert all(b == 5 for b in plan["stage_pred"]) == (Kp in (128, 256) and k <= 128), plan
	elif Kp <= 256:   # the variant the draw names is the kernel that runs (Kp = 512 has one body; qt1 needs Kp >= 128; "" = 16x16x32 up to k = 1024 with the ladder, up to 384 staged)
		want_lg = {"mfma16": (1,), "mfma32": (2,), "qt1": (2,) if Kp >= 128 else (1, 2), "": (1, 2), "staged": (1, 2)}[variant]   # (qt1 at Kp = 64: no such body, the default runs)
		assert plan["lg"] in want_lg and plan["QT"] == (1 if variant == "qt1" and Kp >= 128 else 2), (variant, plan)
	if variant == "staged": assert not plan["ladder"]
	elif variant == "" and plan["lg"] == 1 and Kp <= 256: assert plan["ladder"] and plan["n_stages"] == 1   # the default 16x16x32 body sweeps in one launch (threshold ladder)
	v, i = ops.score_topk_fused(Xp, Etp, I, k, **kw)   # (sweep variants: same answer)
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	scale = float(S.abs().max()) + 1e-30
	got = i.cpu().long()
	assert (got >= 0).all() and (got < I).all() and all(len(set(r.tolist())) == k for r in got)
	assert (v.cpu().double() - rv).abs().max() <= 1e-4 * scale
	assert (torch.gather(S, 1, got) - v.cpu().double()).abs().max() <= 1e-4 * scale
	assert ((v[:, :-1] >= v[:, 1:]).all())                       # sorted descending
	# every selected item beats (up to fp32 round-off) the true k-th score
	assert (torch.gather(S, 1, got).min(dim=1).values >= rv[:, -1] - 1e-4 * scale).all()


@settings(max_examples=(_N // 6) or 16, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(Q=st.integers(1, 700), I=st.integers(2500, 90000), K=st.integers(8, 256), k=st.integers(1, 300), rank=st.integers(2, 48),
	   noise=st.floats(0.0, 0.3), seed=st.integers(0, 10 ** 6), pad=st.integers(0, 3))
def test_eval_fused_random(ops, Q, I, K, k, rank, noise, seed, pad):
	"""anncur_eval_fused (round 4: candidates + error sums of entry A's cell in ONE sweep) on random shapes, pitches and score
	distributions, against fp64: the top-k as the fused top-k's fuzz checks it, the two sums to 1e-4."""
	g = torch.Generator().manual_seed(seed)
	X = torch.randn(Q, K, generator=g).bfloat16()
	E = (torch.randn(K, rank, generator=g) @ torch.randn(rank, I, generator=g) / rank ** 0.5 + noise * torch.randn(K, I, generator=g)).bfloat16()
	A = (X.float() @ E.float() + 0.5 * torch.randn(Q, I, generator=g)).bfloat16()
	Kp = ops.padded_k(K)
	lda = -(-I // 8) * 8 + 8 * pad
	Ap = torch.zeros((Q, lda), dtype=torch.bfloat16, device="cuda"); Ap[:, :I] = A.cuda(); Ad = Ap[:, :I]
	if not ops.eval_fused_ok(Kp, Ad, Q, I, k):
		return
	Xp = ops.pack_bf16(X.cuda(), Kp); Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	(v, i), err, nrm, nfb = ops.eval_fused(Xp, Etp, Ad, I, k, return_fallbacks=True)
	assert nfb.item() < (1 << 30)
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	scale = float(S.abs().max()) + 1e-30
	got = i.cpu().long()
	assert (got >= 0).all() and (got < I).all() and all(len(set(r.tolist())) == k for r in got)
	assert (v.cpu().double() - rv).abs().max() <= 1e-4 * scale
	assert ((v[:, :-1] >= v[:, 1:]).all())
	assert (torch.gather(S, 1, got).min(dim=1).values >= rv[:, -1] - 1e-4 * scale).all()
	A64 = A.double()
	torch.testing.assert_close(err.cpu().double(), ((S - A64) ** 2).sum(1), rtol=1e-4, atol=1e-4 * scale * scale)
	torch.testing.assert_close(nrm.cpu().double(), (A64 ** 2).sum(1), rtol=1e-4, atol=1e-6)


@settings(max_examples=(_N // 8) or 12, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(Q=st.integers(1, 600), I=st.integers(600, 60000), K=st.integers(513, 2300), k=st.integers(1, 300), rank=st.integers(2, 48),
	   noise=st.floats(0.0, 0.3), seed=st.integers(0, 10 ** 6))
def test_wide_score_topk_random(ops, Q, I, K, k, rank, noise, seed):
	"""The K-general kernel (Kp > 512: LDS-tiled GEMM with the filter as epilogue) on random shapes, against fp64."""
	g = torch.Generator().manual_seed(seed)
	X = torch.randn(Q, K, generator=g).bfloat16()
	E = (torch.randn(K, rank, generator=g) @ torch.randn(rank, I, generator=g) / rank ** 0.5 + noise * torch.randn(K, I, generator=g)).bfloat16()
	Kp = ops.padded_k(K)
	assert Kp > 512 and Kp % 128 == 0
	if not ops.fused_supported(Q, I, Kp, k):
		return
	Xp = ops.pack_bf16(X.cuda(), Kp); Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	v, i = ops.score_topk_fused(Xp, Etp, I, k)
	S = X.double() @ E.double()
	rv, ri = torch.topk(S, k, dim=1)
	scale = float(S.abs().max()) + 1e-30
	got = i.cpu().long()
	assert (got >= 0).all() and (got < I).all() and all(len(set(r.tolist())) == k for r in got)
	assert (v.cpu().double() - rv).abs().max() <= 1e-4 * scale
	assert (torch.gather(S, 1, got) - v.cpu().double()).abs().max() <= 1e-4 * scale
	assert ((v[:, :-1] >= v[:, 1:]).all())
	assert (torch.gather(S, 1, got).min(dim=1).values >= rv[:, -1] - 1e-4 * scale).all()


@settings(max_examples=(_N // 10) or 8, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(Q=st.integers(1, 400), I=st.integers(600, 40000), K=st.integers(513, 1100), k=st.integers(1, 300), levels=st.integers(1, 3),
	   kind=st.sampled_from(["ties", "const", "hot"]), seed=st.integers(0, 10 ** 6))
def test_wide_score_topk_random_ties_and_overflow(ops, Q, I, K, k, levels, kind, seed):
	"""Small-integer operands through the K-general kernel: exact integer scores, ties everywhere, whole item ranges above the
	sampled threshold (segment overflow -> the in-call repair).  The result must be THE top-k under the defined order."""
	g = torch.Generator().manual_seed(seed)
	X = torch.randint(0, levels + 1, (Q, K), generator=g).float()
	if kind == "const":
		E = torch.ones(K, I)
	else:
		E = torch.randint(-levels, levels + 1, (K, I), generator=g).float()
		if kind == "hot":
			width = min(3000, I // 3); lo = int(torch.randint(0, I - width, (1,), generator=g)); E[:, lo:lo + width] += levels + 1
	Kp = ops.padded_k(K)
	if not ops.fused_supported(Q, I, Kp, k):
		return
	Xp = ops.pack_bf16(X.cuda(), Kp); Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	v, i = ops.score_topk_fused(Xp, Etp, I, k)
	S = X.double() @ E.double()
	order = torch.argsort(S, dim=1, descending=True, stable=True)[:, :k]
	assert torch.equal(v.cpu().double(), torch.gather(S, 1, order))
	assert torch.equal(i.cpu().long(), order)


@settings(max_examples=_N or 40, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(M=st.integers(1, 300), N=st.integers(1, 300), K=st.integers(1, 200), ta=st.booleans(), tb=st.booleans(), a16=st.booleans(), b16=st.booleans(),
	   c16=st.booleans(), alpha=st.sampled_from([1.0, -0.5, 2.0]), use_cin=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_gemm_random(ops, M, N, K, ta, tb, a16, b16, c16, alpha, use_cin, seed):
	g = torch.Generator().manual_seed(seed)
	A = torch.randn((K, M) if ta else (M, K), generator=g).to(torch.bfloat16 if a16 else torch.float32).cuda()
	B = torch.randn((N, K) if tb else (K, N), generator=g).to(torch.bfloat16 if b16 else torch.float32).cuda()
	Av, Bv = (A.t() if ta else A), (B.t() if tb else B)
	cin = torch.randn(M, N, generator=g).cuda() if use_cin else None
	beta = 0.75 if use_cin else 0.0
	out = ops.gemm(Av, Bv, out_dtype=torch.bfloat16 if c16 else torch.float32, alpha=alpha, beta=beta, cin=cin)
	want = alpha * (Av.double().cpu() @ Bv.double().cpu()) + (beta * cin.double().cpu() if use_cin else 0.0)
	tol = (2e-2 if c16 else 1e-4) * (float(want.abs().max()) + 1.0)
	assert out.dtype == (torch.bfloat16 if c16 else torch.float32) and tuple(out.shape) == (M, N)
	assert (out.double().cpu() - want).abs().max() <= tol


@settings(max_examples=_N or 40, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(Q=st.integers(1, 60), la=st.integers(1, 400), lb=st.integers(1, 400), universe=st.integers(1, 5000), n_pairs=st.integers(1, 6), seed=st.integers(0, 10 ** 6))
def test_overlap_and_rerank_random(ops, Q, la, lb, universe, n_pairs, seed):
	rng = np.random.default_rng(seed)
	universe = max(universe, la, lb)
	a = np.stack([rng.permutation(universe)[:la] for _ in range(Q)]).astype(np.int32)
	b = np.stack([rng.permutation(universe)[:lb] for _ in range(Q)]).astype(np.int32)
	pairs = [(int(rng.integers(0, la + 1)), int(rng.integers(0, lb + 1))) for _ in range(n_pairs)]
	got = ops.overlap_counts(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), pairs).cpu().numpy()
	for p, (ka, kb) in enumerate(pairs):
		assert got[p].tolist() == [len(set(a[q, :ka]) & set(b[q, :kb])) for q in range(Q)]
	# re-rank: the k_out best of approx[:, :k_retvr] by exact score, ties -> smaller index
	A = torch.from_numpy(rng.integers(-8, 9, size=(Q, universe)).astype(np.float32) / 4)
	k_retvr = int(rng.integers(1, la + 1)); k_out = int(rng.integers(1, min(k_retvr, 2048) + 1))
	rr = ops.rerank(A.cuda(), torch.from_numpy(a).cuda(), k_retvr, k_out)
	for q in range(Q):
		cand = a[q, :k_retvr].astype(np.int64)
		order = sorted(cand.tolist(), key=lambda i: (-float(A[q, i]), i))[:k_out]
		assert rr.indices[q].cpu().tolist() == order
		assert rr.values[q].cpu().tolist() == [float(A[q, i]) for i in order]


@settings(max_examples=(_N // 4) or 15, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(Q=st.integers(1, 400), I=st.integers(1, 9000), K=st.integers(1, 512), fp32_exact=st.booleans(), seed=st.integers(0, 10 ** 6))
def test_approx_error_packed_random(ops, Q, I, K, fp32_exact, seed):
	g = torch.Generator().manual_seed(seed)
	X = torch.randn(Q, K, generator=g).bfloat16()
	E = (torch.randn(K, I, generator=g) / K ** 0.5).bfloat16()
	ld = (I + 3) // 4 * 4
	Abuf = torch.zeros(Q, ld, dtype=torch.float32 if fp32_exact else torch.bfloat16)
	Abuf[:, :I] = (X.float() @ E.float() + 0.3 * torch.randn(Q, I, generator=g)).to(Abuf.dtype)
	A = Abuf.cuda()[:, :I]
	Kp = ops.padded_k(K)
	if ops.approx_error_packed_ok(Kp, A):
		err, nrm = ops.approx_error_packed(ops.pack_bf16(X.cuda(), Kp), ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32), A, I)
	else:   # (degenerate row pitch, e.g. a one-row matrix: the callers' fallback)
		err, nrm = ops.approx_error(X.cuda(), E.t().contiguous().cuda(), A)
	S = X.double() @ E.double(); Ad = Abuf[:, :I].double()
	torch.testing.assert_close(err.cpu().double(), ((S - Ad) ** 2).sum(1), rtol=3e-4, atol=1e-4)
	torch.testing.assert_close(nrm.cpu().double(), (Ad ** 2).sum(1), rtol=1e-5, atol=1e-5)


@settings(max_examples=(_N // 8) or 12, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(n=st.integers(12, 300), m=st.integers(40, 3000), rank=st.integers(2, 12), kr_frac=st.floats(0.1, 0.5), kc_frac=st.floats(0.02, 0.3),
	   pref=st.sampled_from(["rows", "cols"]), seed=st.integers(0, 10 ** 6))
def test_cur_operator_api_random_vs_oracle(ops, n, m, rank, kr_frac, kc_frac, pref, seed):
	"""CURApprox (HIP, fp32 route) against the CPU restatement of the reference class on random low-rank + noise matrices with
	over-sampled anchors: U, latent factors, get / get_rows / get_cols / get_complete_* and the top-k sets."""
	from anncur_amd.cur import CURApprox
	from oracle import cur_oracle as O
	g = torch.Generator().manual_seed(seed)
	A = torch.randn(n, rank, generator=g) @ torch.randn(rank, m, generator=g) / rank ** 0.5 + 0.05 * torch.randn(n, m, generator=g)
	rng = np.random.default_rng(seed)
	kc = min(m - 1, max(rank + 2, int(kc_frac * m)))
	kr = min(n, max(2 * kc if pref == "rows" else rank + 2, int(kr_frac * n), 1))   # over-sampled rows keep the intersection well conditioned
	if pref == "cols":
		kc = min(m - 1, max(kc, 2 * kr))
	ri = sorted(rng.choice(n, size=kr, replace=False).tolist()); ci = sorted(rng.choice(m, size=kc, replace=False).tolist())
	ref = O.CURApproxOracle(rows=A[ri, :], cols=A[:, ci], row_idxs=ri, col_idxs=ci, approx_preference=pref)
	cur = CURApprox(rows=A[ri, :], cols=A[:, ci], row_idxs=ri, col_idxs=ci, approx_preference=pref)
	S_ref = ref.get(list(range(n)), list(range(m)))
	scale = float(S_ref.abs().max()) + 1e-6
	def close(a, b, tol=2e-3):
		assert tuple(a.shape) == tuple(b.shape) and float((a.double() - b.double()).abs().max()) <= tol * scale
	S = cur.get(list(range(n)), list(range(m)))
	assert S.device.type == "cpu"
	assert float(torch.linalg.norm(S - S_ref) / torch.linalg.norm(S_ref)) < 2e-3     # (conditioning-dependent; 1e-4 on the pinned goldens)
	sub_r = sorted(rng.choice(n, size=min(n, 5), replace=False).tolist()); sub_c = sorted(rng.choice(m, size=min(m, 7), replace=False).tolist())
	close(cur.get(sub_r, sub_c), ref.get(sub_r, sub_c)); close(cur.get_rows(sub_r), ref.get_rows(sub_r)); close(cur.get_cols(sub_c), ref.get_cols(sub_c))
	k = int(rng.integers(1, min(m, 50) + 1))
	if pref == "rows":
		close(cur.get_complete_row(A[:, ci]), ref.get_complete_row(A[:, ci]))
		tv, ti = cur.topk_in_row(A[:, ci], k)
		rv, _ = torch.topk(S_ref, k, dim=1)
		close(tv, rv)
		close(torch.gather(S_ref, 1, ti), rv)          # the selected items' reference scores are the reference's top-k values
		with pytest.raises(NotImplementedError):
			cur.get_complete_col(A[ri, :][:, :3])
	else:
		cols_in = A[ri, :][:, :min(m, 9)]
		close(cur.get_complete_col(cols_in), ref.get_complete_col(cols_in))
		with pytest.raises(NotImplementedError):
			cur.get_complete_row(A[:, ci])


@settings(max_examples=(_N // 8) or 12, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(Q=st.integers(1, 200), I=st.integers(2500, 70000), K=st.integers(1, 96), k=st.integers(1, 300), levels=st.integers(1, 4),
	   kind=st.sampled_from(["ties", "const", "hot"]), seed=st.integers(0, 10 ** 6))
def test_fused_score_topk_random_ties_and_overflow(ops, Q, I, K, k, levels, kind, seed):
	"""Small-integer operands: every score is an exactly representable integer, ties are everywhere and whole item ranges can
	beat the sampled threshold (segment overflow / ring wrap -> the in-call repair).  The result must be THE top-k under the
	defined order (score descending, then smaller index)."""
	g = torch.Generator().manual_seed(seed)
	X = torch.randint(0, levels + 1, (Q, K), generator=g).float()
	if kind == "const":
		E = torch.ones(K, I)
	else:
		E = torch.randint(-levels, levels + 1, (K, I), generator=g).float()
		if kind == "hot":
			width = min(3000, I // 3); lo = int(torch.randint(0, I - width, (1,), generator=g)); E[:, lo:lo + width] += levels + 1
	Kp = ops.padded_k(K)
	if not ops.fused_supported(Q, I, Kp, k):
		return
	Xp = ops.pack_bf16(X.cuda(), Kp); Etp = ops.pack_bf16(E.t().contiguous().cuda(), Kp, row_multiple=32)
	v, i = ops.score_topk_fused(Xp, Etp, I, k)
	S = X.double() @ E.double()
	order = torch.argsort(S, dim=1, descending=True, stable=True)[:, :k]
	assert torch.equal(v.cpu().double(), torch.gather(S, 1, order))
	assert torch.equal(i.cpu().long(), order)


@settings(max_examples=(_N // 20) or 6, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(n_train=st.integers(60, 200), n_test=st.integers(20, 150), n_ent=st.integers(300, 4000), rank=st.integers(3, 10), seed=st.integers(0, 50),
	   k_retvr=st.sampled_from([20, 64, 100]), bf16=st.booleans())
@example(n_train=111, n_test=22, n_ent=300, rank=3, seed=22, k_retvr=20, bf16=True)      # (fresh-draw finds: bf16 ties, see below)
@example(n_train=170, n_test=56, n_ent=2344, rank=9, seed=9, k_retvr=64, bf16=True)
def test_entry_point_sweeps_random_vs_oracle(ops, n_train, n_test, n_ent, rank, seed, k_retvr, bf16):
	"""The two evaluation sweeps (entry A: one matrix, anchor / non-anchor / all rows; entry B: train/test split over an anchor
	grid) against the CPU restatement of the reference loops on random low-rank + noise inputs."""
	from anncur_amd import harness
	from oracle import cur_oracle as O
	g = torch.Generator().manual_seed(1000 + seed)
	Z = torch.randn(rank, n_ent, generator=g)
	mk = lambda n: torch.randn(n, rank, generator=g) @ Z / rank ** 0.5 + 0.05 * torch.randn(n, n_ent, generator=g)
	A_train, A_test = mk(n_train), mk(n_test)
	key = "exact_vs_reranked_approx_retvr~common_frac_mean"
	# entry B over two anchor counts; bf16 storage is compared with the oracle on the same bf16-rounded values
	if bf16:
		A_train, A_test = A_train.bfloat16().float(), A_test.bfloat16().float()
	dev = lambda t: t.cuda().bfloat16() if bf16 else t.cuda()
	anc_vals = [max(rank + 2, n_train // 4), max(rank + 3, n_train // 2)]
	grids = {"top_k_vals": [1, 10], "top_k_retr_vals": [k_retvr], "n_ent_anchors_vals": anc_vals}
	got = harness.run_eval_method_cur(dev(A_test), dev(A_train), seed, grids)
	if bf16:
		# bf16 exact scores tie often, and the reference-faithful loop inherits torch.topk's arbitrary tie order (its exact top-1 and
		# its re-ranked top-1 can be two items of equal score: recall@1 reads 3-7 % low, DESIGN.md section 2 "Ties"; fresh-draw runs
		# hit that with 22 and 56 queries).  This build orders ties by index, so the bf16 case is judged against the oracle's
		# tie-stable statement of the same loop, over the same anchor sequence (one rng stream across the anchor counts).
		import numpy as np
		rng = np.random.default_rng(seed=seed)
		want = {}
		for n_anc in anc_vals:
			anc = O.select_anchors(rng, n_ent, n_anc)
			ref = O.CURApproxOracle(rows=A_train, cols=A_train[:, anc], row_idxs=np.arange(n_train), col_idxs=anc, approx_preference="rows")
			per_k = O.eval_all_topk_stable(A_test, ref.get_complete_row(A_test[:, anc]), [1, 10], k_retvr)
			for k in (1, 10):
				want.setdefault(f"top_k={k}", {}).setdefault(f"k_retvr={k_retvr}", {})[f"anc_n_m={n_train}_anc_n_e={n_anc}"] = per_k[k]
	else:
		want = O.run_eval_method_cur(A_test, A_train, seed, [1, 10], [k_retvr], anc_vals)
	tol = 0.06 if bf16 else 0.02      # per-query boundary near-ties (and bf16 item embeddings) move single elements in and out
	for k in (1, 10):
		for n_anc in anc_vals:
			cell = f"anc_n_m={n_train}_anc_n_e={n_anc}"
			assert abs(got[f"top_k={k}"][f"k_retvr={k_retvr}"][cell][key] - want[f"top_k={k}"][f"k_retvr={k_retvr}"][cell][key]) <= tol, (k, n_anc)
	# entry A on the test matrix (fp32 route only: the reference's own arithmetic)
	if not bf16:
		n_m, n_e = max(rank + 4, n_test // 3), max(rank + 2, min(n_ent // 10, n_test // 6 + rank))
		a = harness.run_approx_eval_w_seed("cur", A_test.cuda(), n_m, n_e, 10, k_retvr, seed)
		b = O.run_approx_eval_w_seed("cur", A_test, n_m, n_e, 10, k_retvr, seed)
		for subset in ("anchor", "non_anchor", "all"):
			assert abs(float(a[subset][key]) - float(b[subset][key])) <= 0.03, subset
			if subset != "anchor":
				assert float(a[subset]["approx_error_relative"]) == pytest.approx(float(b[subset]["approx_error_relative"]), rel=2e-2, abs=1e-4), subset


@settings(max_examples=(_N // 10) or 12, deadline=None, derandomize=not _FUZZ, database=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(n=st.integers(200, 30000), d=st.integers(3, 200), nlist=st.integers(2, 120), nprobe=st.integers(1, 120), nq=st.integers(1, 60), k=st.integers(1, 300),
	   clusters=st.integers(1, 40), seed=st.integers(0, 10 ** 6), batched=st.booleans())
def test_ivf_flat_random(ops, n, d, nlist, nprobe, nq, k, clusters, seed, batched):
	"""IVF-flat index on random sizes: lists complete and disjoint, the search equals a brute-force search restricted to the probed
	lists (scores exact, FAISS padding), and probing every list equals the exact search.  batched: the list-grouped search whatever the
	number of queries (round 5: anncur_ivf_search_grouped for k <= 128 -- packed score rows, ragged scan --, round 4's sequence above)."""
	from anncur_amd.nearest_nbr import IVFFlatIPIndex
	g = np.random.default_rng(seed)
	nlist = min(nlist, n)
	centers = g.standard_normal((clusters, d)).astype(np.float32) * 2
	X = (centers[g.integers(0, clusters, n)] + g.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
	q = (centers[g.integers(0, clusters, nq)] + g.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
	index = IVFFlatIPIndex(d, nlist, niter=5)
	index.train(X); index.add(X)
	index.nprobe = nprobe
	if batched: index.batched_from = 1
	D, I = index.search(q, k)
	off, ids = index._offsets.cpu().numpy(), index._ids.cpu().numpy()
	assert off[0] == 0 and off[-1] == n and (np.sort(ids) == np.arange(n)).all()
	C = index.centroids.cpu().numpy()
	S = q.astype(np.float64) @ X.astype(np.float64).T
	npr = min(nprobe, nlist)
	probe = np.argsort(-(q @ C.T), axis=1, kind="stable")[:, :npr]
	scale = np.abs(S).max() + 1e-30
	for j in range(nq):
		cand = np.concatenate([ids[off[l]:off[l + 1]] for l in probe[j]])
		found = I[j][I[j] >= 0]
		m = min(k, len(cand))
		if m < k:   # fewer vectors in the probed lists than asked for: (-inf, -1) padding
			assert (I[j, m:] == -1).all() and (D[j, m:] == np.finfo(np.float32).min).all()
		# a near-tie among the centroid scores may swap the last probed list between fp32 GEMM and this numpy reference: compare scores
		assert len(found) >= min(k, 1) and len(set(found.tolist())) == len(found)
		np.testing.assert_allclose(D[j, :len(found)], S[j, found], rtol=0, atol=1e-4 * scale)
		assert (D[j, :len(found) - 1] >= D[j, 1:len(found)]).all()
		if len(found) == m:
			want = np.sort(S[j, cand])[::-1][:m]
			assert np.abs(D[j, :m] - want).max() <= 1e-4 * scale or len(set(found.tolist()) - set(cand.tolist())) > 0
	index.nprobe = nlist
	D2, I2 = index.search(q, min(k, n))
	want = -np.sort(-S, axis=1)[:, :min(k, n)]
	np.testing.assert_allclose(D2, want, rtol=0, atol=1e-4 * scale)
