"""Flat exact inner-product index with the FAISS calling convention the reference relies on
(models/nearest_nbr.py:24-55; call sites utils/data_process.py:343-351,393-397 and
eval/run_cross_encoder_w_binenc_retriever_zeshel.py:125,147):

    index = build_flat_or_ivff_index(embeds, force_exact_search)
    D, I = index.search(queries, k)      # D float32 [nq, k] descending, I int64 [nq, k]   (NumPy, like FAISS)

FAISS itself is a third-party dependency that the reference neither vendors nor pins: parity at this boundary is UNPINNED and
judged against torch.topk(q @ X^T) (tests/).  Like the reference, up to 11 000 vectors (or with force_exact_search) the index is
flat and exact; above, it is an IVF-flat index with nlist = floor(sqrt(n)) lists and nprobe = floor(sqrt(nlist) * probe_mult_factor)
probed lists (models/nearest_nbr.py:40-52), built and searched on the GPU (IVFFlatIPIndex), judged on recall against the exact search.
"""
import logging
import math

import numpy as np
import torch

from . import ops

LOGGER = logging.getLogger(__name__)


PAD_SCORE = np.float32(np.finfo(np.float32).min)   # -FLT_MAX: the neutral element of FAISS' inner-product result heaps (CMin<float>::neutral())


def _faiss_pad(v, i, nq, k, k_eff):
	"""(D float32 [nq, k], I int64 [nq, k]) NumPy arrays; slots without a result hold (-FLT_MAX, -1) as FAISS leaves them for
	METRIC_INNER_PRODUCT (the kernels mark them (-inf, -1))."""
	D = np.full((nq, k), PAD_SCORE, dtype=np.float32)
	I = np.full((nq, k), -1, dtype=np.int64)
	I[:, :k_eff] = i.cpu().numpy().astype(np.int64)
	D[:, :k_eff] = np.where(I[:, :k_eff] >= 0, v.cpu().numpy(), PAD_SCORE)
	return D, I


class FlatIPIndex:
	"""Exact maximum-inner-product search on the GPU.  fp32 by default (dense fp32-MFMA GEMM + exact scan);
	dtype="bf16" uses the fused score+top-k kernel when the dimension fits (d <= 512)."""

	def __init__(self, d, dtype="fp32", device=None):
		self.d = d
		self.dtype = dtype
		self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
		self.ntotal = 0
		self._X = None
		self._Xp = self._ids = None  # packed bf16 copy in descending-norm order + row -> id map (built at the first bf16 search)
		self.nprobe = 1  # accepted for FAISS API compatibility

	def add(self, embeds):
		x = torch.as_tensor(np.ascontiguousarray(embeds, dtype=np.float32)).to(self.device)
		assert x.dim() == 2 and x.shape[1] == self.d, f"expected [n, {self.d}] embeddings"
		self._X = x if self._X is None else torch.cat([self._X, x], dim=0)
		self.ntotal = self._X.shape[0]
		self._Xp = None

	def train(self, embeds):  # FAISS API compatibility (flat index needs no training)
		return None

	def search(self, x, k):
		assert self._X is not None, "index is empty"
		q = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(self.device)
		k_eff = min(k, self.ntotal)
		kp = ops.padded_k(self.d)
		if self.dtype == "bf16" and kp is not None and ops.fused_supported(q.shape[0], self.ntotal, kp, k_eff):
			if self._Xp is None:
				from .cur import _norm_sorted_pack
				self._Xp, self._ids = _norm_sorted_pack(self._X, kp)   # largest-norm vectors first: the likeliest maximum inner products
			v, i = ops.score_topk_fused(ops.pack_bf16(q, kp), self._Xp, self.ntotal, k_eff, leading_sample=True, item_ids=self._ids)
		else:
			v, i = ops.score_topk_dense(q, self._X, k_eff)
		return _faiss_pad(v, i, q.shape[0], k, k_eff)


class IVFFlatIPIndex:
	"""faiss.IndexIVFFlat(IndexFlatIP(d), d, nlist, METRIC_INNER_PRODUCT) on the GPU: train (k-means coarse quantiser) / add
	(inverted lists) / search (exact inner products inside the nprobe best lists).  Restated from FAISS' published algorithm and
	defaults: IndexIVF trains its coarse quantiser through Level1Quantizer, whose ClusteringParameters carry niter = 10 (the
	stand-alone Clustering default is 25: pass niter=25 for that), at most 256 training points per centroid, assignment through the
	inner-product quantiser, centroid = mean of its points RENORMALISED TO UNIT L2 NORM (IndexIVF sets cp.spherical = true for
	METRIC_INNER_PRODUCT: fvec_renorm_L2 after every update, and after the re-seeding of empty lists -- with arg-max inner-product
	assignment an unnormalised mean of large-norm points would keep attracting points and the lists would grow unbalanced; spherical=False
	restores the plain means), an empty list re-seeded by splitting a large one.  FAISS' random draws cannot be reproduced, so parity is
	unpinned and the index is judged on recall and list balance.  Results are deterministic for a given seed."""

	def __init__(self, d, nlist, device=None, niter=10, seed=1234, max_points_per_centroid=256, spherical=True, dtype="fp32"):
		self.d, self.nlist = int(d), int(nlist)
		self.spherical = bool(spherical)
		if dtype not in ("fp32", "bf16"):
			raise ValueError(f"dtype = {dtype} not supported")
		self.dtype = dtype                          # "bf16": the batched search scores bf16 copies of the lists and queries on the bf16 matrix cores
		self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
		self.niter, self.seed, self.max_points_per_centroid = niter, seed, max_points_per_centroid
		self.nprobe = 1
		self.batched_from = 256                     # queries per search() call from which the list-grouped MFMA search runs
		self.grouped_call = True                    # batched search through anncur_ivf_search_grouped (False: round 4's sequence of calls)
		self.ntotal = 0
		self.is_trained = False
		self.centroids = None                       # [nlist x d] fp32
		self._X = None                              # vectors in insertion order
		self._Xs = self._offsets = self._ids = None  # vectors in list order (rows zero-padded to a multiple of 16 floats), list bounds, ids
		self._Xs16 = None                            # dtype "bf16": the same rows rounded to bf16 (batched search)
		# row length of the stored vectors / queries, zero padded: a multiple of 16 floats; of 128 elements for bf16 lists (the 128 x 128-tile
		# GEMM of the batched search streams pairs of 64-wide k-tiles)
		self._dp = -(-self.d // 128) * 128 if dtype == "bf16" else -(-self.d // 16) * 16

	def _dev32(self, x):
		x = torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).to(self.device)
		assert x.dim() == 2 and x.shape[1] == self.d, f"expected [n, {self.d}] vectors"
		return x

	def _assign(self, X):
		"""list of every vector = the centroid of maximum inner product (the IndexFlatIP quantiser's top-1)."""
		return ops.score_topk_dense(X, self.centroids, 1).indices[:, 0].contiguous()

	def train(self, x):
		X = self._dev32(x)
		n = X.shape[0]
		if n < self.nlist:
			raise RuntimeError(f"Number of training points ({n}) should be at least as large as number of clusters ({self.nlist})")
		rng = np.random.default_rng(self.seed)
		cap = self.nlist * self.max_points_per_centroid
		if n > cap:                                                # FAISS subsamples the training set
			X = ops.gather_rows(X, np.sort(rng.choice(n, size=cap, replace=False)))
			n = cap
		self.centroids = ops.gather_rows(X, rng.permutation(n)[:self.nlist])   # initial centroids: distinct random training points
		if self.spherical:
			ops.renorm_rows(self.centroids)   # FAISS post-processes the INITIAL centroids too (Clustering::train): the first assignment already runs on unit vectors
		for it in range(self.niter):
			counts, offsets, ids = ops.ivf_build_lists(self._assign(X), self.nlist)
			ops.ivf_list_means(ops.gather_rows(X, ids), offsets, self.centroids)
			self._split_empty(counts.cpu().numpy(), rng)
			if self.spherical:
				ops.renorm_rows(self.centroids)   # (after the split, as FAISS does: the perturbed copies are renormalised too)
		self.is_trained = True

	def _split_empty(self, counts, rng, eps=1.0 / 1024):
		"""An empty list takes over half of a large one: both get that list's centroid, perturbed symmetrically (FAISS'
		split_clusters).  Rows are edited on the host: a handful of d-vectors per iteration at most."""
		empty = np.flatnonzero(counts == 0)
		if empty.size == 0:
			return
		counts = counts.astype(np.float64).copy()
		C = self.centroids.cpu().numpy()
		for ci in empty:
			w = np.maximum(counts - 1.0, 0.0)
			cj = int(rng.choice(self.nlist, p=w / w.sum()))
			sign = np.where(np.arange(self.d) % 2 == 0, 1.0, -1.0).astype(np.float32)
			C[ci] = C[cj] * (1.0 + sign * eps)
			C[cj] = C[cj] * (1.0 - sign * eps)
			counts[ci] = counts[cj] / 2
			counts[cj] -= counts[ci]
		self.centroids.copy_(torch.from_numpy(C))

	def add(self, x):
		assert self.is_trained, "train() first"
		X = self._dev32(x)
		self._X = X if self._X is None else torch.cat([self._X, X], dim=0)
		self.ntotal = self._X.shape[0]
		counts, self._offsets, self._ids = ops.ivf_build_lists(self._assign(self._X), self.nlist)
		self._sizes = counts.cpu().numpy().astype(np.int64)         # list lengths on the host: the batched search's tile worklist
		self._Xs = torch.zeros((self.ntotal, self._dp), dtype=torch.float32, device=self.device)
		self._Xs[:, :self.d] = ops.gather_rows(self._X, self._ids)
		self._Xs16 = ops.convert(self._Xs, torch.bfloat16) if self.dtype == "bf16" else None

	def search_device(self, q, k, profile=None):
		"""search() on DEVICE-RESIDENT queries q [nq x d] fp32 -> (values f32 [nq x k_eff], ids int32 [nq x k_eff]) on the device, no host copy
		(k_eff = min(k, MAX_TOPK); slots without a result hold (-inf, -1)).  What bench.py times as the kernels' own rate."""
		assert self._Xs is not None, "index is empty"
		nprobe = max(1, min(int(self.nprobe), self.nlist))
		probe = ops.score_topk_dense(q, self.centroids, nprobe).indices        # the nprobe lists of largest <q, centroid>
		if self._dp == self.d:
			qp = q if q.is_contiguous() else q.contiguous()
		else:
			qp = torch.zeros((q.shape[0], self._dp), dtype=torch.float32, device=self.device)
			qp[:, :self.d] = q
		k_eff = min(k, ops._lib.MAX_TOPK)
		if q.shape[0] >= self.batched_from:
			# many queries (hard-negative mining: every mention): pairs grouped by list, each list one MFMA GEMM -- a list's vectors
			# are read once per 64 queries instead of once per query (28 -> ~3 ms for 10^4 queries on 10^5 x 768 vectors)
			if profile is None and self.grouped_call and ops.ivf_search_grouped_ok(k_eff, self.nlist):
				# round 5: the whole batched search behind one library call (packed score rows, 128 x 128 bf16 tiles)
				if self._Xs16 is not None:
					return ops.ivf_search_grouped(self._Xs16, self._offsets, self._ids, self._sizes, ops.convert(qp, torch.bfloat16), probe, k_eff)
				return ops.ivf_search_grouped(self._Xs, self._offsets, self._ids, self._sizes, qp, probe, k_eff)
			return ops.ivf_scan_grouped(self._Xs, self._offsets, self._ids, self._sizes, qp, probe, k_eff, lists_bf16=self._Xs16, profile=profile)
		return ops.ivf_scan(self._Xs, self._offsets, self._ids, qp, probe, k_eff)

	def search(self, x, k):
		assert self._Xs is not None, "index is empty"
		q = self._dev32(x)
		v, i = self.search_device(q, k)
		return _faiss_pad(v, i, q.shape[0], k, min(k, ops._lib.MAX_TOPK))


def build_flat_or_ivff_index(embeds, force_exact_search, probe_mult_factor=1, dtype="fp32", device=None):
	LOGGER.info(f"Beginning indexing given {len(embeds)} embeddings")
	if type(embeds) is not np.ndarray:
		embeds = embeds.detach().cpu().numpy() if torch.is_tensor(embeds) else np.array(embeds)
	d, n = embeds.shape[1], embeds.shape[0]
	if n <= 11000 or force_exact_search:    # if the number of embeddings is small, don't approximate (models/nearest_nbr.py:36)
		index = FlatIPIndex(d, dtype=dtype, device=device)
		index.add(embeds)
	else:
		nlist = int(math.floor(math.sqrt(n)))                                   # number of quantized cells (:41)
		nprobe = int(math.floor(math.sqrt(nlist) * probe_mult_factor))          # number of the quantized cells to probe (:44)
		index = IVFFlatIPIndex(d, nlist, device=device, dtype=dtype)
		index.train(embeds)
		index.add(embeds)
		index.nprobe = nprobe
	LOGGER.info("Finished indexing given embeddings")
	return index
