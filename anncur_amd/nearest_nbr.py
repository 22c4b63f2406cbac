"""Flat exact inner-product index with the FAISS calling convention the reference relies on
(models/nearest_nbr.py:24-55; call sites utils/data_process.py:343-351,393-397 and
eval/run_cross_encoder_w_binenc_retriever_zeshel.py:125,147):

    index = build_flat_or_ivff_index(embeds, force_exact_search)
    D, I = index.search(queries, k)      # D float32 [nq, k] descending, I int64 [nq, k]   (NumPy, like FAISS)

FAISS itself is a third-party dependency that the reference neither vendors nor pins: parity at this boundary is UNPINNED and
judged against torch.topk(q @ X^T) (tests/).  The reference switches to an IVF-flat index above 11 000 vectors; that branch is
approximate by construction.  This build serves every size with the exact search (recall >= any IVF setting) and says so once.
"""
import logging

import numpy as np
import torch

from . import ops

LOGGER = logging.getLogger(__name__)


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
		D = np.full((q.shape[0], k), -np.inf, dtype=np.float32)   # FAISS pads missing results with -inf / -1
		I = np.full((q.shape[0], k), -1, dtype=np.int64)
		D[:, :k_eff] = v.cpu().numpy()
		I[:, :k_eff] = i.cpu().numpy().astype(np.int64)
		return D, I


_warned = False


def build_flat_or_ivff_index(embeds, force_exact_search, probe_mult_factor=1, dtype="fp32", device=None):
	global _warned
	LOGGER.info(f"Beginning indexing given {len(embeds)} embeddings")
	if type(embeds) is not np.ndarray:
		embeds = embeds.detach().cpu().numpy() if torch.is_tensor(embeds) else np.array(embeds)
	d, n = embeds.shape[1], embeds.shape[0]
	if n > 11000 and not force_exact_search and not _warned:
		LOGGER.info("reference would build an approximate IndexIVFFlat here (nlist=floor(sqrt(n))); this build searches exactly instead")
		_warned = True
	index = FlatIPIndex(d, dtype=dtype, device=device)
	index.add(embeds)
	LOGGER.info("Finished indexing given embeddings")
	return index
