"""Synthetic protocol-B score matrices on the device: shared item factors Z, low rank + noise.
Data plumbing for bench.py / tests (torch RNG); not part of the compute path."""
import torch


def protocol_b(n_train, n_test, n_items, device, seed=0, rank=64, noise=0.05, dtype=torch.bfloat16, chunk=2048, row_seed=None):
	"""row_seed: a second seed for the ROW factors and the noise.  The item factors Z always come from `seed`, so processes that
	pass the same seed and different row seeds (the ranks of a row-sharded run) draw different queries of ONE score model."""
	g = torch.Generator(device=device).manual_seed(seed)
	Z = torch.randn(rank, n_items, generator=g, device=device)
	if row_seed is not None:
		g = torch.Generator(device=device).manual_seed(row_seed)

	def make(n):
		out = torch.empty(n, n_items, dtype=dtype, device=device)
		for s in range(0, n, chunk):
			e = min(n, s + chunk)
			out[s:e] = (torch.randn(e - s, rank, generator=g, device=device) @ Z / rank ** 0.5
						+ noise * torch.randn(e - s, n_items, generator=g, device=device)).to(dtype)
		return out

	return make(n_train), make(n_test)
