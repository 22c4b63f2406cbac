"""Synthetic protocol-B score matrices on the device: shared item factors Z, low rank + noise.
Data plumbing for bench.py / tests (torch RNG); not part of the compute path."""
import torch


def protocol_b(n_train, n_test, n_items, device, seed=0, rank=64, noise=0.05, dtype=torch.bfloat16, chunk=2048):
	g = torch.Generator(device=device).manual_seed(seed)
	Z = torch.randn(rank, n_items, generator=g, device=device)

	def make(n):
		out = torch.empty(n, n_items, dtype=dtype, device=device)
		for s in range(0, n, chunk):
			e = min(n, s + chunk)
			out[s:e] = (torch.randn(e - s, rank, generator=g, device=device) @ Z / rank ** 0.5
						+ noise * torch.randn(e - s, n_items, generator=g, device=device)).to(dtype)
		return out

	return make(n_train), make(n_test)
