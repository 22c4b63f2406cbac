"""Dev tool: the IVF-flat branch at the hard-negative-mining size (utils/data_process.py:343-365 of the reference: every mention queries
the entity index): n vectors, d = 768, nq queries, nlist = floor(sqrt(n)), nprobe = floor(sqrt(nlist)).  Prints build and search times, the
bytes model of the list scan (nq x nprobe x mean list length x d x 4 B) and its rate, recall@k against the exact search on a sample."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.nearest_nbr import build_flat_or_ivff_index, FlatIPIndex
n, d, nq, k = int(os.environ.get("IVF_N", 100000)), 768, int(os.environ.get("IVF_NQ", 10000)), 64
g = torch.Generator().manual_seed(0)
C = torch.randn(200, d, generator=g)
X = (C[torch.randint(0, 200, (n,), generator=g)] + 0.7 * torch.randn(n, d, generator=g)).numpy().astype(np.float32)
Qv = (C[torch.randint(0, 200, (nq,), generator=g)] + 0.7 * torch.randn(nq, d, generator=g)).numpy().astype(np.float32)
t0 = time.perf_counter(); index = build_flat_or_ivff_index(X, force_exact_search=False); torch.cuda.synchronize(); t_build = time.perf_counter() - t0
print(f"n={n} d={d} nlist={index.nlist} nprobe={index.nprobe}: train + add {t_build:.3f} s", flush=True)
q = torch.as_tensor(Qv).cuda()
def timed(fn, reps=3):
	fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
	for _ in range(reps): r = fn()
	torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps, r
t_search, (D, I) = timed(lambda: index.search(Qv, k))
sizes = np.diff(index._offsets.cpu().numpy())
probe = ops.score_topk_dense(q, index.centroids, index.nprobe).indices.cpu().numpy()
scanned = sizes[probe].sum()
bytes_model = float(scanned) * index._dp * 4
print(f"search nq={nq} k={k}: {1e3 * t_search:.2f} ms ({nq / t_search:.0f} queries/s); vectors scanned {scanned / nq:.0f} per query; bytes model {bytes_model / 1e9:.1f} GB -> {bytes_model / t_search / 1e12:.2f} TB/s (L2 / MALL traffic: the lists total {n * index._dp * 4 / 1e6:.0f} MB); flops {2 * scanned * d / t_search / 1e12:.1f} TFLOP/s fp32", flush=True)
flat = FlatIPIndex(d); flat.add(X)
t_flat, (Df, If) = timed(lambda: flat.search(Qv[:2000], k), reps=2)
rec = np.mean([len(set(a) & set(b)) / k for a, b in zip(I[:2000], If)])
print(f"exact flat fp32 search of 2000 queries: {1e3 * t_flat:.2f} ms ({2000 / t_flat:.0f} queries/s); IVF recall@{k} vs exact = {rec:.4f}")
