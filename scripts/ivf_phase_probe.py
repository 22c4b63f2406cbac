import os, sys, time
sys.path.insert(0, os.getcwd())
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
	os.environ.setdefault(_v, "8")
import numpy as np, torch
import bench
from anncur_amd import ops
from anncur_amd.nearest_nbr import build_flat_or_ivff_index
device = torch.device("cuda")
r = bench.ivf_sideline(device, 0)
print({k: r[k] for k in ("build_s", "search_ms", "queries_per_s")})
# phases
n, d, nq, k = 100000, 768, 10000, 64
g = torch.Generator(device=device).manual_seed(99)
C = torch.randn(200, d, generator=g, device=device)
X = (C[torch.randint(0, 200, (n,), generator=g, device=device)] + 0.7 * torch.randn(n, d, generator=g, device=device)).cpu().numpy()
Qv = (C[torch.randint(0, 200, (nq,), generator=g, device=device)] + 0.7 * torch.randn(nq, d, generator=g, device=device)).cpu().numpy()
index = build_flat_or_ivff_index(X, force_exact_search=False)
print("sizes: max", index._sizes.max(), "mean", index._sizes.mean(), "min", index._sizes.min())
def T(name, fn, reps=3):
	fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
	for _ in range(reps): r = fn()
	torch.cuda.synchronize(); print("%-30s %.3f ms" % (name, 1e3 * (time.perf_counter() - t0) / reps)); return r
T("search", lambda: index.search(Qv, k))
q = T("H2D queries", lambda: index._dev32(Qv))
probe = T("probe (dense topk)", lambda: ops.score_topk_dense(q, index.centroids, index.nprobe).indices)
qp = torch.zeros((nq, index._dp), dtype=torch.float32, device=device); qp[:, :d] = q
v, i = T("ivf_scan_grouped", lambda: ops.ivf_scan_grouped(index._Xs, index._offsets, index._ids, index._sizes, qp, probe, k))
T("D2H + pad", lambda: (v.cpu().numpy(), i.cpu().numpy()))
