import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.cur import CURApprox, _norm_sorted_pack
from anncur_amd.pinv import pinv_newton_schulz_f64
from anncur_amd.synth import protocol_b
dev = torch.device("cuda")
A_train, A_test = protocol_b(512, 1000, 100000, dev, seed=0)
anc = sorted(np.random.default_rng(0).choice(100000, 256, replace=False))
def T(name, fn, n=5):
	fn(); torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(n): r = fn()
	torch.cuda.synchronize()
	print("%-40s %.3f ms" % (name, 1e3 * (time.perf_counter() - t0) / n), flush=True)
	return r
cols = T("gather_cols", lambda: ops.gather_cols(A_train, anc))
W = T("gather_rows(C, rows) = W", lambda: ops.gather_rows(cols, np.arange(512)))
X, info = pinv_newton_schulz_f64(W, return_info=True); print(info)
U = T("pinv f64 NS", lambda: pinv_newton_schulz_f64(W, return_info=True)[0])
T("pinv numpy host", lambda: torch.from_numpy(np.linalg.pinv(W.float().cpu().numpy())).to(dev))
Et = T("E^T = R^T U^T gemm", lambda: ops.gemm(A_train.t(), U.t()))
T("pack_bf16", lambda: ops.pack_bf16(Et, 256, row_multiple=32))
T("norm sorted pack", lambda: _norm_sorted_pack(Et, 256))
T("torch.equal intersect check", lambda: torch.equal(W, ops.gather_cols(A_train, anc)))
T("CURApprox total", lambda: CURApprox(rows=A_train, cols=cols, row_idxs=np.arange(512), col_idxs=anc, approx_preference="rows", compute_dtype="bf16"))
