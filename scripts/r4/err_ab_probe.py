"""error_lds_kernel at cfg2 size on the bench's own operands (index + synthetic matrix), one library per process (ANNCUR_LIB)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
from anncur_amd.cur import CURApprox   # noqa: E402
import bench   # noqa: E402
dev = torch.device("cuda", 0)
cfg = bench.CONFIGS["cfg2"]
A_train, A = bench.synth_device(cfg, dev, 0, row_seed=None)
rng = np.random.default_rng(0)
anc = sorted(rng.choice(cfg["I"], size=cfg["Ki"], replace=False))
anc_dev = ops.as_index(anc, dev)
cur = CURApprox(rows=A_train, cols=ops.gather_cols(A_train, anc_dev), row_idxs=np.arange(cfg["Kq"]), col_idxs=anc, approx_preference="rows", compute_dtype="bf16")
Xq = ops.gather_cols(A, anc_dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
out = []
for rep in range(3):
	for _ in range(3): ops.approx_error_packed(Xq, cur._Etp, A, cfg["I"])
	ev[0].record()
	for _ in range(20): ops.approx_error_packed(Xq, cur._Etp, A, cfg["I"])
	ev[1].record(); torch.cuda.synchronize()
	out.append(ev[0].elapsed_time(ev[1]) / 20)
print(os.path.basename(os.environ.get("ANNCUR_LIB", "product")), " ".join(f"{x:.4f}" for x in out), "ms")
