"""A few launches of the error kernel and of anncur_eval_fused at cfg2 size on the bench's operands (for rocprofv3 --pmc FETCH_SIZE)."""
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
for _ in range(3): ops.approx_error_packed(Xq, cur._Etp, A, cfg["I"])
for _ in range(3): ops.eval_fused(Xq, cur._Etp, A, cfg["I"], cfg["k_retvr"])
for _ in range(3): ops.score_topk_fused(Xq, cur._Etp_sorted, cfg["I"], cfg["k_retvr"], leading_sample=True, item_ids=cur._item_ids)
torch.cuda.synchronize()
