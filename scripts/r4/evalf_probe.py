"""round-4 probe: one grid cell of entry point A at cfg2 size -- anncur_eval_fused against the two-kernel route (fused top-k + error kernel).
   python scripts/r4/evalf_probe.py [k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.cur import CURRowIndex
from anncur_amd.synth import protocol_b
dev = torch.device("cuda")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
Q, I = 10000, 100000
A_train, A_test = protocol_b(512, Q, I, dev, seed=0)
anc = sorted(np.random.default_rng(0).choice(I, 256, replace=False))
index = CURRowIndex(A_train, anc)
X = ops.gather_cols(A_test, anc)
def timed(fn, n=20):
	for _ in range(3): fn()
	torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	ts = []
	for _ in range(n):
		e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
	return float(np.median(ts))
two = lambda: (ops.score_topk_fused(X, index._Etp_sorted, I, k, leading_sample=True, item_ids=index._item_ids), ops.approx_error_packed(X, index._Etp, A_test, I))
two_item = lambda: (ops.score_topk_fused(X, index._Etp, I, k), ops.approx_error_packed(X, index._Etp, A_test, I))
one = lambda: ops.eval_fused(X, index._Etp, A_test, I, k)
t_topk = timed(lambda: ops.score_topk_fused(X, index._Etp_sorted, I, k, leading_sample=True, item_ids=index._item_ids))
t_err = timed(lambda: ops.approx_error_packed(X, index._Etp, A_test, I))
t_scan = timed(lambda: ops.rowwise_topk(A_test, 10))
print(f"k_retvr = {k}: fused top-k (norm order, default body) {t_topk:.4f} ms, error kernel {t_err:.4f} ms, exact scan {t_scan:.4f} ms")
print(f"  two-kernel route (norm-ordered retrieval + error kernel): {timed(two):.4f} ms;  item-ordered retrieval + error kernel: {timed(two_item):.4f} ms")
print(f"  anncur_eval_fused (one sweep):                            {timed(one):.4f} ms")
(tk, err, nrm) = one(); (tk2, (err2, nrm2)) = two()
print("  index sets equal:", bool(torch.equal(torch.sort(tk.indices, 1).values, torch.sort(tk2.indices, 1).values)), " max |score diff|", float((tk.values - tk2.values).abs().max()),
	  " err rel diff", float(((err - err2).abs() / err2).max()))
