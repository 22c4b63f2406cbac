"""Dev tool: exact-scan variants vs plain torch streaming on the headline matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anncur_amd import ops
dev = torch.device("cuda")
Q, I = 10000, 100000
A = torch.randn(Q, I, device=dev).bfloat16()
def t(name, fn, n=10):
	fn(); torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(n): fn()
	e1.record(); torch.cuda.synchronize()
	ms = e0.elapsed_time(e1) / n
	print("%-28s %.4f ms  %.0f GB/s (read)" % (name, ms, Q * I * 2 / ms / 1e6), flush=True)
t("torch.amax(dim=1)", lambda: torch.amax(A, dim=1))
t("torch clone (r+w)", lambda: A.clone())
t("torch sum", lambda: A.sum())
for k in (1, 10, 100, 128):
	t("scan wave k=%d" % k, lambda: ops.rowwise_topk(A, k))
os.environ["ANNCUR_DEBUG_BLOCK_SCAN"] = "1"
for k in (1, 100):
	t("scan block k=%d" % k, lambda: ops.rowwise_topk(A, k))
t("torch.topk k=100", lambda: torch.topk(A, 100, dim=1), n=2)
