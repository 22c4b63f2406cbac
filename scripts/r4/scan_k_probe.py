"""The exact scan by k on the whole chip (is the k = 10 figure of scan_fixed_probe.py an artefact of being the first launch on a fresh matrix?)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
dev = torch.device("cuda", 0)
Q, I = 10000, 100000
X = torch.randn(Q, I, device=dev).to(torch.bfloat16)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for k in (100, 10, 1, 10, 50, 100, 128, 10):
	for _ in range(3): ops.rowwise_topk(X, k)
	ev[0].record()
	for _ in range(10): ops.rowwise_topk(X, k)
	ev[1].record(); torch.cuda.synchronize()
	print(f"k={k:4d}  {ev[0].elapsed_time(ev[1]) / 10:.4f} ms", flush=True)
