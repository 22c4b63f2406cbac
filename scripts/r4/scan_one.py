"""Launches of the exact scan on short and long iid rows (for rocprofv3 --pmc: instruction counts per launch, in launch order)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
dev = torch.device("cuda", 0)
Q = 10000
for I in (4096, 8192, 16384, 32768, 65536, 100000):
	G = torch.randn(Q, I, device=dev).to(torch.bfloat16)
	for _ in range(3): ops.rowwise_topk(G, 100)
torch.cuda.synchronize()
