"""Dev tool: duration of the exact scan (anncur_rowwise_topk, one wave per row) against the number of rows, I = 100 000 bf16, k = 100:
the fixed cost of a launch and the round structure (rows in flight on the chip) show in the steps of the curve."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anncur_amd import ops
dev = torch.device("cuda")
I = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(20480, I, generator=g, device=dev).bfloat16()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rows in (256, 1024, 2048, 3072, 4096, 5120, 6144, 8192, 10000, 12288, 16384, 20480):
	a = A[:rows]
	for _ in range(3): ops.rowwise_topk(a, 100)
	torch.cuda.synchronize()
	ev0.record()
	for _ in range(10): ops.rowwise_topk(a, 100)
	ev1.record(); torch.cuda.synchronize()
	ms = ev0.elapsed_time(ev1) / 10
	print("rows %6d  %.4f ms  %.0f GB/s  %.1f ns/row" % (rows, ms, rows * I * 2 / ms / 1e6, ms * 1e6 / rows), flush=True)
