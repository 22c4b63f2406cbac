"""anncur_overlap_counts at the bench step's shape (10 000 queries, two lists of 100 ids, four prefix pairs): time per call."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
g = torch.Generator(device="cuda").manual_seed(1)
Q, k, I = 10000, 100, 100000
a = torch.stack([torch.randperm(I, device="cuda", generator=g)[:k] for _ in range(64)]).repeat(Q // 64 + 1, 1)[:Q].to(torch.int32)
b = a.clone(); b[:, ::3] = torch.randint(0, I, (Q, (k + 2) // 3), device="cuda", generator=g, dtype=torch.int32)
pairs = [(1, 100), (10, 100), (50, 100), (100, 100)]
ref = ops.overlap_counts(a, b, pairs)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for rep in range(3):
	for _ in range(50): ops.overlap_counts(a, b, pairs)
	ev[0].record()
	for _ in range(200): out = ops.overlap_counts(a, b, pairs)
	ev[1].record(); torch.cuda.synchronize()
	print(f"overlap_counts {ev[0].elapsed_time(ev[1]) / 200 * 1e3:.1f} us per call   checksum {int(out.sum())}  equal to first: {bool(torch.equal(out, ref))}", flush=True)
