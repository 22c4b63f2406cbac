"""The exact scan at cfg2 size on the bench's matrix, whole chip and 96 CUs, one library per process (ANNCUR_LIB), warm."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts", "r4"))
from anncur_amd import ops   # noqa: E402
import bench   # noqa: E402
from timeline_probe import masked_stream   # noqa: E402
dev = torch.device("cuda", 0)
_, A = bench.synth_device(bench.CONFIGS["cfg2"], dev, 0, row_seed=None)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
res = {}
streams = [("256", torch.cuda.Stream(device=dev)), ("96", masked_stream(0, 96))]
for rep in range(4):
	for name, st in streams:
		with torch.cuda.stream(st):
			for _ in range(20): ops.rowwise_topk(A, 100)
			ev[0].record(st)
			for _ in range(20): ops.rowwise_topk(A, 100)
			ev[1].record(st)
		torch.cuda.synchronize()
		res.setdefault(name, []).append(ev[0].elapsed_time(ev[1]) / 20)
print(os.path.basename(os.environ.get("ANNCUR_LIB", "product")), " | ".join(f"{n} CUs: " + " ".join(f"{x:.4f}" for x in v) for n, v in res.items()), "ms")
