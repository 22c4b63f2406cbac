"""Fixed cost per row of the exact scan (seed select + final select/sort): short rows, 10 000 of them, whole chip and 96 CUs."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts", "r4"))
from anncur_amd import ops   # noqa: E402
from timeline_probe import masked_stream   # noqa: E402

def main():
	dev = torch.device("cuda", 0)
	Q = 10000
	streams = [("256 CUs", torch.cuda.Stream(device=dev), 256), ("96 CUs", masked_stream(0, 96), 96)]
	ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
	for I in (4096, 8192, 16384, 32768, 100000):
		X = torch.randn(Q, I, device=dev).to(torch.bfloat16)
		for k in (10, 100):
			for sname, st, n in streams:
				with torch.cuda.stream(st):
					for _ in range(2): ops.rowwise_topk(X, k)
					ev[0].record(st)
					for _ in range(5): ops.rowwise_topk(X, k)
					ev[1].record(st)
				torch.cuda.synchronize()
				ms = ev[0].elapsed_time(ev[1]) / 5
				waves = n * 16
				print(f"I={I:6d} k={k:3d} {sname:8s} {ms:.4f} ms   per row (at {waves} rows in flight): {1e3 * ms / (Q / waves):.1f} us   {X.numel() * 2 / ms / 1e6 / n:.1f} GB/s per CU", flush=True)

if __name__ == "__main__":
	main()
