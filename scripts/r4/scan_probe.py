"""The exact scan (rowwise_topk_wave_kernel) on the whole chip and on CU-masked streams, by input: bench data (low rank + noise), iid
gaussian, constant rows (nothing passes the prefilter after the seed).  Against scripts/r4/stream_probe (bare streaming, same rows)."""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402
import bench   # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "scripts", "r4"))
from timeline_probe import masked_stream   # noqa: E402

def main():
	dev = torch.device("cuda", 0)
	cfg = bench.CONFIGS["cfg2"]
	_, A = bench.synth_device(cfg, dev, 0, row_seed=None)
	Q, I = A.shape
	datas = {"bench (low rank + noise)": A, "iid gaussian": torch.randn(Q, I, device=dev).to(torch.bfloat16), "constant": torch.full((Q, I), 0.5, device=dev, dtype=torch.bfloat16),
			 "negative constant (every element ties with the threshold)": torch.full((Q, I), -0.5, device=dev, dtype=torch.bfloat16),
			 "negative, few distinct values": (-(torch.randint(1, 40, (Q, I), device=dev).float()) / 8).to(torch.bfloat16),
			 "ascending (every step passes)": torch.arange(I, device=dev, dtype=torch.float32).mul_(1e-3).to(torch.bfloat16).repeat(Q, 1)}
	streams = [("256 CUs", torch.cuda.Stream(device=dev), 256)] + [(f"{n} CUs", masked_stream(0, n), n) for n in (128, 96, 64)]
	ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
	k = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["k"]
	for name, X in datas.items():
		for sname, st, n in streams:
			with torch.cuda.stream(st):
				for _ in range(2): ops.rowwise_topk(X, k)
				ev[0].record(st)
				for _ in range(5): ops.rowwise_topk(X, k)
				ev[1].record(st)
			torch.cuda.synchronize()
			ms = ev[0].elapsed_time(ev[1]) / 5
			print(f"{name:32s} {sname:8s} k={k}  {ms:.3f} ms  {X.numel() * 2 / ms / 1e9:.2f} TB/s  {X.numel() * 2 / ms / 1e6 / n:.1f} GB/s per CU", flush=True)

if __name__ == "__main__":
	main()
