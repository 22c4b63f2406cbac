"""IVF-flat batched search on device-resident queries at bench.py's side-line size (n = 100 000 x 768, nq = 10 000, k = 64): list-size
statistics, the call's time (HIP events) and -- under `rocprofv3 --kernel-trace --stats -- python3 scripts/r5/ivf_probe.py` -- its kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.nearest_nbr import build_flat_or_ivff_index

def main():
	dev = torch.device("cuda", 0)
	n, d, nq, k = 100000, 768, 10000, 64
	g = torch.Generator(device=dev).manual_seed(1234 + 99)
	C = torch.randn(200, d, generator=g, device=dev)
	X = (C[torch.randint(0, 200, (n,), generator=g, device=dev)] + 0.7 * torch.randn(n, d, generator=g, device=dev)).cpu().numpy()
	Qv = (C[torch.randint(0, 200, (nq,), generator=g, device=dev)] + 0.7 * torch.randn(nq, d, generator=g, device=dev)).cpu().numpy()
	reps = int(os.environ.get("REPS", "10"))
	for dtype in os.environ.get("DTYPES", "bf16").split(","):
		index = build_flat_or_ivff_index(X, force_exact_search=False, dtype=dtype)
		s = index._sizes
		print(f"{dtype}: nlist {index.nlist} nprobe {index.nprobe}  list sizes min {s.min()} median {int(np.median(s))} mean {s.mean():.0f} max {s.max()}", flush=True)
		q = torch.as_tensor(Qv).to(dev)
		probe = ops.score_topk_dense(q, index.centroids, index.nprobe).indices.cpu().numpy()
		per_q = s[probe].sum(1)
		print(f"   vectors scanned per query: mean {per_q.mean():.0f} max {per_q.max()}  (nprobe x lmax = {index.nprobe * int(s.max())})", flush=True)
		pairs = np.bincount(probe.reshape(-1), minlength=index.nlist)
		for T in (64, 128, 256):
			nt = int((-(-pairs // T) * -(-s // T)).sum())
			print(f"   tiles of {T} x {T}: {nt}  tile flops / algorithmic flops = {nt * T * T / float((pairs * s).sum()):.3f}", flush=True)
		if dtype == "bf16" and os.environ.get("GEMM_ONLY", "1") == "1":
			qb = ops.convert(q, torch.bfloat16)
			pr = torch.as_tensor(probe).to(dev)
			t = {}
			for skip in (False, True):
				for _ in range(3): ops.ivf_search_grouped(index._Xs16, index._offsets, index._ids, index._sizes, qb, pr, k, _skip_gemm=skip)
				torch.cuda.synchronize()
				ms = []
				for _ in range(reps):
					e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
					e0.record(); ops.ivf_search_grouped(index._Xs16, index._offsets, index._ids, index._sizes, qb, pr, k, _skip_gemm=skip); e1.record(); torch.cuda.synchronize()
					ms.append(e0.elapsed_time(e1))
				t[skip] = float(np.median(ms))
			flops = 2.0 * float((pairs * s).sum()) * d
			print(f"   grouped call {t[False]:.3f} ms, without its tile launch {t[True]:.3f} ms -> tile kernel {t[False] - t[True]:.3f} ms = {flops / (t[False] - t[True]) / 1e9:.0f} TFLOP/s algorithmic", flush=True)
		res = {}
		for grouped in (os.environ.get("MODES", "0,1").split(",")):
			index.grouped_call = grouped == "1"
			for _ in range(3): index.search_device(q, k)
			torch.cuda.synchronize()
			ms = []
			for _ in range(reps):
				e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
				e0.record(); v, i = index.search_device(q, k); e1.record(); torch.cuda.synchronize()
				ms.append(e0.elapsed_time(e1))
			print(f"   grouped_call={index.grouped_call}: search_device median {np.median(ms):.3f} ms  min {min(ms):.3f}", flush=True)
			print("      checksum", float(v[torch.isfinite(v)].sum()), int(i.to(torch.int64).sum()), flush=True)
			res[grouped] = (v, i)
		if len(res) == 2:
			print("   values equal:", bool(torch.equal(res["0"][0], res["1"][0])), " ids equal:", bool(torch.equal(res["0"][1], res["1"][1])), flush=True)

if __name__ == "__main__":
	main()
