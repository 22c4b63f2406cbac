"""Dev tool (GPU box): does partitioning the chip's CUs between the MFMA-bound retrieval and the HBM-bound exact scan pay?
Two streams with CU masks (hipExtStreamCreateWithCUMask): the scan alone on n CUs, the fused retrieval alone on the rest, then both at
once -- against the unmasked streams.  Mask layouts tried: which bits of the mask make up "n CUs" (the bit -> CU map is the driver's).
  python scripts/cumask_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.cur import _norm_sorted_pack
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
dev = torch.device("cuda")
torch.cuda.init(); torch.zeros(1, device=dev)

def masked_stream(bits):
	words = (ctypes.c_uint32 * 8)(*[int(sum(1 << b for b in range(32) if bits[32 * w + b])) for w in range(8)])
	s = ctypes.c_void_p()
	rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
	assert rc == 0, rc
	return torch.cuda.ExternalStream(s.value, device=dev)

Q, I, K, k = 10000, 100000, 256, 100
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, I, generator=g, device=dev)
X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
Etp, ids = _norm_sorted_pack(E.t().contiguous().float(), 256)
Xp = ops.pack_bf16(X, 256)
A = torch.randn(Q, I, generator=g, device=dev).bfloat16()
del Z, E
out_scan = (torch.empty(Q, k, dtype=torch.float32, device=dev), torch.empty(Q, k, dtype=torch.int32, device=dev))

def timed(fn_list, streams, reps=30):
	"""each fn on its stream, all started together, reps times; wall time per rep (host-synchronised)"""
	for fn, s in zip(fn_list, streams):
		with torch.cuda.stream(s): fn()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(reps):
		for fn, s in zip(fn_list, streams):
			with torch.cuda.stream(s): fn()
	torch.cuda.synchronize()
	return (time.perf_counter() - t0) / reps * 1e3

scan = lambda: ops.rowwise_topk(A, k, out=out_scan)
retr = lambda: ops.score_topk_fused(Xp, Etp, I, k, leading_sample=True, item_ids=ids)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
print("unmasked: scan %.3f ms, retrieval %.3f ms, both on two streams %.3f ms" % (timed([scan], [s0]), timed([retr], [s0]), timed([retr, scan], [s0, s1])), flush=True)

# (scripts/cumask_map.py: the first n bits of the mask, n a multiple of 8, are n / 8 CUs on each of the 8 XCDs; sparse masks are ignored)
for n_scan in (64, 80, 88, 96, 104, 112, 120, 128):
	bs = np.zeros(256, dtype=bool); bs[:n_scan] = True
	ss, sr = masked_stream(bs), masked_stream(~bs)
	t_s, t_r = timed([scan], [ss]), timed([retr], [sr])
	t_b = timed([retr, scan], [sr, ss])
	print("scan on %3d CUs: scan alone %.3f ms (%.2f TB/s), retrieval alone on %3d CUs %.3f ms, both %.3f ms" % (n_scan, t_s, Q * I * 2 / t_s / 1e9, 256 - n_scan, t_r, t_b), flush=True)

print("-- scan masked, retrieval on an unmasked stream (the dynamic tile schedule takes what the scan leaves)")
for n_scan in (64, 96, 128, 160):
	bs = np.zeros(256, dtype=bool); bs[:n_scan] = True
	ss = masked_stream(bs)
	print("scan on %3d CUs, retrieval unmasked: both %.3f ms" % (n_scan, timed([retr, scan], [s0, ss])), flush=True)
	print("    scan issued first                : both %.3f ms" % (timed([scan, retr], [ss, s0])), flush=True)
