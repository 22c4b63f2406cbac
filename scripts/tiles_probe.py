"""Dev tool (experiments build): tiles swept and loop start / end per workgroup of the last sweep launch (16x16x32 kernel).
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so python scripts/tiles_probe.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["ANNCUR_DEBUG_STAMPS"] = "1"
from anncur_amd import ops, _lib
from anncur_amd.cur import _norm_sorted_pack
dev = torch.device("cuda")
Q, I, K, k = 10000, 100000, 256, 100
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, I, generator=g, device=dev)
X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
Etp, ids = _norm_sorted_pack(E.t().contiguous().float(), 256)
Xp = ops.pack_bf16(X, 256)
lib = _lib.load()
FLAGS = {"mfma16": True}
for _ in range(20): ops.score_topk_fused(Xp, Etp, I, k, leading_sample=True, item_ids=ids, **FLAGS)
torch.cuda.synchronize()
ops.score_topk_fused_timed(Xp, Etp, I, k, leading_sample=True, item_ids=ids, **FLAGS)
raw = (ctypes.c_ulonglong * (8 * 8192))()
lib.anncur_debug_sweep_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.anncur_debug_sweep_phases(raw) == 0
ph = np.frombuffer(raw, dtype=np.uint64).astype(np.int64).reshape(8192, 8)
raw2 = (ctypes.c_ulonglong * (5 * 8192))()
lib.anncur_debug_sweep_raw.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
assert lib.anncur_debug_sweep_raw(raw2) == 0
a = np.frombuffer(raw2, dtype=np.uint64).astype(np.float64)
tl = a[2 * 8192:].reshape(8192, 3); ok = tl[:, 2] > 0; n = int(ok.sum())
t0 = tl[ok, 0].min()
tiles = ph[0:4 * n:4, 5]
print("workgroups", n, "tiles swept: total", tiles.sum(), "min", tiles.min(), "p50", np.median(tiles), "max", tiles.max())
q8, r8 = n // 8, n % 8; b = np.arange(n); x = b % 8; l = b // 8
wid = np.where(x < r8, x * (q8 + 1), r8 * (q8 + 1) + (x - r8) * q8) + l
n_rb = 40; split, rb = wid // n_rb, wid % n_rb
order = np.argsort(tl[:n, 0])
print("last 12 workgroups to enter: (block, rb, split, entry us, loop start, loop end, tiles)")
for i in order[-12:]: print("  ", i, rb[i], split[i], "%.1f %.1f %.1f" % tuple((tl[i] - t0) / 100), tiles[i])
for r in (0, 1, int(rb[order[-1]])):
	m = rb == r
	print("row block", r, "tiles per split", tiles[m][np.argsort(split[m])], "sum", tiles[m].sum(), "loop end", np.round((tl[:n, 2][m][np.argsort(split[m])] - t0) / 100))
