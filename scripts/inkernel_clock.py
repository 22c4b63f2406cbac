"""Dev tool (experiments build): the clock the chip holds INSIDE the sweep kernel (MI355X guide, 'DVFS give-back' (6)):
Delta s_memtime / Delta s_memrealtime x 100 MHz around the tile loop, median over the workgroups of the last sweep launch, after
>= 2 s of back-to-back calls on random data.  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so python scripts/inkernel_clock.py [K]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["ANNCUR_DEBUG_STAMPS"] = "1"
from anncur_amd import ops, _lib
from anncur_amd.cur import _norm_sorted_pack
dev = torch.device("cuda")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Q, I, k = (10000, 100000, 100) if K <= 256 else (6250, 1000000, 100)
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, I, generator=g, device=dev)
X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
Kp = ops.padded_k(K)
Etp, ids = _norm_sorted_pack(E.t().contiguous().float(), Kp)
Xp = ops.pack_bf16(X, Kp)
del Z, E
FLAGS = {"mfma16": True} if os.environ.get("PH_MFMA16") else {}
lib = _lib.load()
lib.anncur_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 2.5:
	for _ in range(20): ops.score_topk_fused(Xp, Etp, I, k, leading_sample=True, item_ids=ids, **FLAGS)
	torch.cuda.synchronize(); n += 20
(_, _), ms = ops.score_topk_fused_timed(Xp, Etp, I, k, leading_sample=True, item_ids=ids, **FLAGS)
ghz, us, nwg = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
assert lib.anncur_debug_read_stamps(ctypes.byref(ghz), ctypes.byref(us), ctypes.byref(nwg)) == 0
tf = 2.0 * Q * Kp * I / (ms[4] * 1e-3) / 1e12
peak_at_clock = 1024 * 1024 * ghz.value * 1e9 / 1e12   # 1024 SIMDs x 1024 flop / clk
tl = (ctypes.c_double * 6)()
lib.anncur_debug_sweep_timeline.argtypes = [ctypes.POINTER(ctypes.c_double)]
if lib.anncur_debug_sweep_timeline(tl) == 0:
	print(f"timeline of the last sweep launch (us after the first workgroup's entry; median / last workgroup): entry {tl[0]:.1f} / {tl[1]:.1f}, "
		  f"tile loop starts {tl[2]:.1f} / {tl[3]:.1f}, tile loop ends {tl[4]:.1f} / {tl[5]:.1f}; kernel (HIP events incl. launch) {1e3 * ms[4] / ms[5]:.1f} us per launch")
print(f"Kp={Kp}: {n} calls in 2.5 s; last sweep launch: in-kernel clock {ghz.value:.3f} GHz (median of {nwg.value} workgroups, loop {us.value:.1f} us); "
	  f"sweep {tf:.0f} TFLOP/s = {tf / 2500:.3f} of the 2.5 PFLOP/s spec peak = {tf / peak_at_clock:.3f} of the {peak_at_clock:.0f} TFLOP/s the matrix pipes deliver at that clock")

if os.environ.get("ANNCUR_CLOCK_DETAIL"):
	raw = (ctypes.c_ulonglong * (5 * 8192))()
	lib.anncur_debug_sweep_raw.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
	assert lib.anncur_debug_sweep_raw(raw) == 0
	a = np.frombuffer(raw, dtype=np.uint64).astype(np.float64)
	tl = a[2 * 8192:].reshape(8192, 3); ok = tl[:, 2] > 0; n = int(ok.sum())
	dur = (tl[ok, 2] - tl[ok, 1]) / 100.0   # us
	t0_ = tl[ok, 0].min()
	print("entry  us after the first: p10 %.1f p50 %.1f p90 %.1f max %.1f | loop start: p50 %.1f max %.1f | loop end: p10 %.1f p50 %.1f p90 %.1f max %.1f" % (
		*[np.percentile(tl[ok, 0] - t0_, q) / 100 for q in (10, 50, 90, 100)], *[np.percentile(tl[ok, 1] - t0_, q) / 100 for q in (50, 100)], *[np.percentile(tl[ok, 2] - t0_, q) / 100 for q in (10, 50, 90, 100)]))
	b = np.arange(8192)[ok]
	BQ = 256 if Kp <= 256 else 128   # queries per workgroup (two / one 32-query sub-tiles per wave)
	n_rb = (Q + BQ - 1) // BQ; S = n // n_rb
	q8, r8 = n // 8, n % 8; x = b % 8; l = b // 8
	wid = np.where(x < r8, x * (q8 + 1), r8 * (q8 + 1) + (x - r8) * q8) + l     # xcd_remap
	split, rb = wid // n_rb, wid % n_rb
	print(f"{n} workgroups = {n_rb} row blocks x {S} splits; loop us: min {dur.min():.1f} p10 {np.percentile(dur,10):.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f} max {dur.max():.1f}")
	print("by split   :", " ".join(f"{dur[split == s_].mean():.0f}" for s_ in range(S)))
	print("by row blk :", " ".join(f"{dur[rb == r_].mean():.0f}" for r_ in range(n_rb)))
	print("by XCD     :", " ".join(f"{dur[x == i].mean():.0f}" for i in range(8)))
