"""Dev tool (experiments build): where a sweep wave's cycles go, per step of the tile loop -- {ticket read + DMA issue, ring drain, MFMA / filter
section, vmcnt wait, barrier} summed over the loop, per wave, for the stamped sweep launch (ANNCUR_DEBUG_STAMP_STAGE picks the stage).
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so python scripts/sweep_phases.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["ANNCUR_DEBUG_STAMPS"] = "1"
from anncur_amd import ops, _lib
from anncur_amd.cur import _norm_sorted_pack
dev = torch.device("cuda")
Q, I, K, k = 10000, 100000, 256, int(os.environ.get("PH_K", "100"))
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, I, generator=g, device=dev)
X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
Etp, ids = _norm_sorted_pack(E.t().contiguous().float(), 256)
Xp = ops.pack_bf16(X, 256)
lib = _lib.load()
lib.anncur_debug_sweep_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
FLAGS = {"mfma16": True} if os.environ.get("PH_MFMA16") else ({"ring": True} if os.environ.get("PH_RING") else {})
plan = ops.fused_plan(Q, I, 256, k, leading_sample=True, **FLAGS)
print("plan", plan)
for stage in range(plan["n_stages"]):
	os.environ["ANNCUR_DEBUG_STAMP_STAGE"] = str(stage)
	for _ in range(30): ops.score_topk_fused(Xp, Etp, I, k, leading_sample=True, item_ids=ids, **FLAGS)
	torch.cuda.synchronize()
	(_, _), ms = ops.score_topk_fused_timed(Xp, Etp, I, k, leading_sample=True, item_ids=ids, **FLAGS)
	raw = (ctypes.c_ulonglong * (8 * 8192))()
	assert lib.anncur_debug_sweep_phases(raw) == 0
	a = np.frombuffer(raw, dtype=np.uint64).astype(np.float64).reshape(8192, 8)[:, :5]
	a = a[a.sum(1) > 0]
	tot = a.sum(1)
	ring = plan["stage_pred"][stage] == 5   # 8-wave workgroups of 512 queries with the flag-synchronised tile ring (score16r.hpp)
	bq = 512 if ring else 256
	tiles = (plan["stage_end"][stage] - ([0] + plan["stage_end"])[stage]) * ((Q + bq - 1) // bq) * (bq // 64) / len(a)   # mean tiles per wave
	names = (["sequence + FREE wait + DMA issue + landed", "queue drain", "FULL wait", "MFMA/filter section", "ticket publish + poll wait"] if ring
			 else ["ticket+DMA issue", "ring drain", "MFMA/filter section", "vmcnt wait", "barrier"])
	print(f"stage {stage}: launch {1e3 * ms[6 + stage]:.1f} us, {len(a)} waves, mean loop {tot.mean():.0f} cycles (p10 {np.percentile(tot, 10):.0f}, p90 {np.percentile(tot, 90):.0f}); ~{tiles:.1f} tiles per wave -> {tot.mean() / tiles:.0f} cycles per tile (MFMA alone: 1024 per wave, 2048 per SIMD)")
	for i, nm in enumerate(names):
		print(f"    {nm:22s} {a[:, i].mean() / tiles:8.0f} cycles per tile  ({100 * a[:, i].sum() / tot.sum():.1f} %)")
