"""Dev tool (experiments build): the fused call stage by stage at the cfg2 shape, ONE process, interleaved rounds (guide rule 24).
Every setting is a set of ANNCUR_DEBUG_* knobs of the experiments library (read per call) and/or variant flags; reported per
setting: median over rounds of prepass / threshold / sweep launches 1..3 / refinement share / select, and the total.
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so python scripts/stage_probe.py [k] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import ops
from anncur_amd.cur import _norm_sorted_pack
dev = torch.device("cuda")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
Q, I, K = 10000, 100000, 256
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, I, generator=g, device=dev)
X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
Kp = ops.padded_k(K)
Etp, ids = _norm_sorted_pack(E.t().contiguous().float(), Kp)
Xp = ops.pack_bf16(X, Kp)
del Z, E
KNOBS = ["ANNCUR_DEBUG_CHUNK", "ANNCUR_DEBUG_STAGES", "ANNCUR_DEBUG_ALL_PRED", "ANNCUR_DEBUG_TAU_BIAS", "ANNCUR_DEBUG_FLUSH_TILES", "ANNCUR_DEBUG_NOSTORE", "ANNCUR_DEBUG_CONTIG", "ANNCUR_DEBUG_RING_STAGGER", "ANNCUR_DEBUG_RING_SLEEP", "ANNCUR_DEBUG_SLICED", "ANNCUR_DEBUG_WG8"]
def S(name, flags=None, **env):
	return (name, {("ANNCUR_DEBUG_" + a.upper()): str(b) for a, b in env.items()}, flags or {})
settings = [S("default"), S("wg8", wg8=1), S("bare wg8", wg8=1, tau_bias="1e30"), S("unsliced", sliced=0), S("bare unsliced", sliced=0, tau_bias="1e30"), S("stag=10", {"ring": True}, ring_stagger=10), S("stag=20", {"ring": True}, ring_stagger=20), S("stag=30", {"ring": True}, ring_stagger=30), S("stag=20 nosleep", {"ring": True}, ring_stagger=20, ring_sleep=0), S("nosleep", {"ring": True}, ring_sleep=0), S("bare stag=20", {"ring": True}, ring_stagger=20, tau_bias="1e30"), S("ring", {"ring": True}), S("bare ring", {"ring": True}, tau_bias="1e30"), S("mfma32", {"mfma32": True}), S("bare mfma32", {"mfma32": True}, tau_bias="1e30"), S("static", chunk=0), S("chunk=3", chunk=3), S("chunk=5", chunk=5), S("chunk=6", chunk=6), S("chunk=8", chunk=8), S("mfma16", {"mfma16": True}), S("qt1", {"qt1": True}),
			S("static f=.22 p0", chunk=0, stages="0.22", all_pred=0), S("static bare", chunk=0, tau_bias="1e30"),
			S("pred=0", all_pred=0), S("pred=1", all_pred=1),
			S("1 stage", stages="1"), S("f=.06", stages="0.06"), S("f=.10", stages="0.10"), S("f=.15", stages="0.15"), S("f=.22", stages="0.22"), S("f=.30", stages="0.30"),
			S("f=.04,.25", stages="0.04,0.25"), S("f=.06,.30", stages="0.06,0.30"), S("f=.10,.40", stages="0.10,0.40"),
			S("f=.10 pred=0", stages="0.10", all_pred=0), S("f=.22 pred=0", stages="0.22", all_pred=0),
			S("flush=1", flush_tiles=1), S("flush=4", flush_tiles=4), S("f=.25 pred=0", stages="0.25", all_pred=0), S("f=.30 pred=0", stages="0.30", all_pred=0), S("f=.18 pred=0", stages="0.18", all_pred=0),
			S("bare", tau_bias="1e30"), S("bare pred=0", tau_bias="1e30", all_pred=0), S("bare pred=1", tau_bias="1e30", all_pred=1),
			S("bare mfma16", {"mfma16": True}, tau_bias="1e30"), S("bare qt1", {"qt1": True}, tau_bias="1e30"),
			S("static mfma16", {"mfma16": True}, chunk=0), S("bare static mfma16", {"mfma16": True}, chunk=0, tau_bias="1e30"), S("mfma16 chunk=8", {"mfma16": True}, chunk=8),
			S("bare mfma16 chunk=64", {"mfma16": True}, chunk=64, tau_bias="1e30"), S("bare mfma16 chunk=1", {"mfma16": True}, chunk=1, tau_bias="1e30"),
			S("bare static contig", chunk=0, contig=1, tau_bias="1e30"), S("bare chunk=64", chunk=64, tau_bias="1e30"), S("bare 1 stage", tau_bias="1e30", stages="1")]
if os.environ.get("STAGE_PROBE_ONLY"):
	keep = os.environ["STAGE_PROBE_ONLY"].split(";")
	settings = [s for s in settings if s[0] in keep]
res = {s[0]: [] for s in settings}
ref = None
for r in range(rounds + 1):
	for name, env, flags in settings:
		for kn in KNOBS: os.environ.pop(kn, None)
		os.environ.update(env)
		if r == 0: print("warm-up", name, flush=True)
		(v, idx), ms = ops.score_topk_fused_timed(Xp, Etp, I, k, leading_sample=True, item_ids=ids, **flags)
		if r == 0:
			if not env.get("ANNCUR_DEBUG_TAU_BIAS") and not env.get("ANNCUR_DEBUG_NOSTORE"):
				if ref is None: ref = v.clone()
				elif not torch.equal(v, ref): print("!! values differ from the default's:", name, float((v - ref).abs().max()), flush=True)
			continue
		res[name].append(ms)
print(f"k = {k}; ms: prepass  thresh | sweep1  sweep2  sweep3 | refine | select | total   (sweep TFLOP/s over the launches)")
for name, _, _ in settings:
	m = np.median(np.array(res[name]), axis=0)
	refine = m[2] - m[4]
	print("%-14s %.4f  %.4f | %.4f  %.4f  %.4f | %.4f | %.4f | %.4f   (%.0f)" % (name, m[0], m[1], m[6], m[7], m[8], refine, m[3], m[:4].sum(), 2.0 * Q * Kp * I / (m[4] * 1e-3) / 1e12), flush=True)
