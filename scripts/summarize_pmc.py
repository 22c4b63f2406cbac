"""Dev tool: average the rocprofv3 --pmc counter rows per kernel (our kernels only) -> JSON on stdout, including the
`profiles/rNN_pmc_traffic_<config>.json` block bench.py reads for roofline.traffic (HBM bytes per launch of the sweep kernel).
usage: python scripts/summarize_pmc.py gpurun_out/prof_<tag> [cfg2|cfg4_per_gpu]

Corrections per MI355X_MICROARCH.md 'HBM': FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports exactly half the bytes
of wide coalesced streaming reads (16 B per lane, global_load and direct-to-LDS alike), so read bytes = 2 * FETCH_SIZE * 1024;
WRITE_SIZE * 1024 is exact for 16-byte streaming stores (sparse 8-byte candidate stores are counted with their partial-line
amplification).  Each counter group was collected in its own run (--kernel-trace + --pmc only)."""
import csv, collections, glob, json, os, sys
root = sys.argv[1]
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
SHAPES = {"cfg2": dict(Q=10000, I=100000, Kp=256, k=100), "cfg4_per_gpu": dict(Q=6250, I=1000000, Kp=512, k=100)}
shape = SHAPES[cfg]
kp = shape["Kp"]
qt = 2 if kp <= 256 else 1
# the sweep runs as one of four bodies (32x32x16 with the ballot / exec-mask filter, 16x16x32, Kp = 512 with the wave queue -- chosen by the plan): all of them
# are "the sweep kernel" of the roofline, whose per-launch figures average over the launches of a step
sweep_variants = [f"score_kernel<{kp}, 1, 16, false, false, {qt}>", f"score_kernel<{kp}, 1, 16, true, false, {qt}>", f"score16_kernel<{kp}", f"score16r_kernel<{kp}", f"scoreq1_kernel<{kp}>", f"scoreq16_kernel<{kp}>"]
sweep, prepass = f"score_kernel<{kp}, sweep>", f"score_kernel<{kp}, 0, 16, false, false, {qt}>"
names = sweep_variants + [prepass, f"score_kernel<{kp}, 1, 16, false, false, 1>", "rowwise_topk_wave_kernel<unsigned short", "select_wave_kernel<false", "select_wave_kernel<true", "select_stream_kernel<false",
		 "select_stream_kernel<true", "select_candidates_kernel", "kth_value_wave_kernel", "gather_cols_kernel", "overlap_wave_kernel", "copy_bytes_kernel", "wide_kernel", "gemm_f64_kernel"]
def key_of(n): return sweep if n in sweep_variants else n
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/*counter_collection.csv"):
	for r in csv.DictReader(open(f)):
		for n in names:
			if n in r["Kernel_Name"]:
				keys = [key_of(n)] + ([n] if n in sweep_variants else [])   # the sweep as a whole AND each filter variant (= sweep stage) on its own
				for kk in keys:
					acc[kk][r["Counter_Name"]].append(float(r["Counter_Value"]))
					if r["Counter_Name"] in ("FETCH_SIZE", "SQ_WAVE_CYCLES"):
						acc[kk]["_vgpr"].append(float(r["VGPR_Count"])); acc[kk]["_lds"].append(float(r["LDS_Block_Size"]))
				break
out = {n: {c: round(sum(v) / len(v), 1) for c, v in d.items()} | {"launches_seen": len(next(iter(d.values())))} for n, d in acc.items()}
stats = {}
for f in glob.glob(root + "/stats/*kernel_stats.csv"):
	for r in csv.DictReader(open(f)):
		for n in names:
			if n in r["Name"]:
				cur = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "min_us": round(float(r["MinNs"]) / 1e3, 2)}
				if n in sweep_variants:
					stats[n] = cur
					old = stats.get(sweep)
					if old:  # (per-launch average over both variants)
						tot = old["calls"] + cur["calls"]
						cur = {"calls": tot, "avg_us": round((old["avg_us"] * old["calls"] + cur["avg_us"] * cur["calls"]) / tot, 2), "min_us": min(old["min_us"], cur["min_us"])}
					stats[sweep] = cur
				else:
					stats[n] = cur
# per sweep variant (the plan runs the exec-mask filter in the first stage, the ballot filter in the later ones): what the wave cycles went to
per_variant = {}
for n in sweep_variants:
	c, st = out.get(n), stats.get(n)
	if not c or not st: continue
	row = {"avg_us": st["avg_us"], "calls": st["calls"]}
	if "SQ_WAVE_CYCLES" in c:
		row["wait_any_share_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
		row["active_inst_share_of_wave_cycles"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
	if "GRBM_GUI_ACTIVE" in c:
		cyc = c["GRBM_GUI_ACTIVE"] / 8
		row["clock_ghz_from_grbm"] = round(cyc / (st["avg_us"] * 1e-6) / 1e9, 3)
		row["mfma_pipe_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, 3)
		row["valu_insts_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 1) / 32, 1), 2)
		row["salu_insts_per_mfma"] = round(c.get("SQ_INSTS_SALU", 0) / max(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 1) / 32, 1), 2)
	per_variant[n] = row
# ---- the sweep STAGE BY STAGE.  Both stages are launches of one kernel with one grid, so the stats CSV averages a short first stage with
# a long second one; the kernel trace has every dispatch: the i-th sweep dispatch (in start order) is stage i mod n_stages (n_stages from
# the bench line of the same run: every fused call of that run has the same plan; run with --no-k500).
def _bench_line(path):
	try:
		for l in reversed(open(path).read().strip().splitlines()):
			if l.startswith("{"): return json.loads(l)
	except Exception:
		pass
	return None
def _is_sweep(name): return any(v in name for v in sweep_variants)
def stage_split():
	b = _bench_line(root + "/bench_stats.json")
	if not b or "fused_plan" not in b: return None
	plan = b["fused_plan"]; ns = plan["n_stages"]; ends = plan["stage_end"]; tiles = [e - s0 for s0, e in zip([0] + ends[:-1], ends)]
	tile_items = 256 if kp > 512 else 32
	rows = []
	for f in glob.glob(root + "/stats/*kernel_trace.csv"):
		for r in csv.DictReader(open(f)):
			if _is_sweep(r["Kernel_Name"]): rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
	if not rows or len(rows) % ns: return None
	rows.sort()
	per = [[d for i, (_, d) in enumerate(rows) if i % ns == g] for g in range(ns)]
	stages = []
	for g in range(ns):
		avg_ns = sum(per[g]) / len(per[g]); flops = 2.0 * shape["Q"] * kp * tile_items * tiles[g]
		stages.append({"stage": g, "tiles": tiles[g], "calls": len(per[g]), "avg_us": round(avg_ns / 1e3, 2), "min_us": round(min(per[g]) / 1e3, 2),
					   "tflops": round(flops / (avg_ns * 1e-9) / 1e12, 1), "frac_of_2500": round(flops / (avg_ns * 1e-9) / 2.5e15, 4)})
	# PMC rows per stage: dispatch order inside each counter pass
	pm = collections.defaultdict(lambda: collections.defaultdict(list))
	for f in glob.glob(root + "/pmc_*/*counter_collection.csv"):
		disp = collections.defaultdict(dict)
		for r in csv.DictReader(open(f)):
			if _is_sweep(r["Kernel_Name"]): disp[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
		ids = sorted(disp)
		if len(ids) % ns: continue
		for i, d in enumerate(ids):
			for c, v in disp[d].items(): pm[i % ns][c].append(v)
	for g in range(ns):
		c = {k_: sum(v) / len(v) for k_, v in pm[g].items()}
		if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
			cyc = c["GRBM_GUI_ACTIVE"] / 8
			stages[g]["mfma_pipe_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, 3)
			stages[g]["valu_insts_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0) / max(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 32, 1), 2)
		if "SQ_WAVE_CYCLES" in c: stages[g]["wait_any_share_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
		if "FETCH_SIZE" in c: stages[g]["read_bytes"] = round(2 * c["FETCH_SIZE"] * 1024)
		if "WRITE_SIZE" in c: stages[g]["write_bytes"] = round(c["WRITE_SIZE"] * 1024)
	return stages
res = {"config": cfg, **shape, "pmc_avg_per_launch": out, "kernel_stats": stats, "sweep_per_variant": per_variant, "sweep_stages_rocprof": stage_split()}
sw = out.get(sweep)
if sw and "FETCH_SIZE" in sw and "WRITE_SIZE" in sw:
	rd, wr = 2 * sw["FETCH_SIZE"] * 1024, sw["WRITE_SIZE"] * 1024
	res["score_kernel_sweep_read_bytes_per_launch"] = round(rd)
	res["score_kernel_sweep_write_bytes_per_launch"] = round(wr)
	res["score_kernel_sweep_hbm_bytes_per_launch"] = round(rd + wr)
	if "GRBM_GUI_ACTIVE" in sw and sweep in stats:
		clk = sw["GRBM_GUI_ACTIVE"] / 8 / (stats[sweep]["avg_us"] * 1e-6)
		res["score_kernel_sweep_clock_ghz"] = round(clk / 1e9, 3)
		res["score_kernel_sweep_mfma_pipe_busy"] = round(sw.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (sw["GRBM_GUI_ACTIVE"] / 8), 3)
sc = out.get("rowwise_topk_wave_kernel<unsigned short")
if sc and "FETCH_SIZE" in sc:
	res["exact_scan_hbm_bytes_per_launch"] = round(2 * sc["FETCH_SIZE"] * 1024 + sc.get("WRITE_SIZE", 0) * 1024)
print(json.dumps(res, indent=1))
