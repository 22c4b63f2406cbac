"""round-4 dev tool: rocprofv3 evidence for wide_kernel (Kp > 512) from the passes of scripts/r4/profile_wide.sh -> JSON on stdout.
usage: python scripts/r4/summarize_wide.py gpurun_out/r4wide  (Q, I, Kp of the microbench: 10 000 x 100 000 x 1024)"""
import csv, glob, json, sys, collections
root = sys.argv[1]
Q, I, Kp = 10000, 100000, 1024
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
	for r in csv.DictReader(open(f)):
		n = r["Kernel_Name"]
		if "wide_kernel<1" in n: key = "wide_kernel<1,16> (sweep)"
		elif "wide_kernel<0" in n: key = "wide_kernel<0,16> (prepass)"
		else: continue
		acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
		acc[key]["_vgpr"].append(float(r["VGPR_Count"])); acc[key]["_lds"].append(float(r["LDS_Block_Size"]))
stats = {}
for f in glob.glob(root + "/stats/**/*kernel_stats.csv", recursive=True):
	for r in csv.DictReader(open(f)):
		if "wide_kernel<1" in r["Name"]: stats["wide_kernel<1,16> (sweep)"] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3}
		if "wide_kernel<0" in r["Name"]: stats["wide_kernel<0,16> (prepass)"] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3}
out = {"shape": {"Q": Q, "I": I, "Kp": Kp, "k": 100}, "kernel_stats": stats, "kernels": {}}
for key, d in acc.items():
	c = {k: sum(v) / len(v) for k, v in d.items()}
	row = {"launches_seen": len(d.get("GRBM_GUI_ACTIVE", d.get("FETCH_SIZE", [0]))), "vgpr": c.get("_vgpr"), "lds_bytes": c.get("_lds")}
	st = stats.get(key)
	if st and "GRBM_GUI_ACTIVE" in c:
		cyc = c["GRBM_GUI_ACTIVE"] / 8
		row["clock_ghz_from_grbm"] = round(cyc / (st["avg_us"] * 1e-6) / 1e9, 3)
		row["mfma_pipe_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc, 3)
		n_mfma = max(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 1) / 32, 1)
		row["valu_insts_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0) / n_mfma, 2); row["salu_insts_per_mfma"] = round(c.get("SQ_INSTS_SALU", 0) / n_mfma, 2)
	if "SQ_WAVE_CYCLES" in c:
		row["wait_any_share_of_wave_cycles"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
		row["active_inst_share_of_wave_cycles"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
		if "SQ_WAIT_INST_ANY" in c: row["wait_inst_any_share"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
		if "SQ_WAIT_INST_LDS" in c: row["wait_inst_lds_share"] = round(c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"], 3)
	if "SQ_LDS_IDX_ACTIVE" in c: row["lds_bank_conflict_share_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0) / max(c["SQ_LDS_IDX_ACTIVE"], 1), 4)
	if "TCC_HIT_sum" in c: row["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0), 1), 4)
	if "FETCH_SIZE" in c: row["read_bytes_beyond_l2_per_launch"] = round(2 * c["FETCH_SIZE"] * 1024)
	if "WRITE_SIZE" in c: row["write_bytes_per_launch"] = round(c["WRITE_SIZE"] * 1024)
	out["kernels"][key] = row
sw = stats.get("wide_kernel<1,16> (sweep)")
if sw:
	launches_per_call = 2   # (two sweep stages at this shape: the plan line of the microbench)
	flops = 2.0 * Q * Kp * I / launches_per_call
	out["sweep_tflops_from_kernel_stats_avg"] = round(flops / (sw["avg_us"] * 1e-6) / 1e12, 1)
	out["sweep_frac_of_2500"] = round(flops / (sw["avg_us"] * 1e-6) / 2.5e15, 4)
print(json.dumps(out, indent=1))
