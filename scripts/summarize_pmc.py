"""Dev tool: average the rocprofv3 --pmc counter rows per kernel (our kernels only) -> JSON on stdout.
usage: python scripts/summarize_pmc.py gpurun_out/prof_<tag>"""
import csv, collections, glob, json, re, sys
root = sys.argv[1]
names = ["score_kernel<256, 1, 16>", "score_kernel<256, 0, 16>", "rowwise_topk_wave_kernel<unsigned short>", "select_wave_kernel<false>",
		 "select_wave_kernel<true>", "select_candidates_kernel", "kth_value_wave_kernel", "gather_cols_kernel", "overlap_wave_kernel", "copy_bytes_kernel"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/*counter_collection.csv"):
	for r in csv.DictReader(open(f)):
		for n in names:
			if n in r["Kernel_Name"]:
				acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
				if r["Counter_Name"] in ("FETCH_SIZE", "SQ_WAVE_CYCLES"):
					acc[n]["_vgpr"].append(float(r["VGPR_Count"])); acc[n]["_lds"].append(float(r["LDS_Block_Size"]))
				break
out = {n: {c: round(sum(v) / len(v), 1) for c, v in d.items()} | {"launches_seen": len(next(iter(d.values())))} for n, d in acc.items()}
stats = {}
for f in glob.glob(root + "/stats/*kernel_stats.csv"):
	for r in csv.DictReader(open(f)):
		for n in names:
			if n in r["Name"]: stats[n] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "min_us": round(float(r["MinNs"]) / 1e3, 2)}
print(json.dumps({"pmc_avg_per_launch": out, "kernel_stats": stats}, indent=1))
