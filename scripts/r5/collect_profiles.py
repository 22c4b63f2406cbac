"""round-5 dev tool: copy what DESIGN.md quotes from gpurun_out/ (scratch) into profiles/ (tracked) under r05_ names."""
import json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def line(path):
	for l in reversed(open(path).read().strip().splitlines()):
		if l.startswith("{"): return l
	raise SystemExit(f"no JSON line in {path}")
def put(src, dst, json_line=False):
	src = os.path.join(R, src); dst = os.path.join(R, "profiles", dst)
	if not os.path.exists(src): print("missing", src); return
	if json_line: open(dst, "w").write(line(src) + "\n")
	else: shutil.copy(src, dst)
	print("->", dst)
put("gpurun_out/r5_final/bench_cfg2_n1.json", "r05_bench_cfg2_n1.json", True)
put("gpurun_out/r5_final/bench_cfg4_per_gpu_n1.json", "r05_bench_cfg4_per_gpu_n1.json", True)
put("gpurun_out/r5_final/bench_default_2ranks_gloo_one_gpu_rehearsal.json", "r05_bench_default_2ranks_gloo_one_gpu_rehearsal.json", True)
for cfg, tag in (("cfg2", "r05_cfg2"), ("cfg4_per_gpu", "r05_cfg4")):
	d = f"gpurun_out/prof_{tag}"
	put(f"{d}/stats/run_kernel_stats.csv", f"r05_kernel_stats_{cfg}.csv")
	put(f"{d}/summary_{cfg}.json", f"r05_pmc_summary_{cfg}.json")
	put(f"{d}/bench_stats.json", f"r05_bench_under_rocprof_{cfg}.json", True)
	sp = os.path.join(R, d, f"summary_{cfg}.json")
	if os.path.exists(sp):
		s = json.load(open(sp))
		keep = {k: s[k] for k in ("config", "Q", "I", "Kp", "k") + tuple(k for k in s if k.startswith("score_kernel_sweep_") or k.startswith("exact_scan_"))}
		keep["sweep_stages_rocprof"] = s.get("sweep_stages_rocprof")
		json.dump(keep, open(os.path.join(R, "profiles", f"r05_pmc_traffic_{cfg}.json"), "w"), indent=1)
		print("-> profiles/r05_pmc_traffic_%s.json" % cfg)
put("gpurun_out/r5_prof/ceiling_kernel_stats.csv", "r05_sweep_ceiling_kernel_stats.csv")
put("gpurun_out/r5_prof/normal_kernel_stats.csv", "r05_sweep_normal_kernel_stats_same_probe.csv")
put("gpurun_out/r5_c2/survivor_model.json", "r05_survivor_model.json")
put("gpurun_out/r5_final/ladder_probe.json", "r05_ladder_probe.json")
put("gpurun_out/r5_ab_ladder.txt", "r05_ab_ladder_bench.txt")
put("gpurun_out/r5_ladder_period_sweep.txt", "r05_ladder_period_sweep_experiments_lib.txt")
put("gpurun_out/r5_prof_call.txt", "r05_scan_cus_sweep_box1.txt")
put("gpurun_out/r5_final/cumask_repro.log", "r05_cumask_null_stream_repro.txt")
