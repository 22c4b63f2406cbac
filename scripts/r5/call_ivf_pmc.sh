set -e
out=gpurun_out/r5_ivf/pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MODES=1 GEMM_ONLY=0 REPS=3
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS"; do
  name=$(echo $pmc | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d $out/$name -o run -- python3 scripts/r5/ivf_probe.py > $out/$name.txt 2>&1
  echo "pass $name done"
done
find $out -type f ! -name "*.csv" ! -name "*.txt" -delete
python3 - <<'P'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/r5_ivf/pmc/*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'ivf_tile128' in r['Kernel_Name'] or 'rowwise_topk_wave_kernel<float, false, true, true>' in r['Kernel_Name']:
            key = (r['Kernel_Name'][:40], r['Counter_Name'])
            agg[key][0] += float(r['Counter_Value']); agg[key][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        print(f"{k:40s} {c:28s} per launch {v / n:16.1f}  ({n} launches)")
P
