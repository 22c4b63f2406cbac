#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/pmc_err
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_err/$set -o run --output-format csv -- python3 scripts/r4/err_one.py > gpurun_out/pmc_err_$set.log 2>&1 || { echo "pmc $set failed"; tail -5 gpurun_out/pmc_err_$set.log; }
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_err/*/run_counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        for key in ('error_lds_kernel','evalf_kernel','score16_kernel','score_kernel<256, 0'):
            if key in n: agg[(key, r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in sorted(agg.items()): print(f"{k:22s} {c:12s} launches {len(v):3d}  mean per launch {sum(v)/len(v):.4g} (raw counter; FETCH_SIZE in KB: x1024 bytes)")
PY
