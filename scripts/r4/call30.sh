#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf gpurun_out/pmc_scan
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_scan/$tag -o run --output-format csv -- python3 scripts/r4/scan_one.py > gpurun_out/pmc_scan_$tag.log 2>&1 || { echo "pmc $tag failed"; tail -5 gpurun_out/pmc_scan_$tag.log; }
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_scan/*/run_counter_collection.csv')):
    rows=[r for r in csv.DictReader(open(f)) if 'rowwise_topk_wave' in r['Kernel_Name']]
    rows.sort(key=lambda r:int(r['Dispatch_Id']))
    disp=sorted(set(int(r['Dispatch_Id']) for r in rows))
    names=['I=4096','I=8192','I=16384','I=32768','I=65536','I=100000']
    for gi in range(6):
        ids=disp[3*gi:3*gi+3]
        agg=collections.defaultdict(list)
        for r in rows:
            if int(r['Dispatch_Id']) in ids: agg[r['Counter_Name']].append(float(r['Counter_Value']))
        print(names[gi], {k: round(sum(v)/len(v)/1e4) for k,v in sorted(agg.items())}, '(per row)')
PY
