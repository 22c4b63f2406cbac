#!/bin/bash
# Dev tool (GPU box): PMC counters per kernel for one microbench command.  usage: bash scripts/pmc_kernel.sh <outdir> "<counters>" -- <python args...>
# (counters in their own run with --kernel-trace only; the program itself follows `--`)
set -e
out=$1; pmc=$2; shift 3
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc $pmc -d $out -o run -- python3 "$@" > $out/stdout.txt 2> $out/stderr.txt
python3 - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/run_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"][:60]
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen: seen.add(key); cnt[name] += 1
for name in acc:
    if any(t in name for t in ("select", "tau_block", "kth_value", "score_kernel", "rowwise", "wide")):
        print(name, "calls", cnt[name], {c: round(v / cnt[name]) for c, v in acc[name].items()})
PY
find $out -name "*.csv" -size +8M -delete
