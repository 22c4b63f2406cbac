export TMPDIR=/tmp
mkdir -p gpurun_out/rbm
for v in 1 0; do
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so ANNCUR_DEBUG_RB_MAJOR=$v rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/rbm/f$v -o run -- python3 bench.py --config cfg4_per_gpu --steps 3 --warmup 1 --cpu-sample-queries 0 --sustained-seconds 0 --no-k500 --no-graph --no-overlap --no-ivf > gpurun_out/rbm/b$v.json 2> gpurun_out/rbm/e$v.err
  python3 - <<P
import csv,glob
tot={}; n={}
for f in glob.glob("gpurun_out/rbm/f$v/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "scoreq1" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
            tot["q1"]=tot.get("q1",0)+float(r["Counter_Value"]); n["q1"]=n.get("q1",0)+1
print("rb_major=$v  scoreq1 FETCH_SIZE avg per launch: %.2f GB (x2 KB units -> bytes), launches %d" % (2*tot["q1"]*1024/n["q1"]/1e9, n["q1"]))
P
done
AB_FLAGS="--no-ivf --config cfg4_per_gpu" bash scripts/ab_env.sh 2 "rbm1:exp:" "rbm0:exp:ANNCUR_DEBUG_RB_MAJOR=0" 2>&1 | tail -4
AB_FLAGS="--no-ivf" bash scripts/ab_env.sh 2 "rbm1:exp:" "rbm0:exp:ANNCUR_DEBUG_RB_MAJOR=0" 2>&1 | tail -4
