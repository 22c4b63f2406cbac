# Round 5 (GPU box): kernel timeline of bench.py's steps under rocprofv3.  --scan-mode side and --no-graph: rocprofv3 dies in its finaliser (SIGSEGV in
# __cxa_finalize, no output written) when the process has created a CU-masked stream (hipExtStreamCreateWithCUMask), with or without graphs.
set -e
mkdir -p gpurun_out/r5_timeline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_timeline/trace -o run -- python3 bench.py --direct --no-ceiling --steps 40 --warmup 10 --cpu-sample-queries 0 --sustained-seconds 0 --no-k500 --no-ivf --no-graph --scan-mode side > gpurun_out/r5_timeline/bench.json 2> gpurun_out/r5_timeline/bench.err
python3 - <<'P'
import csv, json
rows=list(csv.DictReader(open('gpurun_out/r5_timeline/trace/run_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
# the timed region: find the last 40 occurrences of the sweep kernel, print a window of 3 steps in the middle
idx=[i for i,n in enumerate(names) if 'score16_kernel' in n]
mid=idx[-20]
t0=int(rows[mid]['Start_Timestamp'])
out=[]
for r in rows:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if t0-1200000 <= s <= t0+1800000:
        out.append(f"{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f}  dur {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?'):>3s}  {r['Kernel_Name'][:80]}")
open('gpurun_out/r5_timeline/window.txt','w').write("\n".join(out)+"\n")
print("\n".join(out[:120]))
P
