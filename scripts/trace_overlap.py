"""Dev tool: from a rocprofv3 --kernel-trace CSV, print the timeline of ONE step of bench.py (kernel, start, end relative to the step's
first kernel, queue) so that what actually runs concurrently can be read off.  usage: python scripts/trace_overlap.py <kernel_trace.csv> [step]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a step starts at a gather_cols_kernel
starts = [i for i, r in enumerate(rows) if "gather_cols_kernel" in r["Kernel_Name"]]
i0 = starts[which]; i1 = starts[which + 1] if which + 1 < 0 or which + 1 < len(starts) else len(rows)
if which + 1 == 0: i1 = len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
def short(n):
	n = n.replace("(anonymous namespace)::", "").replace("void ", "")
	return n.split("(")[0][:60]
for r in rows[i0:i1]:
	s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
	print("%8.1f %8.1f  %6.1f us  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
