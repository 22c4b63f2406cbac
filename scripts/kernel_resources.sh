#!/bin/bash
# Dev tool: VGPRs / SGPRs / scratch / occupancy of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage), one line each.
# usage: bash scripts/kernel_resources.sh score_fused.hip [grep pattern] [extra hipcc flags]
cd "$(dirname "$0")/../anncur_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include $3 -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 \
 | python3 -c '
import re, sys, subprocess
cur = None; rows = []
for line in sys.stdin:
	m = re.search(r"remark: +(.*?) \[-Rpass", line)
	if not m: continue
	t = m.group(1).strip()
	if t.startswith("Function Name:"):
		cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
	elif cur is not None and ":" in t:
		k, v = t.split(":", 1); cur[k.strip()] = v.strip()
for r in rows:
	name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
	name = name.split("(")[0].replace("void ", "")
	print("%-62s vgpr %3s agpr %2s sgpr %3s spill %2s scratch %3s occ %s" % (name[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]")))
' | grep -E "${2:-.}"
