"""Dev tool: for one kernel of a hipcc -S listing, where the scratch (spill) accesses, scalar loads and MFMAs sit relative to the
kernel's basic blocks -- enough to see whether a spill or an s_load landed INSIDE the tile loop.
usage: python scripts/isa_loop_report.py listing.s <mangled-name substring>"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
beg = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0] and ":" in l)
end = next(i for i in range(beg, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[beg:end]
blocks = []; cur = ["entry", 0, 0, 0, 0, 0]
for l in body:
	m = re.match(r"^(\.LBB\d+_\d+):", l)
	if m:
		blocks.append(cur); cur = [m.group(1), 0, 0, 0, 0, 0]
	t = l.strip()
	if t.startswith("v_mfma"): cur[1] += 1
	if t.startswith("scratch_") or (t.startswith("buffer_") and "offen" in t): cur[2] += 1
	if t.startswith("s_load"): cur[3] += 1
	if t.startswith("s_cbranch") or t.startswith("s_branch"): cur[4] += 1
	if t and not t.startswith((";", ".")): cur[5] += 1
blocks.append(cur)
print("block            mfma scratch s_load branches insts")
for b in blocks:
	if b[1] or b[2] or b[3]: print("%-16s %4d %7d %6d %8d %5d" % tuple(b))
print("total insts", sum(b[5] for b in blocks), "mfma", sum(b[1] for b in blocks), "scratch", sum(b[2] for b in blocks))
