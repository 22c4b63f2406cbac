"""Static check of the sweep kernels' inline-asm LDS reads: no instruction may touch the destination registers of a
`ds_read_*` that a preceding `s_waitcnt lgkmcnt(n)` has not yet retired.

The fragment reads of score_kernel / score16_kernel / wide_kernel are inline asm with COUNTED waits; hipcc does not know that the
asm's output register is still in flight after the statement, so nothing but the source's own wait statements stops it from
copying, spilling or consuming such a register early (the hardware has no interlock: the instruction reads stale data).  LDS
instructions of one wave return in order, so the check is a FIFO walk over the disassembly: every DS instruction enters the queue,
`s_waitcnt lgkmcnt(n)` retires all but the youngest n, and any instruction that names a register of a queued ds_read's destination
is a finding.  Loop bodies are walked twice so that reads in flight across the back edge are seen.

usage: python scripts/check_lds_hazards.py [lib.so]
"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_tools import disassemble, functions, KERNELS, _lgkm, SMEM

REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs(text):
	out = set()
	for m in REG.finditer(text):
		if m.group(2) is not None: out.add((m.group(1), int(m.group(2))))
		else: out.update((m.group(1), r) for r in range(int(m.group(3)), int(m.group(4)) + 1))
	return out


def check(lib):
	findings, n_kernels, n_reads = [], 0, 0
	for dis in disassemble(lib):
		for name, body in functions(dis):
			if not KERNELS.search(name) or not body: continue
			n_kernels += 1
			base = body[0][0]
			addr_index = {a: i for i, (a, _, _) in enumerate(body)}
			seen, replayed = set(), set()

			def walk(lo, hi, queue):
				nonlocal n_reads
				i = lo
				while i <= hi:
					a, ins, tail = body[i]
					op = ins.split()[0]
					w = _lgkm(ins)
					if w is not None:
						while len(queue) > w: queue.pop(0)
					elif op.startswith("ds_") or SMEM.match(ins):
						used = regs(ins)
						for (qa, qins, dst) in queue:
							if dst & used and (qa, a) not in seen:
								seen.add((qa, a)); findings.append(f"{name[:70]}: '{ins}' @ {a:x} touches the destination of in-flight '{qins}' @ {qa:x}")
						dst = set()
						if op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle") or "_rtn" in op:
							first = ins[len(op):].split(",")[0]
							dst = regs(first); n_reads += 1
						queue.append((a, ins, dst))
					else:
						used = regs(ins)
						if used:
							for (qa, qins, dst) in queue:
								if dst & used and (qa, a) not in seen:
									seen.add((qa, a)); findings.append(f"{name[:70]}: '{ins}' @ {a:x} touches the destination of in-flight '{qins}' @ {qa:x}")
						if op.startswith("s_cbranch") or op.startswith("s_branch"):
							m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", tail)
							tgt = base + int(m.group(1), 16) if m else None
							if tgt is not None and tgt <= a and tgt in addr_index and i not in replayed:
								replayed.add(i)
								walk(addr_index[tgt], i - 1, list(queue))  # once more around the loop with what is in flight now
					i += 1
				return queue

			walk(0, len(body) - 1, [])
	return findings, n_kernels, n_reads


if __name__ == "__main__":
	lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "anncur_amd", "lib", "libanncur_hip.so")
	f, nk, nr = check(lib)
	print(f"{lib}: {nk} sweep kernels, {nr} LDS reads followed to their wait, {len(f)} findings")
	for x in f[:40]: print("  " + x)
	sys.exit(1 if f else 0)
