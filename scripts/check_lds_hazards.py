"""Static check of the sweep kernels' inline-asm LDS reads: no instruction may touch the destination registers of a
`ds_read_*` that a preceding `s_waitcnt lgkmcnt(n)` has not yet retired.

The fragment reads of score_kernel / score16_kernel / wide_kernel are inline asm with COUNTED waits; hipcc does not know that the
asm's output register is still in flight after the statement, so nothing but the source's own wait statements stops it from
copying, spilling or consuming such a register early (the hardware has no interlock: the instruction reads stale data).  LDS
instructions of one wave return in order, so the check is a FIFO walk over the disassembly: every DS instruction enters the queue,
`s_waitcnt lgkmcnt(n)` retires all but the youngest n, and any instruction that names a register of a queued ds_read's destination
is a finding.  The walk follows the kernel's control-flow graph with the queue as its state (every basic block is visited once per
distinct queue at its entry), so out-of-line blocks are judged on the paths that really reach them.

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


def _is_ds_load(op):
	return op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle") or "_rtn" in op


def check_body(name, body):
	"""Findings for one kernel given as [(address, instruction text, raw objdump tail)]: path-sensitive walk, the state is the FIFO
	of LDS loads in flight (their destination registers); every (basic block, state at its entry) pair is visited once."""
	from check_mfma_hazards import basic_blocks
	findings = []
	blocks, succ = basic_blocks(body)
	seen_states, reported = set(), set()
	work = [(0, ())]
	visits = 0
	while work and visits < 400000:
		k, state = work.pop()
		if (k, state) in seen_states: continue
		seen_states.add((k, state)); visits += 1
		queue = list(state)
		st, en = blocks[k]
		for i in range(st, en + 1):
			a, ins, _ = body[i]
			op = ins.split()[0]
			w = _lgkm(ins)
			if w is not None:
				while len(queue) > w: queue.pop(0)
				continue
			used = regs(ins) if queue else set()
			if used:
				for (qa, qins, dst) in queue:
					if dst & used and (qa, a) not in reported:
						reported.add((qa, a))
						findings.append(f"{name[:70]}: '{ins}' @ {a:x} touches the destination of in-flight '{qins}' @ {qa:x}")
			# (LDS stores / atomics are left out of the queue: most of them sit in conditional blocks, a queue that tracked them
			#  would differ on every path, and without them a counted wait retires FEWER loads here than in the hardware -- the
			#  conservative direction)
			if _is_ds_load(op):
				queue.append((a, ins, frozenset(regs(ins[len(op):].split(",")[0]))))
				if len(queue) > 16: queue.pop(0)  # (the hardware counter saturates; nothing here keeps that many in flight)
		out = tuple(queue)
		for t in succ[k]: work.append((t, out))
	if visits >= 400000: findings.append(f"{name[:70]}: state space not exhausted (walk capped)")
	return findings


def _vmcnt(ins):
	"""vmcnt value of an s_waitcnt (None: it does not wait on vmcnt)."""
	if not ins.lstrip().startswith("s_waitcnt"):
		return None
	m = re.search(r"vmcnt\((\d+)\)", ins)
	if m:
		return int(m.group(1))
	m = re.search(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s", ins + " ")  # raw immediate: vmcnt = bits 0..3 and 14..15
	if m:
		v = int(m.group(1), 0)
		return (v & 0xF) | (((v >> 14) & 0x3) << 4)
	return None if ("lgkmcnt" in ins or "expcnt" in ins) else 0


def check_vm_body(name, body):
	"""The same walk for RETURNING global atomics (the ticket draw of the dynamic tile schedule is an inline-asm `global_atomic_add vD, ...
	sc0` whose result is waited for a whole tile later): nothing may name vD until an s_waitcnt vmcnt(0).  (Only a full drain is taken to
	retire it: younger VMEM operations of unknown number may sit behind it.)"""
	from check_mfma_hazards import basic_blocks
	findings = []
	blocks, succ = basic_blocks(body)
	seen, reported = set(), set()
	work = [(0, ())]
	while work and len(seen) < 200000:
		k, state = work.pop()
		if (k, state) in seen: continue
		seen.add((k, state))
		queue = list(state)
		st, en = blocks[k]
		for i in range(st, en + 1):
			a, ins, _ = body[i]
			op = ins.split()[0]
			w = _vmcnt(ins)
			if w is not None:
				if w == 0: queue = []
				continue
			if queue:
				used = regs(ins)
				for (qa, qins, dst) in queue:
					if dst & used and (qa, a) not in reported:
						reported.add((qa, a))
						findings.append(f"{name[:70]}: '{ins}' @ {a:x} touches the destination of in-flight '{qins}' @ {qa:x}")
			if op.startswith("global_atomic") and " sc0" in ins + " ":
				queue.append((a, ins, frozenset(regs(ins[len(op):].split(",")[0]))))
		out = tuple(queue)
		for t in succ[k]: work.append((t, out))
	return findings


def check(lib, only=None):
	findings, n_kernels, n_reads = [], 0, 0
	for dis in disassemble(lib):
		for name, body in functions(dis):
			if not KERNELS.search(name) or not body: continue
			if only and not re.search(only, name): continue
			n_kernels += 1
			n_reads += sum(1 for _, ins, _ in body if _is_ds_load(ins.split()[0]))
			findings += check_body(name, body)
			findings += check_vm_body(name, body)
	return findings, n_kernels, n_reads


if __name__ == "__main__":
	lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "anncur_amd", "lib", "libanncur_hip.so")
	f, nk, nr = check(lib, sys.argv[2] if len(sys.argv) > 2 else None)
	print(f"{lib}: {nk} sweep kernels, {nr} LDS reads followed to their wait, {len(f)} findings")
	for x in f[:40]: print("  " + x)
	sys.exit(1 if f else 0)
