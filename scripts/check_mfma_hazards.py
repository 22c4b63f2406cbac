"""Static check: wait states between an MFMA and the first non-accumulate use of its result registers.

An XDL MFMA writes its D registers passes after it issues; any instruction other than the next MFMA that takes D whole as its C
operand must be >= N wait states behind it (8-pass 32x32x16 bf16: 12 states, 4-pass 16x16x32 bf16: 8; the MI355X guide, 'What
hipcc does not do' (2)).  hipcc pads this for its own code -- except that around the inline-asm statements of the sweep kernels it
was seen to under-pad one path (Kp = 512 sweep, second unrolled tile: MFMA -> s_cbranch -> v_lshl_or -> s_nop 5 -> v_mov of the
accumulator = 8 states), and whether the missing states were there at run time depended on instruction-fetch timing, i.e. on code
placement: a diagnostic build that shifted the loop by 8 bytes returned scores without the last k-step's contribution.

The check builds each kernel's control-flow graph from the disassembly, propagates "wait states since the MFMA that writes
register r issued" forward (one state per instruction, N + 1 for `s_nop N`, the SHORTEST distance over all paths into a block) and
reports every reader / writer of an MFMA's D registers that can come earlier than hipcc's own padding floor for that MFMA.
usage: python scripts/check_mfma_hazards.py [lib.so]
"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_tools import disassemble, functions
from check_lds_hazards import regs

# wait states hipcc itself keeps between an XDL MFMA and a VALU / LDS / VMEM instruction that reads or writes its D registers
# (LLVM GCNHazardRecognizer, gfx940 family: passes + 3; 16-pass 18 for a write): used as the floor here
NEED = [(re.compile(r"v_mfma_f32_32x32x16"), 11), (re.compile(r"v_mfma_f32_16x16x32"), 7), (re.compile(r"v_mfma_f32_32x32x"), 18),
		(re.compile(r"v_mfma_f32_16x16x"), 11), (re.compile(r"v_mfma_f64_16x16x4"), 18), (re.compile(r"v_mfma_f32_4x4x"), 5), (re.compile(r"v_mfma"), 18)]
HORIZON = 24  # states after which a pending MFMA result is certainly written


def operands(ins):
	op = ins.split()[0]
	rest = ins[len(op):]
	parts, depth, cur = [], 0, ""
	for ch in rest:
		if ch == "[": depth += 1
		if ch == "]": depth -= 1
		if ch == "," and depth == 0: parts.append(cur); cur = ""
		else: cur += ch
	parts.append(cur)
	return op, [p.strip() for p in parts]


def basic_blocks(body):
	"""(start, end) index ranges and successor lists from the branch targets objdump prints as <symbol+0xoff>."""
	base = body[0][0]
	addr_index = {a: i for i, (a, _, _) in enumerate(body)}
	leaders, tgt_of = {0}, {}
	for i, (a, ins, tail) in enumerate(body):
		op = ins.split()[0]
		if op.startswith("s_cbranch") or op == "s_branch":
			m = re.search(r"<[^>+]*(?:\+0x([0-9a-fA-F]+))?>", tail)
			t = base + (int(m.group(1), 16) if m and m.group(1) else 0) if m else None
			if t in addr_index:
				tgt_of[i] = addr_index[t]; leaders.add(addr_index[t])
			if i + 1 < len(body): leaders.add(i + 1)
		elif op in ("s_endpgm", "s_setpc_b64"):
			if i + 1 < len(body): leaders.add(i + 1)
	starts = sorted(leaders)
	blocks = [(st, (starts[k + 1] if k + 1 < len(starts) else len(body)) - 1) for k, st in enumerate(starts)]
	bidx = {st: k for k, (st, _) in enumerate(blocks)}
	succ = []
	for st, en in blocks:
		op = body[en][1].split()[0]
		out = []
		if op == "s_branch":
			if en in tgt_of: out.append(bidx[tgt_of[en]])
		elif op in ("s_endpgm", "s_setpc_b64"):
			pass
		else:
			if op.startswith("s_cbranch") and en in tgt_of: out.append(bidx[tgt_of[en]])
			if en + 1 < len(body): out.append(bidx[en + 1])
		succ.append(out)
	return blocks, succ


def transfer(body, st, en, state, report):
	"""state: reg -> (states since the MFMA that writes it issued, needed, mfma text, address).  Returns the block's out-state."""
	pending = dict(state)
	for i in range(st, en + 1):
		a, ins, _ = body[i]
		op, ops = operands(ins)
		states = int(ops[0], 0) + 1 if op == "s_nop" else 1
		if op.startswith("v_mfma"):
			need = next(n for rx, n in NEED if rx.search(op))
			d, srca, srcb = regs(ops[0]), regs(ops[1]), regs(ops[2])
			srcc = regs(ops[3]) if len(ops) > 3 else set()
			touched = srca | srcb | (set() if srcc == d else (srcc | d))  # (C == D whole: the accumulate chain needs no states)
			for r in touched:
				if r in pending and pending[r][0] < pending[r][1]:
					report(a, ins, r, pending[r]); break
			pending = {k: (v[0] + 1, *v[1:]) for k, v in pending.items() if v[0] + 1 < HORIZON}
			for r in d: pending[r] = (0, need, ins, a)
			continue
		if pending:
			for r in regs(ins):
				if r in pending and pending[r][0] < pending[r][1]:
					report(a, ins, r, pending[r]); break
			pending = {k: (v[0] + states, *v[1:]) for k, v in pending.items() if v[0] + states < HORIZON}
	return pending


def check_body(name, body):
	"""Findings for one kernel given as [(address, instruction text, raw objdump tail)]."""
	findings = []
	blocks, succ = basic_blocks(body)
	instate = [None] * len(blocks)  # None = not reached yet
	instate[0] = {}
	work = [0]
	while work:  # forward dataflow, meet = the SHORTEST distance over the predecessors (the worst case)
		k = work.pop()
		out = transfer(body, blocks[k][0], blocks[k][1], instate[k], lambda *x: None)
		for t in succ[k]:
			if instate[t] is None:
				instate[t] = dict(out); work.append(t)
			else:
				changed = False
				for r, v in out.items():
					if r not in instate[t] or v[0] < instate[t][r][0]:
						instate[t][r] = v; changed = True
				if changed: work.append(t)
	seen = set()
	def report(a, ins, r, pend):
		if (a, pend[3]) in seen: return
		seen.add((a, pend[3]))
		findings.append(f"{name[:70]}: '{ins}' @ {a:x} touches {r[0]}{r[1]} {pend[0]} states after '{pend[2]}' @ {pend[3]:x} (hipcc's own floor: {pend[1]})")
	for k, (st, en) in enumerate(blocks):
		if instate[k] is not None: transfer(body, st, en, instate[k], report)
	return findings


def check(lib, only=None):
	findings, n_kernels, n_mfma = [], 0, 0
	for dis in disassemble(lib):
		for name, body in functions(dis):
			if only and not re.search(only, name): continue
			if not body or not any("v_mfma" in ins for _, ins, _ in body): continue
			n_kernels += 1
			n_mfma += sum(1 for _, ins, _ in body if ins.startswith("v_mfma"))
			findings += check_body(name, body)
	return findings, n_kernels, n_mfma


if __name__ == "__main__":
	lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "anncur_amd", "lib", "libanncur_hip.so")
	f, nk, nm = check(lib, sys.argv[2] if len(sys.argv) > 2 else None)
	print(f"{lib}: {nk} kernels with MFMAs, {nm} MFMAs followed, {len(f)} findings")
	for x in f[:60]: print("  " + x)
	sys.exit(1 if f else 0)
