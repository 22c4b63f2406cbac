"""Shared helpers of the static ISA checks (check_mfma_hazards.py, check_lds_hazards.py): extract the gfx950 code objects of a built
library with llvm-objdump --offloading, disassemble them, split the listing into functions of (address, instruction, raw tail)."""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
SMEM = re.compile(r"^\s*(s_load_|s_buffer_load_|s_memtime|s_memrealtime|s_scratch_load|s_atc_probe|s_dcache)")
KERNELS = re.compile(r"ivf_tile128_kernel|score_kernel|score16_kernel|score16r_kernel|evalf_kernel|error_lds_kernel|scoreq1_kernel|scoreq16_kernel|wide_kernel|error_kernel")


def disassemble(lib: str):
	tmp = tempfile.mkdtemp(prefix="anncur_co_")
	try:
		shutil.copy(lib, os.path.join(tmp, "lib.so"))
		subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
		out = []
		for f in sorted(os.listdir(tmp)):
			if "amdgcn" in f and os.path.getsize(os.path.join(tmp, f)) > 0:
				out.append(subprocess.run([f"{LLVM}/llvm-objdump", "-d", f], cwd=tmp, check=True, capture_output=True, text=True).stdout)
		return out
	finally:
		shutil.rmtree(tmp, ignore_errors=True)


def functions(dis: str):
	name, body = None, []
	for line in dis.splitlines():
		m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
		if m:
			if name: yield name, body
			name, body = m.group(1), []
		elif name and "//" in line:
			ins, _, tail = line.partition("//")
			am = re.match(r"\s*([0-9A-Fa-f]+):", tail)
			if am: body.append((int(am.group(1), 16), ins.strip(), tail))
	if name: yield name, body


def _lgkm(ins: str):
	"""lgkmcnt value of an s_waitcnt (None: the instruction does not wait on lgkmcnt)."""
	if not ins.lstrip().startswith("s_waitcnt"):
		return None
	m = re.search(r"lgkmcnt\((\d+)\)", ins)
	if m:
		return int(m.group(1))
	m = re.search(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s", ins + " ")  # raw immediate: lgkmcnt = bits 8..11
	if m:
		return (int(m.group(1), 0) >> 8) & 0xF
	return None if ("vmcnt" in ins or "expcnt" in ins) else 0


