"""Dev tool (experiments build, GPU box): which CUs a bit of hipExtStreamCreateWithCUMask's mask stands for -- one stream per bit (and a few
multi-bit masks), a kernel whose workgroups report HW_REG_XCC_ID / HW_REG_HW_ID.
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so python scripts/cumask_map.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anncur_amd import _lib
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipStreamDestroy.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda"); torch.zeros(1, device=dev)
lib = _lib.load()
lib.anncur_debug_where.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
N = 4096
out = torch.zeros(2 * N, dtype=torch.int32, device=dev)
def where(bits):
	words = (ctypes.c_uint32 * 8)(*[int(sum(1 << b for b in range(32) if 32 * w + b in bits)) for w in range(8)])
	s = ctypes.c_void_p()
	assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words) == 0
	out.zero_(); torch.cuda.synchronize()
	assert lib.anncur_debug_where(out.data_ptr(), N, s) == 0
	torch.cuda.synchronize()
	hip.hipStreamDestroy(s)
	a = out.cpu().numpy().reshape(N, 2)
	xcc = a[:, 0] & 0xf; hw = a[:, 1]
	cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
	return sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
full = where(set(range(256)))
print("no restriction:", len(full), "distinct (xcc, se, sh, cu)")
for b in list(range(0, 40)) + [63, 64, 65, 127, 128, 200, 255]:
	w = where({b})
	print("bit %3d -> %s" % (b, w if len(w) <= 4 else f"{len(w)} places"), flush=True)
for name, bits in (("low 64", set(range(64))), ("low 96", set(range(96))), ("every 4th", set(range(0, 256, 4))), ("every 8th", set(range(0, 256, 8)))):
	w = where(bits)
	per_xcc = {x: sum(1 for t in w if t[0] == x) for x in range(8)}
	print(name, len(w), "CUs; per XCC:", per_xcc, flush=True)
