"""Dev tool (experiments build): s_memtime phase stamps of the wave-level select kernels (100 MHz ticks: x ~21 = shader cycles)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anncur_amd import ops, _lib
from anncur_amd.cur import _norm_sorted_pack
dev = torch.device("cuda"); k = int(sys.argv[1]) if len(sys.argv) > 1 else 500
Q, I, K = 10000, 100000, 256
g = torch.Generator(device=dev).manual_seed(0)
Z = torch.randn(64, I, generator=g, device=dev)
X = (torch.randn(Q, 64, generator=g, device=dev) @ torch.randn(64, K, generator=g, device=dev) / 8).bfloat16()
E = (torch.randn(K, 64, generator=g, device=dev) @ Z / 8 / 16 + 0.003 * torch.randn(K, I, generator=g, device=dev)).bfloat16()
Etp, ids = _norm_sorted_pack(E.t().contiguous().float(), 256); Xp = ops.pack_bf16(X, 256)
lib = _lib.load(); lib.anncur_debug_sel_stamps.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
for _ in range(3): ops.score_topk_fused(Xp, Etp, I, k, leading_sample=True, item_ids=ids)
torch.cuda.synchronize()
assert lib.anncur_debug_sel_stamps(1, None) == 0
ops.score_topk_fused(Xp, Etp, I, k, leading_sample=True, item_ids=ids); torch.cuda.synchronize()
out = (ctypes.c_double * 5)()
assert lib.anncur_debug_sel_stamps(0, out) == 0
print("k", k, "last wave-level select launch (final select): median cycles prologue / load loop / finish:", list(out)[:3], "| first batch: search", out[3], "loads in flight", out[4])
