"""Dev tool (GPU box): a11 on the fused path's operands (anncur_approx_error_packed) at cfg2 / cfg4-per-GPU size, bf16 and fp32 exact matrix.
  python scripts/error_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anncur_amd import ops
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
for Q, I, K in ((10000, 100000, 256), (10000, 100000, 128), (6250, 200000, 512)):
	X = torch.randn(Q, K, generator=g, device=dev).bfloat16()
	E = (torch.randn(I, K, generator=g, device=dev) / K ** 0.5).bfloat16()
	Xp, Etp = ops.pack_bf16(X, K), ops.pack_bf16(E, K, row_multiple=32)
	for adt in (torch.bfloat16, torch.float32):
		A = torch.randn(Q, I, generator=g, device=dev).to(adt)
		for _ in range(3): ops.approx_error_packed(Xp, Etp, A, I)
		torch.cuda.synchronize()
		ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
		ev[0].record()
		for _ in range(20): ops.approx_error_packed(Xp, Etp, A, I)
		ev[1].record(); torch.cuda.synchronize()
		ms = ev[0].elapsed_time(ev[1]) / 20
		flops, byts = 2.0 * Q * K * I, Q * I * A.element_size()
		print(f"Q={Q} I={I} Kp={K} exact {str(adt)[6:]:9s}: {ms:.3f} ms = {flops / ms / 1e9:.0f} TFLOP/s ({flops / ms / 1e9 / 2500:.2f} of peak), exact matrix {byts / ms / 1e9:.2f} TB/s", flush=True)
		del A
