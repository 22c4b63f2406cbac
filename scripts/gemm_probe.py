"""Dev tool: speed of the strided fp32-MFMA GEMM and of approx_error on headline-sized operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anncur_amd import ops
dev = torch.device("cuda")
def t(fn, n=3):
	fn(); torch.cuda.synchronize()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	for _ in range(n): fn()
	e1.record(); torch.cuda.synchronize()
	return e0.elapsed_time(e1) / n
Q, I = 10000, 100000
A = torch.randn(Q, I, device=dev).bfloat16()
for K in (256, 1024):
	X = torch.randn(Q, K, device=dev).bfloat16(); Et = torch.randn(I, K, device=dev).bfloat16()
	if K <= 512:
		Xp = ops.pack_bf16(X, K); Etp = ops.pack_bf16(Et, K, row_multiple=32)
		ms = t(lambda: ops.approx_error_packed(Xp, Etp, A, I), n=10)
		print("approx_error_packed bf16 K=%d: %.3f ms  %.1f TFLOP/s" % (K, ms, 2.0 * Q * I * K / ms / 1e9), flush=True)
	ms = t(lambda: ops.approx_error(X, Et, A))
	print("approx_error bf16 K=%d: %.2f ms  %.1f TFLOP/s" % (K, ms, 2.0 * Q * I * K / ms / 1e9), flush=True)
	Xf, Etf = X.float(), Et.float()
	ms = t(lambda: ops.approx_error(Xf, Etf, A))
	print("approx_error fp32 operands K=%d: %.2f ms  %.1f TFLOP/s" % (K, ms, 2.0 * Q * I * K / ms / 1e9), flush=True)
Qs = 2000
X = torch.randn(Qs, 1024, device=dev).bfloat16(); Et = torch.randn(I, 1024, device=dev).bfloat16()
ms = t(lambda: ops.gemm(X, Et.t()))
print("gemm bf16 -> fp32 [%d x %d x 1024]: %.2f ms  %.1f TFLOP/s" % (Qs, I, ms, 2.0 * Qs * I * 1024 / ms / 1e9), flush=True)
ms = t(lambda: torch.matmul(X, Et.t()))
print("torch.matmul bf16 (hipBLASLt) same shape: %.2f ms  %.1f TFLOP/s" % (ms, 2.0 * Qs * I * 1024 / ms / 1e9), flush=True)
