"""error_lds_kernel under the experiment modes of ANNCUR_DEBUG_ERR_MODE (experiments build): what bounds it at cfg2 size.
  ANNCUR_LIB=anncur_amd/lib/libanncur_hip_exp.so python scripts/r4/err_modes_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from anncur_amd import ops   # noqa: E402

def main():
	dev = torch.device("cuda", 0)
	Q, I = 10000, 100000
	torch.manual_seed(0)
	A = (torch.randn(Q, 64, device=dev) @ torch.randn(64, I, device=dev) / 8).to(torch.bfloat16)
	ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
	for Kp in (128, 256, 512):
		X = torch.randn(Q, Kp, device=dev).to(torch.bfloat16)
		Et = (torch.randn(I, Kp, device=dev) / Kp ** 0.5).to(torch.bfloat16)
		ref = None
		for mode in (0, 1, 2, 4):
			os.environ["ANNCUR_DEBUG_ERR_MODE"] = str(mode)
			for _ in range(3): out = ops.approx_error_packed(X, Et, A, I)
			ev[0].record()
			for _ in range(10): out = ops.approx_error_packed(X, Et, A, I)
			ev[1].record(); torch.cuda.synchronize()
			ms = ev[0].elapsed_time(ev[1]) / 10
			note = ""
			if mode == 0: ref = [o.clone() for o in out]
			if mode == 99: note = "  equal to mode 0: %s" % all(torch.equal(a, b) for a, b in zip(ref, out))
			print(f"Kp {Kp} mode {mode}: {ms:.3f} ms  ({2e-9 * Q * I * Kp / ms:.0f} TFLOP/s){note}", flush=True)

if __name__ == "__main__":
	main()
