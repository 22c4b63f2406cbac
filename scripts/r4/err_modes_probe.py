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
		# (every mode three times round robin, 30 untimed launches before each timing: the first seconds of a process run 10-15 % slow --
		#  the first version of this probe timed mode 0 cold and read that as the mode's cost)
		res = {}
		for rep in range(3):
			for mode in (0, 1, 2, 4):
				os.environ["ANNCUR_DEBUG_ERR_MODE"] = str(mode)
				for _ in range(30): ops.approx_error_packed(X, Et, A, I)
				ev[0].record()
				for _ in range(20): ops.approx_error_packed(X, Et, A, I)
				ev[1].record(); torch.cuda.synchronize()
				res.setdefault(mode, []).append(ev[0].elapsed_time(ev[1]) / 20)
		for mode, v in res.items():
			print(f"Kp {Kp} mode {mode}: " + " ".join(f"{x:.3f}" for x in v) + f" ms  ({2e-9 * Q * I * Kp / v[-1]:.0f} TFLOP/s)", flush=True)

if __name__ == "__main__":
	main()
