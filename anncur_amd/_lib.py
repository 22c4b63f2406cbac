"""ctypes binding of libanncur_hip.so (the C ABI declared in include/anncur_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call
fails, this module raises.  Build with ``python __graft_entry__.py`` (or
``make -C anncur_amd/csrc``).
"""
import ctypes
import os
from ctypes import c_char_p, c_int, c_int32, c_int64, c_size_t, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# ANNCUR_LIB: measurement scripts point this at the -DANNCUR_TIMING_EXPERIMENTS build (`make -C anncur_amd/csrc experiments`)
LIB_PATH = os.environ.get("ANNCUR_LIB") or os.path.join(_HERE, "lib", "libanncur_hip.so")
# the experiments library (ablation knobs, A/B variants such as the tile-ring sweep body; never loaded by the product)
IS_EXPERIMENTS_LIB = os.path.basename(LIB_PATH).startswith(("libanncur_hip_exp", "libanncur_hip_v_"))

F32, BF16, F64 = 0, 1, 2
TOPK_LEADING_SAMPLE = 1
TOPK_MFMA16 = 2
TOPK_QT1 = 4
TOPK_MFMA32 = 8
TOPK_RING = 16
TOPK_STAGED = 32
MAX_TOPK = 2048

_p32 = POINTER(c_int32)

# name -> (restype, argtypes); mirrors include/anncur_hip.h one to one
SIGNATURES = {
	"anncur_version": (c_int, []),
	"anncur_last_error": (c_char_p, []),
	"anncur_device_info": (c_int, [POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
	"anncur_gather_cols": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int, c_int64, c_void_p]),
	"anncur_gather_rows": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_int, c_int64, c_void_p]),
	"anncur_gemm": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64,
							c_int64, c_int64, c_int64, c_void_p]),
	"anncur_gemm_ex": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64,
							   c_int64, c_int64, c_int64, ctypes.c_float, ctypes.c_float, c_void_p, c_int64, c_int64, c_void_p]),
	"anncur_sumsq": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
	"anncur_scale_copy": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, ctypes.c_float, c_void_p, c_void_p]),
	"anncur_gemm_f64": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64,
								ctypes.c_double, ctypes.c_double, c_void_p, c_int64, c_int64, c_void_p]),
	"anncur_convert_f64": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int, c_int64, c_int64, c_int64, c_int64, ctypes.c_double, c_void_p, c_void_p]),
	"anncur_diff_sumsq_f64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
	"anncur_approx_error_packed": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
	"anncur_eval_fused_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32, c_int32]),
	"anncur_eval_fused": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_int64, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p,
								  c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
	"anncur_eval_fused_ex": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p,
								  c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
	"anncur_approx_error": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int64, c_int64,
									c_int64, c_void_p, c_void_p, c_void_p]),
	"anncur_rowwise_topk": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
	"anncur_gather_tables": (c_int, [c_void_p, c_int32, c_int64, c_int, c_void_p, c_void_p]),
	"anncur_rowwise_topk_gather": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_int32, c_void_p, c_void_p,
										   c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
	"anncur_score_topk_workspace_bytes": (c_size_t, [c_int64, c_int64, c_int32, c_int32]),
	"anncur_score_topk_supported": (c_int, [c_int64, c_int64, c_int32, c_int32]),
	"anncur_score_topk": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p,
								  c_void_p, c_size_t, c_void_p]),
	"anncur_score_topk_ex": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p,
									 c_void_p, c_size_t, c_int32, c_void_p, c_void_p]),
	"anncur_eval_topk": (c_int, [c_void_p, c_int, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32,
								 c_void_p, c_void_p, c_void_p, c_size_t, c_int32, c_void_p, c_void_p, c_void_p]),
	"anncur_score_topk_timed": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int32, c_int32, c_void_p, c_void_p,
										c_void_p, c_size_t, c_int32, c_void_p, c_void_p, POINTER(ctypes.c_float)]),
	"anncur_score_topk_plan": (c_int, [c_int64, c_int64, c_int32, c_int32, _p32]),
	"anncur_score_topk_plan_ex": (c_int, [c_int64, c_int64, c_int32, c_int32, c_int32, _p32, c_int32]),
	"anncur_score_topk_survivors": (c_int, [c_void_p, c_int64, c_int64, c_int32, c_int32, c_int32, ctypes.POINTER(ctypes.c_double), c_void_p]),
	"anncur_rerank": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
	"anncur_overlap_counts": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int64, _p32, _p32, c_int32, c_void_p, c_void_p]),
	"anncur_copy_bytes": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
	"anncur_ivf_build_lists": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
	"anncur_ivf_list_means": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
	"anncur_renorm_rows": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p]),
	"anncur_ivf_scan": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
	"anncur_ivf_group_scores": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
	"anncur_ivf_group_scores_bf16": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
	"anncur_ivf_group_scores_dev": (c_int, [c_void_p, c_int, c_int64, c_int32, c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
	"anncur_rowwise_topk_ragged": (c_int, [c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
	"anncur_ivf_search_tile": (c_int32, [c_int, c_int32, c_int64, c_int64, c_int64]),
	"anncur_ivf_search_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32, c_int32, c_int64]),
	"anncur_ivf_search_grouped": (c_int, [c_void_p, c_int, c_int64, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_int32, c_int64,
										  c_void_p, c_int64, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
	"anncur_ivf_map_ids": (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
	"anncur_norm_buckets": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
	"anncur_convert": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int, c_int64, c_int64, c_int64, c_void_p]),
}

_lib = None


class AnncurHipError(RuntimeError):
	pass


def load():
	"""Load (once) and return the ctypes handle.  Raises if the library is absent."""
	global _lib
	if _lib is not None:
		return _lib
	if not os.path.isfile(LIB_PATH):
		raise AnncurHipError(
			f"{LIB_PATH} not found: the HIP extension is not built. Run `python __graft_entry__.py` "
			"(or `make -C anncur_amd/csrc`). anncur_amd has no CPU fallback.")
	# torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  The library binds to whichever libamdhip64 the
	# process loaded first; had it pulled in /opt/rocm's copy before torch loaded its own, the process would hold two runtimes and
	# the kernels here would launch on one that never saw torch's device context ("no ROCm-capable device is detected").
	import torch  # noqa: F401
	lib = ctypes.CDLL(LIB_PATH)
	for name, (res, args) in SIGNATURES.items():
		fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
		fn.restype = res
		fn.argtypes = args
	_lib = lib
	return lib


def check(rc, what=""):
	if rc != 0:
		msg = load().anncur_last_error()
		raise AnncurHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def device_info():
	lib = load()
	n_cu, wave = c_int(0), c_int(0)
	buf = ctypes.create_string_buffer(64)
	check(lib.anncur_device_info(ctypes.byref(n_cu), ctypes.byref(wave), buf, 64), "device_info")
	return {"n_cu": n_cu.value, "wave_size": wave.value, "arch": buf.value.decode()}
