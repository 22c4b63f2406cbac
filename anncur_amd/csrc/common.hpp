// Shared helpers for libanncur_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "anncur_hip.h"

void anncur_set_error(const char *fmt, ...);
// internal cross-file helper (topk.hip): out[q*out_stride] = k-th largest of G[q, :n] (n <= 2048), one wave per row
int anncur_internal_kth_value(const float *G, int64_t Q, int n, int64_t ldg, int k, float *out, int64_t out_stride, hipStream_t st);
// internal cross-file helper (gemm.hip): anncur_approx_error without zeroing the sums (adds a column range to them)
int anncur_internal_approx_error_acc(const void *X, int x_dtype, int64_t ldx, const void *Et, int e_dtype, int64_t lde, const void *Aex, int a_dtype,
									 int64_t lda, int64_t Q, int64_t I, int64_t K, float *err_sq, float *norm_sq, void *stream);

// misc.hip: per-device caches (one process may drive several GPUs)
int anncur_ensure_dyn_lds(const void *fn, int bytes);  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (function, device)
int anncur_num_cu();                                   // CU count of the current device

#define ANNCUR_REQUIRE(cond, code, ...)                 \
	do {                                                \
		if (!(cond)) {                                  \
			anncur_set_error(__VA_ARGS__);              \
			return (code);                              \
		}                                               \
	} while (0)

#define ANNCUR_HIP_OK(expr)                                                              \
	do {                                                                                 \
		hipError_t e__ = (expr);                                                         \
		if (e__ != hipSuccess) {                                                         \
			anncur_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
			return ANNCUR_E_HIP;                                                         \
		}                                                                                \
	} while (0)

#define ANNCUR_LAUNCH_OK()                                                               \
	do {                                                                                 \
		hipError_t e__ = hipGetLastError();                                              \
		if (e__ != hipSuccess) {                                                         \
			anncur_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
			return ANNCUR_E_HIP;                                                         \
		}                                                                                \
	} while (0)

static inline bool dtype_ok(int d) { return d == ANNCUR_F32 || d == ANNCUR_BF16; }
static inline size_t dtype_size(int d) { return d == ANNCUR_F32 ? 4 : 2; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ device helpers
#define WAVE 64

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// round-to-nearest-even, NaN stays NaN
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
	uint32_t u = __float_as_uint(f);
	if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
	return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// order-preserving map float -> uint32 (larger float => larger uint)
__device__ __forceinline__ uint32_t f32_sortable(float f) {
	const uint32_t u = __float_as_uint(f);
	return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);  // negative: flip all bits, positive: set the top bit (3 VALU ops)
}
__device__ __forceinline__ float f32_unsortable(uint32_t s) {
	uint32_t u = (s & 0x80000000u) ? (s & 0x7fffffffu) : ~s;
	return __uint_as_float(u);
}
// composite key: larger = better score, ties -> smaller index wins.  All keys of one row are distinct.
__device__ __forceinline__ uint64_t make_key(float v, uint32_t idx) {
	return ((uint64_t)f32_sortable(v) << 32) | (uint64_t)(0xffffffffu - idx);
}
__device__ __forceinline__ float key_val(uint64_t k) { return f32_unsortable((uint32_t)(k >> 32)); }
__device__ __forceinline__ uint32_t key_idx(uint64_t k) { return 0xffffffffu - (uint32_t)k; }

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

template <typename T>
__device__ __forceinline__ float load_as_f32(const T *p);
template <>
__device__ __forceinline__ float load_as_f32<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float load_as_f32<uint16_t>(const uint16_t *p) { return bf16_bits_to_f32(*p); }
