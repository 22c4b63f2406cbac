// Shared helpers for libanncur_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include "anncur_hip.h"

void anncur_set_error(const char *fmt, ...);
// internal cross-file helper (topk.hip): out[q*out_stride] = k-th largest of G[q, :n] (n <= 2048), one wave per row
// coarse: a lower bound within one bf16 ulp of the exact value (two histogram passes instead of four)
constexpr int LADDER_LEVELS = 8;   // threshold ladder of the 16x16x32 sweep (score16.hpp): levels per query, counted in eight 16-bit fields
int anncur_internal_kth_value(const float *G, int64_t Q, int n, int64_t ldg, int k, float *out, int64_t out_stride, hipStream_t st, int coarse = 0,
							  float *ladder = nullptr, int k2 = 1, float *tau2 = nullptr);
// internal cross-file helper (gemm.hip): anncur_approx_error without zeroing the sums (adds a column range to them)
int anncur_internal_approx_error_acc(const void *X, int x_dtype, int64_t ldx, const void *Et, int e_dtype, int64_t lde, const void *Aex, int a_dtype,
									 int64_t lda, int64_t Q, int64_t I, int64_t K, float *err_sq, float *norm_sq, void *stream);

// misc.hip: per-device caches (one process may drive several GPUs)
int anncur_ensure_dyn_lds(const void *fn, int bytes);  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (function, device)
int anncur_num_cu();                                   // CU count of the current device
int anncur_event_pool(hipEvent_t **out, int n);        // n (<= 16) cached timing-less events of the current device
// topk.hip: rows the one-wave-per-row exact scan keeps in flight on the whole chip (occupancy x CUs x 4 rows per workgroup)
int anncur_internal_scan_rows_in_flight(int dtype);

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

// ---- cross-lane scans / reductions on the VALU (DPP).  A `__shfl*` is a ds_bpermute: an LDS round trip (~100+ cycles exposed in a
// dependent chain), and the LDS pipe turned out to be the limit of the candidate-select kernels (a seven-round segment search cost
// 5 k cycles per wave with 16 waves per CU competing for it).  These are the classic gfx9 wave64 sequences: row_shr 1/2/3 of the
// input, row_shr 4 and 8 of the running value under bank masks, then row_bcast 15 and 31 under row masks; a lane whose source is
// masked off or out of range keeps `identity`.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t identity, uint32_t x) {
	return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)x, CTRL, ROW_MASK, BANK_MASK, false);
}
struct DppAdd { static constexpr uint32_t ID = 0u; __device__ static uint32_t op(uint32_t a, uint32_t b) { return a + b; } };
struct DppMin { static constexpr uint32_t ID = 0xffffffffu; __device__ static uint32_t op(uint32_t a, uint32_t b) { return a < b ? a : b; } };
struct DppMax { static constexpr uint32_t ID = 0u; __device__ static uint32_t op(uint32_t a, uint32_t b) { return a > b ? a : b; } };
// inclusive scan over the lanes 0..lane (lane 63 holds the wave's total)
template <typename OP>
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
	uint32_t t = x;
	t = OP::op(t, dpp_mov<0x111>(OP::ID, x));              // row_shr:1
	t = OP::op(t, dpp_mov<0x112>(OP::ID, x));              // row_shr:2
	t = OP::op(t, dpp_mov<0x113>(OP::ID, x));              // row_shr:3
	t = OP::op(t, dpp_mov<0x114, 0xf, 0xe>(OP::ID, t));    // row_shr:4  bank_mask:0xe
	t = OP::op(t, dpp_mov<0x118, 0xf, 0xc>(OP::ID, t));    // row_shr:8  bank_mask:0xc
	t = OP::op(t, dpp_mov<0x142, 0xa, 0xf>(OP::ID, t));    // row_bcast:15 row_mask:0xa
	t = OP::op(t, dpp_mov<0x143, 0xc, 0xf>(OP::ID, t));    // row_bcast:31 row_mask:0xc
	return t;
}
template <typename OP>
__device__ __forceinline__ uint32_t wave_reduce(uint32_t x) {  // the same value in every lane (wave-uniform)
	return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl<OP>(x), WAVE - 1);
}

// value of lane (lane ^ stride): strides 1, 2 are one quad_perm, 4 = row_half_mirror (^7) then a quad reversal (^3), 8 = row_mirror
// (^15) then row_half_mirror (^7) -- VALU only; 16 and 32 go through ds_bpermute (`stride` must fold to a constant)
__device__ __forceinline__ uint32_t lane_xor(uint32_t x, int stride) {
	if (stride == 1) return dpp_mov<0xB1>(x, x);                       // quad_perm:[1,0,3,2]
	if (stride == 2) return dpp_mov<0x4E>(x, x);                       // quad_perm:[2,3,0,1]
	if (stride == 4) return dpp_mov<0x1B>(x, dpp_mov<0x141>(x, x));    // row_half_mirror, quad_perm:[3,2,1,0]
	if (stride == 8) return dpp_mov<0x141>(x, dpp_mov<0x140>(x, x));   // row_mirror, row_half_mirror
	return (uint32_t)__shfl_xor((int)x, stride);
}

template <typename T>
__device__ __forceinline__ float load_as_f32(const T *p);
template <>
__device__ __forceinline__ float load_as_f32<float>(const float *p) { return *p; }
template <>
__device__ __forceinline__ float load_as_f32<uint16_t>(const uint16_t *p) { return bf16_bits_to_f32(*p); }
