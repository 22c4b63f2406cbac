// Strided GEMM with exact fp32 accumulation on v_mfma_f32_32x32x2_f32 (gfx950), and the
// fused approximation-error reduction.  Used for the index build (E^T = R^T U^T), the dense
// reconstruction API (get / get_rows / get_cols / get_complete_*), cur_oracle's products and
// small / fp32 problems.  The bf16 hot path (fused score + top-k) lives in score_fused.hip.
//
// Tile: 128 x 128 x 16 per 256-thread workgroup, 4 waves as 2 x 2, each wave 64 x 64 =
// 2 x 2 MFMA tiles (64 accumulator VGPRs).  LDS holds A and B k-major so a wave's operand
// read is 32 consecutive floats per half-wave (conflict-free ds_read_b32).
#include <type_traits>
#include "common.hpp"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128, BK = 16, PADM = 1;
constexpr int LDT = BM + PADM;  // LDS row pitch in floats (k-major tiles)

template <typename T>
__device__ __forceinline__ float ld_elem(const T *p, int64_t off, bool ok) {
	return ok ? load_as_f32<T>(p + off) : 0.f;
}

template <typename T>
__device__ __forceinline__ void st_elem(T *p, int64_t off, float v);
template <>
__device__ __forceinline__ void st_elem<float>(float *p, int64_t off, float v) { p[off] = v; }
template <>
__device__ __forceinline__ void st_elem<uint16_t>(uint16_t *p, int64_t off, float v) { p[off] = f32_to_bf16_bits(v); }

// Loads one BM x BK (or BK x BN) operand tile into 8 registers per thread.
// kfast: the operand's k index is the unit-stride one -> walk k fastest for coalescing.
template <typename T>
__device__ __forceinline__ void load_tile(const T *__restrict__ P, int64_t s_mn, int64_t s_k, int64_t mn0, int64_t k0,
										   int64_t MN, int64_t K, bool kfast, float (&r)[8]) {
	const int t = threadIdx.x;
#pragma unroll
	for (int p = 0; p < 8; ++p) {
		int mn, k;
		if (kfast) { k = t & 15; mn = (t >> 4) + 16 * p; }
		else { mn = t & 127; k = (t >> 7) + 2 * p; }
		const int64_t gm = mn0 + mn, gk = k0 + k;
		r[p] = ld_elem<T>(P, gm * s_mn + gk * s_k, gm < MN && gk < K);
	}
}
__device__ __forceinline__ void store_tile(float *__restrict__ S, bool kfast, const float (&r)[8]) {
	const int t = threadIdx.x;
#pragma unroll
	for (int p = 0; p < 8; ++p) {
		int mn, k;
		if (kfast) { k = t & 15; mn = (t >> 4) + 16 * p; }
		else { mn = t & 127; k = (t >> 7) + 2 * p; }
		S[k * LDT + mn] = r[p];
	}
}

// MODE 0: store C.  MODE 1: err/norm reduction against Aex (C is never written).
template <typename TA, typename TB, typename TC, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(const TA *__restrict__ A, int64_t a_sm, int64_t a_sk,
													const TB *__restrict__ B, int64_t b_sk, int64_t b_sn,
													TC *__restrict__ C, int64_t c_sm, int64_t c_sn, int64_t M, int64_t N,
													int64_t K, float *__restrict__ err_sq, float *__restrict__ norm_sq, float alpha,
													float beta, const float *__restrict__ Cin, int64_t i_sm, int64_t i_sn) {
	__shared__ float As[BK * LDT];
	__shared__ float Bs[BK * LDT];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;
	const int r = lane & 31, h = lane >> 5;
	const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
	const bool a_kfast = (a_sk == 1), b_kfast = (b_sk == 1);

	f32x16 acc[2][2];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < 2; ++j)
#pragma unroll
			for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

	float ra[8], rb[8];
	const int64_t nkt = (K + BK - 1) / BK;
	load_tile<TA>(A, a_sm, a_sk, m0, 0, M, K, a_kfast, ra);
	load_tile<TB>(B, b_sn, b_sk, n0, 0, N, K, b_kfast, rb);
	store_tile(As, a_kfast, ra);
	store_tile(Bs, b_kfast, rb);
	__syncthreads();
	for (int64_t kt = 0; kt < nkt; ++kt) {
		const bool more = kt + 1 < nkt;
		if (more) {
			load_tile<TA>(A, a_sm, a_sk, m0, (kt + 1) * BK, M, K, a_kfast, ra);
			load_tile<TB>(B, b_sn, b_sk, n0, (kt + 1) * BK, N, K, b_kfast, rb);
		}
#pragma unroll
		for (int kk = 0; kk < BK / 2; ++kk) {
			float av[2], bv[2];
#pragma unroll
			for (int i = 0; i < 2; ++i) av[i] = As[(2 * kk + h) * LDT + wm * 64 + i * 32 + r];
#pragma unroll
			for (int j = 0; j < 2; ++j) bv[j] = Bs[(2 * kk + h) * LDT + wn * 64 + j * 32 + r];
#pragma unroll
			for (int i = 0; i < 2; ++i)
#pragma unroll
				for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
		}
		__syncthreads();
		if (more) {
			store_tile(As, a_kfast, ra);
			store_tile(Bs, b_kfast, rb);
		}
		__syncthreads();
	}

	// C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
	for (int i = 0; i < 2; ++i) {
#pragma unroll
		for (int e = 0; e < 16; ++e) {
			const int64_t gm = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
			if (MODE == 0) {
#pragma unroll
				for (int j = 0; j < 2; ++j) {
					const int64_t gn = n0 + wn * 64 + j * 32 + r;
					if (gm < M && gn < N) {
						float v = alpha * acc[i][j][e];
						if (Cin) v += beta * Cin[gm * i_sm + gn * i_sn];   // out = alpha * A.B + beta * Cin (Cin fp32, may alias C element-wise)
						st_elem<TC>(C, gm * c_sm + gn * c_sn, v);
					}
				}
			} else {
				// C doubles as the exact matrix (same strides); reduce (acc - exact)^2 and exact^2 over the row segment
				float se = 0.f, sn = 0.f;
#pragma unroll
				for (int j = 0; j < 2; ++j) {
					const int64_t gn = n0 + wn * 64 + j * 32 + r;
					if (gm < M && gn < N) {
						const float ex = load_as_f32<TC>(C + gm * c_sm + gn * c_sn);
						const float d = acc[i][j][e] - ex;
						se += d * d;
						sn += ex * ex;
					}
				}
#pragma unroll
				for (int d = 16; d > 0; d >>= 1) {  // reduce across the 32 lanes of this half-wave
					se += __shfl_xor(se, d);
					sn += __shfl_xor(sn, d);
				}
				if (r == 0 && gm < M) {
					atomicAdd(&err_sq[gm], se);
					atomicAdd(&norm_sq[gm], sn);
				}
			}
		}
	}
}

template <int MODE, typename F>
int dispatch3(int ad, int bd, int cd, F &&f) {
#define D3(TA, TB, TC) return f((const TA *)nullptr, (const TB *)nullptr, (TC *)nullptr)
	if (ad == ANNCUR_F32) {
		if (bd == ANNCUR_F32) { if (cd == ANNCUR_F32) D3(float, float, float); else D3(float, float, uint16_t); }
		else { if (cd == ANNCUR_F32) D3(float, uint16_t, float); else D3(float, uint16_t, uint16_t); }
	} else {
		if (bd == ANNCUR_F32) { if (cd == ANNCUR_F32) D3(uint16_t, float, float); else D3(uint16_t, float, uint16_t); }
		else { if (cd == ANNCUR_F32) D3(uint16_t, uint16_t, float); else D3(uint16_t, uint16_t, uint16_t); }
	}
#undef D3
}

}  // namespace

extern "C" int anncur_gemm_ex(const void *A, int a_dtype, int64_t a_sm, int64_t a_sk, const void *B, int b_dtype,
							  int64_t b_sk, int64_t b_sn, void *C, int c_dtype, int64_t c_sm, int64_t c_sn, int64_t M,
							  int64_t N, int64_t K, float alpha, float beta, const float *Cin, int64_t i_sm, int64_t i_sn, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(a_dtype) && dtype_ok(b_dtype) && dtype_ok(c_dtype), ANNCUR_E_INVALID, "gemm: bad dtype");
	ANNCUR_REQUIRE(M >= 0 && N >= 0 && K >= 0, ANNCUR_E_INVALID, "gemm: negative dimension");
	if (M == 0 || N == 0) return ANNCUR_OK;  // (empty tensors have null data pointers: nothing to do comes first)
	ANNCUR_REQUIRE(C && (K == 0 || (A && B)), ANNCUR_E_INVALID, "gemm: null pointer");
	const int64_t gx = ceil_div64(N, BN), gy = ceil_div64(M, BM);
	ANNCUR_REQUIRE(gy <= 65535 && gx < (int64_t)0x7fffffff, ANNCUR_E_INVALID, "gemm: M too large for one launch (M <= %d)", 65535 * BM);
	hipStream_t st = (hipStream_t)stream;
	dispatch3<0>(a_dtype, b_dtype, c_dtype, [&](auto *a, auto *b, auto *c) {
		using TA = std::remove_cv_t<std::remove_pointer_t<decltype(a)>>;
		using TB = std::remove_cv_t<std::remove_pointer_t<decltype(b)>>;
		using TC = std::remove_pointer_t<decltype(c)>;
		hipLaunchKernelGGL((gemm_kernel<TA, TB, TC, 0>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, st, (const TA *)A, a_sm,
						   a_sk, (const TB *)B, b_sk, b_sn, (TC *)C, c_sm, c_sn, M, N, K, (float *)nullptr, (float *)nullptr, alpha, beta, Cin, i_sm, i_sn);
		return 0;
	});
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_gemm(const void *A, int a_dtype, int64_t a_sm, int64_t a_sk, const void *B, int b_dtype,
						   int64_t b_sk, int64_t b_sn, void *C, int c_dtype, int64_t c_sm, int64_t c_sn, int64_t M,
						   int64_t N, int64_t K, void *stream) {
	return anncur_gemm_ex(A, a_dtype, a_sm, a_sk, B, b_dtype, b_sk, b_sn, C, c_dtype, c_sm, c_sn, M, N, K, 1.0f, 0.0f, nullptr, 0, 0, stream);
}

// err_sq / norm_sq are ACCUMULATED into (the public entry point zeroes them first; score_fused.hip adds a column range)
int anncur_internal_approx_error_acc(const void *X, int x_dtype, int64_t ldx, const void *Et, int e_dtype, int64_t lde,
									 const void *Aex, int a_dtype, int64_t lda, int64_t Q, int64_t I, int64_t K, float *err_sq,
									 float *norm_sq, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(x_dtype) && dtype_ok(e_dtype) && dtype_ok(a_dtype), ANNCUR_E_INVALID, "approx_error: bad dtype");
	ANNCUR_REQUIRE(Q >= 0 && I >= 1 && K >= 1 && ldx >= K && lde >= K && lda >= I, ANNCUR_E_INVALID, "approx_error: bad shape");
	if (Q == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(X && Et && Aex && err_sq && norm_sq, ANNCUR_E_INVALID, "approx_error: null pointer");
	const int64_t gx = ceil_div64(I, BN), gy = ceil_div64(Q, BM);
	ANNCUR_REQUIRE(gy <= 65535, ANNCUR_E_INVALID, "approx_error: Q too large for one launch");
	hipStream_t st = (hipStream_t)stream;
	dispatch3<1>(x_dtype, e_dtype, a_dtype, [&](auto *a, auto *b, auto *c) {
		using TA = std::remove_cv_t<std::remove_pointer_t<decltype(a)>>;
		using TB = std::remove_cv_t<std::remove_pointer_t<decltype(b)>>;
		using TC = std::remove_pointer_t<decltype(c)>;
		// A operand = X (m = query, k), B operand: B(k, n) = Et[n][k]; "C" = exact matrix, read-only in MODE 1
		hipLaunchKernelGGL((gemm_kernel<TA, TB, TC, 1>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, st, (const TA *)X, ldx,
						   (int64_t)1, (const TB *)Et, (int64_t)1, lde, (TC *)const_cast<void *>(Aex), lda, (int64_t)1, Q, I, K, err_sq,
						   norm_sq, 1.0f, 0.0f, (const float *)nullptr, (int64_t)0, (int64_t)0);
		return 0;
	});
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_approx_error(const void *X, int x_dtype, int64_t ldx, const void *Et, int e_dtype, int64_t lde,
								   const void *Aex, int a_dtype, int64_t lda, int64_t Q, int64_t I, int64_t K, float *err_sq,
								   float *norm_sq, void *stream) {
	ANNCUR_REQUIRE(err_sq && norm_sq && Q >= 0, ANNCUR_E_INVALID, "approx_error: null pointer");
	if (Q == 0) return ANNCUR_OK;
	ANNCUR_HIP_OK(hipMemsetAsync(err_sq, 0, (size_t)Q * 4, (hipStream_t)stream));
	ANNCUR_HIP_OK(hipMemsetAsync(norm_sq, 0, (size_t)Q * 4, (hipStream_t)stream));
	return anncur_internal_approx_error_acc(X, x_dtype, ldx, Et, e_dtype, lde, Aex, a_dtype, lda, Q, I, K, err_sq, norm_sq, stream);
}

// ------------------------------------------------------------------ small helpers for the on-device pseudo-inverse
namespace {
__global__ __launch_bounds__(256) void sumsq_kernel(const float *__restrict__ A, int64_t n_rows, int64_t n_cols, int64_t lda, float *__restrict__ out) {
	float acc = 0.f;
	const int64_t total = n_rows * n_cols;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
		const float v = A[(i / n_cols) * lda + (i % n_cols)];
		acc += v * v;
	}
	for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
	if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
__global__ __launch_bounds__(256) void scale_copy_kernel(const float *__restrict__ src, int64_t s0, int64_t s1, float *__restrict__ dst, int64_t d0,
														  int64_t d1, int64_t M, int64_t N, float alpha, const float *__restrict__ inv_scale) {
	const float a = inv_scale ? alpha / inv_scale[0] : alpha;
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i < M * N) dst[(i / N) * d0 + (i % N) * d1] = a * src[(i / N) * s0 + (i % N) * s1];
}
}  // namespace

extern "C" int anncur_sumsq(const float *A, int64_t n_rows, int64_t n_cols, int64_t lda, float *out, void *stream) {
	ANNCUR_REQUIRE(A && out && n_rows >= 0 && n_cols >= 0 && lda >= n_cols, ANNCUR_E_INVALID, "sumsq: bad arguments");
	hipStream_t st = (hipStream_t)stream;
	ANNCUR_HIP_OK(hipMemsetAsync(out, 0, 4, st));
	if (n_rows * n_cols == 0) return ANNCUR_OK;
	const int64_t blocks = ceil_div64(n_rows * n_cols, 256 * 8);
	hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, st, A, n_rows, n_cols, lda, out);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_scale_copy(const float *src, int64_t s0, int64_t s1, float *dst, int64_t d0, int64_t d1, int64_t M, int64_t N, float alpha,
								 const float *divide_by, void *stream) {
	ANNCUR_REQUIRE(src && dst && M >= 0 && N >= 0, ANNCUR_E_INVALID, "scale_copy: bad arguments");
	if (M * N == 0) return ANNCUR_OK;
	hipLaunchKernelGGL(scale_copy_kernel, dim3((unsigned)ceil_div64(M * N, 256)), dim3(256), 0, (hipStream_t)stream, src, s0, s1, dst, d0, d1, M, N, alpha,
					   divide_by);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}
