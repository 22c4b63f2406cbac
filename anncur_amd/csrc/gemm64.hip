// fp64 building blocks of the parity-grade on-device pseudo-inverse (anncur_amd/pinv.py): strided GEMM on the fp64 matrix
// cores (v_mfma_f64_16x16x4_f64), conversions, norms.  Sizes are the anchor blocks' (<= a few thousand squared): the kernels are
// written for correctness and any strides, not for the fp64 roofline (a 2048 x 1024 pseudo-inverse is ~40 products of 4-9 GFlop).
// Replaces the host call numpy.linalg.pinv of eval/matrix_approx_zeshel.py:47,49 when the caller asks for it.
#include "common.hpp"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int DBM = 64, DBN = 64, DBK = 32;
constexpr int DPITCH = 80;  // doubles per LDS row: 160 dwords = 32 (mod 64 banks), the four k-rows of a fragment read do not collide

// C(m,n) = alpha * sum_k A(m,k) B(k,n) + beta * Cin(m,n); element (i,j) of an operand at base + i*s0 + j*s1.
// Workgroup = 64 x 64 outputs, wave (wm, wn) = 32 x 32 = 2 x 2 MFMA tiles.  Fragment maps (MI355X guide, f64 MFMA):
// A[row l&15][k l>>4], B[k l>>4][col l&15], C/D col = l&15, row = (l>>4) + 4*reg.
// The grids are tiny (a 512 x 256 anchor block: 16 workgroups) and the kernel is latency-bound per k-tile: global loads -> LDS ->
// barrier -> MFMAs.  Round 3: 32-deep k-tiles and the NEXT k-tile's global loads issued before the current one's MFMAs (registers),
// so a k-tile costs one load latency per 32 k instead of one per 16 with nothing overlapped (64 -> ~25 us per product at 512 x 256).
__global__ __launch_bounds__(256) void gemm_f64_kernel(const double *__restrict__ A, int64_t a_sm, int64_t a_sk, const double *__restrict__ B,
														int64_t b_sk, int64_t b_sn, double *C, int64_t c_sm, int64_t c_sn, int64_t M, int64_t N,
														int64_t K, double alpha, double beta, const double *Cin, int64_t i_sm, int64_t i_sn) {
	__shared__ double As[DBK][DPITCH], Bs[DBK][DPITCH];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
	const int64_t m0 = (int64_t)blockIdx.y * DBM, n0 = (int64_t)blockIdx.x * DBN;
	f64x4 acc[2][2];
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < 2; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
	const bool a_kfast = a_sk == 1, b_kfast = b_sk == 1;  // which index runs along consecutive threads (coalescing only)
	constexpr int PER = DBM * DBK / 256;  // elements of each operand tile per thread
	double ra[PER], rb[PER];
	auto load = [&](int64_t k0) {
#pragma unroll
		for (int i = 0; i < PER; ++i) {
			const int e = tid + 256 * i;
			const int am = a_kfast ? e / DBK : e % DBM, ak = a_kfast ? e % DBK : e / DBM;
			ra[i] = (m0 + am < M && k0 + ak < K) ? A[(m0 + am) * a_sm + (k0 + ak) * a_sk] : 0.0;
			const int bn = b_kfast ? e / DBK : e % DBN, bk = b_kfast ? e % DBK : e / DBN;
			rb[i] = (n0 + bn < N && k0 + bk < K) ? B[(k0 + bk) * b_sk + (n0 + bn) * b_sn] : 0.0;
		}
	};
	auto stash = [&]() {
#pragma unroll
		for (int i = 0; i < PER; ++i) {
			const int e = tid + 256 * i;
			const int am = a_kfast ? e / DBK : e % DBM, ak = a_kfast ? e % DBK : e / DBM;
			As[ak][am] = ra[i];
			const int bn = b_kfast ? e / DBK : e % DBN, bk = b_kfast ? e % DBK : e / DBN;
			Bs[bk][bn] = rb[i];
		}
	};
	load(0);
	for (int64_t k0 = 0; k0 < K; k0 += DBK) {
		stash();
		__syncthreads();
		if (k0 + DBK < K) load(k0 + DBK);   // in flight during the MFMAs below
#pragma unroll
		for (int ks = 0; ks < DBK / 4; ++ks) {
			double a[2], b[2];
#pragma unroll
			for (int i = 0; i < 2; ++i) a[i] = As[ks * 4 + (lane >> 4)][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
			for (int j = 0; j < 2; ++j) b[j] = Bs[ks * 4 + (lane >> 4)][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
			for (int i = 0; i < 2; ++i)
#pragma unroll
				for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
		}
		__syncthreads();
	}
#pragma unroll
	for (int i = 0; i < 2; ++i)
#pragma unroll
		for (int j = 0; j < 2; ++j)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int64_t m = m0 + wm * 32 + i * 16 + (lane >> 4) + 4 * r, n = n0 + wn * 32 + j * 16 + (lane & 15);
				if (m < M && n < N) {
					double v = alpha * acc[i][j][r];
					if (Cin) v += beta * Cin[m * i_sm + n * i_sn];
					C[m * c_sm + n * c_sn] = v;
				}
			}
}

template <typename TS, typename TD>
__device__ __forceinline__ TD cvt64(TS x);
template <> __device__ __forceinline__ double cvt64<float, double>(float x) { return (double)x; }
template <> __device__ __forceinline__ double cvt64<uint16_t, double>(uint16_t x) { return (double)bf16_bits_to_f32(x); }
template <> __device__ __forceinline__ float cvt64<double, float>(double x) { return (float)x; }  // round to nearest even, once
template <> __device__ __forceinline__ double cvt64<double, double>(double x) { return x; }

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert64_kernel(const TS *__restrict__ src, int64_t s0, int64_t s1, TD *__restrict__ dst, int64_t d0, int64_t d1,
														 int64_t M, int64_t N, double alpha, const double *__restrict__ divide_by) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= M * N) return;
	const int64_t r = i / N, c = i % N;
	double v = (double)cvt64<TS, double>(src[r * s0 + c * s1]);
	if (divide_by) v = alpha * v / divide_by[0]; else v = alpha * v;
	dst[r * d0 + c * d1] = (TD)v;
}

// out[0] += sum (x - y)^2, out[1] += sum x^2   (y may be null: out[0] += 0)
__global__ __launch_bounds__(256) void diff_sumsq_f64_kernel(const double *__restrict__ X, const double *__restrict__ Y, int64_t n, double *__restrict__ out) {
	double d2 = 0.0, x2 = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
		const double x = X[i], d = Y ? x - Y[i] : 0.0;
		d2 += d * d;
		x2 += x * x;
	}
	for (int s = 32; s > 0; s >>= 1) { d2 += __shfl_xor(d2, s); x2 += __shfl_xor(x2, s); }
	if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], d2); atomicAdd(&out[1], x2); }
}

}  // namespace

extern "C" int anncur_gemm_f64(const double *A, int64_t a_sm, int64_t a_sk, const double *B, int64_t b_sk, int64_t b_sn, double *C, int64_t c_sm,
							   int64_t c_sn, int64_t M, int64_t N, int64_t K, double alpha, double beta, const double *Cin, int64_t i_sm, int64_t i_sn,
							   void *stream) {
	ANNCUR_REQUIRE(M >= 0 && N >= 0 && K >= 0, ANNCUR_E_INVALID, "gemm_f64: negative dimension");
	if (M == 0 || N == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(C && (K == 0 || (A && B)), ANNCUR_E_INVALID, "gemm_f64: null pointer");
	const int64_t gx = ceil_div64(N, DBN), gy = ceil_div64(M, DBM);
	ANNCUR_REQUIRE(gy <= 65535 && gx < (int64_t)0x7fffffff, ANNCUR_E_INVALID, "gemm_f64: M too large for one launch");
	hipLaunchKernelGGL(gemm_f64_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream, A, a_sm, a_sk, B, b_sk, b_sn, C, c_sm, c_sn,
					   M, N, K, alpha, beta, Cin, i_sm, i_sn);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_convert_f64(const void *src, int src_dtype, int64_t s0, int64_t s1, void *dst, int dst_dtype, int64_t d0, int64_t d1, int64_t M,
								  int64_t N, double alpha, const double *divide_by, void *stream) {
	ANNCUR_REQUIRE(M >= 0 && N >= 0, ANNCUR_E_INVALID, "convert_f64: negative dimension");
	if (M * N == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(src && dst, ANNCUR_E_INVALID, "convert_f64: null pointer");
	hipStream_t st = (hipStream_t)stream;
	const dim3 grid((unsigned)ceil_div64(M * N, 256));
#define CV(TS, TD) hipLaunchKernelGGL((convert64_kernel<TS, TD>), grid, dim3(256), 0, st, (const TS *)src, s0, s1, (TD *)dst, d0, d1, M, N, alpha, divide_by)
	if (src_dtype == ANNCUR_F32 && dst_dtype == ANNCUR_F64) CV(float, double);
	else if (src_dtype == ANNCUR_BF16 && dst_dtype == ANNCUR_F64) CV(uint16_t, double);
	else if (src_dtype == ANNCUR_F64 && dst_dtype == ANNCUR_F64) CV(double, double);
	else if (src_dtype == ANNCUR_F64 && dst_dtype == ANNCUR_F32) CV(double, float);
	else { anncur_set_error("convert_f64: unsupported dtype pair (%d -> %d)", src_dtype, dst_dtype); return ANNCUR_E_INVALID; }
#undef CV
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_diff_sumsq_f64(const double *X, const double *Y, int64_t n, double *out2, void *stream) {
	ANNCUR_REQUIRE(out2 && n >= 0 && (n == 0 || X), ANNCUR_E_INVALID, "diff_sumsq_f64: bad arguments");
	hipStream_t st = (hipStream_t)stream;
	ANNCUR_HIP_OK(hipMemsetAsync(out2, 0, 16, st));
	if (n == 0) return ANNCUR_OK;
	const int64_t blocks = ceil_div64(n, 256 * 8);
	hipLaunchKernelGGL(diff_sumsq_f64_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, st, X, Y, n, out2);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}
