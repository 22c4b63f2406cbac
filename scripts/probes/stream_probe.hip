// Dev tool: read-bandwidth ceiling of different access patterns on a [Q x I] bf16 matrix (the exact scan's input).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/stream_probe scripts/probes/stream_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ unsigned fold(u32x4 v) { return v[0] ^ v[1] ^ v[2] ^ v[3]; }

// P1: one wave per row, PF loads of 16 B in flight per lane (the scan's pattern).  NT: nontemporal loads.
template <int PF, bool NT>
__global__ __launch_bounds__(256) void wave_per_row(const u32x4 *A, int64_t Q, int64_t nvec, unsigned *out) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t q = (int64_t)blockIdx.x * 4 + wave;
	if (q >= Q) return;
	const u32x4 *row = A + q * nvec;
	unsigned acc = 0;
	u32x4 pf[PF];
#pragma unroll
	for (int d = 0; d < PF; ++d) { const int64_t i = (int64_t)d * 64 + lane; pf[d] = NT ? __builtin_nontemporal_load(row + (i < nvec ? i : nvec - 1)) : row[i < nvec ? i : nvec - 1]; }
	for (int64_t s = 0; s * 64 < nvec; s += PF) {
#pragma unroll
		for (int d = 0; d < PF; ++d) {
			const u32x4 cur = pf[d];
			const int64_t in = (s + d + PF) * 64 + lane;
			pf[d] = NT ? __builtin_nontemporal_load(row + (in < nvec ? in : nvec - 1)) : row[in < nvec ? in : nvec - 1];
			acc ^= fold(cur);
		}
	}
	if (acc == 0x12345678u) out[q] = acc;
}
// P2: one 256-thread workgroup per row, 4 KiB contiguous per step.
template <int PF>
__global__ __launch_bounds__(256) void wg_per_row(const u32x4 *A, int64_t Q, int64_t nvec, unsigned *out) {
	const int64_t q = blockIdx.x;
	const u32x4 *row = A + q * nvec;
	unsigned acc = 0;
	u32x4 pf[PF];
#pragma unroll
	for (int d = 0; d < PF; ++d) { const int64_t i = (int64_t)d * 256 + threadIdx.x; pf[d] = row[i < nvec ? i : nvec - 1]; }
	for (int64_t s = 0; s * 256 < nvec; s += PF) {
#pragma unroll
		for (int d = 0; d < PF; ++d) {
			const u32x4 cur = pf[d];
			const int64_t in = (s + d + PF) * 256 + threadIdx.x;
			pf[d] = row[in < nvec ? in : nvec - 1];
			acc ^= fold(cur);
		}
	}
	if (acc == 0x12345678u) out[q] = acc;
}
// P3: flat grid-stride stream.
template <int PF>
__global__ __launch_bounds__(256) void flat(const u32x4 *A, int64_t n, unsigned *out) {
	const int64_t stride = (int64_t)gridDim.x * 256;
	int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	unsigned acc = 0;
	for (; i + (PF - 1) * stride < n; i += PF * stride) {
		u32x4 v[PF];
#pragma unroll
		for (int d = 0; d < PF; ++d) v[d] = A[i + d * stride];
#pragma unroll
		for (int d = 0; d < PF; ++d) acc ^= fold(v[d]);
	}
	for (; i < n; i += stride) acc ^= fold(A[i]);
	if (acc == 0x12345678u) out[0] = acc;
}
// P4: a wave owns a row but the 4 waves of a workgroup advance in lock step over 4 ADJACENT rows?  (no: rows are 200 KB apart)
// P5: two waves per row (each takes alternate 1 KiB pieces) -> 2 KiB contiguous per pair-step.
template <int PF>
__global__ __launch_bounds__(256) void two_waves_per_row(const u32x4 *A, int64_t Q, int64_t nvec, unsigned *out) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int64_t q = (int64_t)blockIdx.x * 2 + (wave >> 1);
	if (q >= Q) return;
	const int half = wave & 1;
	const u32x4 *row = A + q * nvec;
	unsigned acc = 0;
	u32x4 pf[PF];
#pragma unroll
	for (int d = 0; d < PF; ++d) { const int64_t i = (int64_t)(2 * d + half) * 64 + lane; pf[d] = row[i < nvec ? i : nvec - 1]; }
	for (int64_t s = 0; s * 128 < nvec; s += PF) {
#pragma unroll
		for (int d = 0; d < PF; ++d) {
			const u32x4 cur = pf[d];
			const int64_t in = (2 * (s + d + PF) + half) * 64 + lane;
			pf[d] = row[in < nvec ? in : nvec - 1];
			acc ^= fold(cur);
		}
	}
	if (acc == 0x12345678u) out[q] = acc;
}

template <typename F>
float timeit(F f, int n) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	f(); hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	for (int i = 0; i < n; ++i) f();
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	return ms / n;
}
int main() {
	const int64_t Q = 10000, I = 100000, nvec = I * 2 / 16, n = Q * nvec;
	u32x4 *A; unsigned *out;
	OK(hipMalloc(&A, n * 16)); OK(hipMalloc(&out, Q * 4));
	OK(hipMemset(A, 1, n * 16));
	const double gb = (double)n * 16 / 1e9;
#define RUN(name, ...) do { float ms = timeit([&] { __VA_ARGS__; }, 10); printf("%-44s %.4f ms  %.0f GB/s\n", name, ms, gb / ms * 1e3); fflush(stdout); } while (0)
	RUN("wave/row PF=8", hipLaunchKernelGGL((wave_per_row<8, false>), dim3(2500), dim3(256), 0, 0, A, Q, nvec, out));
	RUN("wave/row PF=4", hipLaunchKernelGGL((wave_per_row<4, false>), dim3(2500), dim3(256), 0, 0, A, Q, nvec, out));
	RUN("wave/row PF=16", hipLaunchKernelGGL((wave_per_row<16, false>), dim3(2500), dim3(256), 0, 0, A, Q, nvec, out));
	RUN("wave/row PF=8 nontemporal", hipLaunchKernelGGL((wave_per_row<8, true>), dim3(2500), dim3(256), 0, 0, A, Q, nvec, out));
	RUN("wg/row PF=4", hipLaunchKernelGGL((wg_per_row<4>), dim3(10000), dim3(256), 0, 0, A, Q, nvec, out));
	RUN("wg/row PF=8", hipLaunchKernelGGL((wg_per_row<8>), dim3(10000), dim3(256), 0, 0, A, Q, nvec, out));
	RUN("2 waves/row PF=8", hipLaunchKernelGGL((two_waves_per_row<8>), dim3(5000), dim3(256), 0, 0, A, Q, nvec, out));
	RUN("flat 2048 wg PF=8", hipLaunchKernelGGL((flat<8>), dim3(2048), dim3(256), 0, 0, A, n, out));
	RUN("flat 4096 wg PF=4", hipLaunchKernelGGL((flat<4>), dim3(4096), dim3(256), 0, 0, A, n, out));
	RUN("flat 1024 wg PF=8", hipLaunchKernelGGL((flat<8>), dim3(1024), dim3(256), 0, 0, A, n, out));
	return 0;
}
