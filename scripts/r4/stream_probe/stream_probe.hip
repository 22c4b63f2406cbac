// What one CU can stream from HBM, by load path: register loads (global_load_dwordx4 nt, the exact scan's path) against LDS-DMA
// (global_load_lds_dwordx4 into a wave-private ring, ds_read_b128 behind a counted vmcnt), one wave per 200 KB row as in the scan,
// on the whole chip and on a CU-masked stream.  Build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int PF>
__global__ __launch_bounds__(256) void reg_stream(const u32x4 *__restrict__ A, int64_t rows, int64_t vec_per_row, uint32_t *out) {
	extern __shared__ unsigned char smem[];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int64_t q = (int64_t)blockIdx.x * 4 + wave;
	if (q >= rows) return;
	const u32x4 *row = A + q * vec_per_row;
	u32x4 pf[PF];
#pragma unroll
	for (int d = 0; d < PF; ++d) pf[d] = __builtin_nontemporal_load(row + d * 64 + lane);
	u32x4 acc = {0, 0, 0, 0};
	const int64_t nblk = vec_per_row / (64 * PF);
	for (int64_t b = 1; b < nblk; ++b) {
		const u32x4 *blk = row + b * 64 * PF;
#pragma unroll
		for (int d = 0; d < PF; ++d) { acc ^= pf[d]; pf[d] = __builtin_nontemporal_load(blk + d * 64 + lane); }
	}
#pragma unroll
	for (int d = 0; d < PF; ++d) acc ^= pf[d];
	if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[q] = 1;
	(void)smem;
}

// ring of SLOTS x 1 KB per wave; PFD fills in flight
template <int SLOTS, int PFD>
__global__ __launch_bounds__(256) void dma_stream(const u32x4 *__restrict__ A, int64_t rows, int64_t vec_per_row, uint32_t *out, int wave_lds_stride) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int64_t q = (int64_t)blockIdx.x * 4 + wave;
	if (q >= rows) return;
	const unsigned char *row = reinterpret_cast<const unsigned char *>(A + q * vec_per_row);
	const uint32_t ring = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)smem) + (uint32_t)wave * (uint32_t)wave_lds_stride;
	const uint32_t voff = (uint32_t)lane * 16u;
	const uint32_t rd = ring + voff;
	const int64_t nv = vec_per_row / 64;   // 1 KB pieces
	auto fill = [&](int64_t i) {
		const unsigned char *src = row + i * 1024;
		const uint32_t m0v = ring + (uint32_t)(i % SLOTS) * 1024u;
		asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(m0v), "v"(voff), "s"(src) : "memory", "m0");
	};
	for (int i = 0; i < PFD; ++i) fill(i);
	u32x4 acc = {0, 0, 0, 0};
	for (int64_t i = 0; i < nv; ++i) {
		if (i + PFD < nv) { fill(i + PFD); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PFD) : "memory"); }
		else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		u32x4 v;
		asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(rd + (uint32_t)(i % SLOTS) * 1024u) : "memory");
		acc ^= v;
	}
	if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[q] = 1;
}

static hipStream_t masked(int lo, int hi) {
	uint32_t mask[8] = {0};
	for (int b = lo; b < hi; ++b) mask[b / 32] |= 1u << (b % 32);
	hipStream_t s;
	CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
	return s;
}

int main() {
	const int64_t rows = 10000, I = 100000, vec_per_row = I * 2 / 16;   // bf16 rows of 200 000 bytes = 12 500 vectors
	const size_t bytes = (size_t)rows * vec_per_row * 16;
	u32x4 *A; uint32_t *out;
	CK(hipMalloc(&A, bytes)); CK(hipMalloc(&out, rows * 4));
	CK(hipMemset(A, 1, bytes)); CK(hipMemset(out, 0, rows * 4));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	struct Case { const char *name; int cus; hipStream_t st; } cases[] = {{"256 CUs", 256, nullptr}, {"128 CUs", 128, masked(0, 128)}, {"96 CUs", 96, masked(0, 96)}, {"64 CUs", 64, masked(0, 64)}};
	const int grid = (int)((rows + 3) / 4);
	for (auto &c : cases) {
		auto timeit = [&](const char *what, auto launch) {
			for (int i = 0; i < 2; ++i) launch();
			CK(hipStreamSynchronize(c.st));
			CK(hipEventRecord(e0, c.st));
			for (int i = 0; i < 5; ++i) launch();
			CK(hipEventRecord(e1, c.st)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
			printf("%-8s %-44s %.3f ms  %.2f TB/s  %.1f GB/s per CU\n", c.name, what, ms, bytes / ms / 1e9, bytes / ms / 1e6 / c.cus);
			fflush(stdout);
		};
		// register loads: occupancy by dynamic LDS (40 KB -> 4 workgroups = 16 waves per CU, the scan's; 20 KB -> 32 waves)
		timeit("registers, 8 x 1 KB per wave, 16 waves/CU", [&] { hipLaunchKernelGGL((reg_stream<8>), dim3(grid), dim3(256), 40 * 1024, c.st, A, rows, vec_per_row, out); });
		timeit("registers, 16 x 1 KB per wave, 16 waves/CU", [&] { hipLaunchKernelGGL((reg_stream<16>), dim3(grid), dim3(256), 40 * 1024, c.st, A, rows, vec_per_row, out); });
		timeit("registers, 8 x 1 KB per wave, 32 waves/CU", [&] { hipLaunchKernelGGL((reg_stream<8>), dim3(grid), dim3(256), 20 * 1024, c.st, A, rows, vec_per_row, out); });
		timeit("registers, 4 x 1 KB per wave, 8 waves/CU", [&] { hipLaunchKernelGGL((reg_stream<4>), dim3(grid), dim3(256), 80 * 1024, c.st, A, rows, vec_per_row, out); });
		// LDS-DMA: ring of 16 slots, 8 or 12 fills in flight; 16 waves per CU (40 KB per workgroup: 4 x (16 KB ring) = 64 KB -> use stride 10 KB and 8 slots)
		timeit("LDS-DMA nt, 8 slots / 6 in flight, 16 waves/CU", [&] { hipLaunchKernelGGL((dma_stream<8, 6>), dim3(grid), dim3(256), 40 * 1024, c.st, A, rows, vec_per_row, out, 10 * 1024); });
		timeit("LDS-DMA nt, 16 slots / 12 in flight, 8 waves/CU", [&] { hipLaunchKernelGGL((dma_stream<16, 12>), dim3(grid), dim3(256), 80 * 1024, c.st, A, rows, vec_per_row, out, 20 * 1024); });
		timeit("LDS-DMA nt, 16 slots / 14 in flight, 8 waves/CU", [&] { hipLaunchKernelGGL((dma_stream<16, 14>), dim3(grid), dim3(256), 80 * 1024, c.st, A, rows, vec_per_row, out, 20 * 1024); });
		timeit("LDS-DMA nt, 8 slots / 6 in flight, 8 waves/CU", [&] { hipLaunchKernelGGL((dma_stream<8, 6>), dim3(grid), dim3(256), 80 * 1024, c.st, A, rows, vec_per_row, out, 20 * 1024); });
	}
	return 0;
}
