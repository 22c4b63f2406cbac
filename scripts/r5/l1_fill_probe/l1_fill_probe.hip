// What one CU can pull through its vector memory path into LDS when the lines come from L2 (not HBM): the operand delivery of the LDS-tiled
// GEMMs (wide_kernel, ivf_tile128_kernel).  Workgroups of 4 waves, 2 per CU; per "k-tile" every wave issues P global_load_lds_dwordx4 pieces
// (1 KiB each: 8 lanes per 128-byte line, 8 lines per piece) and the workgroup meets at a barrier behind s_waitcnt vmcnt(0) -- the GEMMs' loop
// without their MFMAs and LDS reads.  Sources: rows of `pitch` bytes in a region that all workgroups of an XCD share (2 MiB per XCD: L2 hits
// after the first touch) or that every workgroup owns (streamed from HBM / the infinity cache); pieces either row-strided (8 rows x one
// 128-byte line, the GEMMs' pattern) or contiguous (8 consecutive lines).
// Build: hipcc --offload-arch=gfx950 -O3 -o l1_fill_probe l1_fill_probe.hip      Run: ./l1_fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// P pieces per wave and k-tile; STRIDED: piece = 8 rows x 128 B at row pitch `pitch`, else 1 KiB contiguous
template <int P, bool STRIDED, bool TO_LDS>
__global__ __launch_bounds__(256, 2) void fill_kernel(const unsigned char *__restrict__ src, int64_t region_bytes, int shared_per_xcd, int pitch, int ktiles, uint32_t *out) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	// region of this workgroup: shared by the XCD (blocks b, b + 8, ... run on one XCD) or private
	const int64_t reg = shared_per_xcd ? (int64_t)(blockIdx.x & 7) : (int64_t)blockIdx.x;
	const unsigned char *base = src + reg * region_bytes;
	const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)smem);
	const int rows_per_kt = 4 * P * 8;                       // rows one k-tile of the workgroup covers (STRIDED)
	const int64_t kt_bytes = STRIDED ? 128 : (int64_t)4 * P * 1024;
	// a workgroup's start is hashed so that the workgroups of an XCD touch different lines at any moment (L2 hits, no L1 sharing)
	const int64_t start = shared_per_xcd ? ((int64_t)(blockIdx.x >> 3) * 40503 % 64) : 0;
	uint32_t acc = 0;
	typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
	u32x4 r[TO_LDS ? 1 : P];
	for (int kt = 0; kt < ktiles; ++kt) {
#pragma unroll
		for (int i = 0; i < P; ++i) {
			int64_t off;
			if (STRIDED) {
				const int row = (wave * P + i) * 8 + (lane >> 3);
				const int64_t rowsel = ((int64_t)row + (start + kt / (pitch / 128)) * rows_per_kt) % (region_bytes / pitch);
				off = rowsel * pitch + (int64_t)(kt % (pitch / 128)) * 128 + (lane & 7) * 16;
			} else {
				off = (((start * 64 + kt) * kt_bytes) + (int64_t)(wave * P + i) * 1024) % (region_bytes - 4 * P * 1024 + 1024);
				off = (off & ~(int64_t)1023) + lane * 16;
			}
			if (TO_LDS) {
				const uint32_t m0v = lds0 + (uint32_t)((kt & 1) * 4 * P + wave * P + i) * 1024u;
				const uint32_t voff = (uint32_t)off;   // (regions < 4 GiB)
				asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(base) : "memory", "m0");
			} else {
				r[i] = *reinterpret_cast<const u32x4 *>(base + off);
			}
		}
		if (TO_LDS) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		} else {
#pragma unroll
			for (int i = 0; i < P; ++i) acc ^= r[i][0] ^ r[i][3];
		}
		__builtin_amdgcn_s_barrier();
	}
	if (TO_LDS) acc = reinterpret_cast<uint32_t *>(smem)[threadIdx.x];
	if (acc == 0x12345u) out[blockIdx.x] = acc;
}

template <int P, bool STRIDED, bool TO_LDS>
static void run(const char *what, const unsigned char *src, int64_t region_bytes, int shared, int pitch, int n_wg, uint32_t *out, double ghz, int cus) {
	const int ktiles = 2000;
	const size_t lds = TO_LDS ? (size_t)2 * 4 * P * 1024 : 0;
	CK(hipFuncSetAttribute((const void *)fill_kernel<P, STRIDED, TO_LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e30f;
	for (int rep = 0; rep < 4; ++rep) {
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL((fill_kernel<P, STRIDED, TO_LDS>), dim3(n_wg), dim3(256), lds, 0, src, region_bytes, shared, pitch, ktiles, out);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best) best = ms;
	}
	const double bytes = (double)n_wg * ktiles * 4 * P * 1024;
	const double gbs = bytes / (best * 1e-3) / 1e9, per_cu = gbs / cus;
	printf("%-72s %2d pieces/wave  %8.3f ms  %7.2f TB/s  %6.1f GB/s per CU = %5.1f B/clk at %.1f GHz\n", what, P, best, gbs / 1e3, per_cu, per_cu / ghz, ghz);
	fflush(stdout);
}

int main() {
	hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount, n_wg = 2 * cus;
	const double ghz = 2.4;
	const int pitch = 1536;
	const int64_t shared_region = (int64_t)2 << 20, private_region = (int64_t)6 << 20;   // multiples of the pitch
	const int64_t sr = shared_region / pitch * pitch, pr = private_region / pitch * pitch;
	unsigned char *a, *b; uint32_t *out;
	CK(hipMalloc(&a, 8 * sr + 4096)); CK(hipMalloc(&b, (size_t)n_wg * pr + 4096)); CK(hipMalloc(&out, n_wg * 4));
	CK(hipMemset(a, 1, 8 * sr + 4096)); CK(hipMemset(b, 1, (size_t)n_wg * pr + 4096));
	printf("%d CUs, %d workgroups of 4 waves\n", cus, n_wg);
	run<8, true, true>("LDS-DMA, L2-resident (2 MiB per XCD), row-strided pieces", a, sr, 1, pitch, n_wg, out, ghz, cus);
	run<8, false, true>("LDS-DMA, L2-resident (2 MiB per XCD), contiguous pieces", a, sr, 1, pitch, n_wg, out, ghz, cus);
	run<4, true, true>("LDS-DMA, L2-resident (2 MiB per XCD), row-strided pieces", a, sr, 1, pitch, n_wg, out, ghz, cus);
	run<8, true, false>("register loads, L2-resident (2 MiB per XCD), row-strided", a, sr, 1, pitch, n_wg, out, ghz, cus);
	run<8, false, false>("register loads, L2-resident (2 MiB per XCD), contiguous", a, sr, 1, pitch, n_wg, out, ghz, cus);
	run<8, true, true>("LDS-DMA, private 6 MiB per workgroup (beyond L2), row-strided", b, pr, 0, pitch, n_wg, out, ghz, cus);
	run<8, false, true>("LDS-DMA, private 6 MiB per workgroup (beyond L2), contiguous", b, pr, 0, pitch, n_wg, out, ghz, cus);
	run<8, true, true>("LDS-DMA, L2-resident, row-strided, ONE workgroup per CU", a, sr, 1, pitch, n_wg / 2, out, ghz, cus);
	run<16, true, true>("LDS-DMA, L2-resident, row-strided, ONE workgroup per CU (64 KiB per round)", a, sr, 1, pitch, n_wg / 2, out, ghz, cus);
	return 0;
}
