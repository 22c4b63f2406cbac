// K-general fused S_hat + threshold filter for wide inner dimensions (Kp > 512: 1000 / 2000 anchor items of the reference's
// default grids, 1024 anchors of BASELINE cfg5, d = 768 bi-encoder embeddings).  Included by score_fused.hip (one translation
// unit: it shares FusedPlan, the threshold / select kernels and the workspace layout with the register-resident kernel).
//
// With Kp > 512 the wave's queries no longer fit in registers, so this is an LDS-tiled GEMM with the filter as its epilogue:
//   workgroup = 8 waves = 256 queries x one item split; block tile = 256 items x 256 queries, K streamed in 64-wide k-tiles
//   through two 64 KiB LDS stages (items 32 KiB + queries 32 KiB, filled by global_load_lds_dwordx4, one barrier per k-tile);
//   wave (wi, wq) owns items [128 wi, +128) x queries [64 wq, +64): 4 x 2 accumulators of v_mfma_f32_32x32x16_bf16 (128 VGPRs).
// Orientation as in score_kernel: D[item][query], the query sits on the LANE, so the threshold compare is lane-local.  After
// the last k-tile of a block tile the 128 accumulator registers are filtered against the lane's threshold; survivors are
// appended straight to the lane's private candidate segment in HBM -- segment (query, lane half h, wave item half wi, item
// split): written by exactly one lane of one workgroup, no atomics.  One compare per Kp/16 MFMAs: the filter is amortised
// over 4-8x more matrix work than in the Kp = 256 kernel.
// Algorithmic work: 2 Q Kp I flops per sweep; bytes through LDS per block tile: (256 + 256) Kp 2.
//
// LDS image of one operand tile (256 rows x 128 B): row-major, the 16-byte chunk index XOR-ed with (row >> 1) & 7, applied on the
// per-lane SOURCE address of the DMA and again by the readers (ds_read_b128 of 32 rows x one chunk column is conflict free).
#pragma once

constexpr int WBM = 256, WBN = 256;             // items per block tile, queries per workgroup
constexpr int W_TILE_BYTES = 256 * 128;         // one operand tile of one 64-wide k-tile
constexpr int W_STAGE_BYTES = 2 * W_TILE_BYTES; // items + queries
constexpr int W_LDS_BYTES = 2 * W_STAGE_BYTES;  // two stages

struct WideParams {
	const uint16_t *X; int64_t ldx;
	const uint16_t *Et; int64_t et_rows;  // rows present in Et (I rounded up to 32): block tiles past it read its last row (discarded)
	int64_t Q, I;
	int Kp;
	int n_rb;
	int S, bt_per_split, bt_begin, bt_end, carry;      // sweep: item splits of the current stage's block-tile range
	int n_st, S0, st_per_split, sample_leading, n_bt_full;  // prepass: sample block tiles and their partition
	float *gmax; int n_groups;
	const float *tau; int tau_stride;
	uint2 *cand; uint32_t *seg_cnt; int capg;
	int n_wg;
};

// work id -> (query block, item split).  Consecutive ids (= one XCD after xcd_remap) cover a compact rectangle of
// ceil(n_rb / 8) query blocks x all splits: the workgroups of one XCD then share both operands' tiles in its L2
// (speed only, never correctness).
__device__ __forceinline__ void wide_map(int wid, int n_rb, int nsplit, int &rb, int &split) {
	const int G = (n_rb + 7) >> 3;
	const int n_full = n_rb / G, rem = n_rb - n_full * G;
	const int full_ids = n_full * G * nsplit;
	if (wid < full_ids) {
		const int g = wid / (G * nsplit), w = wid - g * (G * nsplit);
		split = w / G;
		rb = g * G + (w - split * G);
	} else {
		const int w = wid - full_ids;
		split = w / rem;
		rb = n_full * G + (w - split * rem);
	}
}

// The wave's four 1 KiB pieces (8 rows x 128 B each) of one operand tile: LDS destination = wave-uniform base + lane * 16.
__device__ __forceinline__ void wide_dma(const unsigned char *const (&src)[4], int k_bytes, unsigned char *tile, int wave) {
#pragma unroll
	for (int i = 0; i < 4; ++i)
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + k_bytes),
										 (__attribute__((address_space(3))) void *)(tile + (wave * 4 + i) * 1024), 16, 0, 0);
}

// MODE 0: prepass (group maxima, GROUP = 16 or 4 items).  MODE 1: filter sweep.
template <int MODE, int GROUP>
__global__ __launch_bounds__(512, 2) void wide_kernel(const WideParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int r = lane & 31, h = lane >> 5, wi = wave >> 2, wq = wave & 3;
	int rb, split;
	wide_map(xcd_remap(blockIdx.x, p.n_wg), p.n_rb, MODE == 0 ? p.S0 : p.S, rb, split);
	const int64_t qb = (int64_t)rb * WBN;

	int j_begin, j_end;
	if (MODE == 0) { j_begin = split * p.st_per_split; j_end = min(j_begin + p.st_per_split, p.n_st); }
	else { j_begin = p.bt_begin + split * p.bt_per_split; j_end = min(j_begin + p.bt_per_split, p.bt_end); }
#define bt_of(j) ((MODE == 0 && !p.sample_leading) ? (int)(((int64_t)(j) * p.n_bt_full) / p.n_st) : (j))

	const int nk = p.Kp >> 6;  // 64-wide k-tiles (even: Kp is a multiple of 128)
	const int64_t row_bytes = (int64_t)p.Kp * 2;

	// ---- DMA sources: piece (wave * 4 + i) = tile rows 8 (wave * 4 + i) .. + 8, this lane: row + (lane >> 3), chunk lane & 7
	const unsigned char *asrc[4], *bsrc[4];
	int prow[4], pchunk[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		prow[i] = (wave * 4 + i) * 8 + (lane >> 3);
		pchunk[i] = (lane & 7) ^ ((prow[i] >> 1) & 7);
		int64_t q = qb + prow[i];
		if (q > p.Q - 1) q = p.Q - 1;  // rows past Q re-read the last query (their lanes carry tau = +inf / are never stored)
		bsrc[i] = reinterpret_cast<const unsigned char *>(p.X) + q * p.ldx * 2 + pchunk[i] * 16;
	}
	auto set_asrc = [&](int bt) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			int64_t row = (int64_t)bt * WBM + prow[i];
			if (row > p.et_rows - 1) row = p.et_rows - 1;  // (items >= I are dropped by the filter / never sampled)
			asrc[i] = reinterpret_cast<const unsigned char *>(p.Et) + row * row_bytes + pchunk[i] * 16;
		}
	};

	// ---- this lane's queries
	int64_t qv[2];
	float tau[2];
	uint32_t ncand[2];
	uint2 *seg[2];
	const int nseg = 4 * p.S, sg = (h * 2 + wi) * p.S + split;
#pragma unroll
	for (int t = 0; t < 2; ++t) {
		qv[t] = qb + wq * 64 + 32 * t + r;
		const bool ok = qv[t] < p.Q;
		tau[t] = (MODE == 1 && ok) ? p.tau[qv[t] * p.tau_stride] : INFINITY;
		ncand[t] = (MODE == 1 && p.carry && ok) ? p.seg_cnt[qv[t] * nseg + sg] : 0u;
		seg[t] = (MODE == 1) ? p.cand + ((ok ? qv[t] : 0) * nseg + sg) * (int64_t)p.capg : nullptr;
	}

	// ---- fragment addresses: row * 128 + ((2 s + h) ^ x) * 16, x = (row >> 1) & 7 = (r >> 1) & 7 for every sub-tile
	const int x = (r >> 1) & 7;
	const uint32_t a_base = (uint32_t)(wi * 128 + r) * 128u, b_base = (uint32_t)W_TILE_BYTES + (uint32_t)(wq * 64 + r) * 128u;
	uint32_t coff[4];
#pragma unroll
	for (int s = 0; s < 4; ++s) coff[s] = (uint32_t)(((2 * s + h) ^ x) * 16);

	if (j_begin < j_end) {
		set_asrc(bt_of(j_begin));
		wide_dma(asrc, 0, smem, wave);
		wide_dma(bsrc, 0, smem + W_TILE_BYTES, wave);
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
	__syncthreads();

	for (int j = j_begin; j < j_end; ++j) {
		const int bt = bt_of(j);
		f32x16 acc[4][2];
#pragma unroll
		for (int m = 0; m < 4; ++m)
#pragma unroll
			for (int t = 0; t < 2; ++t)
#pragma unroll
				for (int e = 0; e < 16; ++e) acc[m][t][e] = 0.f;

#define WIDE_COMPUTE(STAGE)                                                                                       \
		do {                                                                                                      \
			const unsigned char *sb = smem + (STAGE) * W_STAGE_BYTES;                                             \
			_Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                       \
				bf16x8 af[4], bfr[2];                                                                             \
				_Pragma("unroll") for (int m = 0; m < 4; ++m)                                                     \
					af[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(sb + a_base + m * 4096 + coff[s])); \
				_Pragma("unroll") for (int t = 0; t < 2; ++t)                                                     \
					bfr[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(sb + b_base + t * 4096 + coff[s])); \
				_Pragma("unroll") for (int m = 0; m < 4; ++m)                                                     \
					_Pragma("unroll") for (int t = 0; t < 2; ++t)                                                 \
						acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m], bfr[t], acc[m][t], 0, 0, 0);   \
			}                                                                                                     \
		} while (0)

		for (int kt = 0; kt < nk; kt += 2) {
			// stage 0 holds k-tile kt: fetch kt + 1 into stage 1 while it is consumed
			wide_dma(asrc, (kt + 1) * 128, smem + W_STAGE_BYTES, wave);
			wide_dma(bsrc, (kt + 1) * 128, smem + W_STAGE_BYTES + W_TILE_BYTES, wave);
			WIDE_COMPUTE(0);
			__builtin_amdgcn_s_waitcnt(0x0F70);
			__syncthreads();
			// stage 1 holds k-tile kt + 1: fetch kt + 2 (or the first k-tile of the next block tile) into stage 0
			const bool last = kt + 2 >= nk;
			if (!last) {
				wide_dma(asrc, (kt + 2) * 128, smem, wave);
				wide_dma(bsrc, (kt + 2) * 128, smem + W_TILE_BYTES, wave);
			} else if (j + 1 < j_end) {
				set_asrc(bt_of(j + 1));
				wide_dma(asrc, 0, smem, wave);
				wide_dma(bsrc, 0, smem + W_TILE_BYTES, wave);
			}
			WIDE_COMPUTE(1);
			if (last) {
				// ---- epilogue of the block tile (the next tile's DMA is in flight).  C/D layout: query = lane & 31,
				// item row = (e & 3) + 8 (e >> 2) + 4 h within the 32-item sub-tile m of the wave's item half wi
				if (MODE == 0) {
					constexpr int GPB = (GROUP == 16) ? 16 : 64;  // groups per block tile
#pragma unroll
					for (int t = 0; t < 2; ++t)
#pragma unroll
						for (int m = 0; m < 4; ++m) {
							const int g = (wi * 4 + m) * 2 + h;
							if (GROUP == 16) {
								float mx = acc[m][t][0];
#pragma unroll
								for (int e = 1; e < 16; ++e) mx = fmaxf(mx, acc[m][t][e]);
								if (qv[t] < p.Q) p.gmax[qv[t] * p.n_groups + (int64_t)j * GPB + g] = mx;
							} else {
								float4 mx;
								mx.x = fmaxf(fmaxf(acc[m][t][0], acc[m][t][1]), fmaxf(acc[m][t][2], acc[m][t][3]));
								mx.y = fmaxf(fmaxf(acc[m][t][4], acc[m][t][5]), fmaxf(acc[m][t][6], acc[m][t][7]));
								mx.z = fmaxf(fmaxf(acc[m][t][8], acc[m][t][9]), fmaxf(acc[m][t][10], acc[m][t][11]));
								mx.w = fmaxf(fmaxf(acc[m][t][12], acc[m][t][13]), fmaxf(acc[m][t][14], acc[m][t][15]));
								if (qv[t] < p.Q) *reinterpret_cast<float4 *>(p.gmax + qv[t] * p.n_groups + ((int64_t)j * GPB + g * 4)) = mx;
							}
						}
				} else {
#pragma unroll
					for (int t = 0; t < 2; ++t)
#pragma unroll
						for (int m = 0; m < 4; ++m) {
							const uint32_t item0 = (uint32_t)bt * WBM + (uint32_t)(wi * 128 + m * 32 + 4 * h);
#pragma unroll
							for (int e = 0; e < 16; ++e) {
								const float v = acc[m][t][e];
								if (__builtin_expect(__ballot(v >= tau[t]) != 0ull, 0)) {
									const uint32_t item = item0 + (uint32_t)((e & 3) + 8 * (e >> 2));
									if (v >= tau[t] && item < (uint32_t)p.I) {
										if (ncand[t] < (uint32_t)p.capg) seg[t][ncand[t]] = make_uint2(__float_as_uint(v), item);
										ncand[t]++;  // (a count above capg marks the overflow: the select kernel repairs this split)
									}
								}
							}
						}
				}
			}
			__builtin_amdgcn_s_waitcnt(0x0F70);
			__syncthreads();
		}
#undef WIDE_COMPUTE
	}
#undef bt_of
	if (MODE == 1) {
#pragma unroll
		for (int t = 0; t < 2; ++t)
			if (qv[t] < p.Q) p.seg_cnt[qv[t] * nseg + sg] = ncand[t];
	}
}
