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
constexpr int W_LDS_TOTAL = W_LDS_BYTES;

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
	float tau_bias;  // 0 in production (ANNCUR_DEBUG_TAU_BIAS of the timing-experiment build: results become wrong)
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

// The wave's four 1 KiB pieces (8 rows x 128 B each) of one operand tile: LDS destination = wave-uniform base + lane * 16,
// source = wave-uniform base (SGPR pair) + the lane's 32-bit byte offset of its (row, chunk).
__device__ __forceinline__ void wide_dma(const unsigned char *base, const uint32_t (&off)[4], unsigned char *tile, int wave) {
#pragma unroll
	for (int i = 0; i < 4; ++i)
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + off[i]),
										 (__attribute__((address_space(3))) void *)(tile + (wave * 4 + i) * 1024), 16, 0, 0);
}

struct WideFrag { u32x4 a[4], b[2]; };   // raw registers: inline-asm read destinations

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

	// ---- DMA sources: piece (wave * 4 + i) = tile rows 8 (wave * 4 + i) .. + 8, this lane: row + (lane >> 3), chunk lane & 7.
	// Offsets are relative to the tile's first row (uniform base): < 256 rows x 8 KiB.
	const unsigned char *xbase = reinterpret_cast<const unsigned char *>(p.X) + qb * p.ldx * 2;
	const unsigned char *abase;
	uint32_t aoff[4], boff[4];
	const int q_rows = (int)min((int64_t)WBN, p.Q - qb);  // rows past Q re-read the block's last query (their lanes carry tau = +inf)
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int prow = (wave * 4 + i) * 8 + (lane >> 3), pchunk = (lane & 7) ^ ((prow >> 1) & 7);
		boff[i] = (uint32_t)min(prow, q_rows - 1) * (uint32_t)(p.ldx * 2) + (uint32_t)pchunk * 16u;
	}
	auto set_asrc = [&](int bt) {
		abase = reinterpret_cast<const unsigned char *>(p.Et) + (int64_t)bt * WBM * row_bytes;
		const int rows = (int)min((int64_t)WBM, p.et_rows - (int64_t)bt * WBM);  // block tiles past Et's last row re-read it (items >= I are dropped)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int prow = (wave * 4 + i) * 8 + (lane >> 3), pchunk = (lane & 7) ^ ((prow >> 1) & 7);
			aoff[i] = (uint32_t)min(prow, rows - 1) * (uint32_t)row_bytes + (uint32_t)pchunk * 16u;
		}
	};

	// ---- this lane's queries
	int64_t qv[2];
	float tau[2];
	uint32_t ncand[2], segoff[2];  // candidate segment of (query, h, wi, split): byte offset from the workgroup's first segment
	const int nseg = 4 * p.S, sg = (h * 2 + wi) * p.S + split;
	unsigned char *cbase = reinterpret_cast<unsigned char *>(p.cand + qb * nseg * (int64_t)p.capg);  // (256 nseg capg 8 < 4 GiB: plan_wide)
#pragma unroll
	for (int t = 0; t < 2; ++t) {
		qv[t] = qb + wq * 64 + 32 * t + r;
		const bool ok = qv[t] < p.Q;
		tau[t] = (MODE == 1 && ok) ? p.tau[qv[t] * p.tau_stride] + p.tau_bias : INFINITY;
		ncand[t] = (MODE == 1 && p.carry && ok) ? p.seg_cnt[qv[t] * nseg + sg] : 0u;
		segoff[t] = (uint32_t)(((wq * 64 + 32 * t + r) * nseg + sg) * p.capg) * 8u;
	}

	// ---- fragment addresses: row * 128 + ((2 s + h) ^ x) * 16, x = (row >> 1) & 7 = (r >> 1) & 7 for every sub-tile.
	// Round 5: the reads are inline asm with COUNTED waits (as the Kp <= 256 sweep's): hipcc waited lgkmcnt(0) in front of every second MFMA
	// group, i.e. for the fragments it had requested one instruction earlier -- two exposed LDS round trips per k-tile.  One address register
	// per (operand, k-step) of stage 0; stage 1 is 64 KiB on, past the 16-bit offset field, so it gets registers of its own (STAGE is a literal).
	const int x = (r >> 1) & 7;
	const uint32_t lds0 = lds_addr(smem);
	uint32_t fa[2][4], fb[2][4];
#pragma unroll
	for (int s = 0; s < 4; ++s) {
		const uint32_t co = (uint32_t)(((2 * s + h) ^ x) * 16);
		fa[0][s] = lds0 + (uint32_t)(wi * 128 + r) * 128u + co;
		fb[0][s] = lds0 + (uint32_t)W_TILE_BYTES + (uint32_t)(wq * 64 + r) * 128u + co;
		fa[1][s] = fa[0][s] + (uint32_t)W_STAGE_BYTES;
		fb[1][s] = fb[0][s] + (uint32_t)W_STAGE_BYTES;
	}
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);
	const uint32_t lds0_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds0);

	if (j_begin < j_end) {
		set_asrc(bt_of(j_begin));
		wide_dma(abase, aoff, smem, wave);
		wide_dma(xbase, boff, smem + W_TILE_BYTES, wave);
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
	__syncthreads();

	for (int j = j_begin; j < j_end; ++j) {
		const int bt = bt_of(j), bt_next = j + 1 < j_end ? bt_of(j + 1) : bt;
		f32x16 acc[4][2];
#pragma unroll
		for (int m = 0; m < 4; ++m)
#pragma unroll
			for (int t = 0; t < 2; ++t)
#pragma unroll
				for (int e = 0; e < 16; ++e) acc[m][t][e] = 0.f;

		// One 64-wide k-tile = 4 k-steps of 6 fragment reads + 8 MFMAs.  The fragments of step s + 1 are requested before the MFMAs of step s
		// (two register sets) and the MFMAs wait with lgkmcnt(6): the six reads just issued stay in flight.
		// (ablation builds, `make variant V=WIDE_<what>`: results become wrong; scripts/r5/wide_ablation.sh)
#if defined(ANNCUR_V_WIDE_NOREAD) || defined(ANNCUR_V_WIDE_MFMAONLY) || defined(ANNCUR_V_WIDE_NODMA_NOREAD)
#define WIDE_LOAD(F, STAGE, s) do { if (kt == 0 && j == j_begin) { WIDE_LOAD_(F, STAGE, s); } } while (0)
#else
#define WIDE_LOAD(F, STAGE, s) WIDE_LOAD_(F, STAGE, s)
#endif
#define WIDE_LOAD_(F, STAGE, s)                                                                                   \
		do {                                                                                                      \
			_Pragma("unroll") for (int m = 0; m < 4; ++m) lds_read_frag_at(F.a[m], fa[STAGE][s], m * 4096);        \
			_Pragma("unroll") for (int t = 0; t < 2; ++t) lds_read_frag_at(F.b[t], fb[STAGE][s], t * 4096);        \
		} while (0)
#define WIDE_WAIT(F, N)                                                                                           \
		asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(F.a[0]), "+v"(F.a[1]), "+v"(F.a[2]), "+v"(F.a[3]), "+v"(F.b[0]), "+v"(F.b[1]) : "n"(N))
#define WIDE_MFMA(F)                                                                                              \
		do {                                                                                                      \
			_Pragma("unroll") for (int m = 0; m < 4; ++m)                                                         \
				_Pragma("unroll") for (int t = 0; t < 2; ++t)                                                     \
					acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, F.a[m]), __builtin_bit_cast(bf16x8, F.b[t]), acc[m][t], 0, 0, 0); \
		} while (0)
		// The wave's DMA pieces of the next k-tile have landed, its fragment reads are done: raw barrier.
		// (Tried and dropped: warming the XCD's L2 two k-tiles ahead with one 4-byte load per 128-byte line and thread, excluded
		//  from the wait by a counted vmcnt: 907 vs 960 TFLOP/s at 10k x 100k x 1024 -- the loop is not waiting on misses.)
#if defined(ANNCUR_V_WIDE_NOBAR) || defined(ANNCUR_V_WIDE_MFMAONLY)
#define WIDE_SYNC() do { } while (0)
#else
#define WIDE_SYNC()                                                                                               \
		do {                                                                                                      \
			asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                           \
			__builtin_amdgcn_s_barrier();                                                                         \
			asm volatile("" ::: "memory");                                                                        \
		} while (0)
#endif
		// DMA of the next k-tile: two of the wave's eight 1 KiB pieces per k-step, issued between the fragment reads and the MFMAs
		// of the step (eight back-to-back issues at the head of the k-tile kept both waves of a SIMD off the matrix pipe together).
		// Hand-placed (round 5): uniform 64-bit base in SGPRs + the lane's 32-bit offset, M0 from a scalar -- the builtin cost two 64-bit
		// VALU adds and two v_readfirstlane per piece.
#if defined(ANNCUR_V_WIDE_NODMA) || defined(ANNCUR_V_WIDE_MFMAONLY) || defined(ANNCUR_V_WIDE_NODMA_NOREAD)
#define WIDE_NODMA_COND && false
#else
#define WIDE_NODMA_COND
#endif
#define WIDE_PIECE(BASE, OFF, TILE_OFF, i)                                                                        \
		do {                                                                                                      \
			const uint32_t m0v_ = lds0_u + (uint32_t)(TILE_OFF) + (uint32_t)(wave_u * 4 + (i)) * 1024u;            \
			asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v_), "v"((OFF)[i]), "s"(BASE) : "memory", "m0"); \
		} while (0)
#define WIDE_DMA2(s)                                                                                              \
		do {                                                                                                      \
			if (den WIDE_NODMA_COND) {                                                                                            \
				if ((s) < 2) { WIDE_PIECE(da, aoff, dd, 2 * ((s) & 1)); WIDE_PIECE(da, aoff, dd, 2 * ((s) & 1) + 1); } \
				else { WIDE_PIECE(db, boff, dd + W_TILE_BYTES, 2 * ((s) & 1)); WIDE_PIECE(db, boff, dd + W_TILE_BYTES, 2 * ((s) & 1) + 1); } \
			}                                                                                                     \
		} while (0)
#define WIDE_COMPUTE(STAGE)                                                                                       \
		do {                                                                                                      \
			WideFrag f0, f1;                                                                                      \
			WIDE_LOAD(f0, STAGE, 0);                                                                              \
			WIDE_LOAD(f1, STAGE, 1); WIDE_DMA2(0); WIDE_WAIT(f0, 6); WIDE_MFMA(f0); __builtin_amdgcn_sched_barrier(0); \
			WIDE_LOAD(f0, STAGE, 2); WIDE_DMA2(1); WIDE_WAIT(f1, 6); WIDE_MFMA(f1); __builtin_amdgcn_sched_barrier(0); \
			WIDE_LOAD(f1, STAGE, 3); WIDE_DMA2(2); WIDE_WAIT(f0, 6); WIDE_MFMA(f0); __builtin_amdgcn_sched_barrier(0); \
			WIDE_DMA2(3); WIDE_WAIT(f1, 0); WIDE_MFMA(f1);                                                         \
		} while (0)

		ANNCUR_PAD_HERE();
		for (int kt = 0; kt < nk; kt += 2) {
			// stage 0 holds k-tile kt: fetch kt + 1 into stage 1 while it is consumed
			const unsigned char *da = abase + (kt + 1) * 128, *db = xbase + (kt + 1) * 128;
			uint32_t dd = (uint32_t)W_STAGE_BYTES;   // LDS byte offset of the stage being filled
			bool den = true;
			WIDE_COMPUTE(0);
			WIDE_SYNC();
			// stage 1 holds k-tile kt + 1: fetch kt + 2 (or the first k-tile of the next block tile) into stage 0
			const bool last = kt + 2 >= nk;
			dd = 0u;
			if (!last) {
				da = abase + (kt + 2) * 128; db = xbase + (kt + 2) * 128;
			} else if (j + 1 < j_end) {
				set_asrc(bt_next);
				da = abase; db = xbase;
			} else {
				den = false;
			}
			WIDE_COMPUTE(1);
#if defined(ANNCUR_V_WIDE_NOFILTER) || defined(ANNCUR_V_WIDE_MFMAONLY)
			if (last) { asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]), "v"(acc[0][1]), "v"(acc[1][1]), "v"(acc[2][1]), "v"(acc[3][1])); }
			if (false) {
#else
			if (last) {
#endif
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
					const uint32_t item_wave = (uint32_t)bt * WBM + (uint32_t)(wi * 128 + 4 * h);
#pragma unroll
					for (int t = 0; t < 2; ++t)
#pragma unroll
						for (int m = 0; m < 4; ++m) {
#pragma unroll
							for (int e = 0; e < 16; ++e) {
								const float v = acc[m][t][e];
								if (__builtin_expect(__ballot(v >= tau[t]) != 0ull, 0)) {
									// (the item id is made opaque HERE: hipcc otherwise precomputes all 128 ids and their
									//  "item < I" masks once per block tile and spills them)
									uint32_t item = item_wave;
#if defined(__HIP_DEVICE_COMPILE__)
									asm volatile("" : "+v"(item));
#endif
									item += (uint32_t)(m * 32 + (e & 3) + 8 * (e >> 2));
									if (v >= tau[t] && item < (uint32_t)p.I) {
										if (ncand[t] < (uint32_t)p.capg)
											*reinterpret_cast<uint2 *>(cbase + (segoff[t] + ncand[t] * 8u)) = make_uint2(__float_as_uint(v), item);
										ncand[t]++;  // (a count above capg marks the overflow: the select kernel repairs this split)
									}
								}
							}
						}
				}
			}
			WIDE_SYNC();
		}
#undef WIDE_COMPUTE
#undef WIDE_MFMA
#undef WIDE_LOAD
#undef WIDE_SYNC
#undef WIDE_PIECE
#undef WIDE_DMA2
#undef WIDE_WAIT
#undef WIDE_NODMA_COND
#undef WIDE_LOAD_
	}
#undef bt_of
	if (MODE == 1) {
#pragma unroll
		for (int t = 0; t < 2; ++t)
			if (qv[t] < p.Q) p.seg_cnt[qv[t] * nseg + sg] = ncand[t];
	}
}

// (Round 2-4: a ping-pong variant of this kernel -- two wave groups one phase apart over a 4-stage ring of 32-wide half k-tiles, counted
//  vmcnt, the 8-phase 256^2 template of the MI355X guide -- was built, parity-exact, and measured level with wide_kernel twice: 929-934 vs
//  944-947 TFLOP/s in round 2, 989-991 vs 969-990 in round 4 (profiles/r04_ab_wide_vs_pingpong.txt, one process, one device).  The
//  pipeline structure is not what limits this kernel -- profiles/r04_pmc_summary_wide.json: matrix pipe busy 0.48 at 2.08 GHz, 43 % of the
//  wave cycles in issue stalls (SQ_WAIT_INST_ANY: MFMA dependency / pipe), 37 % parked (SQ_WAIT_ANY), no LDS bank conflicts, L2 hit rate
//  0.78 with 1.6 GB per launch from beyond L2 -- and the variant was deleted in round 4 instead of shipping as dead code.)
