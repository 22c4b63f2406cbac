// anncur_eval_fused: ONE sweep of the exact matrix and of E^T per grid cell of entry point A.
//
// The reference's cell (eval/run_retrieval_eval_wrt_exact_crossenc.py:84-147) needs, per query row, the top-k_retvr of S_hat = X . E
// (:106) AND sum_i (S_hat - A)^2, sum_i A^2 (:146-147).  Rounds 1-3 computed the S_hat GEMM twice for that -- the fused sweep
// (score16_kernel: candidates) and error_lds_kernel (the two sums) -- 1.1 ms of MFMA-bound kernels at cfg2 size.  Here the kernel that
// already streams the exact tile through LDS beside the MFMA chain (error_lds_kernel: lane = query, 32x32x16 MFMAs, the workgroup's
// 256 x 32 tile of A by direct-to-LDS loads) also runs the sweep's threshold filter on the accumulator it is holding and pushes the
// survivors through the wave-level queue of score16.hpp (rank by mbcnt, dense drain through per-query LDS counters, one candidate
// segment per (query, item split)).  Prepass, threshold, refinement between stages and select are the fused top-k's own launches.
//   * item order, not norm order: a tile of E^T must face the same 32 columns of A.
//   * static contiguous tile shares (no tickets): LDS holds two E^T tiles, two A tiles, the queues and the counters -- 79 KB at Kp = 256,
//     two workgroups per CU.  Kp <= 256 (Kp = 512 would be one workgroup per CU: the two-kernel route stays there).
//   * the matrix' last, partial tile (I % 32 items) is swept for candidates only; its columns' error terms are added by the strided
//     kernel of gemm.hip, as in anncur_approx_error_packed.
//   * values: the same MFMAs in the same order as score_kernel<Kp, 1, 16> (the 32x32x16 body, ANNCUR_TOPK_MFMA32) -- bit for bit.
#pragma once

template <int KP>
struct EvalFCfg {
	static constexpr int QCAP = 448, DRAIN_AT = 128, CHECK_PUSHES = 4;
	static constexpr int ATILE = FusedCfg<KP>::BQ * 64;                       // the workgroup's tile of the exact matrix (bf16): BQ rows x 32 items
	static constexpr int AOFF = 2 * FusedCfg<KP>::TILE_BYTES;
	static constexpr int QUEUE_OFF = AOFF + 2 * ATILE;
	static constexpr int CNT_OFF = QUEUE_OFF + 4 * QCAP * 8;
	static constexpr int LDS_BYTES = CNT_OFF + 4 * 64 * 4;
	static_assert(KP > 256 || LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

template <int KP>
__global__ __launch_bounds__(256, 2) void evalf_kernel(const FusedParams p, const uint16_t *__restrict__ Aex, int64_t lda,
														float *__restrict__ err_sq, float *__restrict__ norm_sq) {
	using Cfg = FusedCfg<KP>;
	using C = EvalFCfg<KP>;
	constexpr int KSTEPS = Cfg::KSTEPS, QT = Cfg::QT, CPR = Cfg::CPR;
	constexpr int ATILE = C::ATILE, PA = ATILE / 4096, AOFF = C::AOFF;
	static_assert(QT == 2, "Kp <= 256");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int r = lane & 31, h = lane >> 5;
	const int wid = xcd_remap(blockIdx.x, p.n_wg);
	const int n_rb = (int)((p.Q + Cfg::BQ - 1) / Cfg::BQ);
	const int split = wid / n_rb, rb = wid - split * n_rb;

	bf16x8 xb[QT][KSTEPS];
	int64_t qv[QT];
	uint32_t aread[QT], asw[QT];  // LDS byte address of this lane's row of the exact tile (+ 8 h) in buffer 0; the row's chunk swizzle
	float tau[QT];
#pragma unroll
	for (int t = 0; t < QT; ++t) {
		const int row = wave * 32 * QT + 32 * t + r;
		qv[t] = (int64_t)rb * Cfg::BQ + row;
		const bool ok = qv[t] < p.Q;
		aread[t] = lds_addr(smem) + (uint32_t)(AOFF + row * 64 + 8 * h);
		asw[t] = (uint32_t)((row >> 2) & 3) << 4;
		tau[t] = ok ? p.tau[qv[t] * p.tau_stride] + p.tau_bias : INFINITY;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(p.X + (ok ? qv[t] : 0) * p.ldx) + h;
#pragma unroll
		for (int s = 0; s < KSTEPS; ++s) {
			const u32x4 zero = {0u, 0u, 0u, 0u};
			const u32x4 w = ok ? src[2 * s] : zero;
			xb[t][s] = __builtin_bit_cast(bf16x8, w);
		}
	}
	// exact-tile DMA (as error_lds_kernel): piece i of this wave fills LDS chunks (wave * PA + i) * 64 + lane: row = chunk >> 2, position = chunk & 3
	const unsigned char *abase = reinterpret_cast<const unsigned char *>(Aex + (int64_t)rb * Cfg::BQ * lda);
	uint32_t asrc[PA];
#pragma unroll
	for (int i = 0; i < PA; ++i) {
		const int ch = (wave * PA + i) * 64 + lane, row = ch >> 2, pos = ch & 3;
		const int64_t last = p.Q - 1 - (int64_t)rb * Cfg::BQ;   // rows past Q re-read the last row (their sums are dropped)
		asrc[i] = (uint32_t)(((int64_t)row < last ? (int64_t)row : last) * lda * 2) + (uint32_t)(16 * (pos ^ ((row >> 2) & 3)));
	}
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);
	const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(smem));
	uint32_t dma_off[Cfg::TILE_BYTES / 4096];
	tile_dma_offsets<KP>(dma_off, wave_u, lane);
#ifdef ANNCUR_TIMING_EXPERIMENTS
	// experiments build, ANNCUR_DEBUG_RING_STAGGER (p.ring_stagger; the ring body is not in play here): 1 = every exact-tile DMA re-reads the
	// split's first tile (L2-hot), 2 = no error sums, 3 = neither sums nor filter (DMAs and MFMAs only)
	const int dbg_mode = p.ring_stagger;
	const int dbg_j0 = p.tile_begin + split * p.tiles_per_split;
#define EVALF_SUMS_ON (dbg_mode < 2)
#define EVALF_FILTER_ON (dbg_mode != 3)
#else
#define EVALF_SUMS_ON true
#define EVALF_FILTER_ON true
#endif
	auto adma = [&](int j, int buf) {
#ifdef ANNCUR_TIMING_EXPERIMENTS
		if (dbg_mode == 1) j = dbg_j0;
#endif
		const unsigned char *src = abase + (int64_t)j * (TILE_I * 2);   // (uniform)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
		for (int i = 0; i < PA; ++i) {
			const uint32_t m0v = lds_base + (uint32_t)(AOFF + buf * ATILE + (wave_u * PA + i) * 1024);
			asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(asrc[i]), "s"(src) : "memory", "m0");
		}
#endif
	};
	__builtin_amdgcn_s_waitcnt(0x0F70);  // see score_kernel: keeps vmcnt(0) out of the tile loop

	// candidate path: the wave's queue and its 64 per-query counts (local query 32 t + r)
	const int64_t q_wave0 = (int64_t)rb * Cfg::BQ + wave_u * 64;
	WaveQueue w;
	w.base = lds_base + (uint32_t)(C::QUEUE_OFF + wave_u * C::QCAP * 8);
	w.limit = w.base + (uint32_t)(C::QCAP - 64 * (C::CHECK_PUSHES + 1)) * 8u;
	w.cnt = lds_base + (uint32_t)(C::CNT_OFF + wave_u * 256);
	w.q_stride8 = (uint32_t)p.nseg * (uint32_t)p.capg * 8u;
	w.seg = p.cand + (q_wave0 * p.nseg + split) * (int64_t)p.capg;
	w.capg = (uint32_t)p.capg; w.n_items = (uint32_t)p.I; w.lane = lane;
	uint32_t fill = w.base;
	{
		const int64_t q = q_wave0 + lane;
		lds_store_u32(w.cnt + (uint32_t)lane * 4u, (p.carry && q < p.Q) ? p.seg_cnt[q * p.nseg + split] : 0u);
	}

	const int j_begin = p.tile_begin + split * p.tiles_per_split, j_end = min(j_begin + p.tiles_per_split, p.tile_end);
	float se[QT], sn[QT];
#pragma unroll
	for (int t = 0; t < QT; ++t) { se[t] = 0.f; sn[t] = 0.f; }
	// The exact tiles run up to two tiles ahead in TWO buffers (round 4): a tile's rows are wave-private (a wave's DMA pieces fill the rows its
	// lanes read), so the wave refills a buffer as soon as its own reads of it have returned -- tile J + 2 goes out in the middle of step J,
	// behind the reads of tile J -- and the end of the step waits with a COUNTED vmcnt that leaves those pieces in flight.  With one tile
	// ahead and vmcnt(0) the exact tile's HBM round trip (256 rows x 64 bytes) had to fit inside one step and did not
	// (profiles/r04_error_kernel_modes.txt, scripts/r4/evalf_modes_probe.py: the call 1.06 ms, 0.875 with the exact tile L2-hot).  Loads return
	// in issue order among themselves; the queue's stores may complete in any order, which the count tolerates: while an item-tile load is
	// outstanding so are the PA younger exact-tile loads, i.e. more than PA operations -- vmcnt(PA) cannot pass before the item tile landed.
	const int a_end = min(j_end, p.n_full_tiles);   // exact tiles exist for the full tiles only
#ifdef ANNCUR_V_EVALF1   // (A/B variant build: one tile ahead, issued at the start of the step, vmcnt(0) at its end)
	constexpr bool AHEAD2 = false;
#else
	constexpr bool AHEAD2 = true;
#endif
	if (j_begin < j_end) {
		tile_dma_s<KP>(p.Et, j_begin, lds_base, wave_u, dma_off);
		if (j_begin < a_end) adma(j_begin, 0);
		if (AHEAD2 && j_begin + 1 < a_end) adma(j_begin + 1, 1);
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__syncthreads();
	uint32_t aoff[Cfg::NAOFF];
#pragma unroll
	for (int s = 0; s < Cfg::NAOFF; ++s) aoff[s] = lds_addr(smem) + (uint32_t)(r * CPR + swz<CPR>(r, 2 * s + h)) * 16u;
	const uint32_t lane_code = (uint32_t)r << WQ_ITEM_BITS;   // local query 32 t + r: t rides in the element code (bit 31)
	__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): error_mfma_tile() counts LDS reads
#define EVALF_STEP(CUR, J)                                                                                                      \
	do {                                                                                                                        \
		if ((J) + 1 < j_end) tile_dma_s<KP>(p.Et, (J) + 1, lds_base + ((CUR) ^ 1) * Cfg::TILE_BYTES, wave_u, dma_off);          \
		if (!AHEAD2 && (J) + 1 < a_end) adma((J) + 1, (CUR) ^ 1);                                                               \
		bool counted = false;   /* (uniform) the newest VMEM instructions of this step are the PA pieces of exact tile J + 2, and no store */ \
		if (fill >= w.base + C::DRAIN_AT * 8u) wq_drain(w, fill);                                                               \
		f32x16 acc[QT];                                                                                                         \
		error_mfma_tile<KP, CUR>(aoff, xb, acc);                                                                                \
		const uint32_t item0c = ((uint32_t)(J) * TILE_I + 4 * h) | lane_code;                                                   \
		if ((J) < p.n_full_tiles) {   /* (uniform) the last, partial tile has no exact tile: its error terms come from the strided kernel */ \
			ExactQuad<uint16_t> ex[QT][4];  /* items 32 j + 8 g + 4 h + {0..3}, g = 0..3, of the lane's query */                \
			_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                      \
				_Pragma("unroll") for (int g = 0; g < 4; ++g)                                                                   \
					asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ex[t][g].w) : "v"(aread[t] + (((uint32_t)g << 4) ^ asw[t])), "n"((CUR) * ATILE)); \
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                  \
			_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                      \
				_Pragma("unroll") for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(ex[t][g].w));                              \
			if (AHEAD2 && (J) + 2 < a_end) { adma((J) + 2, (CUR)); counted = true; }   /* this wave's rows of buffer CUR are in registers now */ \
			if (EVALF_SUMS_ON)                                                                                                  \
			_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                      \
				_Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                                \
					const float x = ex[t][e >> 2].get(e & 3);                                                                   \
					const float d = acc[t][e] - x;                                                                              \
					se[t] = fmaf(d, d, se[t]);                                                                                  \
					sn[t] = fmaf(x, x, sn[t]);                                                                                  \
				}                                                                                                               \
			else { _Pragma("unroll") for (int t = 0; t < QT; ++t) { se[t] += acc[t][0]; sn[t] += ex[t][0].get(0); } }          \
		}                                                                                                                       \
		/* the sweep's filter on the accumulator at hand: survivors to the wave's queue (element e of sub-tile t: item row (e & 3) + 8 (e >> 2)) */ \
		if (EVALF_FILTER_ON)                                                                                                    \
		_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                          \
			_Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                                    \
				filter16_one(acc[t][e], (uint32_t)(((e) & 3) + 8 * ((e) >> 2)) | ((uint32_t)t << 31), tau[t], item0c, w, fill); \
				if ((e % C::CHECK_PUSHES) == C::CHECK_PUSHES - 1 && __builtin_expect(fill > w.limit, 0)) wq_drain(w, fill);     \
			}                                                                                                                   \
		/* this wave's parts of the next item tile (and of exact tile J + 1) have landed; the barrier orders LDS only */        \
		if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA) : "memory");                                                  \
		else __builtin_amdgcn_s_waitcnt(0x0F70);                                                                                \
		asm volatile("" ::: "memory");                                                                                          \
		__builtin_amdgcn_s_barrier();                                                                                           \
		asm volatile("" ::: "memory");                                                                                          \
	} while (0)
	ANNCUR_PAD_HERE();
	for (int j = j_begin; j < j_end; j += 2) {
		EVALF_STEP(0, j);
		if (j + 1 < j_end) EVALF_STEP(1, j + 1);
	}
#undef EVALF_STEP
#undef EVALF_SUMS_ON
#undef EVALF_FILTER_ON
	wq_drain(w, fill);
	{
		const int64_t q = q_wave0 + lane;
		uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
		asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"(w.cnt + (uint32_t)lane * 4u) : "memory");
#endif
		if (q < p.Q) p.seg_cnt[q * p.nseg + split] = c;
	}
#pragma unroll
	for (int t = 0; t < QT; ++t)
		if (qv[t] < p.Q) {
			atomicAdd(&err_sq[qv[t]], se[t]);
			atomicAdd(&norm_sq[qv[t]], sn[t]);
		}
}
