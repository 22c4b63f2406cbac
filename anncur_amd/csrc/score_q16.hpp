// Kp = 512 sweep on v_mfma_f32_16x16x32_bf16 (round 3): the body of score_q1.hpp (one candidate queue per wave, tickets, cross-tile
// pipeline, 32 queries per wave, 80 KB of LDS) with the MFMA shape of score16.hpp -- the chip holds a higher clock on it at equal
// cycles per flop.  A wave's 32 queries are two 16-query sub-tiles (qs), a 32-item tile two 16-item halves (ih); step s = 2 ks + ih of
// a tile (32 steps) reads one A fragment (16 item rows x 32 k) and issues two MFMAs (qs = 0, 1); the previous tile's accumulator
// (2 halves x 2 sub-tiles x 4 registers = 16 elements per lane) is filtered in the shadow, one element every second step.
// C/D layout: col = lane & 15 = query of the sub-tile, row = 4 (lane >> 4) + reg = item of the half.
// Fragment address of step s: row = 16 ih + (lane & 15), chunk = 4 ks + (lane >> 4); the swizzle XORs the chunk's low four bits with
// lane & 15, and chunk = 16 (ks >> 2) + 4 (ks & 3) + (lane >> 4): address = aoff8[s & 7] + (s >> 3) * 256.
#pragma once

template <int KP, int CUR>
__device__ __forceinline__ void staggerq16_tile(const uint32_t (&aoff8)[8], const bf16x8 (&xb)[2][KP / 32], f32x4 (&acc)[2][2], const f32x4 (&accP)[2][2],
												 float tau0, float tau1, uint32_t item0_prev, const WaveQueue &w, uint32_t &fill) {
	using C = FusedQ1Cfg<KP>;
	constexpr int K = KP / 16, AR = 5, DIST = 3, OFF = CUR * C::TILE_BYTES;   // K steps = (k-step of 32, item half) pairs
	static_assert(K == 32, "Kp = 512");
	u32x4 ring[AR];
#define SQ_READ(slot, s) lds_read_frag_at(ring[slot], aoff8[(s) & 7], OFF + ((s) >> 3) * 256)
	SQ_READ(0, 0); SQ_READ(1, 1); SQ_READ(2, 2);
#pragma unroll
	for (int ih = 0; ih < 2; ++ih)
#pragma unroll
		for (int q = 0; q < 2; ++q) acc[ih][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
	for (int g = 0; g < K; ++g) {
		const int nxt = g + DIST;
		if (nxt < K) SQ_READ(nxt % AR, nxt);
#if defined(__HIP_DEVICE_COMPILE__)
		if (g >= 1) asm volatile("" ::"v"(ring[(g - 1) % AR]));
#endif
		// (uniform, cold) every CHECK_PUSHES pushes: the queue must take the next ones
		if (g > 0 && g % (2 * C::CHECK_PUSHES) == 0 && __builtin_expect(fill > w.limit, 0)) wq_drain(w, fill);
		const int after = K - 1 - g;
		lds_wait_frag(ring[g % AR], after < DIST ? after : DIST);
		const bf16x8 a = __builtin_bit_cast(bf16x8, ring[g % AR]);
		const int ks = g >> 1, ih = g & 1;
		acc[ih][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[0][ks], acc[ih][0], 0, 0, 0);
		acc[ih][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[1][ks], acc[ih][1], 0, 0, 0);
		if (g & 1) {  // element e = g >> 1 of the previous tile: item half e >> 3, sub-tile (e >> 2) & 1, register e & 3
			const int e = g >> 1;
			filter16_one(accP[e >> 3][(e >> 2) & 1][e & 3], (uint32_t)(((e >> 3) * 16 + (e & 3)) | ((uint32_t)((e >> 2) & 1) << (WQ_ITEM_BITS + 4))),
						 ((e >> 2) & 1) ? tau1 : tau0, item0_prev, w, fill);
		}
	}
#undef SQ_READ
	// the last MFMAs' results are read by the NEXT tile's filter (a tile later); the caller alternates two accumulator sets (no copy)
}

template <int KP>
__global__ __launch_bounds__(256, 2) void scoreq16_kernel(const FusedParams p) {
	using C = FusedQ1Cfg<KP>;
	constexpr int KS32 = KP / 32, CPR = KP / 8;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int c16 = lane & 15, g4 = lane >> 4;
#ifdef ANNCUR_TIMING_EXPERIMENTS
	unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();   // (diagnostic build: scripts/inkernel_clock.py)
	asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_entry)::"memory");
#endif
	const int wid = xcd_remap(blockIdx.x, p.n_wg);
	const int n_rb = (int)((p.Q + C::BQ - 1) / C::BQ);
	const int split = p.rb_major ? wid % p.S : wid / n_rb, rb = p.rb_major ? wid / p.S : wid - split * n_rb;

	// ---- this lane's two queries: B operand fragments, resident for the whole kernel.  B[k = 8 (lane >> 4) + j][col = lane & 15]
	bf16x8 xb[2][KS32];
	int64_t qv[2];
#pragma unroll
	for (int t = 0; t < 2; ++t) {
		qv[t] = (int64_t)rb * C::BQ + wave * 32 + 16 * t + c16;
		const bool ok = qv[t] < p.Q;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(p.X + (ok ? qv[t] : 0) * p.ldx) + g4;
#pragma unroll
		for (int s = 0; s < KS32; ++s) {
			const u32x4 zero = {0u, 0u, 0u, 0u};
			const u32x4 wv = ok ? src[4 * s] : zero;
			xb[t][s] = __builtin_bit_cast(bf16x8, wv);
		}
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see score_kernel

	const float tau0 = qv[0] < p.Q ? p.tau[qv[0] * p.tau_stride] + p.tau_bias : INFINITY;
	const float tau1 = qv[1] < p.Q ? p.tau[qv[1] * p.tau_stride] + p.tau_bias : INFINITY;
	// candidate path: the wave's queue, its 32 per-query counts (lane l < 32 <-> local query l = 16 * sub-tile + (lane & 15))
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);
	const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(smem));
	const int64_t q_wave0 = (int64_t)rb * C::BQ + wave_u * 32;  // (uniform) the wave's first query
	WaveQueue w;
	w.base = lds_base + (uint32_t)(C::QUEUE_OFF + wave_u * C::QCAP * 8);
	w.limit = w.base + (uint32_t)(C::QCAP - 64 * (C::CHECK_PUSHES + 1)) * 8u;
	w.cnt = lds_base + (uint32_t)(C::CNT_OFF + wave_u * 128);
	w.q_stride8 = (uint32_t)p.nseg * (uint32_t)p.capg * 8u;
	w.seg = p.cand + (q_wave0 * p.nseg + split) * (int64_t)p.capg;
	w.capg = (uint32_t)p.capg; w.n_items = (uint32_t)p.I; w.lane = lane;
	uint32_t fill = w.base;
	if (lane < 32) {
		const int64_t q = q_wave0 + lane;
		lds_store_u32(w.cnt + (uint32_t)lane * 4u, (p.carry && q < p.Q) ? p.seg_cnt[q * p.nseg + split] : 0u);
	}

	// ---- tile schedule: tickets (see score_kernel), or a static contiguous share
	const int j_begin = p.tile_begin + split * p.tiles_per_split, j_end = min(j_begin + p.tiles_per_split, p.tile_end);
	int t_cur = j_begin < j_end ? j_begin : -1, t_cend = j_end, t_next_chunk = -1;
	bool ticket_pending = false;
	const bool dyn = p.chunk_tiles > 0;
	uint32_t ticket_slot = lds_addr(smem + C::TICKET_OFF);
	// XCD-sliced tickets (score_fused.hip 'XCD-sliced tickets'): thread 0's slice state; unsliced = one counter per row block as in round 3
	uint32_t *const ctr_rb = p.chunk_ctr + (p.sliced ? (size_t)rb * N_SLICES : (size_t)rb);
	int slice = p.sliced ? xcc_id() : 0, tried = 0;
	if (dyn) {
		if (tid == 0) {
			uint32_t c0, c1;
			if (p.sliced) {
				c0 = slice_resolve(atomicAdd(ctr_rb + slice, 1u), ctr_rb, p.n_chunks, p.chunks_per_slice, slice, tried);
				c1 = c0 < (uint32_t)p.n_chunks ? slice_resolve(atomicAdd(ctr_rb + slice, 1u), ctr_rb, p.n_chunks, p.chunks_per_slice, slice, tried) : (uint32_t)p.n_chunks;
			} else {
				c0 = atomicAdd(ctr_rb, 2u); c1 = c0 + 1u;
			}
			if (p.chunk_owner) {
				if (c0 < (uint32_t)p.n_chunks) p.chunk_owner[(int64_t)rb * p.n_chunks + c0] = (uint8_t)split;
				if (c1 < (uint32_t)p.n_chunks) p.chunk_owner[(int64_t)rb * p.n_chunks + c1] = (uint8_t)split;
			}
			lds_store_u32(ticket_slot + 8u, c0);   // (the prologue's pair sits in the third and fourth ticket word)
			lds_store_u32(ticket_slot + 12u, c1);
			__builtin_amdgcn_s_waitcnt(0xC07F);
		}
		__syncthreads();
		const uint32_t c = lds_load_u32_uniform(ticket_slot + 8u), c1 = lds_load_u32_uniform(ticket_slot + 12u);
		t_cur = c < (uint32_t)p.n_chunks ? p.tile_begin + (int)c * p.chunk_tiles : -1;
		t_cend = min(t_cur + p.chunk_tiles, p.tile_end);
		t_next_chunk = c1 < (uint32_t)p.n_chunks ? p.tile_begin + (int)c1 * p.chunk_tiles : -1;
	}
	uint32_t dma_off[C::TILE_BYTES / 4096];
	tile_dma_offsets<KP>(dma_off, wave_u, lane);
	if (t_cur >= 0) tile_dma_s<KP>(p.Et, t_cur, lds_base, wave_u, dma_off);
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__syncthreads();

	f32x4 accA[2][2], accB[2][2];
#pragma unroll
	for (int ih = 0; ih < 2; ++ih)
#pragma unroll
		for (int q = 0; q < 2; ++q) { accA[ih][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; accB[ih][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
	float tp0 = INFINITY, tp1 = INFINITY;   // no previous tile yet: the filter never fires
	uint32_t item0_prev = 0;
	uint32_t aoff8[8];   // step s & 7 = 2 (ks & 3) + ih
#pragma unroll
	for (int s = 0; s < 8; ++s) {
		const int row = 16 * (s & 1) + c16;
		aoff8[s] = lds_addr(smem) + (uint32_t)(row * CPR + ((4 * (s >> 1) + g4) ^ c16)) * 16u;
	}
	const uint32_t lane_code = (uint32_t)c16 << WQ_ITEM_BITS;
	__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): staggerq16_tile() counts LDS reads
#define Q16_STEP(CUR, ACC, ACCP)                                                                                                \
	do {                                                                                                                        \
		const int J = t_cur;                                                                                                    \
		if (ticket_pending) {                                                                                                   \
			const uint32_t c = lds_load_u32_uniform(ticket_slot);                                                               \
			t_next_chunk = c < (uint32_t)p.n_chunks ? p.tile_begin + (int)c * p.chunk_tiles : -1;                               \
			ticket_pending = false;                                                                                             \
			ticket_slot ^= 4u;                                                                                                  \
		}                                                                                                                       \
		int nx = J + 1;                                                                                                         \
		bool crossed = false;                                                                                                   \
		if (nx >= t_cend) { nx = t_next_chunk; crossed = dyn && nx >= 0; }                                                      \
		if (nx >= 0) tile_dma_s<KP>(p.Et, nx, lds_base + ((CUR) ^ 1) * C::TILE_BYTES, wave_u, dma_off);                         \
		uint32_t ticket = 0;                                                                                                    \
		if (crossed && tid == 0) ticket_draw(ticket, ctr_rb + slice);                                                           \
		if (fill >= w.base + C::DRAIN_AT * 8u) wq_drain(w, fill);                                                               \
		staggerq16_tile<KP, CUR>(aoff8, xb, ACC, ACCP, tp0, tp1, item0_prev, w, fill);                                          \
		tp0 = tau0; tp1 = tau1; item0_prev = ((uint32_t)J * TILE_I + 4 * g4) | lane_code;                                       \
		ticket_wait(ticket);                                                                                                    \
		if (crossed) {                                                                                                          \
			if (tid == 0) {                                                                                                     \
				if (p.sliced) ticket = slice_resolve(ticket, ctr_rb, p.n_chunks, p.chunks_per_slice, slice, tried);             \
				lds_store_u32(ticket_slot, ticket);                                                                             \
				if (p.chunk_owner && ticket < (uint32_t)p.n_chunks) p.chunk_owner[(int64_t)rb * p.n_chunks + ticket] = (uint8_t)split; \
				__builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
			}                                                                                                                   \
			ticket_pending = true;                                                                                              \
			t_cend = min(nx + p.chunk_tiles, p.tile_end);                                                                       \
		}                                                                                                                       \
		__syncthreads();                                                                                                        \
		t_cur = nx;                                                                                                             \
	} while (0)
#ifdef ANNCUR_TIMING_EXPERIMENTS
	unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
	asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_c0), "+s"(st_r0)::"memory");
#endif
	ANNCUR_PAD_HERE();
	bool last_in_a = false;  // (uniform) which accumulator set holds the last tile
	while (t_cur >= 0) {
		Q16_STEP(0, accA, accB);
		last_in_a = true;
		if (t_cur < 0) break;
		Q16_STEP(1, accB, accA);
		last_in_a = false;
	}
#undef Q16_STEP
#ifdef ANNCUR_TIMING_EXPERIMENTS
	if (tid == 0 && d_sweep_stamps && p.debug_stamp && blockIdx.x < 8192) {
		unsigned long long *stamps = d_sweep_stamps;
		const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
		stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
		stamps[2 * blockIdx.x + 1] = r1 - st_r0;
		stamps[2 * 8192 + 3 * blockIdx.x] = st_entry; stamps[2 * 8192 + 3 * blockIdx.x + 1] = st_r0; stamps[2 * 8192 + 3 * blockIdx.x + 2] = r1;
	}
#endif
	// drain: the last tile's accumulators (16 pushes, the fill checked every CHECK_PUSHES)
	wq_drain(w, fill);
#define Q16_LAST(ACC)                                                                                                           \
	_Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                                            \
		filter16_one(ACC[e >> 3][(e >> 2) & 1][e & 3], (uint32_t)(((e >> 3) * 16 + (e & 3)) | ((uint32_t)((e >> 2) & 1) << (WQ_ITEM_BITS + 4))), \
					 ((e >> 2) & 1) ? tp1 : tp0, item0_prev, w, fill);                                                          \
		if ((e % C::CHECK_PUSHES) == C::CHECK_PUSHES - 1 && fill > w.limit) wq_drain(w, fill);                                  \
	}
	if (last_in_a) { Q16_LAST(accA) } else { Q16_LAST(accB) }
#undef Q16_LAST
	wq_drain(w, fill);
	if (lane < 32) {
		const int64_t q = q_wave0 + lane;
		uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
		asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"(w.cnt + (uint32_t)lane * 4u) : "memory");
#endif
		if (q < p.Q) p.seg_cnt[q * p.nseg + split] = c;
	}
}
