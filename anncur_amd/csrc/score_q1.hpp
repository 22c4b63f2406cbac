// Sweep kernel for Kp = 512 with the wave-level candidate queue and the dynamic tile schedule (round 3): the Kp = 512 body of
// score_kernel (one 32-query sub-tile per wave -- the query operand alone takes 128 VGPRs --, v_mfma_f32_32x32x16_bf16, cross-tile
// software pipeline: while the 32-MFMA chain of tile j runs, the filter of tile j-1's accumulator is issued in its shadow) uses
// exactly 80 KB of LDS per workgroup, 64 KB of tile buffers and 16 KB of per-lane candidate rings: no room for the two ticket words of
// the dynamic schedule (16 bytes more halved the occupancy: -20 %).  With the candidates in ONE queue per wave (score16.hpp:
// WaveQueue, wq_drain, filter16_one) the queues take 14 KB, the per-query counters 0.5 KB, and the tickets fit.
// Lane <-> (query, item) map of the 32x32 MFMA: query = lane & 31, accumulator register e = item row (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
#pragma once

template <int KP>
struct FusedQ1Cfg {
	static constexpr int KSTEPS = KP / 16;
	static constexpr int CPR = KP / 8;
	static constexpr int TILE_BYTES = TILE_I * KP * 2;
	static constexpr int QCAP = 448;                 // entries of a wave's queue
	static constexpr int DRAIN_AT = 128;             // a step drains at its head from this fill on
	static constexpr int CHECK_PUSHES = 4;           // the tile function checks the fill every 4 pushes (64 entries each at most)
	static constexpr int QUEUE_OFF = 2 * TILE_BYTES;
	static constexpr int CNT_OFF = QUEUE_OFF + 4 * QCAP * 8;   // 128 per-query candidate counts of this item split
	static constexpr int TICKET_OFF = CNT_OFF + 128 * 4;
	static constexpr int LDS_BYTES = TICKET_OFF + 16;
	static constexpr int BQ = 128;
	static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

// item row of accumulator register e within the lane's half of a 32-item tile
#define Q1_ROW(e) ((uint32_t)(((e) & 3) + 8 * ((e) >> 2)))

#ifdef ANNCUR_TIMING_EXPERIMENTS   // (round 5) superseded by scoreq16_kernel (score_q16.hpp, which keeps FusedQ1Cfg): the 32x32x16 body is an A/B variant of the experiments library
// One tile: MFMA chain of this tile into `acc`, filter of the previous tile's accumulator `accP` in its shadow (one element every
// second k-step); A fragments through the counted-wait register ring (see stagger1_tile).
template <int KP, int CUR>
__device__ __forceinline__ void stagger1q_tile(const uint32_t (&aoff8)[8], const bf16x8 (&xb)[FusedQ1Cfg<KP>::KSTEPS], f32x16 &acc, const f32x16 &accP,
												float tau, uint32_t item0_prev, const WaveQueue &w, uint32_t &fill) {
	using C = FusedQ1Cfg<KP>;
	constexpr int K = C::KSTEPS, AR = 5, DIST = 3, OFF = CUR * C::TILE_BYTES;
	static_assert(K == 32, "Kp = 512");
	u32x4 ring[AR];
#define S1Q_READ(slot, s) lds_read_frag_at(ring[slot], aoff8[(s) & 7], OFF + ((s) >> 3) * 256)
	S1Q_READ(0, 0); S1Q_READ(1, 1); S1Q_READ(2, 2);
#pragma unroll
	for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
	for (int g = 0; g < K; ++g) {
		const int nxt = g + DIST;
		if (nxt < K) S1Q_READ(nxt % AR, nxt);
#if defined(__HIP_DEVICE_COMPILE__)
		if (g >= 1) asm volatile("" ::"v"(ring[(g - 1) % AR]));
#endif
		// (uniform, cold) every CHECK_PUSHES pushes: the queue must take the next ones
		if (g > 0 && g % (2 * C::CHECK_PUSHES) == 0 && __builtin_expect(fill > w.limit, 0)) wq_drain(w, fill);
		const int after = K - 1 - g;
		lds_wait_frag(ring[g % AR], after < DIST ? after : DIST);
		acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[g % AR]), xb[g], acc, 0, 0, 0);
		if (g & 1) filter16_one(accP[g >> 1], Q1_ROW(g >> 1), tau, item0_prev, w, fill);
	}
#undef S1Q_READ
	mfma_chain_done(acc);
}

template <int KP>
__global__ __launch_bounds__(256, 2) void scoreq1_kernel(const FusedParams p) {
	using C = FusedQ1Cfg<KP>;
	constexpr int KSTEPS = C::KSTEPS, CPR = C::CPR;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int r = lane & 31, h = lane >> 5;
	const int wid = xcd_remap(blockIdx.x, p.n_wg);
	const int n_rb = (int)((p.Q + C::BQ - 1) / C::BQ);
	// work id -> (row block, split): split-major (p.rb_major = 1, row-block-major, is an experiments knob: measured worse, see launch_fused)
	const int split = p.rb_major ? wid % p.S : wid / n_rb, rb = p.rb_major ? wid / p.S : wid - split * n_rb;

	// ---- this lane's query: B operand fragments, resident for the whole kernel (as score_kernel, QT = 1)
	bf16x8 xb[KSTEPS];
	const int64_t qv = (int64_t)rb * C::BQ + wave * 32 + r;
	{
		const bool ok = qv < p.Q;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(p.X + (ok ? qv : 0) * p.ldx) + h;
#pragma unroll
		for (int s = 0; s < KSTEPS; ++s) {
			const u32x4 zero = {0u, 0u, 0u, 0u};
			const u32x4 wv = ok ? src[2 * s] : zero;
			xb[s] = __builtin_bit_cast(bf16x8, wv);
		}
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see score_kernel

	const float tau = qv < p.Q ? p.tau[qv * p.tau_stride] + p.tau_bias : INFINITY;
	// candidate path: the wave's queue, its 32 per-query counts (lane l < 32 <-> local query l)
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

	// ---- tile schedule: tickets of p.chunk_tiles tiles from the row block's counter (see score_kernel), or a static contiguous share
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

	f32x16 accA, accB;
#pragma unroll
	for (int e = 0; e < 16; ++e) { accA[e] = 0.f; accB[e] = 0.f; }
	float tau_prev = INFINITY;   // no previous tile yet: the filter never fires
	uint32_t item0_prev = 0;
	uint32_t aoff8[8];
#pragma unroll
	for (int s = 0; s < 8; ++s) aoff8[s] = lds_addr(smem) + (uint32_t)(r * CPR + ((2 * s + h) ^ (r & 15))) * 16u;
	const uint32_t lane_code = (uint32_t)r << WQ_ITEM_BITS;
	__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): stagger1q_tile() counts LDS reads
#define Q1_STEP(CUR, ACC, ACCP)                                                                                                 \
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
		stagger1q_tile<KP, CUR>(aoff8, xb, ACC, ACCP, tau_prev, item0_prev, w, fill);                                           \
		tau_prev = tau; item0_prev = ((uint32_t)J * TILE_I + 4 * h) | lane_code;                                                \
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
	ANNCUR_PAD_HERE();
	bool last_in_a = false;  // (uniform) which accumulator holds the last tile
	while (t_cur >= 0) {
		Q1_STEP(0, accA, accB);
		last_in_a = true;
		if (t_cur < 0) break;
		Q1_STEP(1, accB, accA);
		last_in_a = false;
	}
#undef Q1_STEP
	// drain: the last tile's accumulator (16 pushes, the fill checked every CHECK_PUSHES)
	wq_drain(w, fill);
#define Q1_LAST(ACC)                                                                                                            \
	_Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                                            \
		filter16_one(ACC[e], Q1_ROW(e), tau_prev, item0_prev, w, fill);                                                         \
		if ((e % C::CHECK_PUSHES) == C::CHECK_PUSHES - 1 && fill > w.limit) wq_drain(w, fill);                                  \
	}
	if (last_in_a) { Q1_LAST(accA) } else { Q1_LAST(accB) }
#undef Q1_LAST
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
#endif  // ANNCUR_TIMING_EXPERIMENTS
#undef Q1_ROW
