// Sweep kernel on v_mfma_f32_16x16x32_bf16 with a FLAG-SYNCHRONISED TILE RING (round 4; Kp = 128 / 256, k <= 128).
//
// What round 3's body (score16.hpp) left on the table, by its own phase stamps: every 32-item tile ended in a workgroup barrier that cost
// 633 (first stage) / 308 (second stage) of the tile's 4773 / 2929 cycles -- the four waves find different numbers of survivors in a tile
// and the barrier makes each tile as slow as its slowest wave -- and 362 cycles of ticket read + DMA issue (four 1 KB direct-to-LDS pieces
// per wave and tile).  Round 4 first tried to take the drains out of the barrier (all four waves draining in the same tile): no gain, the
// imbalance is the hits themselves.  So the barrier goes:
//   * ONE workgroup of EIGHT waves per CU (512 queries), two waves per SIMD as before.  A tile is staged once per 512 queries instead
//     of once per 256: two DMA pieces per wave and tile instead of four, half the L2 -> LDS traffic.
//   * The item tiles stream through a ring of NS = 4 LDS slots.  No barrier in the tile loop.  Each wave publishes two counters in LDS:
//       landed[w] = number of tiles whose pieces issued by wave w are in LDS (written after the wave's s_waitcnt vmcnt(0) at the end of a step),
//       done[w]   = number of tiles wave w has finished reading.
//     A wave reads tile n when min(landed) > n, and issues its pieces of tile n + D (D = 2) into slot (n + D) % NS -- the slot of tile
//     n + D - NS -- when min(done) > n + D - NS.  A wave's pieces of tile n + D are issued at the head of its step n and known to have landed
//     at the end of that step, so a wave can read up to two tiles ahead of the slowest one and issue one tile ahead of it: the per-tile
//     differences in hit counts average out over the tiles instead of being paid at every tile.
//     (ring rule of the MI355X guide, 'ring-gemm': slots >= fills in flight + 2.)
//   * The tile SEQUENCE is dynamic as before (tickets of CHUNK_TILES tiles from the row block's counter), drawn by wave 0 alone and
//     published to the other waves through a 16-entry ring of tile ids in LDS (seq[n & 15], -1 = end) plus a count of published entries.
//   * One LDS instruction polls everything: lane l < 8 reads landed[l], 8..15 done[l - 8], 16..31 seq[l - 16], 32 the published count;
//     the two minima are three DPP v_min each.  The poll for the NEXT step is issued inside the tile function behind the last fragment
//     read (its latency hides under the tile's last MFMAs); a wave that finds its condition unmet polls again in a sleep loop.
// Exactness is untouched: which tiles a workgroup sweeps never mattered (its survivors go to its own segments, the repair path reads the
// chunk-owner map), and a tile's arithmetic is stagger16_tile() of score16.hpp, instruction for instruction.
// Orderings the protocol relies on (all inside ONE workgroup, i.e. one CU's LDS, which executes a wave's LDS instructions in order):
//   landed: DMA pieces (vmcnt) -> s_waitcnt vmcnt(0) at the end of the step -> ds_write landed.
//   done:   the tile function's last fragment read has returned (its final wait) -> ds_write done.
//   reader: ds_read of the counters -> (values sufficient) -> ds_read of the tile / DMA into the slot.
#pragma once
#ifdef ANNCUR_TIMING_EXPERIMENTS   // (round 5) measured 3-4 % slower than the barrier body: compiled into the experiments library only

template <int KP>
struct Ring16Cfg {
	static constexpr int NS = 4, D = 2, WAVES = 8;
	static constexpr int KS32 = KP / 32, K = KP / 16, CPR = KP / 8;
	static constexpr int TILE_BYTES = TILE_I * KP * 2;
	static constexpr int PIECES = TILE_BYTES / 1024 / WAVES;            // DMA pieces per wave and tile (Kp = 256: 2, 128: 1)
	static constexpr int QCAP = 1024, DRAIN_AT = 192;
	static constexpr int QUEUE_OFF = NS * TILE_BYTES;
	static constexpr int CNT_OFF = QUEUE_OFF + WAVES * QCAP * 8;        // 8 x 64 per-query candidate counts of this item split
	static constexpr int FLAG_OFF = CNT_OFF + WAVES * 64 * 4;           // landed[8], done[8]
	static constexpr int SEQ_OFF = FLAG_OFF + 64;                       // tile id of sequence number n at seq[n & 15]
	static constexpr int PUB_OFF = SEQ_OFF + 64;                        // number of sequence entries published
	static constexpr int LDS_BYTES = PUB_OFF + 16;
	static constexpr int BQ = 64 * WAVES;
	static constexpr int NA = K < 8 ? K : 8;                            // fragment address registers (Kp = 256: steps s and s + 8 are 256 bytes apart)
	static constexpr uint32_t DONE_ALL = 0x7fffffffu;
	static_assert(PIECES >= 1 && (NS - 1) * TILE_BYTES < 65536 && LDS_BYTES <= 160 * 1024, "Kp = 128 or 256: slot offsets are ds_read immediates");
	static_assert(CHUNK_TILES <= 4, "the sequence ring holds 16 entries: two chunks of look-ahead plus the waves' spread");
};

// what one poll sees (wave-uniform scalars)
struct RingView {
	uint32_t min_landed, min_done, pub;
	uint32_t v;   // the polled words, one per lane (see the header): seq entries are read out of it by lane index
};

// issue the poll read (destination in flight until ring_poll_wait)
__device__ __forceinline__ void ring_poll_issue(uint32_t &v, uint32_t addr) {
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
#endif
}
// min over the lanes {0..7} and {8..15} of a row: row_half_mirror (i <-> 7 - i), reverse within the quad, swap neighbours
__device__ __forceinline__ void ring_poll_reduce(RingView &r) {
	uint32_t m = r.v;
	m = min(m, dpp_mov<0x141>(m, m));   // row_half_mirror
	m = min(m, dpp_mov<0x1B>(m, m));    // quad_perm:[3,2,1,0]
	m = min(m, dpp_mov<0xB1>(m, m));    // quad_perm:[1,0,3,2]
	r.min_landed = (uint32_t)__builtin_amdgcn_readlane((int)m, 0);
	r.min_done = (uint32_t)__builtin_amdgcn_readlane((int)m, 8);
	r.pub = (uint32_t)__builtin_amdgcn_readlane((int)r.v, 32);
}
__device__ __forceinline__ void ring_poll_wait(RingView &r) {
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.v)::"memory");
#endif
	ring_poll_reduce(r);
}
__device__ __forceinline__ void ring_poll(RingView &r, uint32_t addr) {
	ring_poll_issue(r.v, addr);
	ring_poll_wait(r);
}
__device__ __forceinline__ int ring_seq(const RingView &r, uint32_t n) {   // tile id of sequence number n (valid while n < r.pub)
	return __builtin_amdgcn_readlane((int)r.v, 16 + (int)(n & 15u));
}

// filter16_one() of score16.hpp with the entry's item word split into a lane constant (item row of the lane's group | query code) and a
// wave-uniform part (first item of the tile + the element's row code): the two tile bases live in SGPRs instead of two VGPRs.
__device__ __forceinline__ void filter16r_one(float v, uint32_t code, float tau, uint32_t item_lane, uint32_t tile_base, const WaveQueue &w, uint32_t &fill) {
	const bool hit = v >= tau;   // (false for the NaN accumulators that stand for "no previous tile")
	const unsigned long long m = __ballot(hit);
	if (__builtin_expect(m != 0ull, 0)) {
		if (hit) {
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			lds_store_2x32(fill + rank * 8u, __float_as_uint(v), item_lane + (tile_base + code));
		}
		fill += 8u * (uint32_t)__builtin_popcountll(m);
	}
}

// stagger16_tile() with the next step's poll read issued behind the last fragment read: its return is counted in the fragment waits of
// the last DIST steps (one more operation in flight) and waited for by the caller.
template <int KP, int CUR>
__device__ __forceinline__ void stagger16r_tile(const uint32_t (&aoff)[Ring16Cfg<KP>::NA], const bf16x8 (&xb)[4][Fused16Cfg<KP>::KS32], f32x4 (&accP)[2][2],
												 const float (&tau)[4], uint32_t item_lane, uint32_t base, uint32_t base_prev, const WaveQueue &w,
												 uint32_t &fill, uint32_t &poll_v, uint32_t poll_addr) {
	using C = Ring16Cfg<KP>;
	constexpr int K = C::K, NA = C::NA, AR = 5, DIST = 3, OFF = CUR * C::TILE_BYTES;
	constexpr int EPS = 16 / K > 0 ? 16 / K : 1;
	static_assert(K <= 16 && 2 * K >= DIST, "staggered path: 4..16 steps per half");
	u32x4 ring[AR];
	// fragment of step s = (k-step s >> 1, item half s & 1): address register (s % K) % NA, + 256 bytes per NA steps (see the kernel)
#define R16_READ(slot, s) lds_read_frag_at(ring[slot], aoff[((s) % K) % NA], OFF + (((s) % K) / NA) * 256)
#pragma unroll
	for (int i = 0; i < DIST; ++i) R16_READ(i, i);
	f32x4 accA[2][2], accB[2][2];
#pragma unroll
	for (int ih = 0; ih < 2; ++ih)
#pragma unroll
		for (int q = 0; q < 2; ++q) { accA[ih][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; accB[ih][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#define F16R_ELEM(ACC, e, QS0, BASE)                                                                                            \
	filter16r_one(ACC[(e) >> 3][((e) >> 2) & 1][(e) & 3],                                                                       \
				  (uint32_t)((((e) >> 3) * 16 + ((e) & 3)) | ((uint32_t)((QS0) + (((e) >> 2) & 1)) << (WQ_ITEM_BITS + 4))),    \
				  tau[(QS0) + (((e) >> 2) & 1)], item_lane, BASE, w, fill)
#pragma unroll
	for (int g = 0; g < 2 * K; ++g) {
		const int nxt = g + DIST;
		if (nxt < 2 * K) R16_READ(nxt % AR, nxt);
		if (nxt == 2 * K) ring_poll_issue(poll_v, poll_addr);   // behind the last fragment read: one more LDS operation in flight from here on
#if defined(__HIP_DEVICE_COMPILE__)
		if (g >= 1) asm volatile("" ::"v"(ring[(g - 1) % AR]));
#endif
		if (g > 0 && (g * EPS) % 8 == 0 && __builtin_expect(fill > w.limit, 0)) wq_drain(w, fill);
		const int after = 2 * K - 1 - g;
		// fragment reads still in flight behind this one: `after` (< DIST) in the tail, plus the poll read once it is issued
		lds_wait_frag(ring[g % AR], after < DIST ? after + 1 : DIST);
		const bf16x8 a = __builtin_bit_cast(bf16x8, ring[g % AR]);
		const int s = g % K, ks = s >> 1, ih = s & 1;
		if (g < K) {
			accA[ih][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[0][ks], accA[ih][0], 0, 0, 0);
			accA[ih][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[1][ks], accA[ih][1], 0, 0, 0);
#pragma unroll
			for (int e = (g == 1 ? 0 : g) * EPS; e < (g == 0 ? 0 : g + 1) * EPS; ++e) F16R_ELEM(accP, e, 2, base_prev);
		} else {
			accB[ih][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[2][ks], accB[ih][0], 0, 0, 0);
			accB[ih][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[3][ks], accB[ih][1], 0, 0, 0);
#pragma unroll
			for (int e = (g - K == 1 ? 0 : g - K) * EPS; e < (g == K ? 0 : g - K + 1) * EPS; ++e) F16R_ELEM(accA, e, 0, base);
		}
	}
#undef F16R_ELEM
#undef R16_READ
#pragma unroll
	for (int ih = 0; ih < 2; ++ih)
#pragma unroll
		for (int q = 0; q < 2; ++q) accP[ih][q] = accB[ih][q];
}

template <int KP>
__global__ __launch_bounds__(512, 2) void score16r_kernel(const FusedParams p) {
	using C = Ring16Cfg<KP>;
	constexpr int K = C::K, KS32 = C::KS32, CPR = C::CPR, NS = C::NS, D = C::D, PIECES = C::PIECES;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int c16 = lane & 15, g4 = lane >> 4;
	const int wid = xcd_remap(blockIdx.x, p.n_wg);
	const int n_rb = (int)((p.Q + C::BQ - 1) / C::BQ);
	const int split = wid / n_rb, rb = wid - split * n_rb;   // split-major work ids (see launch_fused)

	// ---- this lane's four queries: B operand fragments, resident for the whole kernel (as score16_kernel)
	bf16x8 xb[4][KS32];
	int64_t qv[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		qv[t] = (int64_t)rb * C::BQ + wave * 64 + 16 * t + c16;
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

	float tau[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) tau[t] = qv[t] < p.Q ? p.tau[qv[t] * p.tau_stride] + p.tau_bias : INFINITY;
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);
	const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(smem));
	const int64_t q_wave0 = (int64_t)rb * C::BQ + wave_u * 64;
	WaveQueue w;
	w.base = lds_base + (uint32_t)(C::QUEUE_OFF + wave_u * C::QCAP * 8);
	w.limit = w.base + (uint32_t)(C::QCAP - 64 * (8 + 2 * (16 / K > 0 ? 16 / K : 1))) * 8u;
	w.cnt = lds_base + (uint32_t)(C::CNT_OFF + wave_u * 256);
	w.q_stride8 = (uint32_t)p.nseg * (uint32_t)p.capg * 8u;
	w.seg = p.cand + (q_wave0 * p.nseg + split) * (int64_t)p.capg;
	w.capg = (uint32_t)p.capg; w.n_items = (uint32_t)p.I; w.lane = lane;
	uint32_t fill = w.base;
	{
		const int64_t q = q_wave0 + lane;
		lds_store_u32(w.cnt + (uint32_t)lane * 4u, (p.carry && q < p.Q) ? p.seg_cnt[q * p.nseg + split] : 0u);
	}
	const uint32_t flag_base = lds_base + (uint32_t)C::FLAG_OFF;
	const uint32_t my_landed = flag_base + (uint32_t)wave_u * 4u, my_done = flag_base + 32u + (uint32_t)wave_u * 4u;
	const uint32_t poll_addr = flag_base + 4u * (uint32_t)(lane < 32 ? lane : 32);   // (one VGPR; hipcc rematerialises it where that is cheaper)
	const bool leader = wave_u == 0;

	// ---- prologue: counters to zero, the first two tickets, their tiles published (wave 0); one barrier; then no more barriers
	uint32_t pub_n = 0;      // (leader) sequence entries published so far
	bool ended = false;      // (leader) the terminator is published
	if (tid < 16) lds_store_u32(flag_base + (uint32_t)tid * 4u, 0u);
	if (tid == 0) {
		const uint32_t c = atomicAdd(p.chunk_ctr + rb, 2u);   // the first two chunks
		uint32_t np = 0;
		for (uint32_t cc = c; cc < c + 2u; ++cc) {
			if (cc < (uint32_t)p.n_chunks) {
				if (p.chunk_owner) p.chunk_owner[(int64_t)rb * p.n_chunks + cc] = (uint8_t)split;
				for (int i = 0; i < p.chunk_tiles; ++i) {
					const int t = p.tile_begin + (int)cc * p.chunk_tiles + i;
					if (t < p.tile_end) { lds_store_u32(lds_base + (uint32_t)C::SEQ_OFF + (np & 15u) * 4u, (uint32_t)t); ++np; }
				}
			} else {   // the first exhausted ticket of the two publishes the terminator
				lds_store_u32(lds_base + (uint32_t)C::SEQ_OFF + (np & 15u) * 4u, 0xffffffffu); ++np;
				break;
			}
		}
		lds_store_u32(lds_base + (uint32_t)C::PUB_OFF, np);
		__builtin_amdgcn_s_waitcnt(0xC07F);
	}
	__syncthreads();
	RingView rv;
	ring_poll(rv, poll_addr);
	if (leader) {
		pub_n = rv.pub;
		ended = pub_n > 0 && ring_seq(rv, pub_n - 1u) < 0;
	}
	// DMA offsets of this wave's PIECES pieces: piece = wave * PIECES + i covers LDS chunks piece * 64 + lane of the tile image
	uint32_t dma_off[PIECES];
#pragma unroll
	for (int i = 0; i < PIECES; ++i) {
		const int pch = (wave_u * PIECES + i) * 64 + lane;
		const int row = pch / CPR, cs = pch % CPR;
		dma_off[i] = (uint32_t)(row * CPR + swz<CPR>(row, cs)) * 16u;
	}
	auto dma = [&](int tile, uint32_t slot_base) {
		const unsigned char *src = reinterpret_cast<const unsigned char *>(p.Et) + (int64_t)tile * C::TILE_BYTES;  // (uniform)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
		for (int i = 0; i < PIECES; ++i) {
			const uint32_t m0v = slot_base + (uint32_t)(wave_u * PIECES + i) * 1024u;
			asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(dma_off[i]), "s"(src) : "memory", "m0");
		}
#endif
	};
	// tiles of the sequence numbers n, n + 1 (t0, t1); n + 2 is read at the head of step n.  -1 = past the end.
	uint32_t n = 0;
	int t0 = rv.pub > 0u ? ring_seq(rv, 0u) : -1, t1 = -1;
	if (t0 >= 0) {
		// (at least two entries are published whenever the first is a tile: the second chunk's tiles, or the terminator)
		t1 = rv.pub > 1u ? ring_seq(rv, 1u) : -1;
		dma(t0, lds_base);
		if (t1 >= 0) dma(t1, lds_base + C::TILE_BYTES);
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);   // the first two tiles' pieces of this wave
	lds_store_u32(my_landed, 2u);

	// accP = sub-tiles {2,3} of the PREVIOUS tile, filtered in the shadow of this tile's first half.  No previous tile yet: NaN, which no
	// threshold compare passes (round 3 kept a second pair of thresholds at +inf for this: two VGPRs the 256-register ring kernel needs)
	// Stagger: waves w and w + 4 share a SIMD (a workgroup's waves go to the SIMDs in a cyclic order of four).  Started together they run in
	// lockstep -- both in their DMA / poll phase, then both on the matrix pipe -- and nothing overlaps; in the 4-wave kernel the SIMD partner
	// belongs to ANOTHER workgroup and is at a random phase.  Waves 4..7 therefore start p.ring_stagger x 64 cycles late (about half a tile);
	// the ring's slack (two tiles on the landed side, one on the done side) keeps the offset without forcing it.
	if (wave_u >= 4 && p.ring_stagger > 0) {
		for (int i = 0; i < p.ring_stagger; ++i) __builtin_amdgcn_s_sleep(1);
	}
	f32x4 accP[2][2];
#pragma unroll
	for (int ih = 0; ih < 2; ++ih)
#pragma unroll
		for (int q = 0; q < 2; ++q) accP[ih][q] = (f32x4){__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
	uint32_t base_prev = 0;   // (uniform) first item of the previous tile
	// A fragment of step s = (k-step ks = s >> 1, item half s & 1): row 16 (s & 1) + (lane & 15), chunk (4 ks + (lane >> 4)) ^ (row & 15).  The XOR
	// touches the chunk's low four bits only, and 4 ks + g4 = 16 (ks >> 2) + (4 (ks & 3) + g4): steps s and s + 8 are 256 bytes apart -- eight
	// address registers and an immediate serve the sixteen steps of Kp = 256 (the eight registers this saves are the ring kernel's margin)
	static_assert(CPR >= 16, "Kp >= 128: the swizzle XORs four bits");
	uint32_t aoff[C::NA];
#pragma unroll
	for (int s = 0; s < C::NA; ++s) {
		const int row = 16 * (s & 1) + c16;
		aoff[s] = lds_addr(smem) + (uint32_t)(row * CPR + swz<CPR>(row, 4 * (s >> 1) + g4)) * 16u;
	}
	const uint32_t item_lane = (uint32_t)(4 * g4) | ((uint32_t)c16 << WQ_ITEM_BITS);   // item row of the lane's group | query code
	__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the tile function counts LDS reads
	const uint32_t drain_level = w.base + (uint32_t)C::DRAIN_AT * 8u;
#ifdef ANNCUR_TIMING_EXPERIMENTS
	uint32_t ph_acc[5] = {0u, 0u, 0u, 0u, 0u};   // phase stamps (diagnostic build): {sequence + FREE wait + DMA issue + landed, drain, FULL wait, tile section, ticket}
	uint32_t ph_t = (uint32_t)__builtin_amdgcn_s_memtime();
	uint32_t ph_tiles = 0;
#define PHR(i) do { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); ph_acc[i] += now_ - ph_t; ph_t = now_; } while (0)
#else
#define PHR(i) do { } while (0)
#endif
	// Every spin is bounded: a wave that has polled 2^20 times (~0.1 s) for a condition gives up ALL waiting for the rest of the kernel and
	// adds 2^30 to the call's fallback counter (results are then wrong and the caller sees a non-zero, absurd n_fallback) -- a protocol bug
	// must end the launch, never hang the GPU.
	bool dead = false;
#define RING_WAIT(COND)                                                                                                         \
	for (uint32_t spins_ = 0; !dead && !(COND); ) {                                                                             \
		if (++spins_ > (1u << 20)) {                                                                                            \
			dead = true;                                                                                                        \
			if (lane == 0 && p.nfb) atomicAdd(p.nfb, 1u << 30);                                                                 \
			break;                                                                                                              \
		}                                                                                                                       \
		if (p.ring_spin_sleep) __builtin_amdgcn_s_sleep(1);                                                                     \
		ring_poll(rv, poll_addr);                                                                                               \
	}
#define RING_STEP(CUR)                                                                                                          \
	do {                                                                                                                        \
		/* (rv: the poll issued inside the previous tile function, or the prologue's) */                                        \
		/* the tile of sequence number n + D: published by wave 0 at least a chunk ahead; -1 once the sequence has ended */      \
		int t2 = -1;                                                                                                            \
		if (t1 >= 0) {                                                                                                          \
			RING_WAIT(rv.pub > n + (uint32_t)D);                                                                                \
			t2 = ring_seq(rv, n + (uint32_t)D);                                                                                 \
		}                                                                                                                       \
		if (t2 >= 0) {                                                                                                          \
			/* FREE: slot (n + D) % NS held tile n + D - NS; every wave has finished reading it when min(done) > n + D - NS */   \
			RING_WAIT(rv.min_done + (uint32_t)(NS - D) > n);                                                                    \
			dma(t2, lds_base + (uint32_t)(((CUR) + D) % NS) * C::TILE_BYTES);                                                   \
		}                                                                                                                       \
		uint32_t ticket = 0;                                                                                                    \
		const bool draw = leader && !ended && pub_n <= n + (uint32_t)(D + 1 + CHUNK_TILES);                                     \
		if (draw && lane == 0) ticket_draw(ticket, p.chunk_ctr + rb);   /* in flight until ticket_wait() below */               \
		PHR(0);                                                                                                                 \
		if (fill >= drain_level) wq_drain(w, fill);                                                                             \
		PHR(1);                                                                                                                 \
		/* FULL: every wave's pieces of tile n are in LDS */                                                                    \
		RING_WAIT(rv.min_landed > n);                                                                                           \
		PHR(2);                                                                                                                 \
		const uint32_t base = (uint32_t)t0 * TILE_I;                                                                            \
		stagger16r_tile<KP, CUR>(aoff, xb, accP, tau, item_lane, base, base_prev, w, fill, rv.v, poll_addr);                    \
		base_prev = base;                                                                                                       \
		lds_store_u32(my_done, n + 1u);   /* (the tile function's last wait covers every fragment read) */                      \
		PHR(3);                                                                                                                 \
		/* vmcnt(0): this wave's pieces of tile n + D (issued a whole tile ago), its candidate stores and the ticket have landed.  The */ \
		/* wait is unconditional and names the ticket register, as in score16_kernel: the register then stays the ticket's from the draw */ \
		/* to here on every path (with the wait under the leader's branch hipcc handed it to the tile's accumulators meanwhile -- the */ \
		/* static check of in-flight returning atomics, scripts/check_lds_hazards.py, caught that in the first build of this kernel) */ \
		ticket_wait(ticket);                                                                                                    \
		lds_store_u32(my_landed, t2 >= 0 ? n + (uint32_t)(D + 1) : C::DONE_ALL);   /* tiles <= n + D of this wave are in LDS */  \
		if (draw) {   /* (leader) publish the chunk the ticket stands for, or the terminator */                                 \
			const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);                                           \
			if (c < (uint32_t)p.n_chunks) {                                                                                     \
				if (lane == 0 && p.chunk_owner) p.chunk_owner[(int64_t)rb * p.n_chunks + c] = (uint8_t)split;                   \
				for (int i = 0; i < p.chunk_tiles; ++i) {                                                                       \
					const int t = p.tile_begin + (int)c * p.chunk_tiles + i;                                                    \
					if (t < p.tile_end) { lds_store_u32(lds_base + (uint32_t)C::SEQ_OFF + (pub_n & 15u) * 4u, (uint32_t)t); ++pub_n; } \
				}                                                                                                               \
			} else {                                                                                                            \
				lds_store_u32(lds_base + (uint32_t)C::SEQ_OFF + (pub_n & 15u) * 4u, 0xffffffffu); ++pub_n;                      \
				ended = true;                                                                                                   \
			}                                                                                                                   \
			lds_store_u32(lds_base + (uint32_t)C::PUB_OFF, pub_n);   /* after the entries: one wave's LDS stores execute in order */ \
		}                                                                                                                       \
		ring_poll_wait(rv);   /* the poll issued inside the tile function */                                                    \
		PHR(4);                                                                                                                 \
		++n; t0 = t1; t1 = t2;                                                                                                  \
	} while (0)
	ANNCUR_PAD_HERE();
	while (t0 >= 0) {
		RING_STEP(0);
		if (t0 < 0) break;
		RING_STEP(1);
		if (t0 < 0) break;
		RING_STEP(2);
		if (t0 < 0) break;
		RING_STEP(3);
	}
#undef RING_STEP
#undef RING_WAIT
	// this wave is through: nothing of it is awaited any more (its pieces of every tile up to the end are in LDS: the last step's vmcnt(0))
	__builtin_amdgcn_s_waitcnt(0x0F70);
	lds_store_u32(my_landed, C::DONE_ALL);
	lds_store_u32(my_done, C::DONE_ALL);
#ifdef ANNCUR_TIMING_EXPERIMENTS
	if (lane == 0 && d_sweep_stamps && p.debug_stamp && blockIdx.x * 8 + wave < 8192) {
		unsigned long long *ph = d_sweep_stamps + 5 * 8192 + (size_t)(blockIdx.x * 8 + wave) * 8;
		for (int i = 0; i < 5; ++i) ph[i] = ph_acc[i];
		ph[5] = n;
	}
	(void)ph_tiles;
#endif
#undef PHR
	// drain: sub-tiles {2,3} of the last tile (16 pushes: at most the whole queue)
	wq_drain(w, fill);
#define F16R_LAST(e)                                                                                                            \
	filter16r_one(accP[(e) >> 3][((e) >> 2) & 1][(e) & 3],                                                                      \
				  (uint32_t)((((e) >> 3) * 16 + ((e) & 3)) | ((uint32_t)(2 + (((e) >> 2) & 1)) << (WQ_ITEM_BITS + 4))),        \
				  tau[2 + (((e) >> 2) & 1)], item_lane, base_prev, w, fill)
	F16R_LAST(0); F16R_LAST(1); F16R_LAST(2); F16R_LAST(3); F16R_LAST(4); F16R_LAST(5); F16R_LAST(6); F16R_LAST(7);
	F16R_LAST(8); F16R_LAST(9); F16R_LAST(10); F16R_LAST(11); F16R_LAST(12); F16R_LAST(13); F16R_LAST(14); F16R_LAST(15);
#undef F16R_LAST
	wq_drain(w, fill);
	{
		// (the lane id afresh from mbcnt: carried from the prologue it cost three VGPRs across the tile loop, i.e. three spilled registers)
		const uint32_t lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
		const int64_t q = q_wave0 + lane_e;
		uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
		asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"(w.cnt + lane_e * 4u) : "memory");
#endif
		if (q < p.Q) p.seg_cnt[q * p.nseg + split] = c;
	}
}
#endif  // ANNCUR_TIMING_EXPERIMENTS
