// Sweep kernel on v_mfma_f32_16x16x32_bf16 (Kp <= 256).  Same pipeline position, operands, LDS tile image and DMA as
// score_kernel<KP, 1> (included by score_fused.hip); what changes is the MFMA shape and with it the lane <-> (query, item) map:
//   D[16 items][16 queries] per MFMA, C/D layout col = lane & 15 = query, row = 4 (lane >> 4) + reg = item;
//   a wave owns 64 queries as FOUR 16-query sub-tiles (the lane serves one query of each), a 32-item tile is two 16-item halves.
// Why: under MFMA load the chip lowers its clock, and it holds a higher one on the 16x16x32 shape than on 32x32x16 at equal
// cycles per flop (MI355X guide, 'DVFS give-back' (7): 1.12-1.15x flop/s with every operand re-read from LDS) -- fragments read
// per MFMA cycle, accumulator registers and filter compares per flop are the same as in the 32x32 kernel.
//
// Stagger as before, in units of a 16-item half x query pair: steps 0..K-1 run the MFMAs of query sub-tiles {0,1} (two per step,
// sharing the step's A fragment) while the filter of sub-tiles {2,3} of the PREVIOUS tile is issued in their shadow; steps
// K..2K-1 do sub-tiles {2,3} beside the filter of {0,1}.  K = Kp / 16 steps per half, step s = (k-step s >> 1, item half s & 1).
//
// Candidates: a WAVE-level queue instead of per-lane rings.  A lane serves four queries with one item group, so a per-lane ring pools
// four streams and its drain has to sort every entry into one of four segments (22 vector instructions per slot, for the fullest of
// the 64 rings: the drain was 22 % of a first-stage tile in the phase stamps, most lanes idle).  Here the lanes that pass a compare
// append their survivors to ONE queue per wave in LDS (rank among the hitting lanes by mbcnt, wave-uniform fill pointer in an SGPR);
// an entry carries its query in the item word (bits 26..31: query sub-tile and lane & 15; I < 2^26).  The drain is dense -- 64 entries
// per pass, one entry per lane: the entry's query draws its slot from the wave's 64 per-query counters in LDS (ds_add_rtn) and the entry
// goes to the segment of (query, item split): ONE segment per (query, split), S per query.  The queue cannot overflow: the tile
// function checks the fill a few times per tile (every 8 pushes of at most 64 entries) and drains on the spot (cold path) while the
// next pushes might not fit, and a step drains at its head when the queue holds DRAIN_AT entries or more.
//
// Threshold LADDER (round 5).  The prepass bounds a query's k-th best score from below by tau0 = the k-th largest group maximum of an 8 % sample;
// against tau0 about 1.3 % of the leading tiles' scores pass (one 64-lane compare in two takes the hit branch), and rounds 1-4 raised the
// threshold only BETWEEN sweep launches (a wave-per-query refinement kernel, 38 us, for one exact update at 22 % of the tiles).  Now the
// threshold kernel also leaves eight LEVELS per query above tau0 (value-linear between the sample's k-th and a higher order statistic:
// roughly geometric in rank), and the sweep counts what it keeps: the drain adds every candidate at or above a level the wave has not
// reached yet to that level's 16-bit field of the query's four counter words (one no-return global_atomic_add per candidate, device scope).
// Every fourth tile a wave fetches its 64 queries' counter words (one LDS-DMA piece, sc1) and, a tile later, moves each query's threshold up
// to the highest level at or above which k candidates have been counted by ANY workgroup -- k distinct items score >= that level, so it is a
// valid lower bound on the k-th best, whatever the timing (a stale count only means a later move).  One launch sweeps all the tiles: no
// stage boundary, no refinement launch.  cfg2 (model: scripts/r5/survivor_model.py): 442 -> ~340 candidates per query; a field cannot
// overflow (a wave stops counting a level once its own threshold is there; between two refreshes it adds <= 128 per query).
#pragma once

typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int KP, int NW = 4>
struct Fused16Cfg {
	static constexpr int WAVES = NW;                 // waves per workgroup: 4 (256 queries, two workgroups per CU) or 8 (512 queries, one per CU: the tile is staged
	                                                 // once per 512 queries -- half the DMA pieces per wave and half the L2 -> LDS traffic; round 4, see score16_kernel)
	static constexpr int KS32 = KP / 32;             // MFMA k-steps
	static constexpr int K = KP / 16;                // stagger steps per half = (k-step, item half) pairs
	static constexpr int CPR = KP / 8;
	static constexpr int TILE_BYTES = TILE_I * KP * 2;
	static constexpr int QCAP = 1024;                // entries of a wave's queue
	static constexpr int DRAIN_AT = 192;             // a step drains at its head from this fill on (three full passes; 128 / 320 measured: see DESIGN 4.1)
	static constexpr int QUEUE_OFF = 2 * TILE_BYTES;
	static constexpr int CNT_OFF = QUEUE_OFF + NW * QCAP * 8;       // 64 NW per-query candidate counts of this item split
	static constexpr int TICKET_OFF = CNT_OFF + NW * 64 * 4;        // ticket words of the dynamic tile schedule (score_kernel)
	static constexpr int PIECES = TILE_BYTES / 1024 / NW;           // DMA pieces per wave and tile
	static_assert(PIECES >= 1, "a tile holds at least one 1 KB piece per wave");
	// threshold ladder (wave-private rows, local query l of wave w at (w * 64 + l)):
	static constexpr int LVL_OFF = TICKET_OFF + 16;                 // LADDER_LEVELS floats per query (32 bytes), DMA'd once in the prologue
	static constexpr int LCNT_OFF = LVL_OFF + NW * 64 * 4 * LADDER_LEVELS;   // landing zone of the counter words: 16 bytes per query, refetched every LADDER_PERIOD tiles
	static constexpr int JCUR_OFF = LCNT_OFF + NW * 64 * 16;        // the level this wave's threshold of the query has reached (0 = tau0): what the drain still counts
	static constexpr int LDS_BYTES = JCUR_OFF + NW * 64 * 4;
	static_assert(NW != 4 || LDS_BYTES <= 80 * 1024, "two workgroups per CU");
	static_assert(LADDER_LEVELS == 8, "two 128-bit reads per query; eight 16-bit fields in four counter words");
	static constexpr int BQ = 64 * NW;
};
#ifndef ANNCUR_LADDER_PERIOD
#define ANNCUR_LADDER_PERIOD 16
#endif
constexpr int LADDER_PERIOD = ANNCUR_LADDER_PERIOD;   // tiles between two fetches of a wave's counter words (a power of two)
constexpr uint32_t WQ_ITEM_BITS = 26, WQ_ITEM_MASK = (1u << WQ_ITEM_BITS) - 1u;

// wave-uniform state of the candidate path (scalars)
struct WaveQueue {
	uint32_t base, limit;      // LDS byte address of the queue, of the fill from which a step of the tile function drains first
	uint32_t cnt;              // LDS byte address of the wave's 64 per-query counts
	uint2 *seg;                // segment of the wave's first query for this item split
	uint32_t q_stride8;        // BYTES between the segments of consecutive queries (segments per query x capg x 8; x 63 queries < 2^32: nseg <= 255, capg <= 16384)
	uint32_t capg, n_items;
	int lane;
	// threshold ladder (lad_on = 0: none): LDS byte addresses of the wave's level rows and reached-level words, the wave's first query's counter words
	uint32_t lad_on, lvl, jcur;
	uint32_t *lcnt;
};

// Drain: entry i of the queue -> lane i & 63 of pass i >> 6.  Segment address = uniform base + 32-bit byte offset (round 4; round 3 built a
// 64-bit address per entry with two v_mad_u64_u32).  (Round 4 also tried two passes per iteration -- both entries read back to back, both
// counter draws back to back: one LDS round trip less per 128 entries, seven more live registers where the drain is inlined into the tile
// function, which the 256-register ring kernel does not have; no measurable gain in the 4-wave kernel either.)
template <bool LAD = false>
__device__ __forceinline__ void wq_drain(const WaveQueue &w, uint32_t &fill) {
	const uint32_t n = (fill - w.base) >> 3;  // (uniform)
	unsigned char *const segb = reinterpret_cast<unsigned char *>(w.seg);
#pragma nounroll
	for (uint32_t i0 = 0; i0 < n; i0 += 64) {
		const uint32_t i = i0 + (uint32_t)w.lane;
		if (i < n) {
			const uint2 e = lds_load_u64(w.base + i * 8u);
			const uint32_t item = e.y & WQ_ITEM_MASK, ql = e.y >> WQ_ITEM_BITS;
			if (item < w.n_items) {  // (the matrix' last tile may be partial)
				uint32_t pos = 0;
#if defined(__HIP_DEVICE_COMPILE__)
				const uint32_t one = 1u;
				asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pos) : "v"(w.cnt + ql * 4u), "v"(one) : "memory");
#endif
				if (pos < w.capg) *reinterpret_cast<uint2 *>(segb + (ql * w.q_stride8 + pos * 8u)) = make_uint2(e.x, item);
				if (LAD && w.lad_on) {   // (uniform) count the candidate at the highest ladder level it reaches, if the wave is not there yet
					f32x4 la = {0.f, 0.f, 0.f, 0.f}, lb = {0.f, 0.f, 0.f, 0.f};
					uint32_t jc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
					asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b32 %2, %4\n\ts_waitcnt lgkmcnt(0)"
								 : "=&v"(la), "=&v"(lb), "=&v"(jc) : "v"(w.lvl + ql * 32u), "v"(w.jcur + ql * 4u) : "memory");
#endif
					const float v = __uint_as_float(e.x);
					const uint32_t j = (uint32_t)(v >= la[0]) + (uint32_t)(v >= la[1]) + (uint32_t)(v >= la[2]) + (uint32_t)(v >= la[3])
									   + (uint32_t)(v >= lb[0]) + (uint32_t)(v >= lb[1]) + (uint32_t)(v >= lb[2]) + (uint32_t)(v >= lb[3]);   // levels ascend: j = the highest reached
					if (j > jc)
						(void)__hip_atomic_fetch_add(w.lcnt + ql * 4u + ((j - 1u) >> 1), 1u << (16u * ((j - 1u) & 1u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
		}
	}
	fill = w.base;
}

// one accumulator element against the lane's threshold of its sub-tile; code = item row within the tile's lane group | sub-tile << 30
// (a compile-time constant after unrolling); item0c = first item of the lane's group | (lane & 15) << 26
__device__ __forceinline__ void filter16_one(float v, uint32_t code, float tau, uint32_t item0c, const WaveQueue &w, uint32_t &fill) {
	const bool hit = v >= tau;
	const unsigned long long m = __ballot(hit);
	if (__builtin_expect(m != 0ull, 0)) {
		if (hit) {
			const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			lds_store_2x32(fill + rank * 8u, __float_as_uint(v), item0c + code);
		}
		fill += 8u * (uint32_t)__builtin_popcountll(m);
	}
}

__device__ __forceinline__ uint32_t opaque_u32(uint32_t x) {   // the value, hidden from the optimiser (no hoisting, no CSE with earlier uses)
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("" : "+v"(x));
#endif
	return x;
}
__device__ __forceinline__ void ladder_fetch(const uint32_t *lcnt_wave, uint32_t land, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t voff = (uint32_t)lane * 16u;
	asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" ::"s"(land), "v"(voff), "s"(lcnt_wave) : "memory", "m0");
#endif
}

// Ladder refresh (lane l <-> local query l of the wave): the counter words fetched LADDER_PERIOD tiles ago (this wave's own DMA piece, behind its
// own vmcnt(0) at the end of that tile) -> the highest level at or above which k candidates have been counted -> the thresholds of the
// MFMA layout (lane (g4, c16) serves queries 16 t + c16).  tn = -inf where no level qualifies.
__device__ __forceinline__ void ladder_refresh(uint32_t lcnt_lane, uint32_t lvl_lane, uint32_t jcur_lane, uint32_t k, int c16, float (&tau)[4]) {
	u32x4 c = {0u, 0u, 0u, 0u};
	f32x4 la = {0.f, 0.f, 0.f, 0.f}, lb = {0.f, 0.f, 0.f, 0.f};
	uint32_t jc = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %5 offset:16\n\tds_read_b32 %3, %6\n\ts_waitcnt lgkmcnt(0)"
				 : "=&v"(c), "=&v"(la), "=&v"(lb), "=&v"(jc) : "v"(lcnt_lane), "v"(lvl_lane), "v"(jcur_lane) : "memory");
#endif
	const float lv[8] = {la[0], la[1], la[2], la[3], lb[0], lb[1], lb[2], lb[3]};
	uint32_t cum = 0, jn = 0;
	float tn = -INFINITY;
#pragma unroll
	for (int j = 8; j >= 1; --j) {
		cum += (c[(j - 1) >> 1] >> (16 * ((j - 1) & 1))) & 0xffffu;
		const bool take = jn == 0u && cum >= k;
		jn = take ? (uint32_t)j : jn;
		tn = take ? lv[j - 1] : tn;
	}
	if (jn > jc) lds_store_u32(jcur_lane, jn);
#pragma unroll
	for (int t = 0; t < 4; ++t) {
		const float o = __int_as_float(__builtin_amdgcn_ds_bpermute((16 * t + c16) * 4, __float_as_int(tn)));
		tau[t] = fmaxf(tau[t], o);
	}
}

// One 32-item tile.  accP = sub-tiles {2,3} of the previous tile (filtered here, then replaced by this tile's).
template <int KP, int CUR>
__device__ __forceinline__ void stagger16_tile(const uint32_t (&aoff)[Fused16Cfg<KP>::K], const bf16x8 (&xb)[4][Fused16Cfg<KP>::KS32], f32x4 (&accP)[2][2],
												const float (&tau)[4], const float (&tau_prev)[2], uint32_t item0, uint32_t item0_prev, const WaveQueue &w,
												uint32_t &fill) {
	using C = Fused16Cfg<KP>;
	constexpr int K = C::K, AR = 5, DIST = 3, OFF = CUR * C::TILE_BYTES;
	constexpr int EPS = 16 / K > 0 ? 16 / K : 1;  // filter elements per step (Kp = 64: 4, 128: 2, 256: 1)
	static_assert(K <= 16 && 2 * K >= DIST, "staggered path: 4..16 steps per half");
	u32x4 ring[AR];
#pragma unroll
	for (int i = 0; i < DIST; ++i) lds_read_frag<OFF>(ring[i], aoff[i % K]);
	f32x4 accA[2][2], accB[2][2];
#pragma unroll
	for (int ih = 0; ih < 2; ++ih)
#pragma unroll
		for (int q = 0; q < 2; ++q) { accA[ih][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; accB[ih][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
	// element e of a finished half: item half e >> 3, query of the pair (e >> 2) & 1, register e & 3
#define F16_ELEM(ACC, e, QS0, TAU, ITEM0)                                                                                       \
	filter16_one(ACC[(e) >> 3][((e) >> 2) & 1][(e) & 3],                                                                        \
				 (uint32_t)((((e) >> 3) * 16 + ((e) & 3)) | ((uint32_t)((QS0) + (((e) >> 2) & 1)) << (WQ_ITEM_BITS + 4))),     \
				 TAU[((e) >> 2) & 1], ITEM0, w, fill)
#pragma unroll
	for (int g = 0; g < 2 * K; ++g) {
		const int nxt = g + DIST;
		if (nxt < 2 * K) lds_read_frag<OFF>(ring[nxt % AR], aoff[nxt % K]);
#if defined(__HIP_DEVICE_COMPILE__)
		if (g >= 1) asm volatile("" ::"v"(ring[(g - 1) % AR]));
#endif
		// (uniform, cold) every 8 pushes: the queue must take the next 8 + EPS (64 entries each at most) -- checked a few times per tile, not
		// per push (two scalar instructions per step cost the bare loop 10 %; per push the 48 inlined drains ran the kernel out of SGPRs)
		// (this drain does not count for the ladder: an uncounted candidate only delays a move, and the level reads would be live beside the accumulators)
		if (g > 0 && (g * EPS) % 8 == 0 && __builtin_expect(fill > w.limit, 0)) wq_drain<false>(w, fill);
		const int after = 2 * K - 1 - g;
		lds_wait_frag(ring[g % AR], after < DIST ? after : DIST);
		const bf16x8 a = __builtin_bit_cast(bf16x8, ring[g % AR]);
		const int s = g % K, ks = s >> 1, ih = s & 1;
		if (g < K) {
			accA[ih][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[0][ks], accA[ih][0], 0, 0, 0);
			accA[ih][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[1][ks], accA[ih][1], 0, 0, 0);
#pragma unroll
			for (int e = (g == 1 ? 0 : g) * EPS; e < (g == 0 ? 0 : g + 1) * EPS; ++e) F16_ELEM(accP, e, 2, tau_prev, item0_prev);
		} else {
			accB[ih][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[2][ks], accB[ih][0], 0, 0, 0);
			accB[ih][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, xb[3][ks], accB[ih][1], 0, 0, 0);
#pragma unroll
			for (int e = (g - K == 1 ? 0 : g - K) * EPS; e < (g == K ? 0 : g - K + 1) * EPS; ++e) F16_ELEM(accA, e, 0, tau, item0);
		}
	}
#pragma unroll
	for (int ih = 0; ih < 2; ++ih)
#pragma unroll
		for (int q = 0; q < 2; ++q) accP[ih][q] = accB[ih][q];
}

#define S16_SLICED (p.sliced)
template <int KP, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void score16_kernel(const FusedParams p) {
	using C = Fused16Cfg<KP, NW>;
	constexpr int K = C::K, KS32 = C::KS32, CPR = C::CPR;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int c16 = lane & 15, g4 = lane >> 4;
#ifdef ANNCUR_TIMING_EXPERIMENTS
	unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
	asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_entry)::"memory");
#endif
	// static priority (experiment, round 5): the two workgroups of a CU share each SIMD's matrix pipe; at equal priority their waves fall into
	// step -- both in the MFMA section (each at half rate), then both in the tile's ~600 cycles of DMA issue / checks / barrier with the pipe idle.
	// One of the two at a higher priority runs its MFMA section at full rate and does its bookkeeping while the other has the pipe.
#ifdef ANNCUR_TIMING_EXPERIMENTS   // measured NULL (ANNCUR_DEBUG_PRIO = 1 / 2 / 3, by dispatch half or by parity: sweep 0.4511-0.4543 ms against 0.4517-0.4542 without; profiles/r05_static_priority_null.txt)
	if (p.prio_mode) {   // (uniform)
		const int sel = (p.prio_mode & 4) ? (int)(blockIdx.x & 1u) : (int)((blockIdx.x >> 8) & 1u);   // which co-resident workgroup: blocks b and b + 256 share a CU under round-robin dispatch (speed only)
		if (sel) { if ((p.prio_mode & 3) == 1) __builtin_amdgcn_s_setprio(1); else if ((p.prio_mode & 3) == 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
	}
#endif
	const int wid = xcd_remap(blockIdx.x, p.n_wg);
	const int n_rb = (int)((p.Q + C::BQ - 1) / C::BQ);
	// work id -> (row block, split): split-major (p.rb_major = 1, row-block-major, is an experiments knob: measured worse, see launch_fused)
	const int split = p.rb_major ? wid % p.S : wid / n_rb, rb = p.rb_major ? wid / p.S : wid - split * n_rb;

	// ---- this lane's four queries: B operand fragments, resident for the whole kernel.  B[k = 8 (lane >> 4) + j][col = lane & 15]
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
			const u32x4 w = ok ? src[4 * s] : zero;
			xb[t][s] = __builtin_bit_cast(bf16x8, w);
		}
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see score_kernel

	const int j_begin = p.tile_begin + split * p.tiles_per_split, j_end = min(j_begin + p.tiles_per_split, p.tile_end);
	float tau[4];
#pragma unroll
	for (int t = 0; t < 4; ++t) tau[t] = qv[t] < p.Q ? p.tau[qv[t] * p.tau_stride] + p.tau_bias : INFINITY;
	// candidate path: the wave's queue, its 64 per-query counts (lane l <-> local query l = 16 * sub-tile + (lane & 15) of the wave)
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);
	const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(smem));
	const int64_t q_wave0 = (int64_t)rb * C::BQ + wave_u * 64;  // (uniform) the wave's first query
	WaveQueue w;
	w.base = lds_base + (uint32_t)(C::QUEUE_OFF + wave_u * C::QCAP * 8);
	w.limit = w.base + (uint32_t)(C::QCAP - 64 * (8 + 2 * (16 / K > 0 ? 16 / K : 1))) * 8u;  // see stagger16_tile: 8 + 2 EPS pushes between two checks
	w.cnt = lds_base + (uint32_t)(C::CNT_OFF + wave_u * 256);
	w.q_stride8 = (uint32_t)p.nseg * (uint32_t)p.capg * 8u;
	w.seg = p.cand + (q_wave0 * p.nseg + split) * (int64_t)p.capg;
	w.capg = (uint32_t)p.capg; w.n_items = (uint32_t)p.I; w.lane = lane;
	w.lad_on = (uint32_t)p.ladder_on;
	w.lvl = lds_base + (uint32_t)(C::LVL_OFF + wave_u * 64 * 4 * LADDER_LEVELS);
	w.jcur = lds_base + (uint32_t)(C::JCUR_OFF + wave_u * 256);
	w.lcnt = p.ladder_cnt + q_wave0 * 4;   // (the arrays are padded to whole row blocks: rows past Q are never counted, their thresholds are +inf)
	const uint32_t lcnt_land = lds_base + (uint32_t)(C::LCNT_OFF + wave_u * 1024);
	uint32_t fill = w.base;
	{
		const int64_t q = q_wave0 + lane;
		lds_store_u32(w.cnt + (uint32_t)lane * 4u, (p.carry && q < p.Q) ? p.seg_cnt[q * p.nseg + split] : 0u);
		lds_store_u32(w.jcur + (uint32_t)lane * 4u, 0u);
	}
	if (p.ladder_on) {   // (uniform) the wave's 64 level rows: 2 KB = two DMA pieces, waited for with the first tile
#if defined(__HIP_DEVICE_COMPILE__)
		const unsigned char *lsrc = reinterpret_cast<const unsigned char *>(p.ladder + q_wave0 * LADDER_LEVELS);
#pragma unroll
		for (int i = 0; i < 2; ++i) {
			const uint32_t m0v = w.lvl + (uint32_t)i * 1024u, voff = (uint32_t)(i * 1024 + lane * 16);
			asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(lsrc) : "memory", "m0");
		}
#endif
	}
	uint32_t lad_tick = 0;
	bool lad_pending = false;

	// ---- tile schedule (see score_kernel): static contiguous share, or tickets of p.chunk_tiles tiles from the row block's counter
	int t_cur = j_begin < j_end ? j_begin : -1, t_cend = j_end, t_next_chunk = -1;
	bool ticket_pending = false;
	const bool dyn = p.chunk_tiles > 0;
	uint32_t ticket_slot = lds_addr(smem + C::TICKET_OFF);
	// XCD-sliced tickets (score_fused.hip 'XCD-sliced tickets'): thread 0's slice state; unsliced = one counter per row block as in round 3
	uint32_t *const ctr_rb = p.chunk_ctr + (S16_SLICED ? (size_t)rb * N_SLICES : (size_t)rb);
	int slice = S16_SLICED ? xcc_id() : 0, tried = 0;
	if (dyn) {
		if (tid == 0) {
			uint32_t c0, c1;
			if (S16_SLICED) {
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
	// DMA: piece wave * PIECES + i of a tile covers LDS chunks piece * 64 + lane of the tile image (swizzle on the source address: tile_dma_offsets)
	uint32_t dma_off[C::PIECES];
#pragma unroll
	for (int i = 0; i < C::PIECES; ++i) {
		const int pch = (wave_u * C::PIECES + i) * 64 + lane;
		const int row = pch / CPR, cs = pch % CPR;
		dma_off[i] = (uint32_t)(row * CPR + swz<CPR>(row, cs)) * 16u;
	}
	auto tile_dma_w = [&](int tile, uint32_t lds_buf) {
		const unsigned char *src = reinterpret_cast<const unsigned char *>(p.Et) + (int64_t)tile * C::TILE_BYTES;  // (uniform)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
		for (int i = 0; i < C::PIECES; ++i) {
			const uint32_t m0v = lds_buf + (uint32_t)(wave_u * C::PIECES + i) * 1024u;
			asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(dma_off[i]), "s"(src) : "memory", "m0");
		}
#endif
	};
	if (t_cur >= 0) tile_dma_w(t_cur, lds_base);
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__syncthreads();

	f32x4 accP[2][2];
#pragma unroll
	for (int ih = 0; ih < 2; ++ih)
#pragma unroll
		for (int q = 0; q < 2; ++q) accP[ih][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
	float tau_prev[2] = {INFINITY, INFINITY};  // no previous tile yet: the filter of accP never fires
	uint32_t item0_prev = 0;
	// A fragment of step s = (k-step s >> 1, item half s & 1): row 16 (s & 1) + (lane & 15), 16-byte chunk 4 (s >> 1) + (lane >> 4)
	uint32_t aoff[K];
#pragma unroll
	for (int s = 0; s < K; ++s) {
		const int row = 16 * (s & 1) + c16;
		aoff[s] = lds_addr(smem) + (uint32_t)(row * CPR + swz<CPR>(row, 4 * (s >> 1) + g4)) * 16u;
	}
	const uint32_t lane_code = (uint32_t)c16 << WQ_ITEM_BITS;
	__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): stagger16_tile() counts LDS reads
	float tau_hi[2] = {tau[2], tau[3]};
#ifdef ANNCUR_TIMING_EXPERIMENTS
	uint32_t ph_acc[5] = {0u, 0u, 0u, 0u, 0u};   // phase stamps of the diagnostic build: see score_kernel
	uint32_t ph_t = (uint32_t)__builtin_amdgcn_s_memtime();
	uint32_t ph_tiles = 0;
#define PH16(i) do { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); ph_acc[i] += now_ - ph_t; ph_t = now_; } while (0)
#define PH16_TILE() do { ++ph_tiles; } while (0)
#else
#define PH16(i) do { } while (0)
#define PH16_TILE() do { } while (0)
#endif
	// (Round 4 measured a SCHEDULED drain here -- all four waves of the workgroup draining in the same tile, by a uniform countdown planned for
	//  ~160 entries, instead of each wave draining when ITS queue holds DRAIN_AT entries: sweep launches 0.4992 ms with it, 0.4893 without,
	//  0.4888 at a period of 24 tiles; the per-tile barrier wait it was meant to remove is the waves' HIT imbalance, not their drains -- phase
	//  stamps: barrier 546 -> 441 cycles per first-stage tile, drains 206 -> 274 -- and the countdown's own instructions cost 1.3 % of the
	//  sweep even when switched off.  Removed; profiles/r04_ring_and_drain_experiments.txt.)
#define R16_DRAIN_CHECK(J) do { if (fill >= w.base + C::DRAIN_AT * 8u) wq_drain<true>(w, fill); } while (0)
	// ladder, at the head of a tile (uniform branches): consume the counter words fetched a tile ago; every LADDER_PERIOD tiles fetch them again
	// (ONE piece: 64 queries x 16 bytes, sc1 = served by L2 / the fabric, never by this CU's L1; it lands before this tile's closing vmcnt(0))
#define LADDER_STEP()                                                                                                           \
	do {                                                                                                                        \
		if (p.ladder_on) {                                                                                                      \
			const uint32_t ll_ = opaque_u32((uint32_t)lane);   /* (the per-lane addresses below must not become loop-invariant registers of the tile loop) */ \
			if (lad_pending) {                                                                                                  \
				ladder_refresh(lcnt_land + ll_ * 16u, w.lvl + ll_ * 32u, w.jcur + ll_ * 4u, p.ladder_k, (int)(ll_ & 15u), tau);     \
				tau_hi[0] = tau[2]; tau_hi[1] = tau[3];                                                                         \
				tau_prev[0] = fmaxf(tau_prev[0], tau[2]); tau_prev[1] = fmaxf(tau_prev[1], tau[3]);                             \
				lad_pending = false;                                                                                            \
			}                                                                                                                   \
			if ((++lad_tick & p.ladder_mask) == 0u) {                                                                           \
				ladder_fetch(w.lcnt, lcnt_land, (int)ll_);                                                                      \
				lad_pending = true;                                                                                             \
			}                                                                                                                   \
		}                                                                                                                       \
	} while (0)
#define STAGGER16_STEP(CUR)                                                                                                     \
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
		if (nx >= 0) tile_dma_w(nx, lds_base + ((CUR) ^ 1) * C::TILE_BYTES);                                                    \
		uint32_t ticket = 0;                                                                                                    \
		if (crossed && tid == 0) ticket_draw(ticket, ctr_rb + slice);                                                           \
		PH16(0);                                                                                                                \
		LADDER_STEP();                                                                                                          \
		R16_DRAIN_CHECK(J);                                                                                                     \
		const uint32_t item0 = ((uint32_t)J * TILE_I + 4 * g4) | lane_code;                                                     \
		PH16(1);                                                                                                                \
		stagger16_tile<KP, CUR>(aoff, xb, accP, tau, tau_prev, item0, item0_prev, w, fill);                                        \
		tau_prev[0] = tau_hi[0]; tau_prev[1] = tau_hi[1]; item0_prev = item0;                                                   \
		PH16(2);                                                                                                                \
		ticket_wait(ticket);                                                                                                    \
		PH16(3);                                                                                                                \
		if (crossed) {                                                                                                          \
			if (tid == 0) {                                                                                                     \
				if (S16_SLICED) ticket = slice_resolve(ticket, ctr_rb, p.n_chunks, p.chunks_per_slice, slice, tried);             \
				lds_store_u32(ticket_slot, ticket);                                                                             \
				if (p.chunk_owner && ticket < (uint32_t)p.n_chunks) p.chunk_owner[(int64_t)rb * p.n_chunks + ticket] = (uint8_t)split; \
				__builtin_amdgcn_s_waitcnt(0xC07F);                                                                             \
			}                                                                                                                   \
			ticket_pending = true;                                                                                              \
			t_cend = min(nx + p.chunk_tiles, p.tile_end);                                                                       \
		}                                                                                                                       \
		__syncthreads();                                                                                                        \
		PH16(4);                                                                                                                \
		PH16_TILE();                                                                                                            \
		t_cur = nx;                                                                                                             \
	} while (0)
#ifdef ANNCUR_TIMING_EXPERIMENTS
	unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
	asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_c0), "+s"(st_r0)::"memory");
#endif
	ANNCUR_PAD_HERE();
	while (t_cur >= 0) {
		STAGGER16_STEP(0);
		if (t_cur < 0) break;
		STAGGER16_STEP(1);
	}
#undef STAGGER16_STEP
#undef LADDER_STEP
#ifdef ANNCUR_TIMING_EXPERIMENTS
	if (lane == 0 && d_sweep_stamps && p.debug_stamp && blockIdx.x * NW + wave < 8192) {
		unsigned long long *ph = d_sweep_stamps + 5 * 8192 + (size_t)(blockIdx.x * NW + wave) * 8;
		for (int i = 0; i < 5; ++i) ph[i] = ph_acc[i];
		ph[5] = ph_tiles;
	}
	if (tid == 0 && d_sweep_stamps && p.debug_stamp && blockIdx.x < 8192) {
		unsigned long long *stamps = d_sweep_stamps;
		const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
		stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
		stamps[2 * blockIdx.x + 1] = r1 - st_r0;
		stamps[2 * 8192 + 3 * blockIdx.x] = st_entry; stamps[2 * 8192 + 3 * blockIdx.x + 1] = st_r0; stamps[2 * 8192 + 3 * blockIdx.x + 2] = r1;
	}
#endif
#undef PH16
#undef PH16_TILE
	// drain: sub-tiles {2,3} of the last tile (16 pushes: at most the whole queue)
	wq_drain<true>(w, fill);
#define F16_LAST(e)                                                                                                             \
	filter16_one(accP[(e) >> 3][((e) >> 2) & 1][(e) & 3],                                                                       \
				 (uint32_t)((((e) >> 3) * 16 + ((e) & 3)) | ((uint32_t)(2 + (((e) >> 2) & 1)) << (WQ_ITEM_BITS + 4))),         \
				 tau_prev[((e) >> 2) & 1], item0_prev, w, fill)
	F16_LAST(0); F16_LAST(1); F16_LAST(2); F16_LAST(3); F16_LAST(4); F16_LAST(5); F16_LAST(6); F16_LAST(7);
	F16_LAST(8); F16_LAST(9); F16_LAST(10); F16_LAST(11); F16_LAST(12); F16_LAST(13); F16_LAST(14); F16_LAST(15);
#undef F16_LAST
#undef F16_ELEM
	wq_drain<true>(w, fill);
	if (p.ladder_on) {   // the thresholds this wave ended with: the select's prefilter is the highest any workgroup reached (all of them are valid)
		// one last look at the counters, now that this wave's own adds have completed (vmcnt): the workgroup that finishes last sees every count
		// of the launch, and the select takes the maximum over the workgroups
		// (query ids recomputed from an opaque lane id: kept from the prologue they were eight VGPRs live across the tile loop -- spills at Kp = 256)
		const int le = (int)opaque_u32((uint32_t)lane);
		__builtin_amdgcn_s_waitcnt(0x0F70);
		ladder_fetch(w.lcnt, lcnt_land, le);
		__builtin_amdgcn_s_waitcnt(0x0F70);
		ladder_refresh(lcnt_land + (uint32_t)le * 16u, w.lvl + (uint32_t)le * 32u, w.jcur + (uint32_t)le * 4u, p.ladder_k, le & 15, tau);
		if ((le >> 4) == 0) {
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				const int64_t qe = q_wave0 + 16 * t + (le & 15);
				if (qe < p.Q) {
					if (tau[t] >= 0.f) atomicMax(reinterpret_cast<int *>(p.tau_final + qe), __float_as_int(tau[t]));
					else atomicMin(reinterpret_cast<unsigned int *>(p.tau_final + qe), __float_as_uint(tau[t]));
				}
			}
		}
	}
	{
		const int64_t q = q_wave0 + lane;
		uint32_t c = 0;
#if defined(__HIP_DEVICE_COMPILE__)
		asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c) : "v"(w.cnt + (uint32_t)lane * 4u) : "memory");
#endif
		if (q < p.Q) p.seg_cnt[q * p.nseg + split] = c;
	}
}
