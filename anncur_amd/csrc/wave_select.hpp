// Wave-level exact top-k selection (k <= 128): ONE wave owns one row / one query.  No workgroup barrier, no atomics
// that return: the wave's candidate buffer, digit histogram and sort scratch live in a wave-private LDS region, the
// candidate count in a (wave-uniform) register, counting is ballot + popcount.  Same-wave LDS accesses execute in
// program order, so the only synchronisation needed is keeping the compiler from reordering them (wave_barrier).
//
// Candidate = 64-bit composite key split in two 32-bit halves: hi = order-preserving map of the fp32 score,
// lo = 0xffffffff - index  (larger key = better score, ties -> smaller index).  Keys of one row are distinct.
#pragma once
#include "common.hpp"

namespace anncur {

constexpr int WSEL_K = 128;  // largest k of the wave-level path

template <int CAP>
struct WaveSelLayout {
	static constexpr int BYTES = CAP * 8 + 128 * 8 + 256 * 4;
};

struct WaveSel {
	uint32_t *whi, *wlo;  // [CAP]
	uint2 *sort_buf;      // [128]
	uint32_t *hist;       // [256]
	uint32_t cnt;         // wave-uniform
	uint32_t tau_hi, tau_lo;  // running threshold key (k-th best so far); 0 = none yet
	float tau;                // its score
};

template <int CAP>
__device__ __forceinline__ WaveSel wsel_init(unsigned char *wave_lds) {
	WaveSel w;
	w.whi = reinterpret_cast<uint32_t *>(wave_lds);
	w.wlo = w.whi + CAP;
	w.sort_buf = reinterpret_cast<uint2 *>(w.wlo + CAP);
	w.hist = reinterpret_cast<uint32_t *>(w.sort_buf + 128);
	w.cnt = 0; w.tau_hi = 0; w.tau_lo = 0; w.tau = -INFINITY;
	return w;
}

// Append the lanes of `m` (= __ballot(hit), non-zero; wave-uniform call; caller guarantees room).  v_mbcnt gives the lane's rank
// among the set bits in two instructions.
__device__ __forceinline__ void wsel_push_mask(WaveSel &w, unsigned long long m, bool hit, uint32_t hi, uint32_t lo) {
	if (hit) {
		const uint32_t pos = w.cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
		w.whi[pos] = hi;
		w.wlo[pos] = lo;
	}
	w.cnt += (uint32_t)__popcll(m);
}
// Append the lanes with `hit` (wave-uniform call; caller guarantees room).
__device__ __forceinline__ void wsel_push(WaveSel &w, bool hit, uint32_t hi, uint32_t lo) {
	const unsigned long long m = __ballot(hit);
	if (m == 0ull) return;
	wsel_push_mask(w, m, hit, hi, lo);
}

// Offer one element per lane: cheap float filter first, exact composite-key comparison only when some lane passes.
__device__ __forceinline__ void wsel_offer(WaveSel &w, bool valid, float v, uint32_t idx) {
	const bool maybe = valid && (v >= w.tau);
	if (__ballot(maybe) == 0ull) return;
	const uint32_t hi = f32_sortable(v), lo = 0xffffffffu - idx;
	wsel_push(w, maybe && (hi > w.tau_hi || (hi == w.tau_hi && lo > w.tau_lo)), hi, lo);
}

// In-order stream variant: every candidate already in the buffer has a smaller index than this element, so a score tie
// with the threshold loses and one strict float compare is exact.
__device__ __forceinline__ void wsel_offer_inorder(WaveSel &w, bool valid, float v, uint32_t idx) {
	const bool hit = valid && v == v && (v > w.tau || !(w.tau > -INFINITY));  // no threshold yet: -inf is a candidate too; NaN never
	if (__ballot(hit) == 0ull) return;
	wsel_push(w, hit, f32_sortable(v), 0xffffffffu - idx);
}

// The bin of a 256-bin digit histogram that holds the need-th LARGEST key; `above` = keys in higher bins, `in_bin` = keys in it.
// Lane L owns the four bins 252 - 4 L .. 255 - 4 L, so that an inclusive prefix sum over the lanes (DPP, no LDS round trip) counts
// the keys from the top bin down.
__device__ __forceinline__ void wsel_find_bin(const uint32_t *hist, uint32_t lane, uint32_t need, uint32_t &bin_out, uint32_t &above, uint32_t &in_bin) {
	const uint32_t b0 = (WAVE - 1 - lane) * 4;
	const uint32_t h0 = hist[b0], h1 = hist[b0 + 1], h2 = hist[b0 + 2], h3 = hist[b0 + 3];
	const uint32_t c4 = h0 + h1 + h2 + h3;
	const uint32_t pre = wave_scan_incl<DppAdd>(c4);  // keys in this lane's bins and above
	uint32_t a = pre - c4, bin = 0, hb = 0;
	const bool mine = a < need && pre >= need;
	if (mine) {
		if (a + h3 >= need) { bin = b0 + 3; hb = h3; }
		else { a += h3;
			if (a + h2 >= need) { bin = b0 + 2; hb = h2; }
			else { a += h2;
				if (a + h1 >= need) { bin = b0 + 1; hb = h1; }
				else { a += h1; bin = b0; hb = h0; } } }
	}
	const int src = __ffsll((long long)__ballot(mine)) - 1;  // exactly one lane (wave-uniform index: v_readlane, not a shuffle)
	bin_out = (uint32_t)__builtin_amdgcn_readlane((int)bin, src);
	above = (uint32_t)__builtin_amdgcn_readlane((int)a, src);
	in_bin = (uint32_t)__builtin_amdgcn_readlane((int)hb, src);
}

// k-th largest among key[0..n) (optionally only where gate[j] == gate_val): MSB-first radix select, 8-bit digits,
// histogram filled with non-returning LDS atomics.  Returns the key; need_out = copies of it that belong to the top-k.
// PASSES < 4: the keys are known to be zero below bit 32 - 8*PASSES (bf16 scores: 2 passes).
template <bool GATED, int PASSES = 4>
__device__ __forceinline__ uint32_t wsel_kth(const WaveSel &w, const uint32_t *key, uint32_t n, uint32_t k, uint32_t &need_out,
											  const uint32_t *gate, uint32_t gate_val, uint32_t *in_bin_out = nullptr) {
	const uint32_t lane = (uint32_t)lane_id();
	uint32_t prefix = 0, need = k, hb_last = 0;
	for (int pass = 0; pass < PASSES; ++pass) {
		const int shift = 24 - 8 * pass;
#pragma unroll
		for (int i = 0; i < 4; ++i) w.hist[lane * 4 + i] = 0;
		__builtin_amdgcn_wave_barrier();
#pragma unroll 4
		for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
			const uint32_t j = j0 + lane;
			if (j < n) {
				const uint32_t x = key[j];
				bool c = (pass == 0) || ((x >> (shift + 8)) == prefix);
				if (GATED) c = c && (gate[j] == gate_val);
				if (c) atomicAdd(&w.hist[(x >> shift) & 255u], 1u);
			}
		}
		__builtin_amdgcn_wave_barrier();
		uint32_t bin, a, hb;
		wsel_find_bin(w.hist, lane, need, bin, a, hb);
		need -= a;
		prefix = (prefix << 8) | bin;
		hb_last = hb;
		__builtin_amdgcn_wave_barrier();
	}
	need_out = need;
	if (in_bin_out) *in_bin_out = hb_last;   // keys (passing the gate) whose significant bits equal the returned key's
	return prefix << (32 - 8 * PASSES);
}

// ONE pass for keys with 16 significant bits (bf16 scores) that are known to lie at or above `base16` (the 16-bit prefix of the threshold every
// candidate of an in-order stream has passed): the digit is (key >> 16) - base16, the last bin collects everything 255 or more above the base.
// Returns false -- nothing decided -- when the k-th largest key falls into that last bin (keys spread over more than 255 bf16 values above the
// threshold: the caller takes the two fixed-digit passes).  Round 5: candidates above a threshold share their sign /
// exponent byte, so the first of the two fixed-digit passes told next to nothing and serialised its atomics on one or two LDS words.
__device__ __forceinline__ bool wsel_kth16_based(const WaveSel &w, const uint32_t *key, uint32_t n, uint32_t k, uint32_t base16, uint32_t &T,
												  uint32_t &need_out, uint32_t &in_bin_out) {
	const uint32_t lane = (uint32_t)lane_id();
#pragma unroll
	for (int i = 0; i < 4; ++i) w.hist[lane * 4 + i] = 0;
	__builtin_amdgcn_wave_barrier();
#pragma unroll 4
	for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
		const uint32_t j = j0 + lane;
		if (j < n) {
			const uint32_t x = key[j] >> 16;
			const uint32_t d = x >= base16 ? x - base16 : 0u;   // (keys below the base do not occur; they would count as the base)
			atomicAdd(&w.hist[d < 255u ? d : 255u], 1u);
		}
	}
	__builtin_amdgcn_wave_barrier();
	uint32_t bin, a, hb;
	wsel_find_bin(w.hist, lane, k, bin, a, hb);
	__builtin_amdgcn_wave_barrier();
	if (bin == 255u) return false;
	T = (base16 + bin) << 16;
	need_out = k - a;
	in_bin_out = hb;
	return true;
}

// The same for keys that crowd into a narrow range (candidates of the fused sweep: every score is >= the sweep's threshold, so the
// sign / exponent byte -- and mostly the next one -- is common to all of them and the fixed-digit histogram above serialises 64 lanes on
// ONE LDS word per instruction).  Digits are taken from (key - min) instead, as many 8-bit digits as (max - min) has bytes: the first
// pass spreads the keys over up to 256 bins, later passes only touch the k-th key's bin.
__device__ __forceinline__ uint32_t wsel_kth_ranged(const WaveSel &w, const uint32_t *key, uint32_t n, uint32_t k, uint32_t &need_out) {
	const uint32_t lane = (uint32_t)lane_id();
	uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll 4
	for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
		const uint32_t j = j0 + lane;
		if (j < n) { const uint32_t x = key[j]; mn = x < mn ? x : mn; mx = x > mx ? x : mx; }
	}
	mn = wave_reduce<DppMin>(mn); mx = wave_reduce<DppMax>(mx);
	const uint32_t range = mx - mn;
	const int passes = range == 0u ? 0 : (32 - __clz(range) + 7) / 8;  // (uniform)
	uint32_t prefix = 0, need = k;
	for (int pass = 0; pass < passes; ++pass) {
		const int shift = 8 * (passes - 1 - pass);
#pragma unroll
		for (int i = 0; i < 4; ++i) w.hist[lane * 4 + i] = 0;
		__builtin_amdgcn_wave_barrier();
#pragma unroll 4
		for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
			const uint32_t j = j0 + lane;
			if (j < n) {
				const uint32_t x = key[j] - mn;
				// (shift + 8 == 32 only in pass 0 of a four-byte range, where every key takes part)
				if (pass == 0 || (x >> (shift + 8)) == prefix) atomicAdd(&w.hist[(x >> shift) & 255u], 1u);
			}
		}
		__builtin_amdgcn_wave_barrier();
		uint32_t bin, a, hb;
		wsel_find_bin(w.hist, lane, need, bin, a, hb);
		need -= a;
		prefix = (prefix << 8) | bin;
		__builtin_amdgcn_wave_barrier();
	}
	need_out = need;
	return mn + prefix;
}

// Cut the buffer down to its k best candidates (cnt > k), in place; the k-th best becomes the threshold.
// KEEP_TIES (mid-stream use only): keep every candidate that ties with the k-th score instead of resolving the tie by
// index, as long as they fit below `tie_limit`; the buffer then holds >= k entries and the threshold is the k-th SCORE,
// which is all an in-order stream needs (later elements lose score ties).  The final call must be exact.
// RANGED: the keys are expected in a narrow range (wsel_kth_ranged; full 32-bit keys only).
// BASED (HI_PASSES = 2 only): every key is known to be at or above base16 << 16 (wsel_kth16_based: one pass where that decides).
template <int HI_PASSES = 4, bool KEEP_TIES = false, bool RANGED = false, bool BASED = false>
__device__ __forceinline__ void wsel_compact(WaveSel &w, uint32_t k, uint32_t tie_limit = 0, uint32_t base16 = 0) {
	static_assert(!RANGED || HI_PASSES == 4, "the ranged radix select takes whole keys");
	static_assert(!BASED || HI_PASSES == 2, "the based select takes 16-bit keys");
	const uint32_t lane = (uint32_t)lane_id();
	const uint32_t n = w.cnt;
	__builtin_amdgcn_wave_barrier();
	uint32_t need, need2;
	// with HI_PASSES < 4 only the top 8*HI_PASSES key bits are significant (and the low bits of all keys of one sign agree)
	constexpr uint32_t M = HI_PASSES >= 4 ? 0xffffffffu : (0xffffffffu << (32 - 8 * HI_PASSES));
	uint32_t T = 0, cnt_eq = 0;   // cnt_eq: keys that tie with the k-th score = the keys in the last digit's bin (round 5: a pass over the buffer counted them)
	if constexpr (RANGED) {
		T = wsel_kth_ranged(w, w.whi, n, k, need);
#pragma unroll 4
		for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
			const uint32_t j = j0 + lane;
			cnt_eq += (uint32_t)__popcll(__ballot(j < n && (w.whi[j] & M) == T));
		}
	} else {
		bool done = false;
		if constexpr (BASED) done = wsel_kth16_based(w, w.whi, n, k, base16, T, need, cnt_eq);   // (uniform)
		if (!done) T = wsel_kth<false, HI_PASSES>(w, w.whi, n, k, need, w.whi, 0u, &cnt_eq);
	}
	uint32_t Tlo = 0;  // ties at the k-th score: the `need` smallest indices (largest lo) win
	if (cnt_eq > need && !(KEEP_TIES && (k - need) + cnt_eq <= tie_limit)) {
		// gate on the full key of the tie group: all its members share one score, hence one full hi key
		uint32_t tie_hi = 0;
		for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
			const uint32_t j = j0 + lane;
			const uint32_t x = j < n ? w.whi[j] : 0u;
			const unsigned long long m = __ballot(j < n && (x & M) == T);
			if (m) { tie_hi = __shfl(x, __ffsll((long long)m) - 1); break; }
		}
		Tlo = wsel_kth<true>(w, w.wlo, n, need, need2, w.whi, tie_hi);
	}
	// in-place forward compaction: chunk j0 is read into registers before anything at or below it is overwritten
	uint32_t base = 0;
	uint64_t kmin = ~0ull;
	for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
		const uint32_t j = j0 + lane;
		uint32_t h = 0, l = 0;
		if (j < n) { h = w.whi[j]; l = w.wlo[j]; }
		const bool sel = (j < n) && ((h & M) > T || ((h & M) == T && l >= Tlo));
		const unsigned long long m = __ballot(sel);
		__builtin_amdgcn_wave_barrier();
		if (sel) {
			const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			w.whi[pos] = h;
			w.wlo[pos] = l;
			if constexpr (!KEEP_TIES) {
				const uint64_t key = ((uint64_t)h << 32) | l;
				if (key < kmin) kmin = key;
			}
		}
		base += (uint32_t)__popcll(m);
	}
	w.cnt = base;  // == k (or k + extra score ties under KEEP_TIES)
	if constexpr (KEEP_TIES) {
		// mid-stream: the threshold an in-order stream needs is the k-th SCORE, and that is T itself -- a key that ties with it is always kept,
		// so the smallest kept key's score bits are T (round 5: two wave reductions over the kept keys found the same value)
		// (T carries the significant key bits only: the full key of a NEGATIVE bf16 score has its low bits set -- left clear, the threshold would sit a
		//  few fp32 ulps below the k-th score and every later element that ties with it would pass the stream's strict compare)
		w.tau_hi = (HI_PASSES < 4 && T < 0x80000000u) ? (T | ~M) : T;
		w.tau_lo = Tlo;
	} else {
		// smallest kept key = smallest hi, then the smallest lo among the lanes that hold it (two DPP reductions, no shuffles)
		const uint32_t min_hi = wave_reduce<DppMin>((uint32_t)(kmin >> 32));
		const uint32_t min_lo = wave_reduce<DppMin>((uint32_t)(kmin >> 32) == min_hi ? (uint32_t)kmin : 0xffffffffu);
		w.tau_hi = min_hi;  // the smallest kept key: exactly the k-th best in exact mode
		w.tau_lo = min_lo;
	}
	w.tau = f32_unsortable(w.tau_hi);
	__builtin_amdgcn_wave_barrier();
}

// base16 of an in-order stream whose candidates all passed `v > tau` (or every element, while tau is -inf): the 16-bit key prefix of tau --
// a bf16 score above tau has a key prefix at or above it.
__device__ __forceinline__ uint32_t wsel_base16(float tau) { return tau > -INFINITY ? f32_sortable(tau) >> 16 : 0u; }

// Out-of-line compaction for kernels that reach it from many unrolled sites (the scan: one inline copy is ~6 KB of code, a
// dozen of them no longer fit the instruction cache).  The selector's state crosses the call as plain scalars: the LDS offset
// of the wave's region going in, (count, threshold key) coming back through the histogram words.
template <int CAP, int HI_PASSES, bool KEEP_TIES>
__device__ __attribute__((noinline)) void wsel_compact_outlined(uint32_t lds_off, uint32_t n, uint32_t k, uint32_t tie_limit, uint32_t base16) {
	WaveSel w;
	w.whi = (uint32_t *)(__attribute__((address_space(3))) uint32_t *)(uintptr_t)lds_off;
	w.wlo = w.whi + CAP;
	w.sort_buf = reinterpret_cast<uint2 *>(w.wlo + CAP);
	w.hist = reinterpret_cast<uint32_t *>(w.sort_buf + 128);
	w.cnt = n; w.tau_hi = 0; w.tau_lo = 0; w.tau = 0.f;
	wsel_compact<HI_PASSES, KEEP_TIES, false, (HI_PASSES == 2)>(w, k, tie_limit, base16);
	if (lane_id() == 0) { w.hist[0] = w.cnt; w.hist[1] = w.tau_hi; w.hist[2] = w.tau_lo; }
	__builtin_amdgcn_wave_barrier();
}
// base16: see wsel_compact (16-bit keys only; 0 = nothing known: the fixed-digit passes run after one wasted pass at most)
template <int CAP, int HI_PASSES, bool KEEP_TIES>
__device__ __forceinline__ void wsel_compact_call(WaveSel &w, uint32_t k, uint32_t tie_limit = 0, uint32_t base16 = 0) {
	wsel_compact_outlined<CAP, HI_PASSES, KEEP_TIES>((uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)w.whi, w.cnt, k, tie_limit, base16);
	__builtin_amdgcn_wave_barrier();
	w.cnt = __builtin_amdgcn_readfirstlane(w.hist[0]);
	w.tau_hi = __builtin_amdgcn_readfirstlane(w.hist[1]);
	w.tau_lo = __builtin_amdgcn_readfirstlane(w.hist[2]);
	w.tau = f32_unsortable(w.tau_hi);
	__builtin_amdgcn_wave_barrier();
}

// Bitonic sort (descending) of 128 64-bit keys held as (hi, lo) pairs, two per lane: element i = e*64 + lane.
__device__ __forceinline__ void wave_sort128_desc(uint32_t (&hi)[2], uint32_t (&lo)[2]) {
	const int lane = lane_id();
#pragma unroll
	for (int size = 2; size <= 128; size <<= 1) {
#pragma unroll
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			if (stride == 64) {
				// partner is the other register of this lane; size == 128 here -> whole sequence descending
				const uint64_t a = ((uint64_t)hi[0] << 32) | lo[0], b = ((uint64_t)hi[1] << 32) | lo[1];
				if (a < b) {
					const uint32_t th = hi[0], tl = lo[0];
					hi[0] = hi[1]; lo[0] = lo[1]; hi[1] = th; lo[1] = tl;
				}
			} else {
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					const uint32_t oh = lane_xor(hi[e], stride), ol = lane_xor(lo[e], stride);
					const uint64_t mine = ((uint64_t)hi[e] << 32) | lo[e], other = ((uint64_t)oh << 32) | ol;
					const int i = e * 64 + lane;
					const bool desc = (i & size) == 0;
					const bool lower = (lane & stride) == 0;
					const bool keep_max = (desc == lower);
					const bool take_other = keep_max ? (other > mine) : (other < mine);
					if (take_other) { hi[e] = oh; lo[e] = ol; }
				}
			}
		}
	}
}

// The same for E * 64 keys, E per lane (element i = e*64 + lane): strides below 64 are lane exchanges, strides of 64 and more
// pair two registers of the same lane.  E = 8: the 512-key sort of the candidate select for 128 < k <= 512.
template <int E>
__device__ __forceinline__ void wave_sort_desc(uint32_t (&hi)[E], uint32_t (&lo)[E]) {
	const int lane = lane_id();
#pragma unroll
	for (int size = 2; size <= E * 64; size <<= 1) {
#pragma unroll
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			if (stride >= 64) {
				const int es = stride >> 6;
#pragma unroll
				for (int e = 0; e < E; ++e) {
					if ((e & es) == 0) {
						const int f = e | es;
						const bool desc = ((e * 64) & size) == 0;  // (bits of the lane are below `size` here)
						const uint64_t a = ((uint64_t)hi[e] << 32) | lo[e], b = ((uint64_t)hi[f] << 32) | lo[f];
						if (desc ? (a < b) : (a > b)) {
							const uint32_t th = hi[e], tl = lo[e];
							hi[e] = hi[f]; lo[e] = lo[f]; hi[f] = th; lo[f] = tl;
						}
					}
				}
			} else {
#pragma unroll
				for (int e = 0; e < E; ++e) {
					// (E = 16: hipcc does not unroll the stage loops, `stride` stays a run-time value and the DPP selection a chain of
				//  branches -- 554 k cycles per sort, measured; the shuffle takes a run-time stride as it is)
				uint32_t oh, ol;
				if constexpr (E <= 8) { oh = lane_xor(hi[e], stride); ol = lane_xor(lo[e], stride); }
				else { oh = (uint32_t)__shfl_xor((int)hi[e], stride); ol = (uint32_t)__shfl_xor((int)lo[e], stride); }
					const uint64_t mine = ((uint64_t)hi[e] << 32) | lo[e], other = ((uint64_t)oh << 32) | ol;
					const int i = e * 64 + lane;
					const bool desc = (i & size) == 0;
					const bool lower = (lane & stride) == 0;
					const bool keep_max = (desc == lower);
					const bool take_other = keep_max ? (other > mine) : (other < mine);
					if (take_other) { hi[e] = oh; lo[e] = ol; }
				}
			}
		}
	}
}

// Reduce to the k best, sort them, write the output row.  Fewer than k candidates -> (-inf, -1) padding.
// E = keys per lane of the final sort (k <= 64 E).
template <int OUTLINED_CAP = 0, int HI_PASSES = 4, int E = 2, bool RANGED = false>
__device__ __forceinline__ void wsel_finish(WaveSel &w, uint32_t k, float *out_val, int32_t *out_idx, const int32_t *__restrict__ remap = nullptr, uint32_t base16 = 0) {
	const int lane = lane_id();
	if (w.cnt > k) {
		if constexpr (OUTLINED_CAP > 0) wsel_compact_call<OUTLINED_CAP, HI_PASSES, false>(w, k, 0, base16);
		else wsel_compact<HI_PASSES, false, RANGED>(w, k);
	}
	__builtin_amdgcn_wave_barrier();
	uint32_t sh[E], sl[E];
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const uint32_t i = (uint32_t)(e * WAVE + lane);
		const bool in = i < w.cnt;
		sh[e] = in ? w.whi[i] : 0u;
		sl[e] = in ? w.wlo[i] : 0u;
	}
	if constexpr (E == 2) wave_sort128_desc(sh, sl);
	else wave_sort_desc<E>(sh, sl);
#pragma unroll
	for (int e = 0; e < E; ++e) {
		const uint32_t i = (uint32_t)(e * WAVE + lane);
		if (i < k) {
			const bool real = i < w.cnt;
			out_val[i] = real ? f32_unsortable(sh[e]) : -INFINITY;
			// remap (fused top-k with an item-order hint): the selected index is a ROW of the reordered operand, reported as the caller's item id
			const int32_t id = (int32_t)(0xffffffffu - sl[e]);
			out_idx[i] = real ? (remap ? remap[id] : id) : -1;
		}
	}
}

}  // namespace anncur
