// Streaming candidate select of the fused score + top-k (a7/a8): ONE wave per query, NO candidate buffer in LDS.
//
// The buffer-and-compact selector of wave_select.hpp is built for a stream whose threshold tightens as it goes (the exact scan).
// The candidates of the fused sweep are different: all of them are already >= the sweep's threshold, there are only a few times k
// of them, and they sit in HBM/L2.  Copying them into a 2048-entry LDS buffer (16 KB per wave -> 2 waves per SIMD) and compacting
// it with LDS-resident radix passes ran at one LDS round trip per step with nothing to hide it behind: s_memtime stamps at k = 500
// showed 76 k cycles in the load loop (one compaction inside) and 66 k in the final compaction + sort, per query.
//
// Here the radix select is a fixed sequence of passes:
//   pass A   one walk over the query's segments (8 loads of 64 entries in flight per lane): the 32-bit score keys at or above the
//            running threshold go to a 2048-key LDS buffer; their min / max and count
//   levels   8-bit digits of (key - min), as many as (max - min) has bytes (the keys crowd into a narrow range: see
//            wsel_kth_ranged), from the key buffer; each level only counts the keys inside the k-th key's bin of the level before
//            (more keys than the buffer holds: the levels walk the segments again instead -- slower, never wrong)
//   ties     (rare) the same on the index half among the candidates that tie with the k-th score
//   final    second walk: the k survivors are collected (ballot rank) into the LDS buffer, sorted in registers, written out
// TAU_ONLY (threshold refinement between sweep stages) stops after the levels: the k-th best score is the new threshold.
// LDS per wave: 9 KB whatever k (the old buffer: 10 KB for k <= 128, 18 KB for k <= 512).
#pragma once
#include "wave_select.hpp"
#ifndef SEL_STAMP
#define SEL_STAMP(i) do { } while (0)
#endif

namespace anncur {

// (the bin search of a digit histogram is wsel_find_bin of wave_select.hpp)
__device__ __forceinline__ uint32_t hist_find_bin(const uint32_t *hist, uint32_t lane, uint32_t need, uint32_t &above, uint32_t &in_bin) {
	uint32_t bin;
	wsel_find_bin(hist, lane, need, bin, above, in_bin);
	return bin;
}

// Visit the query's candidates segment by segment, 64 entries of ONE segment per visit (the last visit of a segment is partial),
// STREAM_U visits per batch with the batch's loads issued back to back.  The walk over the segment counts is scalar (the counts sit
// in lane sg of `c`; v_readlane with a uniform index): no per-entry segment search.  A flat index -> (segment, entry) binary search
// per lane costs seven ds_bpermute per 64 candidates and pass, which made the LDS pipe the limit of the whole kernel (measured: 15 k
// cycles per 512 candidates whether the searches ran one after the other or in lockstep).
// f(entry, valid) is called wave-uniformly once per visit.
constexpr int STREAM_U = 8;
template <bool NEED_IDX, typename F>
__device__ __forceinline__ void walk_candidates(const uint2 *__restrict__ qc, uint32_t c, int nseg, int capg, int lane, F &&f) {
	int sg = -1;
	uint32_t cnt = 0, e0 = 0;  // (uniform) current segment, its count, next entry
	bool more = true;
	while (more) {
		uint32_t off[STREAM_U], nv[STREAM_U];  // (uniform) first entry of the visit (in entries from qc), valid lanes
#pragma unroll
		for (int u = 0; u < STREAM_U; ++u) {
			while (more && e0 >= cnt) {
				if (++sg >= nseg) more = false;
				else { cnt = __builtin_amdgcn_readlane(c, sg); e0 = 0; }
			}
			off[u] = more ? (uint32_t)sg * (uint32_t)capg + e0 : 0u;
			nv[u] = more ? (cnt - e0 < (uint32_t)WAVE ? cnt - e0 : (uint32_t)WAVE) : 0u;
			e0 += WAVE;
		}
		uint2 e[STREAM_U];
#pragma unroll
		for (int u = 0; u < STREAM_U; ++u) {
			const uint2 *src = qc + off[u] + ((uint32_t)lane < nv[u] ? lane : 0);
			if (NEED_IDX) e[u] = *src;
			else { e[u].x = reinterpret_cast<const uint32_t *>(src)[0]; e[u].y = 0u; }
		}
#pragma unroll
		for (int u = 0; u < STREAM_U; ++u)
			if (nv[u] != 0u) f(e[u], (uint32_t)lane < nv[u]);
	}
}

// LDS per wave: 256-bin histogram + the key buffer (KCAP sortable score keys of pass A; reused as the output buffer of the final pass)
constexpr int STREAM_KCAP = 2048;
struct StreamSelLayout { static constexpr int BYTES = 256 * 4 + STREAM_KCAP * 4; };

// One radix level over the keys (k-th largest): visit(g) calls g(key, valid) for every key.  Returns the updated (prefix, need, n_eq).
template <typename V>
__device__ __forceinline__ void stream_level(uint32_t *hist, int lane, bool first, int shift, uint32_t mn, uint32_t &prefix, uint32_t &need,
											  uint32_t &n_eq, V &&visit) {
#pragma unroll
	for (int i = 0; i < 4; ++i) hist[lane * 4 + i] = 0;
	__builtin_amdgcn_wave_barrier();
	visit([&](uint32_t key, bool valid) {
		const uint32_t x = key - mn;
		// (shift + 8 == 32 only in the first level of a four-byte range, where every key takes part)
		if (valid && (first || (x >> (shift + 8)) == prefix)) atomicAdd(&hist[(x >> shift) & 255u], 1u);
	});
	__builtin_amdgcn_wave_barrier();
	uint32_t above;
	const uint32_t bin = hist_find_bin(hist, (uint32_t)lane, need, above, n_eq);
	need -= above;
	prefix = (prefix << 8) | bin;
	__builtin_amdgcn_wave_barrier();
}

// E = keys per lane of the final sort (k <= 64 E <= STREAM_KCAP / 2).
template <bool TAU_ONLY, int E>
__global__ __launch_bounds__(256) void select_stream_kernel(const uint2 *__restrict__ cand, const uint32_t *__restrict__ seg_cnt, int nseg,
															 int capg, int64_t Q, uint32_t k, float *__restrict__ out_val,
															 int32_t *__restrict__ out_idx, uint32_t *__restrict__ hard_cnt,
															 int32_t *__restrict__ hard_list, float *__restrict__ tau, int tau_stride, int prefilter,
															 const int32_t *__restrict__ remap) {
	static_assert(E * WAVE * 8 <= STREAM_KCAP * 4, "the output buffer reuses the key buffer");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = lane_id(), wave = threadIdx.x >> 6;
	const int64_t q = (int64_t)blockIdx.x * 4 + wave;
	if (q >= Q) return;
	uint32_t *hist = reinterpret_cast<uint32_t *>(smem + wave * StreamSelLayout::BYTES);
	uint32_t *keys = hist + 256;
	const uint32_t c = (lane < nseg) ? seg_cnt[q * nseg + lane] : 0u;
	const uint32_t total = wave_reduce<DppAdd>(c);
	auto defer = [&]() {  // the workgroup-level kernel recomputes this query exactly
		if (!TAU_ONLY && lane == 0) hard_list[atomicAdd(hard_cnt, 1u)] = (int32_t)q;
	};
	if (__ballot(c > (uint32_t)capg) != 0ull || total < k) { defer(); return; }  // TAU_ONLY: keep the old (still valid) threshold
	// the threshold the last sweep stage ran with is a valid lower bound on the k-th best: candidates of earlier stages below it
	// are ignored (at least k candidates are >= it by construction)
	const uint32_t floor_key = (tau && prefilter) ? f32_sortable(tau[q * tau_stride]) : 0u;
	const uint2 *qc = cand + q * nseg * (int64_t)capg;

	SEL_STAMP(0);
	// ---- pass A: the keys at or above the floor go to the LDS key buffer (as many as fit); their range and count
	uint32_t mn = 0xffffffffu, mx = 0u, n_in = 0;
	walk_candidates<false>(qc, c, nseg, capg, lane, [&](const uint2 &e, bool valid) {
		const uint32_t x = f32_sortable(__uint_as_float(e.x));
		const bool in = valid && x >= floor_key;
		const unsigned long long m = __ballot(in);
		if (in) {
			mn = x < mn ? x : mn; mx = x > mx ? x : mx;
			const uint32_t pos = n_in + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			if (pos < (uint32_t)STREAM_KCAP) keys[pos] = x;
		}
		n_in += (uint32_t)__popcll(m);
	});
	mn = wave_reduce<DppMin>(mn); mx = wave_reduce<DppMax>(mx);
	__builtin_amdgcn_wave_barrier();
	SEL_STAMP(1);
	if (n_in < k) { defer(); return; }

	// ---- levels: 8-bit digits of (key - mn), most significant first; from the key buffer, or (more keys than it holds) from HBM again
	const uint32_t range = mx - mn;
	const int passes = range == 0u ? 0 : (32 - __clz(range) + 7) / 8;  // (uniform)
	uint32_t prefix = 0, need = k, n_eq = n_in;  // n_eq: keys that share the digits fixed so far
	const bool buffered = n_in <= (uint32_t)STREAM_KCAP;  // (uniform)
	for (int pass = 0; pass < passes; ++pass) {
		const int shift = 8 * (passes - 1 - pass);
		if (buffered)
			stream_level(hist, lane, pass == 0, shift, mn, prefix, need, n_eq, [&](auto &&g) {
#pragma unroll 4
				for (uint32_t j0 = 0; j0 < n_in; j0 += WAVE) {
					const uint32_t j = j0 + (uint32_t)lane;
					g(keys[j < n_in ? j : 0u], j < n_in);
				}
			});
		else
			stream_level(hist, lane, pass == 0, shift, mn, prefix, need, n_eq, [&](auto &&g) {
				walk_candidates<false>(qc, c, nseg, capg, lane, [&](const uint2 &e, bool valid) {
					const uint32_t x = f32_sortable(__uint_as_float(e.x));
					g(x, valid && x >= floor_key);
				});
			});
	}
	SEL_STAMP(2);
	const uint32_t T = mn + prefix;  // key of the k-th best score; `need` of the n_eq candidates that carry it belong to the top-k
	if (TAU_ONLY) {
		if (lane == 0) tau[q * tau_stride] = fmaxf(tau[q * tau_stride], f32_unsortable(T));
		SEL_STAMP(3);
		return;
	}

	// ---- ties at the k-th score (rare): the `need` smallest indices (largest lo = ~index) win
	uint32_t Tlo = 0;
	if (n_eq > need) {
		uint32_t lprefix = 0, lneed = need, dummy;
		for (int pass = 0; pass < 4; ++pass)
			stream_level(hist, lane, pass == 0, 24 - 8 * pass, 0u, lprefix, lneed, dummy, [&](auto &&g) {
				walk_candidates<true>(qc, c, nseg, capg, lane, [&](const uint2 &e, bool valid) {
					g(0xffffffffu - e.y, valid && f32_sortable(__uint_as_float(e.x)) == T);
				});
			});
		Tlo = lprefix;
	}

	// ---- final pass: collect the k survivors (the key buffer is dead: it becomes the output buffer), sort, write
	if constexpr (!TAU_ONLY) {
		uint2 *obuf = reinterpret_cast<uint2 *>(keys);
		uint32_t base = 0;
		__builtin_amdgcn_wave_barrier();
		walk_candidates<true>(qc, c, nseg, capg, lane, [&](const uint2 &e, bool valid) {
			const uint32_t hi = f32_sortable(__uint_as_float(e.x)), lo = 0xffffffffu - e.y;
			const bool sel = valid && (hi > T || (hi == T && lo >= Tlo));
			const unsigned long long m = __ballot(sel);
			if (sel) {
				const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
				if (pos < (uint32_t)(E * WAVE)) obuf[pos] = make_uint2(hi, lo);
			}
			base += (uint32_t)__popcll(m);
		});
		__builtin_amdgcn_wave_barrier();
		// (base == k: keys of one query are distinct)
		uint32_t sh[E], sl[E];
#pragma unroll
		for (int e = 0; e < E; ++e) {
			const uint32_t i = (uint32_t)(e * WAVE + lane);
			const uint2 v = i < base ? obuf[i] : make_uint2(0u, 0u);
			sh[e] = v.x; sl[e] = v.y;
		}
		if constexpr (E == 2) wave_sort128_desc(sh, sl);
		else wave_sort_desc<E>(sh, sl);
		float *ov = out_val + q * (int64_t)k;
		int32_t *oi = out_idx + q * (int64_t)k;
#pragma unroll
		for (int e = 0; e < E; ++e) {
			const uint32_t i = (uint32_t)(e * WAVE + lane);
			if (i < k) {
				const bool real = i < base;
				ov[i] = real ? f32_unsortable(sh[e]) : -INFINITY;
				const int32_t id = (int32_t)(0xffffffffu - sl[e]);
				oi[i] = real ? (remap ? remap[id] : id) : -1;
			}
		}
		SEL_STAMP(3);
	}
}

}  // namespace anncur
