// Workgroup-level exact top-k selection over a stream of (score, index) pairs.
//
// One 256-thread workgroup owns one query row.  Elements that may still belong to the
// row's top-k (score >= running k-th best) are appended, wave-aggregated, to an LDS
// candidate array of 64-bit composite keys; whenever the array could overflow before the
// next check it is cut back to exactly k entries with an MSB-first radix select (8-bit
// digits, early exit as soon as a digit bucket is taken whole) and the running threshold
// rises.  The final k keys are bitonic-sorted in LDS.  Exact for any input order and any
// distribution (sorted-ascending input just compacts more often).
#pragma once
#include "common.hpp"

namespace anncur {

constexpr int SEL_THREADS = 256;
constexpr int SEL_PASS = 4096;  // upper bound on pushes between two trigger checks

// PASS = upper bound on pushes between two trigger checks of the kernel that uses the state
template <int KMAX, int PASS = SEL_PASS>
struct SelCfg {
	static constexpr int TRIGGER = 2 * KMAX;
	static constexpr int CAP = PASS + 2 * KMAX;
	static constexpr size_t LDS_BYTES = (size_t)CAP * 8 + (size_t)KMAX * 8 + 256 * 4 + 16 * 4;
};

struct SelState {
	uint64_t *buf;   // [CAP] candidate keys
	uint64_t *keep;  // [KMAX] scratch for compaction
	uint32_t *hist;  // [256]
	uint32_t *scal;  // [16]: 0 cnt, 1 newcnt, 2 bin, 3 above, 4 bincount, 6..7 min key (u64), 8.. caller scratch
};

template <int KMAX, int PASS = SEL_PASS>
__device__ __forceinline__ SelState sel_carve(unsigned char *smem) {
	SelState s;
	s.buf = reinterpret_cast<uint64_t *>(smem);
	s.keep = s.buf + SelCfg<KMAX, PASS>::CAP;
	s.hist = reinterpret_cast<uint32_t *>(s.keep + KMAX);
	s.scal = s.hist + 256;
	return s;
}

__device__ __forceinline__ void sel_init(const SelState &s) {
	if (threadIdx.x < 16) s.scal[threadIdx.x] = 0;
	__syncthreads();
}

// Must be reached by every lane of the wave (wave-uniform control flow).
__device__ __forceinline__ void sel_push(bool hit, uint64_t key, uint64_t *dst, uint32_t *cnt_ptr) {
	const unsigned long long mask = __ballot(hit);
	if (mask == 0ull) return;
	const int lane = lane_id();
	const int leader = __ffsll((long long)mask) - 1;
	uint32_t base = 0;
	if (lane == leader) base = atomicAdd(cnt_ptr, (uint32_t)__popcll(mask));
	base = __shfl(base, leader);
	if (hit) dst[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = key;
}

// Cut buf[0..cnt) down to its k largest keys (cnt > k on entry, pushes already visible).
// Returns the k-th largest key (the new threshold).  Keys must be distinct.
template <int KMAX>
__device__ uint64_t sel_compact(const SelState &s, uint32_t k) {
	const int tid = threadIdx.x;
	const uint32_t cnt = s.scal[0];
	uint64_t prefix = 0;
	uint32_t need = k;
	int shift = 56;
	for (int pass = 0; pass < 8; ++pass) {
		shift = 56 - 8 * pass;
		s.hist[tid] = 0;
		__syncthreads();
		for (uint32_t i = tid; i < cnt; i += SEL_THREADS) {
			const uint64_t key = s.buf[i];
			const bool match = (pass == 0) || ((key >> (shift + 8)) == prefix);
			if (match) atomicAdd(&s.hist[(uint32_t)(key >> shift) & 255u], 1u);
		}
		__syncthreads();
		if (tid < WAVE) {
			const int b0 = 4 * tid;
			const uint32_t h0 = s.hist[b0], h1 = s.hist[b0 + 1], h2 = s.hist[b0 + 2], h3 = s.hist[b0 + 3];
			const uint32_t c = h0 + h1 + h2 + h3;
			uint32_t suf = c;  // inclusive suffix sum over lanes >= tid
			for (int d = 1; d < WAVE; d <<= 1) {
				const uint32_t t = __shfl_down(suf, d);
				if (tid + d < WAVE) suf += t;
			}
			uint32_t a = suf - c;
			if (a < need && suf >= need) {
				uint32_t bin, bc;
				if (a + h3 >= need) { bin = b0 + 3; bc = h3; }
				else { a += h3;
					if (a + h2 >= need) { bin = b0 + 2; bc = h2; }
					else { a += h2;
						if (a + h1 >= need) { bin = b0 + 1; bc = h1; }
						else { a += h1; bin = b0; bc = h0; } } }
				s.scal[2] = bin; s.scal[3] = a; s.scal[4] = bc;
			}
		}
		__syncthreads();
		const uint32_t bin = s.scal[2], above = s.scal[3], bc = s.scal[4];
		need -= above;
		prefix = (prefix << 8) | bin;
		if (bc == need) break;  // the whole bucket is taken: every key with this prefix is kept
	}
	const uint64_t T = prefix << shift;
	if (tid == 0) {
		s.scal[1] = 0;
		*reinterpret_cast<uint64_t *>(&s.scal[6]) = ~0ull;
	}
	__syncthreads();
	uint64_t lmin = ~0ull;
	for (uint32_t i0 = 0; i0 < cnt; i0 += SEL_THREADS) {
		const uint32_t i = i0 + tid;
		const bool in = i < cnt;
		const uint64_t key = in ? s.buf[i] : 0ull;
		const bool kp = in && key >= T;
		sel_push(kp, key, s.keep, &s.scal[1]);
		if (kp && key < lmin) lmin = key;
	}
	for (int d = WAVE / 2; d > 0; d >>= 1) {
		const uint64_t o = __shfl_xor(lmin, d);
		if (o < lmin) lmin = o;
	}
	if (lane_id() == 0) atomicMin(reinterpret_cast<unsigned long long *>(&s.scal[6]), (unsigned long long)lmin);
	__syncthreads();
	uint32_t newcnt = s.scal[1];
	if (newcnt > (uint32_t)KMAX) newcnt = KMAX;  // unreachable with distinct keys; never write past keep[]
	for (uint32_t i = tid; i < newcnt; i += SEL_THREADS) s.buf[i] = s.keep[i];
	const uint64_t tau = *reinterpret_cast<uint64_t *>(&s.scal[6]);
	__syncthreads();
	if (tid == 0) s.scal[0] = newcnt;
	__syncthreads();
	return tau;
}

// Sort buf[0..n) descending (n <= kp, kp a power of two <= CAP); slots n..kp-1 become 0.
__device__ inline void sel_sort_desc(uint64_t *buf, uint32_t n, uint32_t kp) {
	const uint32_t tid = threadIdx.x;
	for (uint32_t i = n + tid; i < kp; i += SEL_THREADS) buf[i] = 0ull;
	__syncthreads();
	for (uint32_t size = 2; size <= kp; size <<= 1) {
		for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
			for (uint32_t t = tid; t < (kp >> 1); t += SEL_THREADS) {
				const uint32_t i = 2 * t - (t & (stride - 1));
				const uint32_t j = i + stride;
				const bool desc = (i & size) == 0;
				const uint64_t a = buf[i], b = buf[j];
				if ((a < b) == desc) { buf[i] = b; buf[j] = a; }
			}
			__syncthreads();
		}
	}
}

// After the stream: reduce to k, sort, write row q of the outputs.
template <int KMAX>
__device__ void sel_finish(const SelState &s, uint32_t k, float *out_val, int32_t *out_idx, const int32_t *__restrict__ remap = nullptr) {
	__syncthreads();
	if (s.scal[0] > k) sel_compact<KMAX>(s, k);
	const uint32_t n = s.scal[0];
	uint32_t kp = 1;
	while (kp < k) kp <<= 1;
	sel_sort_desc(s.buf, n, kp);
	for (uint32_t j = threadIdx.x; j < k; j += SEL_THREADS) {
		if (j < n) {
			const uint64_t key = s.buf[j];
			out_val[j] = key_val(key);
			out_idx[j] = remap ? remap[key_idx(key)] : (int32_t)key_idx(key);
		} else {  // fewer than k selectable elements in the row
			out_val[j] = -INFINITY;
			out_idx[j] = -1;
		}
	}
}

// Offer one element per lane (wave-uniform call).  `tau` / `tau_key` are the caller's
// register copies of the running threshold.
__device__ __forceinline__ void sel_offer(const SelState &s, bool valid, float v, uint32_t idx, float tau,
										   uint64_t tau_key) {
	const bool maybe = valid && (v >= tau);
	if (__ballot(maybe) == 0ull) return;
	const uint64_t key = make_key(v, idx);
	sel_push(maybe && key > tau_key, key, s.buf, &s.scal[0]);
}

// Check after a batch of offers (<= SEL_PASS pushes since the last check).  Uniform.
template <int KMAX>
__device__ __forceinline__ void sel_maybe_compact(const SelState &s, uint32_t k, float &tau, uint64_t &tau_key) {
	__syncthreads();
	if (s.scal[0] > (uint32_t)SelCfg<KMAX>::TRIGGER) {
		tau_key = sel_compact<KMAX>(s, k);
		tau = key_val(tau_key);
	}
}

}  // namespace anncur
