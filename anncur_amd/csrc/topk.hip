// Exact row-wise top-k (HBM-streaming scan), re-rank, overlap counts, gathers, dtype conversion.
#include <stdlib.h>
#include <atomic>
#include "select.hpp"
#include "wave_select.hpp"

using namespace anncur;

namespace {

constexpr int SCAN_PASS = 2048;  // elements per trigger check of the scan kernel (LDS: 20.5 KB at k <= 128 -> 7 workgroups per CU)

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<uint16_t> { static constexpr int N = 8; };

// native 16-byte vector: HIP's uint4 struct gets scalarised into four branched dword loads when used conditionally
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ __forceinline__ float vec_elem(const u32x4 &r, int e);
template <>
__device__ __forceinline__ float vec_elem<float>(const u32x4 &r, int e) { return __uint_as_float(r[e]); }
template <>
__device__ __forceinline__ float vec_elem<uint16_t>(const u32x4 &r, int e) {
	const uint32_t w = r[e >> 1];
	return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16));
}

// ------------------------------------------------------------------ a7/a8: exact scan
// One workgroup per row; the row is read once with 16-byte coalesced loads (one vector per thread per
// iteration, the next one prefetched into registers before the current one is filtered).
// SEED (k <= 128): the first two vectors of every thread give 512 group maxima; their k-th largest is a
// valid lower bound on the row's k-th best, so the stream starts with a tight threshold instead of
// pushing (and then radix-selecting) the first 4096 elements.
// Algorithmic HBM traffic: I*sizeof(T) bytes per row (+ 8*k bytes written).
template <typename T, int KMAX, bool SEED>
__global__ __launch_bounds__(SEL_THREADS) void rowwise_topk_kernel(const T *__restrict__ A, int64_t I, int64_t lda,
																	uint32_t k, float *__restrict__ out_val,
																	int32_t *__restrict__ out_idx) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const SelState s = sel_carve<KMAX, SCAN_PASS>(smem);
	sel_init(s);
	const int tid = threadIdx.x;
	const int64_t q = blockIdx.x;
	const T *row = A + q * lda;
	constexpr int VEC = VecOf<T>::N;
	float tau = -INFINITY;
	uint64_t tau_key = 0;

	const uintptr_t addr = reinterpret_cast<uintptr_t>(row);
	int64_t head = (int64_t)(((16 - (addr & 15)) & 15) / sizeof(T));
	if (head > I) head = I;
	const int64_t nvec = (I - head) / VEC;
	const int64_t tail0 = head + nvec * VEC;
	const u32x4 *vp = reinterpret_cast<const u32x4 *>(row + head);
	const int64_t vlast = nvec > 0 ? nvec - 1 : 0;  // loads are unconditional on a clamped index (one dwordx4 each); `ok` masks the offers
	int64_t base = 0;

	if (SEED && nvec >= 2 * SEL_THREADS) {
		// ---- threshold seed from 512 group maxima (group = one 16-byte vector)
		const u32x4 r0 = vp[tid], r1 = vp[SEL_THREADS + tid];
		float m0 = vec_elem<T>(r0, 0), m1 = vec_elem<T>(r1, 0);
		int a0 = 0, a1 = 0;
#pragma unroll
		for (int e = 1; e < VEC; ++e) {
			const float x0 = vec_elem<T>(r0, e), x1 = vec_elem<T>(r1, e);
			if (x0 > m0) { m0 = x0; a0 = e; }
			if (x1 > m1) { m1 = x1; a1 = e; }
		}
		// NaN-free maxima as composite keys (value, argmax index): real elements of the row
		s.buf[tid] = make_key(m0, (uint32_t)(head + (int64_t)tid * VEC + a0));
		s.buf[SEL_THREADS + tid] = make_key(m1, (uint32_t)(head + ((int64_t)SEL_THREADS + tid) * VEC + a1));
		if (tid == 0) s.scal[0] = 2 * SEL_THREADS;
		__syncthreads();
		const uint64_t kth = sel_compact<KMAX>(s, k);  // k <= 128 < 512 groups
		tau_key = kth - 1;                              // admit the k-th group maximum itself again below
		tau = key_val(kth);
		if (tid == 0) s.scal[0] = 0;                    // drop the maxima: the two vectors are re-offered in full
		__syncthreads();
#pragma unroll
		for (int e = 0; e < VEC; ++e) sel_offer(s, true, vec_elem<T>(r0, e), (uint32_t)(head + (int64_t)tid * VEC + e), tau, tau_key);
#pragma unroll
		for (int e = 0; e < VEC; ++e)
			sel_offer(s, true, vec_elem<T>(r1, e), (uint32_t)(head + ((int64_t)SEL_THREADS + tid) * VEC + e), tau, tau_key);
		// at most k groups * VEC elements can pass: <= 1024 < CAP
		base = 2 * SEL_THREADS;
	}
	{  // unaligned head and the tail: fewer than 2*VEC elements in total
		int64_t i = -1;
		if (tid < head) i = tid;
		else if (tid - head < I - tail0) i = tail0 + (tid - head);
		const bool in = i >= 0;
		const float v = in ? load_as_f32<T>(row + i) : 0.f;
		sel_offer(s, in, v, (uint32_t)i, tau, tau_key);
	}
	// ---- stream: SCAN_PASS elements per step (U vectors per thread), PF steps of loads kept in flight per thread
	// (the scan is latency-bound otherwise: one 16-byte load per thread in flight gave 2.6 TB/s)
	constexpr int U = SCAN_PASS / (SEL_THREADS * VEC);
	constexpr int PF = 4;
	constexpr int64_t STEP = (int64_t)SEL_THREADS * U;
	u32x4 pf[PF][U];
#pragma unroll
	for (int d = 0; d < PF; ++d)
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int64_t iv = base + d * STEP + (int64_t)u * SEL_THREADS + tid;
			pf[d][u] = vp[iv < nvec ? iv : vlast];
		}
	for (; base < nvec; base += PF * STEP) {
#pragma unroll
		for (int d = 0; d < PF; ++d) {  // statically indexed rotation over the PF register sets; steps past the row are masked
			const int64_t sb = base + d * STEP;  // (no branch around a step: a PHI copy of the prefetch registers would force vmcnt(0))
#pragma unroll
			for (int u = 0; u < U; ++u) {
				const int64_t iv = sb + (int64_t)u * SEL_THREADS + tid;
				const bool ok = iv < nvec;
				const int64_t i0 = head + iv * VEC;
				const u32x4 cur = pf[d][u];
				const int64_t ivn = iv + PF * STEP;
				pf[d][u] = vp[ivn < nvec ? ivn : vlast];
#pragma unroll
				for (int e = 0; e < VEC; ++e) sel_offer(s, ok, vec_elem<T>(cur, e), (uint32_t)(i0 + e), tau, tau_key);
			}
			sel_maybe_compact<KMAX>(s, k, tau, tau_key);
		}
	}
	sel_finish<KMAX>(s, k, out_val + q * (int64_t)k, out_idx + q * (int64_t)k);
}

// ------------------------------------------------------------------ a7/a8: exact scan, ONE WAVE PER ROW (k <= 128)
// Barrier-free scan: each wave streams its own row with 16-byte nontemporal loads (WS_PF in flight per lane), filters against
// its running threshold and keeps candidates in a wave-private LDS buffer (wave_select.hpp).  4 rows per 256-thread
// workgroup, 11 KB LDS per wave.  The wave-per-row access pattern itself streams at 6.2 TB/s (6.9 nontemporal) on MI355X
// (scripts/probes/stream_probe.hip); what the kernel adds is instruction issue, so the stream is built to issue little:
//  * seed: the maxima of the lane's first WS_PF vectors (packed max, values only) -> their k-th largest is a valid lower bound
//    on the row's k-th best; the threshold starts just below it with an EMPTY buffer and the stream starts at element 0, so
//    indices are in order from the first element on and a strict score compare is exact throughout;
//  * per vector a prefilter on the raw words ("does any of this lane's elements beat the threshold?"); the per-element path
//    runs only for vectors in which some lane has a candidate, and compares integers (bf16) without unpacking;
//  * the threshold used inside a block of WS_PF vectors is the one at block start: thresholds only rise, a stale one lets a
//    superset through, and the compaction (exact composite keys) sorts that out;
//  * the compaction is out of line (wsel_compact_call).
// Rounds (round 3, scripts/scan_rows_probe.py): the kernel keeps 4096 rows in flight; a launch costs whole rounds of them -- a full round
// is bandwidth-bound (4096 x 200 KB in 134-145 us), a partial one latency-bound (92 us for ONE row as for 1800: a wave alone streams
// 2.2 GB/s) -- so 10 000 rows cost 2 + 0.65 rounds (36.4 ns per row against 32.7 at 20 000 rows).  Cutting rows into 2 or 4 shares
// (one wave each, merged through LDS) to make the rounds finer was built and measured: WORSE everywhere (10 000 x 100 000: 0.387 -> 0.443 /
// 0.596 ms): every share pays its own seed threshold, candidate upkeep and final sort, ~24 us per share.
// bf16 ordering trick: y = x ^ M per 16-bit pattern with M = 0x8000 if tau >= 0 else 0xffff.  For tau >= 0 an element beats tau
// iff it is positive and larger, i.e. iff y > tau ^ 0x8000 as UNSIGNED (negative floats get the top bit cleared); for tau < 0
// iff y > ~tau as unsigned (positives become >= 0x8000, negatives order by falling magnitude).  NaN patterns can pass these
// integer tests; they are dropped at the push (v == v).
typedef unsigned short h16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short h16x2 __attribute__((ext_vector_type(2)));

constexpr int WS_CAP = 1024;  // compaction trigger (<= 512 candidates) + at most 512 pushes per step
// The trigger is k + max(64, 1.5 k): the threshold only tightens at a compaction, so a lazy trigger (512) let ~2.5x more
// elements through than a continuously updated threshold would; a compaction costs about as much as 50 pushes.
static inline uint32_t ws_trigger(uint32_t k) { const uint32_t m = k + (k >> 1) > 64u ? k + (k >> 1) : 64u; return k + m < 512u ? k + m : 512u; }
constexpr int WS_PF = 8;

template <typename T> struct ScanPre;
template <> struct ScanPre<float> {  // fp32 rows: plain float compares
	float pre;
	bool all;  // (wave-uniform) no threshold yet: every real number is a candidate, -inf included
	static constexpr bool pos = false;   // (no separate path for positive thresholds: the float compare is already three instructions)
	__device__ __forceinline__ void set(float tau) { pre = tau; all = !(tau > -INFINITY); }
	__device__ __forceinline__ bool any_pos(const u32x4 &c) const { return any(c); }
	__device__ __forceinline__ u32x4 xform(const u32x4 &c) const { return c; }
	__device__ __forceinline__ bool any(const u32x4 &y) const {
		return all || fmaxf(fmaxf(fmaxf(__uint_as_float(y[0]), __uint_as_float(y[1])), __uint_as_float(y[2])), __uint_as_float(y[3])) > pre;
	}
	__device__ __forceinline__ uint32_t word_hits(uint32_t yw) const { return (all || __uint_as_float(yw) > pre) ? 1u : 0u; }  // one element per word
	__device__ __forceinline__ bool elem_hit(uint32_t hw, int) const { return hw != 0u; }
	static __device__ __forceinline__ uint32_t group_max_key(const u32x4 &c) {  // NaN-free maximum as a sortable key
		return f32_sortable(fmaxf(fmaxf(fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1])), __uint_as_float(c[2])), __uint_as_float(c[3])));
	}
};
template <> struct ScanPre<uint16_t> {  // bf16 rows: packed 16-bit integer compares on y = x ^ M
	uint32_t mask, pre_pk;  // wave-uniform
	uint32_t pre_s;         // (pos) the threshold's own 16 bits in both halves
	bool all;               // (wave-uniform) no threshold yet: every real number is a candidate, -inf included
	bool pos;               // (wave-uniform) a threshold >= +0: x > tau iff x's bits exceed tau's as SIGNED 16-bit integers (negative
	                        // values are negative integers, positive floats order like their bits; NaN patterns with the sign clear pass and
	                        // are dropped at the push) -- the stream's steps then skip the xor: three packed maxima, one more against the
	                        // threshold, one compare (round 4: the scan on part of the chip is bound by instruction issue, not by loads in flight)
	__device__ __forceinline__ void set(float tau) {
		const uint32_t b = __builtin_amdgcn_readfirstlane(__float_as_uint(tau));
		all = b == 0xff800000u || (b & 0x7fffffffu) > 0x7f800000u;  // -inf (or, defensively, a NaN threshold)
		const bool neg = (b >> 31) != 0u;
		uint32_t t16 = b >> 16;                        // largest bf16 <= tau (tau is a bf16 value, -inf, or the fp32 just below one)
		if (neg && (b & 0xffffu) != 0u) t16 += 1u;
		const uint32_t m16 = neg ? 0xffffu : 0x8000u;
		mask = m16 | (m16 << 16);
		const uint32_t p16 = t16 ^ m16;
		pre_pk = p16 | (p16 << 16);
		pos = !neg && !all;
#ifdef ANNCUR_V_SCANPTR
		pos = false;
#endif
		pre_s = t16 | (t16 << 16);
	}
	__device__ __forceinline__ bool any_pos(const u32x4 &c) const {
		typedef short i16x8 __attribute__((ext_vector_type(8)));
		typedef short i16x4 __attribute__((ext_vector_type(4)));
		typedef short i16x2 __attribute__((ext_vector_type(2)));
		const i16x8 v = __builtin_bit_cast(i16x8, c);
		const i16x4 a = __builtin_elementwise_max(v.lo, v.hi);
		const i16x2 m = __builtin_elementwise_max(a.lo, a.hi);
		return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(m, __builtin_bit_cast(i16x2, pre_s))) != pre_s;
	}
	__device__ __forceinline__ u32x4 xform(const u32x4 &c) const { return c ^ mask; }
	__device__ __forceinline__ bool any(const u32x4 &y) const {
		const h16x8 v = __builtin_bit_cast(h16x8, y);
		const h16x4 a = __builtin_elementwise_max(v.lo, v.hi);
		const h16x2 m = __builtin_elementwise_max(a.lo, a.hi);
		return all || __builtin_bit_cast(uint32_t, __builtin_elementwise_max(m, __builtin_bit_cast(h16x2, pre_pk))) != pre_pk;
	}
	__device__ __forceinline__ uint32_t word_hits(uint32_t yw) const {  // non-zero half-word = that element beats the threshold
		if (all) return 0x00010001u;
		return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(h16x2, yw), __builtin_bit_cast(h16x2, pre_pk))) ^ pre_pk;
	}
	__device__ __forceinline__ bool elem_hit(uint32_t hw, int e) const { return ((e & 1) ? (hw >> 16) : (hw & 0xffffu)) != 0u; }
	static __device__ __forceinline__ uint32_t group_max_key(const u32x4 &c) {  // maximum of the 8 sortable 16-bit keys, as a 32-bit key
		u32x4 k;
#pragma unroll
		for (int j = 0; j < 4; ++j) {  // per half-word: x ^ (sign ? 0xffff : 0x8000); NaN patterns -> 0 (never the maximum)
			const uint32_t x = c[j];
			const uint32_t sgn = (x >> 15) & 0x00010001u;
			uint32_t s = x ^ (sgn * 0x7fffu) ^ 0x80008000u;
			const uint32_t mag = x & 0x7fff7fffu;
			if ((mag & 0xffffu) > 0x7f80u) s &= 0xffff0000u;
			if ((mag >> 16) > 0x7f80u) s &= 0x0000ffffu;
			k[j] = s;
		}
		const h16x8 v = __builtin_bit_cast(h16x8, k);
		const h16x4 a = __builtin_elementwise_max(v.lo, v.hi);
		const h16x2 m = __builtin_elementwise_max(a.lo, a.hi);
		const uint32_t mm = __builtin_bit_cast(uint32_t, m);
		const uint32_t top = (mm >> 16) > (mm & 0xffffu) ? (mm >> 16) : (mm & 0xffffu);
		return top << 16;
	}
};

// element e of a 16-byte vector, e not a compile-time constant (select chain: the vector stays in registers)
template <typename T> __device__ __forceinline__ T dyn_elem(const u32x4 &v, int e);
template <> __device__ __forceinline__ uint16_t dyn_elem<uint16_t>(const u32x4 &v, int e) {
	const int wi = e >> 1;
	const uint32_t x = wi == 0 ? v[0] : (wi == 1 ? v[1] : (wi == 2 ? v[2] : v[3]));
	return (uint16_t)((e & 1) ? (x >> 16) : (x & 0xffffu));
}
template <> __device__ __forceinline__ float dyn_elem<float>(const u32x4 &v, int e) {
	return __uint_as_float(e == 0 ? v[0] : (e == 1 ? v[1] : (e == 2 ? v[2] : v[3])));
}

// element e of a 16-byte vector as fp32: e a compile-time constant / a per-lane value (select chain: the vector stays in registers)
template <typename T> __device__ __forceinline__ float vec_elem_f32(const u32x4 &v, int e);
template <> __device__ __forceinline__ float vec_elem_f32<uint16_t>(const u32x4 &v, int e) {
	const uint32_t x = v[e >> 1];
	return __uint_as_float((e & 1) ? (x & 0xffff0000u) : (x << 16));
}
template <> __device__ __forceinline__ float vec_elem_f32<float>(const u32x4 &v, int e) { return __uint_as_float(v[e]); }
template <typename T> __device__ __forceinline__ float vec_elem_f32_dyn(const u32x4 &v, int e);
template <> __device__ __forceinline__ float vec_elem_f32_dyn<uint16_t>(const u32x4 &v, int e) {
	const int wi = e >> 1;
	const uint32_t x = wi == 0 ? v[0] : (wi == 1 ? v[1] : (wi == 2 ? v[2] : v[3]));
	return __uint_as_float((e & 1) ? (x & 0xffff0000u) : (x << 16));
}
template <> __device__ __forceinline__ float vec_elem_f32_dyn<float>(const u32x4 &v, int e) {
	return __uint_as_float(e == 0 ? v[0] : (e == 1 ? v[1] : (e == 2 ? v[2] : v[3])));
}

// the two tables of ScanGather from the ascending anchor columns: one thread per vector
__global__ __launch_bounds__(256) void gather_tables_kernel(const int32_t *__restrict__ col_idx, int n_idx, int64_t n_vec, int vec,
															uint32_t *__restrict__ vtab) {
	const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (v >= n_vec) return;
	const int64_t c0 = v * vec;
	int lo = 0, hi = n_idx;   // first anchor >= c0
	while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int64_t)col_idx[mid] < c0) lo = mid + 1; else hi = mid; }
	uint32_t m = 0u;
	for (int j = lo; j < n_idx && (int64_t)col_idx[j] < c0 + vec; ++j) m |= 1u << (int)((int64_t)col_idx[j] - c0);
	vtab[v] = ((uint32_t)lo << 8) | m;
}

// GATHER (round 3, a2 folded into the first pass over A -- SURVEY a2, reference ...splits.py:297,300): the wave also copies the row's
// anchor columns to cq[q, 0..n_idx) as their vectors stream past, so C_q = A[:, anchors] costs no second read of the row's sectors (the
// separate gather re-reads one 64-byte sector per anchor: 164 MB at cfg2).  The anchors arrive as a table over the row's 16-byte
// vectors (anncur_gather_tables: which elements of vector v are anchors, and how many anchors lie before it); a vector's table word
// is prefetched with the vector.  Rows must start 16-byte aligned (one vector grid for all rows).
struct ScanGather {
	const uint32_t *vtab;    // [ceil(I / VEC)]: bits 0..7: bit e = element e of the vector is an anchor column; bits 8..: anchors in the
	                         // vectors before it.  ONE word, prefetched with the vector: a rank fetched where it is needed would be a
	                         // dependent load whose wait also drains the eight prefetches in flight (loads return in order)
	const int32_t *col_idx;  // the n_idx anchor columns, ascending (the row's tail elements are looked up here)
	int32_t n_idx;
	void *cq;                // [Q x ldo] of A's element type
	int64_t ldo;
};
// RAGGED (round 5, the IVF search's packed score rows): row q holds row_len[q] elements (k <= row_len[q] <= I_all, the caller's contract).
template <typename T, bool GATHER = false, bool BUF = true, bool RAGGED = false, int E = 2>   // E: keys per lane of the final sort (k <= 64 E; E = 1 from the ragged entry only)
__global__ __launch_bounds__(256) void rowwise_topk_wave_kernel(const T *__restrict__ A, int64_t Q, int64_t I_all, int64_t lda, uint32_t k,
																 uint32_t trig, float *__restrict__ out_val, int32_t *__restrict__ out_idx,
																 const ScanGather gt = ScanGather{}, const int32_t *__restrict__ row_len = nullptr) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	// (the wave index through readfirstlane: the compiler then knows the row pointer is wave-uniform and keeps the stream's base address
	//  in scalar registers -- without it every load of the stream carried 64-bit per-lane address arithmetic)
	const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int64_t q = (int64_t)blockIdx.x * 4 + wave;
	if (q >= Q) return;
	const int64_t I = RAGGED ? min((int64_t)max(__builtin_amdgcn_readfirstlane(row_len[q]), 0), I_all) : I_all;   // (a length outside 0..I_all is clamped: no read past the row)
	// (A start stagger -- the workgroups resident at launch sleeping hashed offsets of up to 8..48 us so that the select phases of a CU's waves
	//  do not coincide -- was measured in round 4 and dropped: 0.580 -> 0.589..0.621 ms on 96 CUs, 0.330 -> 0.337..0.340 on the chip.  The
	//  phases are not what holds a part of the chip at 35 GB/s per CU: see DESIGN.md 4.4, 'what bounds the scan on part of the chip'.)
	WaveSel w = wsel_init<WS_CAP>(smem + wave * WaveSelLayout<WS_CAP>::BYTES);
	const T *row = A + q * lda;
	constexpr int VEC = VecOf<T>::N;
	constexpr int HP = sizeof(T) == 2 ? 2 : 4;  // radix passes over the score key: bf16 scores carry 16 key bits
	const uintptr_t addr = reinterpret_cast<uintptr_t>(row);
	int64_t head = (int64_t)(((16 - (addr & 15)) & 15) / sizeof(T));
	if (head > I) head = I;
	const int64_t nvec = (I - head) / VEC;
	const int64_t tail0 = head + nvec * VEC;
	const u32x4 *vp = reinterpret_cast<const u32x4 *>(row + head);
	const int64_t vlast = nvec > 0 ? nvec - 1 : 0;
	const int64_t nsteps = (nvec + WAVE - 1) / WAVE;
	// (BUF, round 4) the stream's loads as raw buffer loads: the row's vectors behind a resource in scalar registers, the lane's 16 bytes as
	// a 32-bit offset, the step as an immediate and the block as a scalar offset -- no per-lane 64-bit address arithmetic (three to four
	// VALU instructions per step of the pointer form: a fifth of the loop).  Rows under 2 GB (the launcher picks the pointer form otherwise).
	const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4 *>(vp), 0, BUF ? (int)(nvec * 16) : 0, 0x00027000);
	const uint32_t voff = (uint32_t)lane * 16u;

	// the first WS_PF vectors of every lane: they seed the threshold AND are the first prefetch set of the stream
	u32x4 pf[WS_PF];
	uint32_t pm[GATHER ? WS_PF : 1];   // (GATHER) anchor mask of the prefetched vector; 0 for the clamped duplicates past the row
	T *cqrow = GATHER ? reinterpret_cast<T *>(gt.cq) + q * gt.ldo : nullptr;
#pragma unroll
	for (int d = 0; d < WS_PF; ++d) {
		const int64_t iv = (int64_t)d * WAVE + lane;
		if (GATHER) pm[d] = iv < nvec ? gt.vtab[iv] : 0u;
		pf[d] = vp[iv < nvec ? iv : vlast];
	}
	if (nvec >= (int64_t)WS_PF * WAVE && k <= (uint32_t)(WS_PF * WAVE)) {
		// ---- seed: WS_PF*64 group maxima (real elements of the row); the k-th largest bounds the row's k-th best from below
#pragma unroll
		for (int d = 0; d < WS_PF; ++d) w.whi[d * WAVE + lane] = ScanPre<T>::group_max_key(pf[d]);
		__builtin_amdgcn_wave_barrier();
		uint32_t need;
		const uint32_t kth = wsel_kth<false, HP>(w, w.whi, (uint32_t)(WS_PF * WAVE), k, need, w.whi, 0u);
		// admit the bound itself: the threshold is the fp32 value just below it (the buffer is still empty).  If fewer than k
		// maxima are finite the bound is -inf (or the NaN key 0): the threshold stays at -inf = "none yet".
		if (kth > f32_sortable(-INFINITY)) {  // (a bound of -inf or below is no bound: stay unthresholded)
			float t = f32_unsortable(kth - 1u);
			if (fabsf(t) < 1.17549435e-38f) t = -1.17549435e-38f;  // bound = +-0 or a denormal: any negative normal admits it (and +0 == -0)
			w.tau = t;
		}
	}
	{  // unaligned head (fewer than VEC elements): lowest indices first
		const bool in = lane < head;
		wsel_offer_inorder(w, in, in ? load_as_f32<T>(row + lane) : 0.f, (uint32_t)lane);
	}
	// ---- stream, element 0 onwards, blocks of WS_PF vectors per lane.
	// Candidate path in two phases (round 2: the per-element work used to run for every step in which ANY lane passed -- ~60 % of
	// the steps, ~70 wave instructions each, one or two useful lanes).  Phase A, in the stream: a lane whose vector passes the
	// prefilter RECORDS it (16 bytes + element index of its first element) in a 64-entry staging area, ~10 instructions per
	// passing step.  Phase B, when the next step's vectors would not fit (and at the end): every lane takes ONE recorded vector
	// and the elements are tested lane-locally, 64 vectors at a time with all lanes busy.  Exactness: batches are processed in
	// stream order and the threshold only changes at a compaction, which runs between batches -- every element of a batch has a
	// larger index than everything that was in the buffer at the last compaction, so the strict compare against that threshold
	// stays exact whatever the order inside the batch.
	// Staging = the layout's sort scratch (64 x 16 B) + the first 64 histogram words (the histogram is only live inside a
	// compaction, which finds the staging area empty).
	ScanPre<T> sp;
	const uint32_t tie_limit = k + (trig - k) / 2;  // ties at the k-th score are kept while they fit below
	u32x4 *stage_vec = reinterpret_cast<u32x4 *>(w.sort_buf);
	uint32_t *stage_idx = w.hist;
	const uint32_t stage_vec_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) const char *)stage_vec;   // LDS byte addresses of the staging area
	const uint32_t stage_idx_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) const char *)stage_idx;
	uint32_t scnt = 0;  // staged vectors (wave-uniform)
	// (Round 4, measured and dropped: a threshold-only refresh -- the k-th select without the pass that rewrites the buffer -- every 16 / 32 / 50 /
	//  64 / 100 candidates, compaction only when the buffer runs out of room at 512: 0.578 -> 0.674 / 0.664 / 0.631 / 0.599 / 0.594 ms on 96 CUs,
	//  0.331 -> 0.367 ... 0.335 on the chip.  The select IS the cost of a compaction, and a buffer that is not trimmed makes every later one longer.)
	// (Round 5: the staged vector comes back in ONE ds_read_b128 and its elements are judged in registers -- a bit per element that passes --, then
	//  every lane pushes its passing elements one per round: as many rounds as the fullest lane has hits, two or three, where the loop over the
	//  eight element positions read LDS and ran the push path eight times.  The order of the pushes inside a batch is free: see above.)
#define SCAN_DRAIN()                                                                                                            \
	{                                                                                                                           \
		__builtin_amdgcn_wave_barrier();                                                                                        \
		const bool have = (uint32_t)lane < scnt;                                                                                \
		const u32x4 zero_ = {0u, 0u, 0u, 0u};                                                                                   \
		const u32x4 sv = have ? stage_vec[lane] : zero_;                                                                        \
		const uint32_t i0 = have ? stage_idx[lane] : 0u;                                                                        \
		const bool nothr = !(w.tau > -INFINITY);  /* no threshold yet: every real number is a candidate, -inf included */        \
		uint32_t pmask = 0u;                                                                                                    \
		_Pragma("unroll") for (int e = 0; e < VEC; ++e) {                                                                       \
			const float v = vec_elem_f32<T>(sv, e);                                                                             \
			const bool hit = nothr ? (v == v) : (v > w.tau);   /* (NaN fails both) */                                           \
			pmask |= hit ? (1u << e) : 0u;                                                                                      \
		}                                                                                                                       \
		if (!have) pmask = 0u;                                                                                                  \
		for (;;) {                                                                                                              \
			const bool hit = pmask != 0u;                                                                                       \
			const unsigned long long hm = __ballot(hit);                                                                        \
			if (hm == 0ull) break;                                                                                              \
			const int e = hit ? __builtin_ctz(pmask) : 0;                                                                       \
			wsel_push_mask(w, hm, hit, f32_sortable(vec_elem_f32_dyn<T>(sv, e)), 0xffffffffu - (i0 + (uint32_t)e));             \
			pmask &= pmask - 1u;                                                                                                \
		}                                                                                                                       \
		scnt = 0;                                                                                                               \
		__builtin_amdgcn_wave_barrier();                                                                                        \
		if (w.cnt > trig) wsel_compact_call<WS_CAP, HP, true>(w, k, tie_limit, wsel_base16(w.tau));                             \
	}
#define SCAN_STEP(d, FULL, POS)                                                                                                 \
	{                                                                                                                           \
		const u32x4 cur = pf[d];                                                                                                \
		const uint32_t mcur = GATHER ? pm[GATHER ? (d) : 0] : 0u;                                                               \
		bool pass;                                                                                                              \
		if (POS) pass = sp.any_pos(cur);                                                                                        \
		else pass = sp.any(sp.xform(cur));                                                                                      \
		if (FULL) {  /* the block and its prefetch lie inside the row: wave-uniform base + lane, nothing to clamp */            \
			if (GATHER) pm[GATHER ? (d) : 0] = mblk[((d) + WS_PF) * WAVE + lane];                                     \
			if (BUF) pf[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + (uint32_t)(((d) & 3) * 1024), soff + (((d) & 4) ? 4096u : 0u), 2); \
			else pf[d] = __builtin_nontemporal_load(blk + ((d) + WS_PF) * WAVE + lane);                                         \
		} else {  /* the last blocks: prefetches clamped, steps past the row masked (never branched around) */                  \
			const int64_t iv = (s0 + (d)) * WAVE + lane;                                                                        \
			const int64_t ivn = iv + (int64_t)WS_PF * WAVE;                                                                     \
			if (GATHER) pm[GATHER ? (d) : 0] = ivn < nvec ? gt.vtab[ivn] : 0u;                                       \
			pf[d] = vp[ivn < nvec ? ivn : vlast];                                                                               \
			pass = pass && iv < nvec;                                                                                           \
		}                                                                                                                       \
		if (GATHER) {  /* this vector's anchor elements -> cq[q, rank of the column], in column order */                        \
			if (__ballot((mcur & 0xffu) != 0u) != 0ull) {                                                                       \
				if ((mcur & 0xffu) != 0u) {                                                                                     \
					uint32_t pos = mcur >> 8;                                                                                   \
					for (uint32_t mm = mcur & 0xffu; mm != 0u; mm &= mm - 1u) cqrow[pos++] = dyn_elem<T>(cur, __builtin_ctz(mm)); \
				}                                                                                                               \
			}                                                                                                                   \
		}                                                                                                                       \
		const unsigned long long pm = __ballot(pass);                                                                           \
		if (pm != 0ull) {                                                                                                       \
			const uint32_t np = (uint32_t)__popcll(pm);                                                                         \
			if (scnt + np > (uint32_t)WAVE) SCAN_DRAIN()                                                                        \
			if (pass) {                                                                                                         \
				const uint32_t pos = scnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u)); \
				/* (explicit 32-bit LDS addresses: hipcc derived the second address from the first with a 64-bit multiply-add, a quarter-rate instruction) */ \
				*reinterpret_cast<__attribute__((address_space(3))) u32x4 *>((uintptr_t)(stage_vec_lds + pos * 16u)) = cur;            \
				*reinterpret_cast<__attribute__((address_space(3))) uint32_t *>((uintptr_t)(stage_idx_lds + pos * 4u)) = (uint32_t)(head + ((s0 + (d)) * WAVE + lane) * VEC); \
			}                                                                                                                   \
			scnt += np;                                                                                                         \
		}                                                                                                                       \
	}
	static_assert(WS_PF == 8, "the block spells out eight steps");
	// Two loops (round 3): the blocks whose prefetches lie inside the row -- all but the last one or two -- carry no per-lane index
	// arithmetic and no clamp (the one-loop version spent ~130 instructions per 16-byte step, most of them 64-bit address and mask
	// work for a case that arises in the last block only; that did not matter with 256 CUs on the stream, HBM-bound either way, but it
	// caps what a PART of the chip can stream: 31 GB/s per CU -- see bench.py --scan-mode partition).
	int64_t s0 = 0;
	for (; (s0 + 2 * WS_PF) * WAVE <= nvec; s0 += WS_PF) {
		sp.set(w.tau);  // frozen for the block
		const u32x4 *blk = vp + s0 * WAVE;  // (uniform)
		const uint32_t *mblk = GATHER ? gt.vtab + s0 * WAVE : nullptr;
		const uint32_t soff = (uint32_t)((s0 + WS_PF) * (WAVE * 16));  // (uniform) byte offset of the block the steps prefetch
		(void)mblk; (void)blk; (void)soff;
		if (sp.pos) {
			SCAN_STEP(0, true, true) SCAN_STEP(1, true, true) SCAN_STEP(2, true, true) SCAN_STEP(3, true, true) SCAN_STEP(4, true, true) SCAN_STEP(5, true, true) SCAN_STEP(6, true, true) SCAN_STEP(7, true, true)
		} else {
			SCAN_STEP(0, true, false) SCAN_STEP(1, true, false) SCAN_STEP(2, true, false) SCAN_STEP(3, true, false) SCAN_STEP(4, true, false) SCAN_STEP(5, true, false) SCAN_STEP(6, true, false) SCAN_STEP(7, true, false)
		}
	}
	for (; s0 < nsteps; s0 += WS_PF) {
		sp.set(w.tau);
		const u32x4 *blk = vp;  // (unused)
		const uint32_t *mblk = nullptr;
		const uint32_t soff = 0u;
		(void)blk; (void)mblk; (void)soff;
		SCAN_STEP(0, false, false) SCAN_STEP(1, false, false) SCAN_STEP(2, false, false) SCAN_STEP(3, false, false) SCAN_STEP(4, false, false) SCAN_STEP(5, false, false) SCAN_STEP(6, false, false) SCAN_STEP(7, false, false)
	}
	if (scnt > 0u) SCAN_DRAIN()
#undef SCAN_STEP
#undef SCAN_DRAIN
	{  // the tail (fewer than VEC elements), still in index order
		const bool in = lane < I - tail0;
		wsel_offer_inorder(w, in, in ? load_as_f32<T>(row + tail0 + lane) : 0.f, (uint32_t)(tail0 + lane));
	}
	if (GATHER) {  // anchors among the tail elements (no vector covers them): looked up directly
		for (int j = lane; j < gt.n_idx; j += WAVE) {
			const int32_t c = gt.col_idx[j];
			if ((int64_t)c >= tail0 && (int64_t)c < I) cqrow[j] = row[c];
		}
	}
	wsel_finish<WS_CAP, HP, E>(w, k, out_val + q * (int64_t)k, out_idx + q * (int64_t)k, nullptr, wsel_base16(w.tau));
}

// SHORT rows (round 5: I <= WS_CAP = 1024 -- the IVF probe's [queries x nlist] centroid scores, the k-means assignment, small matrices): the
// whole row goes into the wave's candidate buffer, then the final select + sort of the stream kernel (wsel_finish): no seed, no prefetch
// set, no staging -- the stream kernel spent ~2 500 instructions per 316-element row on its loop's fixed parts (42 us per 10 000 rows).
// Same result: exact, score descending, ties by ascending index, NaN never selected, -inf an ordinary candidate, (-inf, -1) padding.
template <typename T, int E = 2>   // E: keys per lane of the final sort (k <= 64 E)
__global__ __launch_bounds__(256) void rowwise_topk_short_kernel(const T *__restrict__ A, int64_t Q, int32_t I, int64_t lda, uint32_t k,
																  float *__restrict__ out_val, int32_t *__restrict__ out_idx) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int64_t q = (int64_t)blockIdx.x * 4 + wave;
	if (q >= Q) return;
	WaveSel w = wsel_init<WS_CAP>(smem + wave * WaveSelLayout<WS_CAP>::BYTES);
	constexpr int HP = sizeof(T) == 2 ? 2 : 4;
	const T *row = A + q * lda;
	for (int32_t i0 = 0; i0 < I; i0 += WAVE) {
		const int32_t i = i0 + lane;
		const float v = i < I ? load_as_f32<T>(row + i) : 0.f;
		wsel_push(w, i < I && v == v, f32_sortable(v), 0xffffffffu - (uint32_t)i);
	}
	__builtin_amdgcn_wave_barrier();
	wsel_finish<WS_CAP, HP, E>(w, k, out_val + q * (int64_t)k, out_idx + q * (int64_t)k);
}

// ------------------------------------------------------------------ k-th largest VALUE of short fp32 rows, one wave per row
// (the fused path's threshold step: rows of a few hundred group maxima).  The row's sortable keys go to a wave-private LDS
// array; MSB-first radix select with 8-bit digits (wsel_kth: four histogram passes instead of 32 bit-by-bit ballot rounds).
// No workgroup barrier.  NaN is never selected.
// Round 2: the key array is dynamic LDS sized to n (it was a static 4 x 2048 array: 36 KB per workgroup whatever n) and n may be
// 4096: the threshold of k = 1000 (4000 group maxima per query) no longer goes through the workgroup-level top-k (0.74 -> 0.21 ms).
constexpr int KTH_MAX_N = 4096;
// The coarse (16-bit key prefix) k-th AND k2-th largest keys (k2 <= k) from the SAME two passes over the keys: the second pass fills one
// histogram for the k-th key's top byte and, where the k2-th key has another top byte (the sortable key's top byte is the sign and seven
// exponent bits: thresholds around 2.0 straddle two of them), a second histogram for that one (a second wsel_kth cost the threshold kernel
// 15 us at cfg2: as much as the first).  hist2: 256 more words of the wave's LDS.
__device__ __forceinline__ void kth2_coarse(const WaveSel &w, uint32_t *hist2, const uint32_t *key, uint32_t n, uint32_t k, uint32_t k2, uint32_t &T, uint32_t &G) {
	const uint32_t lane = (uint32_t)lane_id();
	uint32_t bin0 = 0, a0 = 0, hb = 0, binG0 = 0, aG0 = 0;
#pragma unroll
	for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
		for (int i = 0; i < 4; ++i) { w.hist[lane * 4 + i] = 0; hist2[lane * 4 + i] = 0; }
		__builtin_amdgcn_wave_barrier();
#pragma unroll 4
		for (uint32_t j0 = 0; j0 < n; j0 += WAVE) {
			const uint32_t j = j0 + lane;
			if (j < n) {
				const uint32_t x = key[j];
				if (pass == 0) atomicAdd(&w.hist[x >> 24], 1u);
				else {
					if ((x >> 24) == bin0) atomicAdd(&w.hist[(x >> 16) & 255u], 1u);
					else if ((x >> 24) == binG0) atomicAdd(&hist2[(x >> 16) & 255u], 1u);
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
		if (pass == 0) {
			wsel_find_bin(w.hist, lane, k, bin0, a0, hb);
			wsel_find_bin(w.hist, lane, k2, binG0, aG0, hb);
		} else {
			uint32_t bin1, a1, binG1 = 0;
			wsel_find_bin(w.hist, lane, k - a0, bin1, a1, hb);
			T = ((bin0 << 8) | bin1) << 16;
			wsel_find_bin(binG0 == bin0 ? w.hist : hist2, lane, k2 - aG0, binG1, a1, hb);   // (uniform choice; aG0 == a0 when the top bytes agree)
			G = ((binG0 << 8) | binG1) << 16;
		}
		__builtin_amdgcn_wave_barrier();
	}
}

// Threshold LADDER (round 5, score16.hpp): with ladder != nullptr the wave also selects the k2-th largest value g (k2 < k) and writes
// ladder[q][j - 1] = tau + (g - tau) j / LADDER_LEVELS, j = 1..LADDER_LEVELS: candidate thresholds ABOVE tau that the sweep may move up to once it has
// counted k candidates at or above one of them.  Any values would be valid there (the count makes the move valid, not the level); value-linear
// spacing between two order statistics of the sample puts roughly geometric rank spacing on a score tail.  tau2[q] = tau (the cell the sweep's
// waves raise to the threshold they ended with: the select's prefilter).
__global__ __launch_bounds__(256) void kth_value_wave_kernel(const float *__restrict__ G, int64_t Q, int n, int64_t ldg, uint32_t k,
															  float *__restrict__ out, int64_t out_stride, int n_pad, int coarse,
															  float *__restrict__ ladder, uint32_t k2, float *__restrict__ tau2) {
	extern __shared__ __attribute__((aligned(16))) unsigned char kth_smem[];
	const uint32_t lane = (uint32_t)lane_id();
	const int wave = threadIdx.x >> 6;
	uint32_t *hist = reinterpret_cast<uint32_t *>(kth_smem) + wave * (256 + n_pad + (ladder ? 256 : 0));
	uint32_t *keys = hist + 256;
	const int64_t q = (int64_t)blockIdx.x * 4 + wave;
	if (q >= Q) return;
	const float *g = G + q * ldg;
	for (uint32_t j = lane; j < (uint32_t)n; j += WAVE) {
		const float v = g[j];
		keys[j] = (v == v) ? f32_sortable(v) : 0u;
	}
	__builtin_amdgcn_wave_barrier();
	WaveSel w;
	w.whi = keys; w.wlo = nullptr; w.sort_buf = nullptr; w.hist = hist;
	w.cnt = (uint32_t)n; w.tau_hi = 0; w.tau_lo = 0; w.tau = 0.f;
	uint32_t need;
	// (fixed sign/exponent digits: the range-adaptive digits of wsel_kth_ranged cost more here -- min / max pass and two reductions
	//  per row -- than the histogram conflicts they avoid: 0.034 vs 0.030 ms at n = 512, 0.23 vs 0.21 ms at n = 4000, measured)
	// PASSES: 4 = the exact k-th largest value; 2 = the k-th largest 16-bit key PREFIX with the low half zero -- in the sortable map that
	// is a value <= the exact one by less than one bf16 ulp (0.8 %), i.e. still a valid lower bound, for half the histogram passes
	// (the fused top-k's threshold: 23.6 -> 13 us at 512 keys, 91 -> 48 us at 2000; ~2 % more survivors in the first sweep stage)
	uint32_t kk, gk = 0;
	if (ladder) kth2_coarse(w, hist + 256 + n_pad, w.whi, (uint32_t)n, k, k2 < k ? k2 : k, kk, gk);   // (the ladder's call is the coarse one; its second histogram sits behind the keys)
	else if (coarse) kk = wsel_kth<false, 2>(w, w.whi, (uint32_t)n, k, need, w.whi, 0u);
	else kk = wsel_kth<false, 4>(w, w.whi, (uint32_t)n, k, need, w.whi, 0u);
	if (lane == 0) out[q * out_stride] = f32_unsortable(kk);
	if (ladder) {   // (uniform)
		const float t0 = f32_unsortable(kk), g = f32_unsortable(gk);
		float d = g - t0;
		if (!(d > 0.f) || !(d < INFINITY)) d = 0.f;   // (ties, +-inf: a flat ladder -- every level equals tau, which nothing can be counted against wrongly)
		if (lane < (uint32_t)LADDER_LEVELS) ladder[q * LADDER_LEVELS + lane] = t0 + d * ((float)(lane + 1) * (1.0f / LADDER_LEVELS));
		if (lane == 0 && tau2) tau2[q] = t0;
	}
}

// ------------------------------------------------------------------ a8: exact re-rank
// rerank = the k_out best (by exact score) among the k_retvr approximately retrieved items.
template <typename T, int KMAX>
__global__ __launch_bounds__(SEL_THREADS) void rerank_kernel(const T *__restrict__ A, int64_t I, int64_t lda,
															  const int32_t *__restrict__ approx_idx, int64_t ld_idx,
															  uint32_t k_retvr, uint32_t k_out, float *__restrict__ out_val,
															  int32_t *__restrict__ out_idx) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const SelState s = sel_carve<KMAX>(smem);
	sel_init(s);
	const int tid = threadIdx.x;
	const int64_t q = blockIdx.x;
	const T *row = A + q * lda;
	const int32_t *ai = approx_idx + q * ld_idx;
	float tau = -INFINITY;
	uint64_t tau_key = 0;
	for (uint32_t j0 = 0; j0 < k_retvr; j0 += SEL_THREADS) {  // k_retvr <= 2048 <= SEL_PASS: no mid-stream compaction needed
		const uint32_t j = j0 + tid;
		int32_t it = (j < k_retvr) ? ai[j] : -1;
		const bool in = it >= 0 && (int64_t)it < I;
		const float v = in ? load_as_f32<T>(row + it) : 0.f;
		sel_offer(s, in, v, (uint32_t)it, tau, tau_key);
	}
	sel_finish<KMAX>(s, k_out, out_val + q * (int64_t)k_out, out_idx + q * (int64_t)k_out);
}

// ------------------------------------------------------------------ a10: overlap counts
// common[p*Q + q] = |set(a[q,:ka[p]]) & set(b[q,:kb[p]])|.  pos[j] = first position of a[q,j] in b[q,:] (or lb).
constexpr int OVL_MAX_PAIRS = 64;
struct OvlPairs { int32_t ka[OVL_MAX_PAIRS]; int32_t kb[OVL_MAX_PAIRS]; };

__global__ __launch_bounds__(256) void overlap_kernel(const int32_t *__restrict__ a, int32_t la,
													   const int32_t *__restrict__ b, int32_t lb, int64_t Q,
													   OvlPairs pairs, int32_t n_pairs, int32_t *__restrict__ common) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	int32_t *sb = reinterpret_cast<int32_t *>(smem);  // [lb]
	int32_t *spos = sb + lb;                          // [la]
	int32_t *scnt = spos + la;                        // [n_pairs]
	const int tid = threadIdx.x;
	const int64_t q = blockIdx.x;
	for (int j = tid; j < lb; j += 256) sb[j] = b[q * lb + j];
	for (int p = tid; p < n_pairs; p += 256) scnt[p] = 0;
	__syncthreads();
	for (int j = tid; j < la; j += 256) {
		const int32_t x = a[q * la + j];
		int32_t pos = lb;
		// also require that x did not already appear earlier in a[q,:j] (set semantics); inputs are
		// distinct per row on this path, so this is a no-op scan kept cheap by the early exit below
		if (x >= 0) {
			for (int t = 0; t < lb; ++t)
				if (sb[t] == x) { pos = t; break; }
		}
		spos[j] = pos;
	}
	__syncthreads();
	for (int p = 0; p < n_pairs; ++p) {
		const int ka = pairs.ka[p], kb = pairs.kb[p];
		int c = 0;
		for (int j = tid; j < ka; j += 256) c += (spos[j] < kb) ? 1 : 0;
		for (int d = WAVE / 2; d > 0; d >>= 1) c += __shfl_xor(c, d);
		if (lane_id() == 0 && c) atomicAdd(&scnt[p], c);
	}
	__syncthreads();
	for (int p = tid; p < n_pairs; p += 256) common[(int64_t)p * Q + q] = scnt[p];
}

// Wave-per-row variant for short lists (la <= 128, lb <= 128): b in registers, broadcast with readlane.
__global__ __launch_bounds__(256) void overlap_wave_kernel(const int32_t *__restrict__ a, int32_t la, const int32_t *__restrict__ b,
															int32_t lb, int64_t Q, OvlPairs pairs, int32_t n_pairs,
															int32_t *__restrict__ common) {
	const int lane = lane_id();
	const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (q >= Q) return;
	int32_t av[2], bv[2], pos[2];
#pragma unroll
	for (int e = 0; e < 2; ++e) {
		const int j = e * WAVE + lane;
		av[e] = j < la ? a[q * la + j] : -1;
		bv[e] = j < lb ? b[q * lb + j] : -2;
		pos[e] = 0x7fffffff;
	}
	// position of every a-element in b.  Round 5: b's elements from the LAST to the first, each broadcast as a scalar and compared against the
	// lanes' a-elements with a compare that yields a lane mask -- a set bit is one v_writelane (the earliest position is written last, so it
	// stays); the version before kept a running minimum in every lane: two compares, two conditional moves per element and list half.
#pragma unroll
	for (int e = 1; e >= 0; --e) {
		const int lim = min(WAVE, lb - e * WAVE);
		for (int t = lim - 1; t >= 0; --t) {
			const int32_t x = __builtin_amdgcn_readlane(bv[e], t);
			const int p = e * WAVE + t;
			const unsigned long long m0 = __ballot(av[0] == x), m1 = __ballot(av[1] == x);
#if defined(__HIP_DEVICE_COMPILE__)
			if (m0 != 0ull) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(pos[0]) : "s"(p), "s"((int)__builtin_ctzll(m0)) : "m0");   // (more than one lane: a's -1 padding, never counted)
			if (m1 != 0ull) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(pos[1]) : "s"(p), "s"((int)__builtin_ctzll(m1)) : "m0");
#endif
		}
	}
	for (int p = 0; p < n_pairs; ++p) {
		const int ka = pairs.ka[p], kb = pairs.kb[p];
		const int c = __popcll(__ballot(lane < ka && av[0] >= 0 && pos[0] < kb)) +
					  __popcll(__ballot(WAVE + lane < ka && av[1] >= 0 && pos[1] < kb));
		if (lane == 0) common[(int64_t)p * Q + q] = c;
	}
}

// ------------------------------------------------------------------ a2: gathers, conversion
template <typename TS, typename TD>
__device__ __forceinline__ TD cvt(TS x);
template <> __device__ __forceinline__ float cvt<float, float>(float x) { return x; }
template <> __device__ __forceinline__ uint16_t cvt<uint16_t, uint16_t>(uint16_t x) { return x; }
template <> __device__ __forceinline__ float cvt<uint16_t, float>(uint16_t x) { return bf16_bits_to_f32(x); }
template <> __device__ __forceinline__ uint16_t cvt<float, uint16_t>(float x) { return f32_to_bf16_bits(x); }

// out[r, j] = A[r, col_idx[j]]: one WAVE per row, four rows per workgroup (a workgroup per row made the launch dispatch-bound:
// 10 000 workgroups of one load per thread); the anchor indices come out of L1/L2, the reads are strided element loads.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void gather_cols_kernel(const TS *__restrict__ A, int64_t n_rows, int64_t n_cols,
														   int64_t lda, const int32_t *__restrict__ col_idx,
														   int32_t n_idx, TD *__restrict__ out, int64_t ldo) {
	const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_rows) return;
	const TS *row = A + r * lda;
	// four elements per lane and pass, their loads issued back to back (round 3: one element per loop iteration, index load -> element
	// load -> store in a dependent chain, left one or two loads in flight per lane: 0.069 ms for 10 000 x 256 elements)
	constexpr int U = 4;
	for (int j0 = threadIdx.x & 63; j0 < n_idx; j0 += 64 * U) {
		int32_t c[U];
		TS v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) c[u] = (j0 + 64 * u < n_idx) ? col_idx[j0 + 64 * u] : -1;
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const bool in = c[u] >= 0 && (int64_t)c[u] < n_cols;
			v[u] = in ? row[c[u]] : cvt<float, TS>(0.f);
		}
#pragma unroll
		for (int u = 0; u < U; ++u)
			if (j0 + 64 * u < n_idx) out[r * ldo + j0 + 64 * u] = cvt<TS, TD>(v[u]);
	}
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void gather_rows_kernel(const TS *__restrict__ A, int64_t n_rows, int64_t n_cols,
														   int64_t lda, const int32_t *__restrict__ row_idx,
														   TD *__restrict__ out, int64_t ldo) {
	const int64_t j = blockIdx.y;
	const int32_t r = row_idx[j];
	const bool in = r >= 0 && (int64_t)r < n_rows;
	const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (c < n_cols) out[j * ldo + c] = in ? cvt<TS, TD>(A[(int64_t)r * lda + c]) : cvt<float, TD>(0.f);
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS *__restrict__ src, int64_t lds_, TD *__restrict__ dst,
													   int64_t ldd, int64_t n_rows, int64_t n_cols) {
	const int64_t r = blockIdx.y;
	const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (c < n_cols) dst[r * ldd + c] = cvt<TS, TD>(src[r * lds_ + c]);
}

// dst may be mapped (pinned) host memory: 16-byte coalesced writes over the host link, no copy engine involved
__global__ __launch_bounds__(256) void copy_bytes_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
														  const unsigned char *__restrict__ src_tail, unsigned char *__restrict__ dst_tail, int n_tail) {
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i < n16) dst[i] = src[i];
	if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

template <typename F>
int dispatch2(int sd, int dd, F &&f) {
	if (sd == ANNCUR_F32 && dd == ANNCUR_F32) return f((const float *)nullptr, (float *)nullptr);
	if (sd == ANNCUR_F32 && dd == ANNCUR_BF16) return f((const float *)nullptr, (uint16_t *)nullptr);
	if (sd == ANNCUR_BF16 && dd == ANNCUR_F32) return f((const uint16_t *)nullptr, (float *)nullptr);
	return f((const uint16_t *)nullptr, (uint16_t *)nullptr);
}

int kmax_class(int k) { return k <= 128 ? 128 : (k <= 512 ? 512 : 2048); }

}  // namespace

// internal (not part of the C ABI): tau[q] = k-th largest of G[q, :n], n <= 2048
int anncur_internal_kth_value(const float *G, int64_t Q, int n, int64_t ldg, int k, float *out, int64_t out_stride, hipStream_t st, int coarse,
							  float *ladder, int k2, float *tau2) {
	ANNCUR_REQUIRE(n >= 1 && n <= KTH_MAX_N && k >= 1 && k <= n, ANNCUR_E_INVALID, "kth_value: need 1 <= k <= n <= 4096");
	const unsigned grid = (unsigned)ceil_div64(Q, 4);
	const int n_pad = (n + 63) & ~63;
	const int lds = 4 * (256 + n_pad + (ladder ? 256 : 0)) * 4;  // (<= 72 KB: above the 64 KB default only for n > 3584)
	int rc;
	if (lds > 64 * 1024 && (rc = anncur_ensure_dyn_lds((const void *)kth_value_wave_kernel, lds)) != ANNCUR_OK) return rc;
	hipLaunchKernelGGL(kth_value_wave_kernel, dim3(grid), dim3(256), lds, st, G, Q, n, ldg, (uint32_t)k, out, out_stride, n_pad, coarse, ladder, (uint32_t)(k2 < 1 ? 1 : k2), tau2);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

// rows of the wave-per-row scan resident on the chip at once (what one "round" of the scan covers): anncur_eval_topk sizes its row
// chunks in multiples of this so that a chunk does not end on a mostly empty round
int anncur_internal_scan_rows_in_flight(int dtype) {
	// per (device, dtype), like the caches of misc.hip (a process may drive GPUs with different CU counts / partition modes); relaxed
	// atomics: two threads racing on the first use both compute the same value
	constexpr int MAX_DEV = 64;
	static std::atomic<int> cached[MAX_DEV][2];
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
	const int di = dtype == ANNCUR_F32 ? 0 : 1;
	std::atomic<int> *slot = (dev >= 0 && dev < MAX_DEV) ? &cached[dev][di] : nullptr;
	int v = slot ? slot->load(std::memory_order_relaxed) : 0;
	if (v == 0) {
		int occ = 0;
		const size_t lds = 4 * (size_t)WaveSelLayout<WS_CAP>::BYTES;
		const hipError_t e = di == 0 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, rowwise_topk_wave_kernel<float>, 256, lds)
									 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, rowwise_topk_wave_kernel<uint16_t>, 256, lds);
		if (e != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 3; }
		v = occ * 4 * anncur_num_cu();
		if (slot) slot->store(v, std::memory_order_relaxed);
	}
	return v;
}

// =================================================================== C ABI
extern "C" int anncur_rowwise_topk(const void *A, int dtype, int64_t Q, int64_t I, int64_t lda, int32_t k,
								   float *out_val, int32_t *out_idx, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(dtype), ANNCUR_E_INVALID, "rowwise_topk: bad dtype %d", dtype);
	ANNCUR_REQUIRE(Q >= 0 && I >= 1 && lda >= I, ANNCUR_E_INVALID, "rowwise_topk: bad shape Q=%lld I=%lld lda=%lld",
				   (long long)Q, (long long)I, (long long)lda);
	ANNCUR_REQUIRE(k >= 1 && k <= ANNCUR_MAX_TOPK && k <= I, ANNCUR_E_INVALID, "rowwise_topk: k=%d out of range (1..min(I,%d))",
				   k, ANNCUR_MAX_TOPK);
	ANNCUR_REQUIRE(I < (int64_t)0x7fffffff, ANNCUR_E_INVALID, "rowwise_topk: I too large for int32 indices");
	ANNCUR_REQUIRE(A && out_val && out_idx, ANNCUR_E_INVALID, "rowwise_topk: null pointer");
	ANNCUR_REQUIRE(((uintptr_t)A % dtype_size(dtype)) == 0, ANNCUR_E_INVALID, "rowwise_topk: misaligned A");
	if (Q == 0) return ANNCUR_OK;
	hipStream_t st = (hipStream_t)stream;
	const int kc = kmax_class(k);
#define LAUNCH_ROWTOPK(T, KM)                                                                                   \
	do {                                                                                                        \
		{ const int rc_ = anncur_ensure_dyn_lds((const void *)rowwise_topk_kernel<T, KM, (KM == 128)>, (int)(SelCfg<KM, SCAN_PASS>::LDS_BYTES)); if (rc_ != ANNCUR_OK) return rc_; } \
		hipLaunchKernelGGL((rowwise_topk_kernel<T, KM, (KM == 128)>), dim3((unsigned)Q), dim3(SEL_THREADS), (SelCfg<KM, SCAN_PASS>::LDS_BYTES), st, \
						   (const T *)A, I, lda, (uint32_t)k, out_val, out_idx);                                \
	} while (0)
	ANNCUR_REQUIRE(Q < (int64_t)0x7fffffff, ANNCUR_E_INVALID, "rowwise_topk: Q too large");
	bool wave_scan = k <= WSEL_K;  // barrier-free path: one wave per row
#ifdef ANNCUR_TIMING_EXPERIMENTS
	if (getenv("ANNCUR_DEBUG_BLOCK_SCAN")) wave_scan = false;
#endif
	if (wave_scan && I <= (int64_t)WS_CAP) {   // short rows: the whole row into the wave's buffer, then the final select + sort
		const size_t lds = 4 * (size_t)WaveSelLayout<WS_CAP>::BYTES;
		const unsigned grid = (unsigned)ceil_div64(Q, 4);
		if (dtype == ANNCUR_F32) {
			if (k <= 64) hipLaunchKernelGGL((rowwise_topk_short_kernel<float, 1>), dim3(grid), dim3(256), lds, st, (const float *)A, Q, (int32_t)I, lda, (uint32_t)k, out_val, out_idx);
			else hipLaunchKernelGGL((rowwise_topk_short_kernel<float, 2>), dim3(grid), dim3(256), lds, st, (const float *)A, Q, (int32_t)I, lda, (uint32_t)k, out_val, out_idx);
		} else {
			if (k <= 64) hipLaunchKernelGGL((rowwise_topk_short_kernel<uint16_t, 1>), dim3(grid), dim3(256), lds, st, (const uint16_t *)A, Q, (int32_t)I, lda, (uint32_t)k, out_val, out_idx);
			else hipLaunchKernelGGL((rowwise_topk_short_kernel<uint16_t, 2>), dim3(grid), dim3(256), lds, st, (const uint16_t *)A, Q, (int32_t)I, lda, (uint32_t)k, out_val, out_idx);
		}
		ANNCUR_LAUNCH_OK();
		return ANNCUR_OK;
	}
	if (wave_scan) {
		size_t lds = 4 * (size_t)WaveSelLayout<WS_CAP>::BYTES;
#ifdef ANNCUR_TIMING_EXPERIMENTS
		if (const char *dbg = getenv("ANNCUR_DEBUG_SCAN_LDS")) lds = (size_t)atoi(dbg);  // occupancy experiment: larger request = fewer waves per SIMD
#endif
		const unsigned grid = (unsigned)ceil_div64(Q, 4);
		uint32_t trig = ws_trigger((uint32_t)k);
#ifdef ANNCUR_TIMING_EXPERIMENTS
		if (const char *dbg = getenv("ANNCUR_DEBUG_SCAN_TRIGGER")) {  // tuning knob: any value in (k, 512] is exact
			const int t = atoi(dbg);
			if (t > k && t <= 512) trig = (uint32_t)t;
		}
#endif
		bool buf = I * (int64_t)dtype_size(dtype) < ((int64_t)1 << 31);   // the stream's loads through a buffer resource (32-bit offsets)
#ifdef ANNCUR_V_SCANPTR   // (A/B variant build: round 3's stream loop -- pointer loads, xor prefilter)
		buf = false;
#endif
		if (dtype == ANNCUR_F32) {
			if (buf) hipLaunchKernelGGL((rowwise_topk_wave_kernel<float, false, true>), dim3(grid), dim3(256), lds, st, (const float *)A, Q, I, lda, (uint32_t)k, trig, out_val, out_idx);
			else hipLaunchKernelGGL((rowwise_topk_wave_kernel<float, false, false>), dim3(grid), dim3(256), lds, st, (const float *)A, Q, I, lda, (uint32_t)k, trig, out_val, out_idx);
		} else {
			if (buf) hipLaunchKernelGGL((rowwise_topk_wave_kernel<uint16_t, false, true>), dim3(grid), dim3(256), lds, st, (const uint16_t *)A, Q, I, lda, (uint32_t)k, trig, out_val, out_idx);
			else hipLaunchKernelGGL((rowwise_topk_wave_kernel<uint16_t, false, false>), dim3(grid), dim3(256), lds, st, (const uint16_t *)A, Q, I, lda, (uint32_t)k, trig, out_val, out_idx);
		}
		ANNCUR_LAUNCH_OK();
		return ANNCUR_OK;
	}
	if (dtype == ANNCUR_F32) {
		if (kc == 128) LAUNCH_ROWTOPK(float, 128); else if (kc == 512) LAUNCH_ROWTOPK(float, 512); else LAUNCH_ROWTOPK(float, 2048);
	} else {
		if (kc == 128) LAUNCH_ROWTOPK(uint16_t, 128); else if (kc == 512) LAUNCH_ROWTOPK(uint16_t, 512); else LAUNCH_ROWTOPK(uint16_t, 2048);
	}
#undef LAUNCH_ROWTOPK
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

/* Ragged rows (round 5: the IVF search's packed score rows, ivf.hip): anncur_rowwise_topk of a matrix whose row q holds row_len[q]
 * elements; the wave-per-row scan only (k <= 128). */
extern "C" int anncur_rowwise_topk_ragged(const void *A, int dtype, int64_t Q, int64_t I_max, int64_t lda, const int32_t *row_len, int32_t k,
										  float *out_val, int32_t *out_idx, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(dtype), ANNCUR_E_INVALID, "rowwise_topk_ragged: bad dtype %d", dtype);
	ANNCUR_REQUIRE(Q >= 0 && Q < (int64_t)0x7fffffff && I_max >= 1 && I_max < (int64_t)0x7fffffff && lda >= I_max, ANNCUR_E_INVALID,
				   "rowwise_topk_ragged: bad shape Q=%lld I_max=%lld lda=%lld", (long long)Q, (long long)I_max, (long long)lda);
	ANNCUR_REQUIRE(k >= 1 && k <= I_max, ANNCUR_E_INVALID, "rowwise_topk_ragged: k=%d out of range (1..I_max)", k);
	if (k > WSEL_K) { anncur_set_error("rowwise_topk_ragged: k = %d above the wave-per-row scan's %d", k, WSEL_K); return ANNCUR_E_UNSUPPORTED; }
	if (Q == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(A && row_len && out_val && out_idx && ((uintptr_t)A % dtype_size(dtype)) == 0, ANNCUR_E_INVALID, "rowwise_topk_ragged: null or misaligned pointer");
	hipStream_t st = (hipStream_t)stream;
	const size_t lds = 4 * (size_t)WaveSelLayout<WS_CAP>::BYTES;
	const unsigned grid = (unsigned)ceil_div64(Q, 4);
	const uint32_t trig = ws_trigger((uint32_t)k);
	const bool buf = I_max * (int64_t)dtype_size(dtype) < ((int64_t)1 << 31);
#define LAUNCH_RAGGED(T, B, EE) hipLaunchKernelGGL((rowwise_topk_wave_kernel<T, false, B, true, EE>), dim3(grid), dim3(256), lds, st, (const T *)A, Q, I_max, lda, (uint32_t)k, trig, out_val, out_idx, ScanGather{}, row_len)
	if (dtype == ANNCUR_F32) {
		if (k <= 64) { if (buf) LAUNCH_RAGGED(float, true, 1); else LAUNCH_RAGGED(float, false, 1); }
		else { if (buf) LAUNCH_RAGGED(float, true, 2); else LAUNCH_RAGGED(float, false, 2); }
	} else {
		if (k <= 64) { if (buf) LAUNCH_RAGGED(uint16_t, true, 1); else LAUNCH_RAGGED(uint16_t, false, 1); }
		else { if (buf) LAUNCH_RAGGED(uint16_t, true, 2); else LAUNCH_RAGGED(uint16_t, false, 2); }
	}
#undef LAUNCH_RAGGED
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

/* a2 folded into a8's scan: see ScanGather.  Tables for anncur_rowwise_topk_gather from the ascending, distinct anchor columns. */
extern "C" int anncur_gather_tables(const int32_t *col_idx, int32_t n_idx, int64_t I, int dtype, uint32_t *vec_tab, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(dtype) && col_idx && vec_tab && n_idx >= 1 && n_idx <= 65535 && I >= 1, ANNCUR_E_INVALID, "gather_tables: bad arguments");
	const int vec = dtype == ANNCUR_F32 ? 4 : 8;
	const int64_t n_vec = ceil_div64(I, vec);
	hipLaunchKernelGGL(gather_tables_kernel, dim3((unsigned)ceil_div64(n_vec, 256)), dim3(256), 0, (hipStream_t)stream, col_idx, (int)n_idx, n_vec, vec, vec_tab);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

/* anncur_rowwise_topk (k <= 128) + cq[q, j] = A[q, col_idx[j]] in the same pass over A.  A 16-byte aligned with 16-byte aligned rows,
 * col_idx ascending and distinct in [0, I), tables from anncur_gather_tables for the same (col_idx, I, dtype); cq has A's element type. */
extern "C" int anncur_rowwise_topk_gather(const void *A, int dtype, int64_t Q, int64_t I, int64_t lda, int32_t k, float *out_val, int32_t *out_idx,
										  const int32_t *col_idx, int32_t n_idx, const uint32_t *vec_tab, void *cq, int64_t ldo, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(dtype), ANNCUR_E_INVALID, "rowwise_topk_gather: bad dtype %d", dtype);
	ANNCUR_REQUIRE(Q >= 0 && I >= 1 && lda >= I && Q < (int64_t)0x7fffffff && I < (int64_t)0x7fffffff, ANNCUR_E_INVALID, "rowwise_topk_gather: bad shape");
	ANNCUR_REQUIRE(k >= 1 && k <= WSEL_K && k <= I, ANNCUR_E_UNSUPPORTED, "rowwise_topk_gather: k=%d outside the wave-level scan (1..%d)", k, WSEL_K);
	ANNCUR_REQUIRE(A && out_val && out_idx && col_idx && vec_tab && cq, ANNCUR_E_INVALID, "rowwise_topk_gather: null pointer");
	ANNCUR_REQUIRE(n_idx >= 1 && n_idx <= 65535 && ldo >= n_idx, ANNCUR_E_INVALID, "rowwise_topk_gather: bad anchor count / output pitch");
	ANNCUR_REQUIRE(((uintptr_t)A % 16) == 0 && (Q == 1 || ((size_t)lda * dtype_size(dtype)) % 16 == 0), ANNCUR_E_UNSUPPORTED,
				   "rowwise_topk_gather: A and its rows must be 16-byte aligned (use anncur_gather_cols + anncur_rowwise_topk)");
	if (Q == 0) return ANNCUR_OK;
	hipStream_t st = (hipStream_t)stream;
	const size_t lds = 4 * (size_t)WaveSelLayout<WS_CAP>::BYTES;
	const unsigned grid = (unsigned)ceil_div64(Q, 4);
	const uint32_t trig = ws_trigger((uint32_t)k);
	ScanGather gt{vec_tab, col_idx, n_idx, cq, ldo};
	if (dtype == ANNCUR_F32)
		hipLaunchKernelGGL((rowwise_topk_wave_kernel<float, true, false>), dim3(grid), dim3(256), lds, st, (const float *)A, Q, I, lda, (uint32_t)k, trig, out_val, out_idx, gt);
	else
		hipLaunchKernelGGL((rowwise_topk_wave_kernel<uint16_t, true, false>), dim3(grid), dim3(256), lds, st, (const uint16_t *)A, Q, I, lda, (uint32_t)k, trig, out_val, out_idx, gt);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_rerank(const void *A, int dtype, int64_t Q, int64_t I, int64_t lda, const int32_t *approx_idx,
							 int64_t ld_idx, int32_t k_retvr, int32_t k_out, float *rerank_val, int32_t *rerank_idx, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(dtype), ANNCUR_E_INVALID, "rerank: bad dtype %d", dtype);
	ANNCUR_REQUIRE(Q >= 0 && I >= 1 && lda >= I && Q < (int64_t)0x7fffffff, ANNCUR_E_INVALID, "rerank: bad shape");
	ANNCUR_REQUIRE(k_retvr >= 1 && k_retvr <= ANNCUR_MAX_TOPK && k_out >= 1 && k_out <= k_retvr, ANNCUR_E_INVALID,
				   "rerank: need 1 <= k_out <= k_retvr <= %d (got %d, %d)", ANNCUR_MAX_TOPK, k_out, k_retvr);
	ANNCUR_REQUIRE(ld_idx >= k_retvr, ANNCUR_E_INVALID, "rerank: ld_idx < k_retvr");
	ANNCUR_REQUIRE(A && approx_idx && rerank_val && rerank_idx, ANNCUR_E_INVALID, "rerank: null pointer");
	if (Q == 0) return ANNCUR_OK;
	hipStream_t st = (hipStream_t)stream;
	const int kc = kmax_class(k_out);
#define LAUNCH_RERANK(T, KM)                                                                                    \
	do {                                                                                                        \
		{ const int rc_ = anncur_ensure_dyn_lds((const void *)rerank_kernel<T, KM>, (int)SelCfg<KM>::LDS_BYTES); if (rc_ != ANNCUR_OK) return rc_; } \
		hipLaunchKernelGGL((rerank_kernel<T, KM>), dim3((unsigned)Q), dim3(SEL_THREADS), SelCfg<KM>::LDS_BYTES, st, \
						   (const T *)A, I, lda, approx_idx, ld_idx, (uint32_t)k_retvr, (uint32_t)k_out, rerank_val, rerank_idx); \
	} while (0)
	if (dtype == ANNCUR_F32) {
		if (kc == 128) LAUNCH_RERANK(float, 128); else if (kc == 512) LAUNCH_RERANK(float, 512); else LAUNCH_RERANK(float, 2048);
	} else {
		if (kc == 128) LAUNCH_RERANK(uint16_t, 128); else if (kc == 512) LAUNCH_RERANK(uint16_t, 512); else LAUNCH_RERANK(uint16_t, 2048);
	}
#undef LAUNCH_RERANK
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_overlap_counts(const int32_t *a, int32_t la, const int32_t *b, int32_t lb, int64_t Q,
									 const int32_t *ka, const int32_t *kb, int32_t n_pairs, int32_t *common, void *stream) {
	ANNCUR_REQUIRE(a && b && ka && kb && common, ANNCUR_E_INVALID, "overlap_counts: null pointer");
	ANNCUR_REQUIRE(la >= 1 && lb >= 1 && la <= 4096 && lb <= 4096, ANNCUR_E_INVALID, "overlap_counts: list lengths must be in 1..4096");
	ANNCUR_REQUIRE(n_pairs >= 1 && n_pairs <= OVL_MAX_PAIRS, ANNCUR_E_INVALID, "overlap_counts: n_pairs must be in 1..%d", OVL_MAX_PAIRS);
	ANNCUR_REQUIRE(Q >= 0 && Q < (int64_t)0x7fffffff, ANNCUR_E_INVALID, "overlap_counts: bad Q");
	OvlPairs pairs;
	for (int p = 0; p < n_pairs; ++p) {
		ANNCUR_REQUIRE(ka[p] >= 0 && ka[p] <= la && kb[p] >= 0 && kb[p] <= lb, ANNCUR_E_INVALID,
					   "overlap_counts: pair %d prefix lengths (%d,%d) exceed list lengths (%d,%d)", p, ka[p], kb[p], la, lb);
		pairs.ka[p] = ka[p];
		pairs.kb[p] = kb[p];
	}
	if (Q == 0) return ANNCUR_OK;
	if (la <= 2 * WAVE && lb <= 2 * WAVE) {
		hipLaunchKernelGGL(overlap_wave_kernel, dim3((unsigned)ceil_div64(Q, 4)), dim3(256), 0, (hipStream_t)stream, a, la, b, lb, Q, pairs,
						   n_pairs, common);
		ANNCUR_LAUNCH_OK();
		return ANNCUR_OK;
	}
	const size_t lds = (size_t)(la + lb + n_pairs) * 4;
	hipLaunchKernelGGL(overlap_kernel, dim3((unsigned)Q), dim3(256), lds, (hipStream_t)stream, a, la, b, lb, Q, pairs, n_pairs, common);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_gather_cols(const void *A, int dtype, int64_t n_rows, int64_t n_cols, int64_t lda,
								  const int32_t *col_idx, int32_t n_idx, void *out, int dst_dtype, int64_t ldo, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(dtype) && dtype_ok(dst_dtype), ANNCUR_E_INVALID, "gather_cols: bad dtype");
	ANNCUR_REQUIRE(n_rows >= 0 && n_cols >= 1 && lda >= n_cols && n_idx >= 0 && ldo >= n_idx && n_rows < (int64_t)0x7fffffff,
				   ANNCUR_E_INVALID, "gather_cols: bad shape");
	if (n_rows == 0 || n_idx == 0) return ANNCUR_OK;  // (empty tensors have null data pointers: nothing to do comes first)
	ANNCUR_REQUIRE(A && col_idx && out, ANNCUR_E_INVALID, "gather_cols: null pointer");
	hipStream_t st = (hipStream_t)stream;
	const int rc = dispatch2(dtype, dst_dtype, [&](auto *s, auto *d) {
		using TS = std::remove_cv_t<std::remove_pointer_t<decltype(s)>>;
		using TD = std::remove_pointer_t<decltype(d)>;
		hipLaunchKernelGGL((gather_cols_kernel<TS, TD>), dim3((unsigned)ceil_div64(n_rows, 4)), dim3(256), 0, st, (const TS *)A, n_rows,
						   n_cols, lda, col_idx, n_idx, (TD *)out, ldo);
		return 0;
	});
	(void)rc;
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_gather_rows(const void *A, int dtype, int64_t n_rows, int64_t n_cols, int64_t lda,
								  const int32_t *row_idx, int32_t n_idx, void *out, int dst_dtype, int64_t ldo, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(dtype) && dtype_ok(dst_dtype), ANNCUR_E_INVALID, "gather_rows: bad dtype");
	ANNCUR_REQUIRE(n_rows >= 1 && n_cols >= 0 && lda >= n_cols && n_idx >= 0 && ldo >= n_cols && n_idx <= 65535,
				   ANNCUR_E_INVALID, "gather_rows: bad shape (n_idx <= 65535)");
	if (n_cols == 0 || n_idx == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(A && row_idx && out, ANNCUR_E_INVALID, "gather_rows: null pointer");
	hipStream_t st = (hipStream_t)stream;
	dispatch2(dtype, dst_dtype, [&](auto *s, auto *d) {
		using TS = std::remove_cv_t<std::remove_pointer_t<decltype(s)>>;
		using TD = std::remove_pointer_t<decltype(d)>;
		hipLaunchKernelGGL((gather_rows_kernel<TS, TD>), dim3((unsigned)ceil_div64(n_cols, 256), (unsigned)n_idx), dim3(256), 0, st,
						   (const TS *)A, n_rows, n_cols, lda, row_idx, (TD *)out, ldo);
		return 0;
	});
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_convert(const void *src, int src_dtype, int64_t lds_, void *dst, int dst_dtype, int64_t ldd,
							  int64_t n_rows, int64_t n_cols, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(src_dtype) && dtype_ok(dst_dtype), ANNCUR_E_INVALID, "convert: bad dtype");
	ANNCUR_REQUIRE(n_rows >= 0 && n_cols >= 0 && lds_ >= n_cols && ldd >= n_cols && n_rows <= 65535 * 1024LL, ANNCUR_E_INVALID, "convert: bad shape");
	if (n_rows == 0 || n_cols == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(src && dst, ANNCUR_E_INVALID, "convert: null pointer");
	hipStream_t st = (hipStream_t)stream;
	// grid.y is limited to 65535: loop over row chunks
	for (int64_t r0 = 0; r0 < n_rows; r0 += 65535) {
		const int64_t nr = (n_rows - r0 < 65535) ? (n_rows - r0) : 65535;
		dispatch2(src_dtype, dst_dtype, [&](auto *s, auto *d) {
			using TS = std::remove_cv_t<std::remove_pointer_t<decltype(s)>>;
			using TD = std::remove_pointer_t<decltype(d)>;
			hipLaunchKernelGGL((convert_kernel<TS, TD>), dim3((unsigned)ceil_div64(n_cols, 256), (unsigned)nr), dim3(256), 0, st,
							   (const TS *)src + r0 * lds_, lds_, (TD *)dst + r0 * ldd, ldd, nr, n_cols);
			return 0;
		});
	}
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_copy_bytes(const void *src, void *dst, size_t nbytes, void *stream) {
	if (nbytes == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(src && dst, ANNCUR_E_INVALID, "copy_bytes: null pointer");
	ANNCUR_REQUIRE(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, ANNCUR_E_INVALID, "copy_bytes: pointers must be 16-byte aligned");
	const size_t n16 = nbytes / 16;
	const int n_tail = (int)(nbytes % 16);
	const size_t blocks = (n16 + 255) / 256;
	ANNCUR_REQUIRE(blocks < (size_t)0x7fffffff, ANNCUR_E_INVALID, "copy_bytes: too large");
	hipLaunchKernelGGL(copy_bytes_kernel, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, (hipStream_t)stream, (const uint4 *)src, (uint4 *)dst, n16,
					   (const unsigned char *)src + n16 * 16, (unsigned char *)dst + n16 * 16, n_tail);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}
