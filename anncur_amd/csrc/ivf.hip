// IVF-flat inner-product index on the GPU: the branch of build_flat_or_ivff_index that the reference takes above 11 000 vectors
//   nlist = floor(sqrt(n)); nprobe = floor(sqrt(nlist) * probe_mult_factor); faiss.IndexIVFFlat(IndexFlatIP(d), d, nlist,
//   METRIC_INNER_PRODUCT); train / add / search                                   models/nearest_nbr.py:40-52
// FAISS is a third-party dependency the reference neither vendors nor pins (parity unpinned): this restates the published
// algorithm -- k-means coarse quantiser (assignment by maximum inner product, centroid = mean of its points), inverted lists,
// search = exact inner products inside the nprobe best lists -- and is judged on recall against the exact search.
// The dense steps (points x centroids, queries x centroids + top-nprobe) run on anncur_gemm / anncur_rowwise_topk; this file holds
// what is particular to the inverted file: list construction, centroid update, list scan.
#include "select.hpp"

using namespace anncur;

namespace {

// counts[l] = #{i : assign[i] == l}; one workgroup per list, a plain pass over the assignment (n * nlist int reads in all:
// n = 1e6, nlist = 1000 -> 4 GB out of L2; deterministic, no atomics)
__global__ __launch_bounds__(256) void ivf_count_kernel(const int32_t *__restrict__ assign, int64_t n, int32_t *__restrict__ counts) {
	const int32_t l = blockIdx.x;
	uint32_t c = 0;
	for (int64_t i = threadIdx.x; i < n; i += 256) c += assign[i] == l;
	for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
	__shared__ uint32_t part[4];
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
	__syncthreads();
	if (threadIdx.x == 0) counts[l] = (int32_t)(part[0] + part[1] + part[2] + part[3]);
}

// offsets[0..nlist] = exclusive prefix sum of counts (single workgroup; nlist is ~sqrt(n))
__global__ __launch_bounds__(256) void ivf_scan_counts_kernel(const int32_t *__restrict__ counts, int32_t nlist, int32_t *__restrict__ offsets) {
	__shared__ int32_t carry;
	__shared__ int32_t wsum[4];
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int32_t b = 0; b < nlist; b += 256) {
		const int32_t i = b + threadIdx.x;
		const int32_t c = i < nlist ? counts[i] : 0;
		int32_t inc = c;
		for (int d = 1; d < WAVE; d <<= 1) {
			const int32_t t = __shfl_up(inc, d);
			if (lane_id() >= d) inc += t;
		}
		if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = inc;
		__syncthreads();
		int32_t base = carry;
		for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
		if (i < nlist) offsets[i] = base + inc - c;
		__syncthreads();
		if (threadIdx.x == 255) carry = base + inc;
		__syncthreads();
	}
	if (threadIdx.x == 0) offsets[nlist] = carry;
}

// ids[offsets[l] ..] = the points of list l in ascending id order (stable).  One workgroup per list; each of its four waves owns a
// contiguous quarter of the assignment: it counts its hits, ONE barrier publishes the four counts, then it writes its hits in order
// behind the earlier quarters' -- no barrier inside the loops (round 2 had two per 256 points: 0.15 ms per call at n = 100 000,
// 1.5 ms at n = 10^6; the assignment is read from L2 either way).
__global__ __launch_bounds__(256) void ivf_fill_kernel(const int32_t *__restrict__ assign, int64_t n, const int32_t *__restrict__ offsets,
													   int32_t *__restrict__ ids) {
	const int32_t l = blockIdx.x;
	__shared__ uint32_t wcnt[4];
	const int wave = threadIdx.x >> 6, lane = lane_id();
	const int64_t per = ((n + 3) / 4 + WAVE - 1) / WAVE * WAVE;   // quarter length, a multiple of 64: a wave step never straddles quarters
	const int64_t beg = per * wave < n ? per * wave : n, end = beg + per < n ? beg + per : n;
	uint32_t c = 0;
	for (int64_t i = beg + lane; i < end; i += WAVE) c += assign[i] == l;
	for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
	if (lane == 0) wcnt[wave] = c;
	__syncthreads();
	uint32_t pos = (uint32_t)offsets[l];
	for (int w = 0; w < wave; ++w) pos += wcnt[w];
	for (int64_t b = beg; b < end; b += WAVE) {
		const int64_t i = b + lane;
		const bool hit = i < end && assign[i] == l;
		const unsigned long long m = __ballot(hit);
		if (hit) ids[pos + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (int32_t)i;
		pos += (uint32_t)__popcll(m);
	}
}

// centroid[l] = mean of the rows of list l of Xs (vectors stored in list order), summed in list order (deterministic);
// an empty list keeps its centroid.
__global__ __launch_bounds__(256) void ivf_means_kernel(const float *__restrict__ Xs, int64_t ldx, int32_t d, const int32_t *__restrict__ offsets,
														float *__restrict__ C, int64_t ldc) {
	const int32_t l = blockIdx.x;
	const int32_t beg = offsets[l], end = offsets[l + 1];
	if (end == beg) return;
	const float inv = 1.0f / (float)(end - beg);
	for (int32_t c = threadIdx.x; c < d; c += 256) {
		float s = 0.f;
		for (int32_t i = beg; i < end; ++i) s += Xs[(int64_t)i * ldx + c];
		C[(int64_t)l * ldc + c] = s * inv;
	}
}

// rows of M rescaled to unit L2 norm in place (a zero row stays): FAISS' fvec_renorm_L2, the "spherical" k-means step that IndexIVF switches
// on for METRIC_INNER_PRODUCT (cp.spherical = true).  One workgroup per row.
__global__ __launch_bounds__(256) void renorm_rows_kernel(float *__restrict__ M, int64_t n_cols, int64_t ld) {
	float *row = M + (int64_t)blockIdx.x * ld;
	float s = 0.f;
	for (int64_t c = threadIdx.x; c < n_cols; c += 256) { const float v = row[c]; s = fmaf(v, v, s); }
	for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
	__shared__ float part[4];
	if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
	__syncthreads();
	const float tot = (part[0] + part[1]) + (part[2] + part[3]);
	if (!(tot > 0.f)) return;
	const float inv = 1.0f / sqrtf(tot);
	for (int64_t c = threadIdx.x; c < n_cols; c += 256) row[c] *= inv;
}

// One workgroup per query: exact inner products with every vector of its nprobe lists, streamed through the workgroup-level
// selector.  Vectors are stored in list order with a row pitch that is a multiple of 16 floats (zero padded), the query likewise:
// four lanes share one vector (each reads 16 bytes per step: 16 vectors x 64 contiguous bytes per wave-instruction).
template <int KMAX>
__global__ __launch_bounds__(SEL_THREADS) void ivf_scan_kernel(const float *__restrict__ Xs, int64_t ldx, int32_t dp, const int32_t *__restrict__ offsets,
															   const int32_t *__restrict__ ids, const float *__restrict__ Q, int64_t ldq,
															   const int32_t *__restrict__ probe, int32_t nprobe, uint32_t k,
															   float *__restrict__ out_val, int32_t *__restrict__ out_idx) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const SelState s = sel_carve<KMAX>(smem);
	float *qs = reinterpret_cast<float *>(smem + SelCfg<KMAX>::LDS_BYTES);  // [dp]
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 3, slot = lane >> 2;
	const int64_t qi = blockIdx.x;
	for (int c = tid; c < dp; c += SEL_THREADS) qs[c] = Q[qi * ldq + c];
	sel_init(s);  // (barrier inside)
	float tau = -INFINITY;
	uint64_t tau_key = 0;
	int since = 0;
	for (int32_t p = 0; p < nprobe; ++p) {
		const int32_t l = probe[qi * nprobe + p];
		if (l < 0) continue;  // (uniform)
		const int32_t beg = offsets[l], end = offsets[l + 1];
		for (int32_t v0 = beg; v0 < end; v0 += 64) {  // 4 waves x 16 vectors
			const int32_t v = v0 + wave * 16 + slot;
			const bool in = v < end;
			const float4 *row = reinterpret_cast<const float4 *>(Xs + (int64_t)(in ? v : beg) * ldx) + sub;
			const float4 *qv = reinterpret_cast<const float4 *>(qs) + sub;
			float acc = 0.f;
			for (int c = 0; c < dp / 16; ++c) {
				const float4 x = row[4 * c], y = qv[4 * c];
				acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
			}
			acc += __shfl_xor(acc, 1);
			acc += __shfl_xor(acc, 2);
			const bool offer = in && sub == 0 && !(acc != acc);  // NaN scores are never selected
			sel_offer(s, offer, acc, offer ? (uint32_t)ids[v] : 0u, tau, tau_key);
			if (++since == 32) {  // <= 2048 pushes since the last check
				since = 0;
				sel_maybe_compact<KMAX>(s, k, tau, tau_key);
			}
		}
	}
	sel_finish<KMAX>(s, k, out_val + qi * (int64_t)k, out_idx + qi * (int64_t)k);
}

// ---- batched search: queries grouped by probed list, scored on the fp32 matrix cores --------------------------------------------
// The per-query kernel above reads every probed list once PER QUERY (nq x nprobe x list bytes through L2: 167 GB for 10^4 queries on
// 10^5 x 768 vectors, 28 ms -- slower than the exact flat search of the same queries).  Batched, the (query, probe slot) pairs are sorted
// by list (anncur_ivf_build_lists on the flattened probe array) and each list is one small GEMM: [pairs of the list x d] . [d x vectors
// of the list], 64 x 64 output tiles from a host-built worklist, products and sums exact fp32 (v_mfma_f32_32x32x2_f32).  Scores go to
// S[pair][position in its list] (row pitch lmax, the caller pre-fills -inf), i.e. query q's row of S holds its nprobe lists side by
// side: anncur_rowwise_topk over [nq x nprobe * lmax] + ivf_map_ids_kernel finish the search.
constexpr int GT = 64, GK = 16, GP = GT + 1;   // tile edge, k-tile depth, LDS pitch (k-major tiles as in gemm.hip)

// Device-built tile worklist (round 4).  Round 3 copied the pairs-per-list counts to the host, enumerated the (list, query tile, vector tile)
// triples there and copied them back: a device synchronisation in the middle of every search.  Now tile_start[l] = number of tiles of the
// lists before l (ivf_tile_starts_kernel: one workgroup, nlist is ~sqrt(n)); the GEMM is launched with an upper bound on the tile count
// that the host knows without looking at the counts, and workgroup b finds its triple by a binary search (b >= tile_start[nlist]: exit).
__global__ __launch_bounds__(256) void ivf_tile_starts_kernel(const int32_t *__restrict__ pair_off, const int32_t *__restrict__ offsets, int32_t nlist,
															   int32_t *__restrict__ tile_start) {
	__shared__ int32_t carry;
	__shared__ int32_t wsum[4];
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int32_t b = 0; b < nlist; b += 256) {
		const int32_t l = b + threadIdx.x;
		int32_t c = 0;
		if (l < nlist) {
			const int32_t qt = (pair_off[l + 1] - pair_off[l] + GT - 1) / GT, vt = (offsets[l + 1] - offsets[l] + GT - 1) / GT;
			c = qt * vt;
		}
		int32_t inc = c;
		for (int d = 1; d < WAVE; d <<= 1) {
			const int32_t t = __shfl_up(inc, d);
			if (lane_id() >= d) inc += t;
		}
		if (lane_id() == WAVE - 1) wsum[threadIdx.x >> 6] = inc;
		__syncthreads();
		int32_t base = carry;
		for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
		if (l < nlist) tile_start[l] = base + inc - c;
		__syncthreads();
		if (threadIdx.x == 255) carry = base + inc;
		__syncthreads();
	}
	if (threadIdx.x == 0) tile_start[nlist] = carry;
}
// workgroup b -> (list, query tile, vector tile) from the host-built triples (nlist = 0) or the device-built starts; false: nothing to do
__device__ __forceinline__ bool ivf_tile_of(const int32_t *__restrict__ tiles, int32_t nlist, const int32_t *__restrict__ offsets, int32_t &l, int32_t &qt, int32_t &vt) {
	const int32_t b = blockIdx.x;
	if (nlist <= 0) { l = tiles[3 * b]; qt = tiles[3 * b + 1]; vt = tiles[3 * b + 2]; return true; }
	if (b >= tiles[nlist]) return false;
	int32_t lo = 0, hi = nlist;   // largest l with tiles[l] <= b (empty lists repeat their start: the search lands past them)
	while (hi - lo > 1) {
		const int32_t mid = (lo + hi) >> 1;
		if (tiles[mid] <= b) lo = mid; else hi = mid;
	}
	l = lo;
	const int32_t vcnt = (offsets[l + 1] - offsets[l] + GT - 1) / GT, within = b - tiles[l];
	qt = within / vcnt; vt = within - qt * vcnt;
	return true;
}
typedef __attribute__((ext_vector_type(16))) float f32x16g;

__global__ __launch_bounds__(256) void ivf_group_scores_kernel(const float *__restrict__ Xs, int64_t ldx, int32_t dp, const int32_t *__restrict__ offsets,
																const float *__restrict__ Q, int64_t ldq, int32_t nprobe, const int32_t *__restrict__ pair_ids,
																const int32_t *__restrict__ pair_off, const int32_t *__restrict__ tiles, int32_t nlist, int64_t lmax,
																float *__restrict__ S) {
	__shared__ float As[GK * GP], Bs[GK * GP];
	__shared__ int32_t arow[GT];   // query row of each pair of the tile (-1 past the list's pairs)
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
	int32_t l, qt, vt;
	if (!ivf_tile_of(tiles, nlist, offsets, l, qt, vt)) return;   // (uniform: before any barrier)
	const int32_t p0 = pair_off[l] + qt * GT, p_end = pair_off[l + 1];
	const int32_t v0 = offsets[l] + vt * GT, v_end = offsets[l + 1];
	if (tid < GT) arow[tid] = p0 + tid < p_end ? pair_ids[p0 + tid] / nprobe : -1;
	__syncthreads();
	// thread -> (row m, k) of the operand tiles: k fastest (16 consecutive floats of a row per 16 lanes)
	const int kk = tid & 15, mm = tid >> 4;
	f32x16g acc;
#pragma unroll
	for (int e = 0; e < 16; ++e) acc[e] = 0.f;
	float ra[4], rb[4];
	auto load = [&](int k0) {
#pragma unroll
		for (int p = 0; p < 4; ++p) {
			const int m = mm + 16 * p;
			const int32_t qr = arow[m];
			ra[p] = (qr >= 0 && k0 + kk < dp) ? Q[(int64_t)qr * ldq + k0 + kk] : 0.f;
			rb[p] = (v0 + m < v_end && k0 + kk < dp) ? Xs[(int64_t)(v0 + m) * ldx + k0 + kk] : 0.f;
		}
	};
	load(0);
	for (int k0 = 0; k0 < dp; k0 += GK) {
#pragma unroll
		for (int p = 0; p < 4; ++p) { As[kk * GP + mm + 16 * p] = ra[p]; Bs[kk * GP + mm + 16 * p] = rb[p]; }
		__syncthreads();
		if (k0 + GK < dp) load(k0 + GK);
#pragma unroll
		for (int ks = 0; ks < GK / 2; ++ks)
			acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(2 * ks + h) * GP + wm * 32 + r], Bs[(2 * ks + h) * GP + wn * 32 + r], acc, 0, 0, 0);
		__syncthreads();
	}
	// C/D layout: col = lane & 31 (vector), row = (e & 3) + 8 (e >> 2) + 4 h (pair)
#pragma unroll
	for (int e = 0; e < 16; ++e) {
		const int m = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, n = wn * 32 + r;
		if (p0 + m < p_end && v0 + n < v_end) S[(int64_t)pair_ids[p0 + m] * lmax + (vt * GT + n)] = acc[e];
	}
}

// The same GEMM per list on the bf16 matrix cores (round 4: the index built with dtype = "bf16" keeps a bf16 copy of the list-ordered
// vectors; the queries are rounded to bf16 per call): 64 x 64 output tiles as above, k-tiles of 32, v_mfma_f32_32x32x16_bf16 with fp32
// accumulation.  Staging: every thread moves ONE 16-byte chunk (8 bf16) of a pair's query row and one of a vector per k-tile -- 64 rows
// x 4 chunks -- into a [64 rows][4 chunks] LDS image whose chunk index is XOR-ed with (row >> 2) & 3 (rows are 64 bytes apart: a
// ds_read_b128 of 16 rows x one chunk column then touches every bank group once); the next k-tile's chunks are in flight during the MFMAs.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8g;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4g;
__global__ __launch_bounds__(256) void ivf_group_scores_bf16_kernel(const uint16_t *__restrict__ Xs, int64_t ldx, int32_t dp, const int32_t *__restrict__ offsets,
																	 const uint16_t *__restrict__ Q, int64_t ldq, int32_t nprobe, const int32_t *__restrict__ pair_ids,
																	 const int32_t *__restrict__ pair_off, const int32_t *__restrict__ tiles, int32_t nlist, int64_t lmax,
																	 float *__restrict__ S) {
	__shared__ __attribute__((aligned(16))) u32x4g As[GT * 4], Bs[GT * 4];
	__shared__ int32_t arow[GT];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
	int32_t l, qt, vt;
	if (!ivf_tile_of(tiles, nlist, offsets, l, qt, vt)) return;   // (uniform: before any barrier)
	const int32_t p0 = pair_off[l] + qt * GT, p_end = pair_off[l + 1];
	const int32_t v0 = offsets[l] + vt * GT, v_end = offsets[l + 1];
	if (tid < GT) arow[tid] = p0 + tid < p_end ? pair_ids[p0 + tid] / nprobe : -1;
	__syncthreads();
	const int row = tid >> 2, ch = tid & 3;                 // this thread's chunk of the operand tiles
	const int slot = row * 4 + (ch ^ ((row >> 2) & 3));
	const int32_t qr = arow[row];
	const uint16_t *ap = qr >= 0 ? Q + (int64_t)qr * ldq + ch * 8 : nullptr;
	const uint16_t *bp = v0 + row < v_end ? Xs + (int64_t)(v0 + row) * ldx + ch * 8 : nullptr;
	f32x16g acc;
#pragma unroll
	for (int e = 0; e < 16; ++e) acc[e] = 0.f;
	const u32x4g zero = {0u, 0u, 0u, 0u};
	u32x4g ra, rb;
	auto load = [&](int k0) {
		const bool in = k0 + ch * 8 < dp;                   // (dp is a multiple of 16: a chunk is inside the row or past it)
		ra = (ap && in) ? *reinterpret_cast<const u32x4g *>(ap + k0) : zero;
		rb = (bp && in) ? *reinterpret_cast<const u32x4g *>(bp + k0) : zero;
	};
	load(0);
	for (int k0 = 0; k0 < dp; k0 += 32) {
		As[slot] = ra; Bs[slot] = rb;
		__syncthreads();
		if (k0 + 32 < dp) load(k0 + 32);
#pragma unroll
		for (int ks = 0; ks < 2; ++ks) {
			const int ra_row = wm * 32 + r, rb_row = wn * 32 + r, c = 2 * ks + h;
			const bf16x8g a = __builtin_bit_cast(bf16x8g, As[ra_row * 4 + (c ^ ((ra_row >> 2) & 3))]);
			const bf16x8g b = __builtin_bit_cast(bf16x8g, Bs[rb_row * 4 + (c ^ ((rb_row >> 2) & 3))]);
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
		}
		__syncthreads();
	}
	// C/D layout: col = lane & 31 (vector), row = (e & 3) + 8 (e >> 2) + 4 h (pair)
#pragma unroll
	for (int e = 0; e < 16; ++e) {
		const int m = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, n = wn * 32 + r;
		if (p0 + m < p_end && v0 + n < v_end) S[(int64_t)pair_ids[p0 + m] * lmax + (vt * GT + n)] = acc[e];
	}
}

// column of the [nq x nprobe * lmax] score matrix -> id of the vector: slot = col / lmax, position = col % lmax in list probe[q, slot]
__global__ __launch_bounds__(256) void ivf_map_ids_kernel(const int32_t *__restrict__ col, int64_t n, int32_t k, int64_t lmax, const int32_t *__restrict__ probe,
														   int32_t nprobe, const int32_t *__restrict__ offsets, const int32_t *__restrict__ ids, const float *__restrict__ val,
														   int32_t *__restrict__ out) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const int32_t c = col[i];
	int32_t id = -1;
	if (c >= 0 && val[i] > -INFINITY) {   // (a -inf score is padding: fewer than k vectors in the probed lists)
		const int64_t q = i / k;
		const int32_t slot = (int32_t)(c / lmax), pos = (int32_t)(c % lmax);
		const int32_t l = probe[q * nprobe + slot];
		if (l >= 0) id = ids[offsets[l] + pos];
	}
	out[i] = id;
}


// ================================================================================================================================
// Round 5: the batched search as ONE call (anncur_ivf_search_grouped) -- packed score rows, a pair sort without passes per list, a
// 128 x 128-tile bf16 GEMM.  What the round-4 pipeline spent per 10^4 queries on 10^5 x 768 vectors (rocprofv3, 1.08 ms in all): pair sort
// 0.25 ms (one workgroup PER LIST passing over all pairs, twice), -inf fill of S 0.095, GEMM 0.33, scan of S 0.145 -- S is [nq x nprobe x
// lmax] there and three quarters of it padding (lists of 1..1026 vectors, 5 400 scanned per query of 17 442 columns).
//   packed rows: query q's row of S holds its probed lists back to back (coff[q, slot] = vectors in its earlier slots), row_len[q] scores
//     in all; nothing is pre-filled (rows shorter than k get -inf up to k) and the scan reads row_len[q] columns (anncur_rowwise_topk_ragged);
//   pair sort: LDS histograms per 256 queries + one global atomic per (workgroup, list present) for the counts, the same again for the
//     positions.  The order of a list's pairs depends on the atomics' timing; a pair's scores do not depend on its position;
//   tiles: tile_start over (list, 128-pair tile, 128-vector tile) for bf16 (64 x 64 for fp32), workgroup b takes tile xcd_remap(b, n_tiles):
//     the tiles of a list run on ONE XCD around the same time and share the list's vectors and its pairs' queries in that L2 (round 4:
//     consecutive tiles on eight XCDs, every operand tile from beyond L2 -- 2.7 GB per search).
// (scripts/placement_sweep.sh: ANNCUR_PLACEMENT_PAD = 1..n extra instructions in front of the tile loop -- a result that depends on code placement
//  is a missing wait state; the tile kernel's fragment reads and row loads are inline asm with counted waits)
#ifdef ANNCUR_PLACEMENT_PAD
#define IVF_PAD_HERE() asm volatile(".rept %0\n\ts_nop 0\n\t.endr" ::"n"(ANNCUR_PLACEMENT_PAD) : "memory")
#else
#define IVF_PAD_HERE() do { } while (0)
#endif
constexpr int IVF_MAX_NLIST_LDS = 8192;   // 2 x nlist words of LDS in the sort kernels
constexpr uint32_t IVF_NO_SLOT = 0xffffffffu;   // coff of a (query, probe slot) pair that takes no part in the search
constexpr int IVF_RAGGED_MAX_K = 128;    // anncur_rowwise_topk_ragged: the wave-per-row scan (WSEL_K of wave_select.hpp)

__device__ __forceinline__ int ivf_xcd_remap(int b, int n) {   // work ids of one XCD contiguous (blocks b and b + 8 share an XCD); a bijection on [0, n)
	const int q = n >> 3, r = n & 7, x = b & 7, l = b >> 3;
	return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}

// coff / row_len / short-row padding and the pairs-per-list counts (counts zeroed by the caller).  One thread per query.
__global__ __launch_bounds__(256) void ivf_layout_kernel(const int32_t *__restrict__ probe, int64_t nq, int32_t nprobe, const int32_t *__restrict__ offsets,
														  int32_t nlist, int32_t k, int64_t pitch, uint32_t *__restrict__ coff, int32_t *__restrict__ row_len,
														  int32_t *__restrict__ counts, float *__restrict__ S) {
	extern __shared__ uint32_t ivf_hist[];
	for (int i = threadIdx.x; i < nlist; i += 256) ivf_hist[i] = 0u;
	__syncthreads();
	const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (q < nq) {
		uint32_t run = 0;
		for (int s = 0; s < nprobe; ++s) {
			const int32_t l = probe[q * nprobe + s];
			uint32_t at = IVF_NO_SLOT;   // a slot that is skipped: list id out of range, or the list does not fit the row any more (a pitch below the contract)
			if (l >= 0 && l < nlist) {
				const uint32_t size = (uint32_t)(offsets[l + 1] - offsets[l]);
				if ((uint64_t)run + size <= (uint64_t)pitch) {
					at = run;
					run += size;
					atomicAdd(&ivf_hist[l], 1u);
				}
			}
			coff[q * nprobe + s] = at;
		}
		for (uint32_t c = run; c < (uint32_t)k; ++c) S[q * pitch + c] = -INFINITY;   // fewer than k vectors in the probed lists
		row_len[q] = (int32_t)(run > (uint32_t)k ? run : (uint32_t)k);
	}
	__syncthreads();
	for (int i = threadIdx.x; i < nlist; i += 256)
		if (ivf_hist[i] != 0u) atomicAdd(&counts[i], (int32_t)ivf_hist[i]);
}

// pair_off = exclusive prefix of the pair counts (+ a copy as the scatter's cursors) and tile_start = exclusive prefix of the lists' tile
// counts, tile = T pairs x T vectors.  One workgroup.
__global__ __launch_bounds__(256) void ivf_pair_offsets_kernel(const int32_t *__restrict__ counts, const int32_t *__restrict__ offsets, int32_t nlist, int32_t T,
																int32_t *__restrict__ pair_off, int32_t *__restrict__ cursor, int32_t *__restrict__ tile_start) {
	__shared__ int32_t carry[2];
	__shared__ int32_t wsum[2][4];
	if (threadIdx.x < 2) carry[threadIdx.x] = 0;
	__syncthreads();
	for (int32_t b = 0; b < nlist; b += 256) {
		const int32_t l = b + threadIdx.x;
		int32_t c0 = 0, c1 = 0;
		if (l < nlist) {
			c0 = counts[l];
			c1 = ((c0 + T - 1) / T) * ((offsets[l + 1] - offsets[l] + T - 1) / T);
		}
		int32_t i0 = c0, i1 = c1;
		for (int d = 1; d < WAVE; d <<= 1) {
			const int32_t t0 = __shfl_up(i0, d), t1 = __shfl_up(i1, d);
			if (lane_id() >= d) { i0 += t0; i1 += t1; }
		}
		if (lane_id() == WAVE - 1) { wsum[0][threadIdx.x >> 6] = i0; wsum[1][threadIdx.x >> 6] = i1; }
		__syncthreads();
		int32_t b0 = carry[0], b1 = carry[1];
		for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) { b0 += wsum[0][w]; b1 += wsum[1][w]; }
		if (l < nlist) { pair_off[l] = b0 + i0 - c0; cursor[l] = b0 + i0 - c0; tile_start[l] = b1 + i1 - c1; }
		__syncthreads();
		if (threadIdx.x == 255) { carry[0] = b0 + i0; carry[1] = b1 + i1; }
		__syncthreads();
	}
	if (threadIdx.x == 0) { pair_off[nlist] = carry[0]; tile_start[nlist] = carry[1]; }
}

// pairs -> their lists' ranges: pair_q[pos] = query row, pair_out[pos] = element offset of the pair's scores in S (q * pitch + coff).
__global__ __launch_bounds__(256) void ivf_scatter_kernel(const int32_t *__restrict__ probe, int64_t nq, int32_t nprobe, int32_t nlist, int64_t pitch,
														   const uint32_t *__restrict__ coff, int32_t *__restrict__ cursor, int32_t *__restrict__ pair_q,
														   uint32_t *__restrict__ pair_out) {
	extern __shared__ uint32_t ivf_hist[];
	uint32_t *base = ivf_hist + nlist;
	for (int i = threadIdx.x; i < nlist; i += 256) ivf_hist[i] = 0u;
	__syncthreads();
	const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (q < nq)
		for (int s = 0; s < nprobe; ++s)
			if (coff[q * nprobe + s] != IVF_NO_SLOT) atomicAdd(&ivf_hist[probe[q * nprobe + s]], 1u);   // (a slot with an offset has a valid list id: ivf_layout_kernel)
	__syncthreads();
	for (int i = threadIdx.x; i < nlist; i += 256) {
		const uint32_t c = ivf_hist[i];
		base[i] = c != 0u ? (uint32_t)atomicAdd(&cursor[i], (int32_t)c) : 0u;
		ivf_hist[i] = 0u;
	}
	__syncthreads();
	if (q < nq)
		for (int s = 0; s < nprobe; ++s) {
			const uint32_t at = coff[q * nprobe + s];
			if (at != IVF_NO_SLOT) {
				const int32_t l = probe[q * nprobe + s];
				const uint32_t pos = base[l] + atomicAdd(&ivf_hist[l], 1u);
				pair_q[pos] = (int32_t)q;
				pair_out[pos] = (uint32_t)(q * pitch) + at;
			}
		}
}

// work id -> (list, pair tile, vector tile): every thread looks at its lists' tile ranges (one round trip; non-empty ranges are disjoint)
__device__ __forceinline__ void ivf_find_tile(const int32_t *__restrict__ tile_start, int32_t nlist, const int32_t *__restrict__ offsets, int32_t T, int32_t wid,
											   int32_t *meta, int32_t &l, int32_t &qt, int32_t &vt) {
	for (int32_t i = threadIdx.x; i < nlist; i += 256) {
		const int32_t ts = tile_start[i], te = tile_start[i + 1];
		if (ts <= wid && wid < te) { meta[0] = i; meta[1] = ts; }
	}
	__syncthreads();
	l = __builtin_amdgcn_readfirstlane(meta[0]);   // (what depends on the tile stays in scalar registers: the DMA's source bases are SGPR pairs)
	const int32_t vcnt = (offsets[l + 1] - offsets[l] + T - 1) / T, within = wid - __builtin_amdgcn_readfirstlane(meta[1]);
	qt = within / vcnt; vt = within - qt * vcnt;
}

// The round-3/4 tile kernels (64 x 64; fp32 lists, and bf16 rows whose length is not a multiple of 128) on the packed layout.
template <typename T>
__global__ __launch_bounds__(256) void ivf_tile64_packed_kernel(const T *__restrict__ Xs, int64_t ldx, int32_t dp, const int32_t *__restrict__ offsets,
																 const T *__restrict__ Q, int64_t ldq, const int32_t *__restrict__ pair_q,
																 const uint32_t *__restrict__ pair_out, const int32_t *__restrict__ pair_off,
																 const int32_t *__restrict__ tile_start, int32_t nlist, float *__restrict__ S) {
	constexpr bool F32 = sizeof(T) == 4;
	__shared__ __attribute__((aligned(16))) unsigned char tiles_lds[F32 ? 2 * GK * GP * 4 : 2 * GT * 64];
	__shared__ int32_t arow[GT];
	__shared__ uint32_t orow[GT];
	__shared__ int32_t meta[2];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
	const int32_t n_tiles = min(tile_start[nlist], (int32_t)gridDim.x);   // (the grid is the caller's bound on the tile count)
	if ((int32_t)blockIdx.x >= n_tiles) return;   // (uniform: before any barrier)
	int32_t l, qt, vt;
	ivf_find_tile(tile_start, nlist, offsets, GT, ivf_xcd_remap((int)blockIdx.x, n_tiles), meta, l, qt, vt);
	const int32_t p0 = pair_off[l] + qt * GT, p_end = pair_off[l + 1];
	const int32_t v0 = offsets[l] + vt * GT, v_end = offsets[l + 1];
	if (tid < GT) {
		const bool in = p0 + tid < p_end;
		arow[tid] = in ? pair_q[p0 + tid] : -1;
		orow[tid] = in ? pair_out[p0 + tid] : 0u;
	}
	__syncthreads();
	f32x16g acc;
#pragma unroll
	for (int e = 0; e < 16; ++e) acc[e] = 0.f;
	if constexpr (F32) {
		float *As = reinterpret_cast<float *>(tiles_lds), *Bs = As + GK * GP;
		const int kk = tid & 15, mm = tid >> 4;
		float ra[4], rb[4];
		auto load = [&](int k0) {
#pragma unroll
			for (int p = 0; p < 4; ++p) {
				const int m = mm + 16 * p;
				const int32_t qr = arow[m];
				ra[p] = (qr >= 0 && k0 + kk < dp) ? Q[(int64_t)qr * ldq + k0 + kk] : 0.f;
				rb[p] = (v0 + m < v_end && k0 + kk < dp) ? Xs[(int64_t)(v0 + m) * ldx + k0 + kk] : 0.f;
			}
		};
		load(0);
		for (int k0 = 0; k0 < dp; k0 += GK) {
#pragma unroll
			for (int p = 0; p < 4; ++p) { As[kk * GP + mm + 16 * p] = ra[p]; Bs[kk * GP + mm + 16 * p] = rb[p]; }
			__syncthreads();
			if (k0 + GK < dp) load(k0 + GK);
#pragma unroll
			for (int ks = 0; ks < GK / 2; ++ks)
				acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(2 * ks + h) * GP + wm * 32 + r], Bs[(2 * ks + h) * GP + wn * 32 + r], acc, 0, 0, 0);
			__syncthreads();
		}
	} else {
		u32x4g *As = reinterpret_cast<u32x4g *>(tiles_lds), *Bs = As + GT * 4;
		const int row = tid >> 2, ch = tid & 3;
		const int slot = row * 4 + (ch ^ ((row >> 2) & 3));
		const int32_t qr = arow[row];
		const uint16_t *ap = qr >= 0 ? reinterpret_cast<const uint16_t *>(Q) + (int64_t)qr * ldq + ch * 8 : nullptr;
		const uint16_t *bp = v0 + row < v_end ? reinterpret_cast<const uint16_t *>(Xs) + (int64_t)(v0 + row) * ldx + ch * 8 : nullptr;
		const u32x4g zero = {0u, 0u, 0u, 0u};
		u32x4g ra, rb;
		auto load = [&](int k0) {
			const bool in = k0 + ch * 8 < dp;
			ra = (ap && in) ? *reinterpret_cast<const u32x4g *>(ap + k0) : zero;
			rb = (bp && in) ? *reinterpret_cast<const u32x4g *>(bp + k0) : zero;
		};
		load(0);
		for (int k0 = 0; k0 < dp; k0 += 32) {
			As[slot] = ra; Bs[slot] = rb;
			__syncthreads();
			if (k0 + 32 < dp) load(k0 + 32);
#pragma unroll
			for (int ks = 0; ks < 2; ++ks) {
				const int ra_row = wm * 32 + r, rb_row = wn * 32 + r, c = 2 * ks + h;
				const bf16x8g a = __builtin_bit_cast(bf16x8g, As[ra_row * 4 + (c ^ ((ra_row >> 2) & 3))]);
				const bf16x8g b = __builtin_bit_cast(bf16x8g, Bs[rb_row * 4 + (c ^ ((rb_row >> 2) & 3))]);
				acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
			}
			__syncthreads();
		}
	}
	// C/D layout: col = lane & 31 (vector), row = (e & 3) + 8 (e >> 2) + 4 h (pair)
#pragma unroll
	for (int e = 0; e < 16; ++e) {
		const int m = wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, n = wn * 32 + r;
		if (p0 + m < p_end && v0 + n < v_end) S[(int64_t)orow[m] + (vt * GT + n)] = acc[e];
	}
}

// ---- bf16, 128 pairs x 128 vectors per tile (round 5).  Four waves, wave (wm, wn) owns pairs [64 wm, +64) x vectors [64 wn, +64): 2 x 2
// accumulators of v_mfma_f32_32x32x16_bf16; d streamed in 64-wide k-tiles through two 32 KiB LDS stages (pairs' query rows 16 KiB + vectors
// 16 KiB) filled by global_load_lds_dwordx4 -- the query rows gathered by the per-lane source address --, one barrier per k-tile, fragment
// reads as inline asm with counted waits (the scheme of score_wide.hpp; the LDS image and its swizzle are that kernel's: row-major 128-byte
// rows, 16-byte chunk index XOR (row >> 1) & 7 on the DMA's source address and again at the reads).  Two workgroups per CU.
// PERSISTENT workgroups: a tile is 12 k-tiles at d = 768 -- about as long as the chain of dependent loads that finds it (tile -> list ->
// its pairs' rows -> first operand bytes); the first version, one workgroup per tile, ran at 0.17 of the bf16 peak.  Here the tiles' (list,
// first pair, first vector, counts) are expanded to 32-byte descriptors by a kernel of their own, workgroup b walks the tiles start_x + (b >> 3),
// + gridDim / 8, ... of its XCD's contiguous share (the tiles of a list run on one XCD around the same time and meet in its L2), reads the next
// tile's descriptor and pair rows while the current one is multiplied, and the last k-tile's DMA slots fetch the next tile's first k-tile.
// Rows: dp a multiple of 128 (an even number of k-tiles: a tile begins in stage 0), 16-byte aligned; nq * ldq * 2 < 2^32.
constexpr int T128 = 128;
constexpr int T128_TILE_BYTES = 128 * 128, T128_STAGE_BYTES = 2 * T128_TILE_BYTES, T128_LDS_BYTES = 2 * T128_STAGE_BYTES;
constexpr int T128_LDS_TOTAL = T128_LDS_BYTES + 3 * 2 * T128 * 4;   // + three generations of (query row, output offset) per tile row
struct T128Frag { u32x4g a[2], b[2]; };
struct T128Desc { int32_t l, p0, p_rows, v0, v_rows, c0, pad0, pad1; };
__device__ __forceinline__ void t128_read(u32x4g &dst, uint32_t addr, int off) {   // `off` folds to an immediate after unrolling
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
#endif
}
// tile id -> descriptor (one thread per tile; tile_start lives in L2)
__global__ __launch_bounds__(256) void ivf_tile_desc_kernel(const int32_t *__restrict__ tile_start, int32_t nlist, const int32_t *__restrict__ offsets,
															 const int32_t *__restrict__ pair_off, int32_t T, int32_t max_tiles, T128Desc *__restrict__ desc) {
	const int32_t t = (int32_t)(blockIdx.x * 256 + threadIdx.x);
	if (t >= min(tile_start[nlist], max_tiles)) return;   // (max_tiles: the caller's bound = the descriptor array's length; a bound that is too small loses tiles, never memory)
	int32_t lo = 0, hi = nlist;   // largest l with tile_start[l] <= t (empty lists repeat their start: the search lands past them)
	while (hi - lo > 1) {
		const int32_t mid = (lo + hi) >> 1;
		if (tile_start[mid] <= t) lo = mid; else hi = mid;
	}
	const int32_t l = lo, vcnt = (offsets[l + 1] - offsets[l] + T - 1) / T, within = t - tile_start[l];
	const int32_t qt = within / vcnt, vt = within - qt * vcnt;
	T128Desc d;
	d.l = l;
	d.p0 = pair_off[l] + qt * T; d.p_rows = min(T, pair_off[l + 1] - d.p0);
	d.v0 = offsets[l] + vt * T; d.v_rows = min(T, offsets[l + 1] - d.v0);
	d.c0 = vt * T; d.pad0 = d.pad1 = 0;
	desc[t] = d;
}

__global__ __launch_bounds__(256, 2) void ivf_tile128_kernel(const uint16_t *__restrict__ Xs, int64_t ldx, int32_t dp, const uint16_t *__restrict__ Q, int64_t ldq,
															  const int32_t *__restrict__ pair_q, const uint32_t *__restrict__ pair_out,
															  const int32_t *__restrict__ tile_start, int32_t nlist, int32_t max_tiles, const T128Desc *__restrict__ desc,
															  float *__restrict__ S) {
	extern __shared__ __attribute__((aligned(16))) unsigned char t128_smem[];   // the two stages at offset 0, then the tiles' rows
	int32_t *rows_lds = reinterpret_cast<int32_t *>(t128_smem + T128_LDS_BYTES);   // generation g: query rows at [g * 256, +128), output offsets at [g * 256 + 128, +128)
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
	// ---- this workgroup's tiles: its XCD's contiguous share of [0, n_tiles), strided by the workgroups of the XCD (gridDim.x is a multiple of 8)
	const int32_t n_tiles = min(tile_start[nlist], max_tiles);
	const int32_t xcd = (int32_t)(blockIdx.x & 7), lw = (int32_t)(blockIdx.x >> 3), wpx = (int32_t)(gridDim.x >> 3);
	const int32_t share = n_tiles >> 3, rem = n_tiles & 7;
	const int32_t t_begin = xcd < rem ? xcd * (share + 1) : rem * (share + 1) + (xcd - rem) * share, t_end = t_begin + share + (xcd < rem ? 1 : 0);
	int32_t t = t_begin + lw;
	if (t >= t_end) return;   // (uniform: before any barrier)

	const unsigned char *abase = reinterpret_cast<const unsigned char *>(Q);
	const uint32_t lda_b = (uint32_t)(ldq * 2), ldb_b = (uint32_t)(ldx * 2);
	const int x = (r >> 1) & 7;
	const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) const char *)t128_smem;
	uint32_t fa[2][4], fb[2][4];
#pragma unroll
	for (int s = 0; s < 4; ++s) {
		const uint32_t co = (uint32_t)(((2 * s + h) ^ x) * 16);
		fa[0][s] = lds0 + (uint32_t)(wm * 64 + r) * 128u + co;
		fb[0][s] = lds0 + (uint32_t)T128_TILE_BYTES + (uint32_t)(wn * 64 + r) * 128u + co;
		fa[1][s] = fa[0][s] + (uint32_t)T128_STAGE_BYTES;
		fb[1][s] = fb[0][s] + (uint32_t)T128_STAGE_BYTES;
	}
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);
	const uint32_t lds0_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds0);

#define T128_PIECE(BASE, OFF, TILE_OFF, i)                                                                        \
	do {                                                                                                          \
		const uint32_t m0v_ = lds0_u + (uint32_t)(TILE_OFF) + (uint32_t)(wave_u * 4 + (i)) * 1024u;               \
		asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v_), "v"((OFF)[i]), "s"(BASE) : "memory", "m0"); \
	} while (0)
	// DMA of the next k-tile: two of the wave's eight 1 KiB pieces per k-step, between the fragment reads and the MFMAs of the step
#ifdef ANNCUR_V_IVF_NODMA   // (ablation build, results wrong: scripts/r5/ivf_ablation.sh)
#define T128_DEN(DEN) ((DEN) && false)
#define T128_ROWS_WAIT 0
#else
#define T128_DEN(DEN) (DEN)
#define T128_ROWS_WAIT 8   // the phase's DMA pieces, issued after the rows' loads
#endif
#define T128_DMA2(s, DA, AOFF, DB, BOFF, DD, DEN)                                                                 \
	do {                                                                                                          \
		if (T128_DEN(DEN)) {                                                                                              \
			if ((s) < 2) { T128_PIECE(DA, AOFF, DD, 2 * ((s) & 1)); T128_PIECE(DA, AOFF, DD, 2 * ((s) & 1) + 1); } \
			else { T128_PIECE(DB, BOFF, (DD) + T128_TILE_BYTES, 2 * ((s) & 1)); T128_PIECE(DB, BOFF, (DD) + T128_TILE_BYTES, 2 * ((s) & 1) + 1); } \
		}                                                                                                         \
	} while (0)
#define T128_LOAD(F, STAGE, s)                                                                                    \
	do {                                                                                                          \
		_Pragma("unroll") for (int m = 0; m < 2; ++m) t128_read(F.a[m], fa[STAGE][s], m * 4096);                  \
		_Pragma("unroll") for (int tt = 0; tt < 2; ++tt) t128_read(F.b[tt], fb[STAGE][s], tt * 4096);             \
	} while (0)
#define T128_WAIT(F, N) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(F.a[0]), "+v"(F.a[1]), "+v"(F.b[0]), "+v"(F.b[1]) : "n"(N))
#define T128_MFMA(F)                                                                                              \
	do {                                                                                                          \
		_Pragma("unroll") for (int m = 0; m < 2; ++m)                                                             \
			_Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                      \
				acc[m][tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8g, F.a[m]), __builtin_bit_cast(bf16x8g, F.b[tt]), acc[m][tt], 0, 0, 0); \
	} while (0)
#define T128_SYNC() T128_SYNC_V(0)
#define T128_SYNC_V(NV)                                                                                           \
	do {                                                                                                          \
		asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NV) : "memory");                                      \
		__builtin_amdgcn_s_barrier();                                                                             \
		asm volatile("" ::: "memory");                                                                            \
	} while (0)
#define T128_COMPUTE(STAGE, DA, AOFF, DB, BOFF, DD, DEN)                                                          \
	do {                                                                                                          \
		T128Frag f0, f1;                                                                                          \
		T128_LOAD(f0, STAGE, 0);                                                                                  \
		T128_LOAD(f1, STAGE, 1); T128_DMA2(0, DA, AOFF, DB, BOFF, DD, DEN); T128_WAIT(f0, 4); T128_MFMA(f0); __builtin_amdgcn_sched_barrier(0); \
		T128_LOAD(f0, STAGE, 2); T128_DMA2(1, DA, AOFF, DB, BOFF, DD, DEN); T128_WAIT(f1, 4); T128_MFMA(f1); __builtin_amdgcn_sched_barrier(0); \
		T128_LOAD(f1, STAGE, 3); T128_DMA2(2, DA, AOFF, DB, BOFF, DD, DEN); T128_WAIT(f0, 4); T128_MFMA(f0); __builtin_amdgcn_sched_barrier(0); \
		T128_DMA2(3, DA, AOFF, DB, BOFF, DD, DEN); T128_WAIT(f1, 0); T128_MFMA(f1);                                \
	} while (0)
	// this lane's DMA source offsets for a tile whose rows sit in generation g of rows_lds: piece (wave * 4 + i) = tile rows 8 (wave * 4 + i) .. + 8,
	// this lane: row + (lane >> 3), chunk lane & 7; vector rows past the tile's last re-read it (their scores are not stored)
#define T128_SOURCES(AOFF, BOFF, g, VROWS)                                                                        \
	do {                                                                                                          \
		_Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                           \
			const int prow = (wave * 4 + i) * 8 + (lane >> 3), pchunk = (lane & 7) ^ ((prow >> 1) & 7);            \
			AOFF[i] = (uint32_t)rows_lds[(g) * 256 + prow] * lda_b + (uint32_t)pchunk * 16u;                       \
			BOFF[i] = (uint32_t)min(prow, (VROWS) - 1) * ldb_b + (uint32_t)pchunk * 16u;                           \
		}                                                                                                         \
	} while (0)

	const int nk = dp >> 6;
	int gen = 0;
	T128Desc d = desc[t];
	if (tid < T128) {   // rows past the tile's pairs re-read its last pair's query (their scores are not stored)
		const int32_t p = d.p0 + min(tid, d.p_rows - 1);
		rows_lds[tid] = pair_q[p];
		rows_lds[128 + tid] = (int32_t)pair_out[p];
	}
	__syncthreads();
	uint32_t aoff[4], boff[4];
	const unsigned char *bbase = reinterpret_cast<const unsigned char *>(Xs) + (int64_t)__builtin_amdgcn_readfirstlane(d.v0) * ldx * 2;   // (SGPR pair: the DMA's base)
	T128_SOURCES(aoff, boff, 0, d.v_rows);
#pragma unroll
	for (int i = 0; i < 4; ++i) { T128_PIECE(abase, aoff, 0u, i); T128_PIECE(bbase, boff, (uint32_t)T128_TILE_BYTES, i); }
	T128_SYNC();

	IVF_PAD_HERE();
	for (;;) {
		const int32_t t_next = t + wpx;
		const bool has_next = t_next < t_end;   // (uniform)
		T128Desc dn = d;
		if (has_next) dn = desc[t_next];
		// The next tile's rows (query row, output offset), by inline-asm loads that every thread issues (the last tile re-reads its own): hipcc
		// tracks the loads it emits and waits for them where it sees fit -- `vmcnt(0)` in front of the LDS write below and, on the path around
		// that write, in front of every store of the epilogue -- and each such wait also drains what this kernel keeps in flight on purpose.
		int32_t nrow, nout;
		{
			const int32_t p = dn.p0 + min(tid & 127, dn.p_rows - 1);
			const int32_t *pq = pair_q + p;
			const uint32_t *po = pair_out + p;
			asm volatile("global_load_dword %0, %2, off\n\tglobal_load_dword %1, %3, off" : "=&v"(nrow), "=&v"(nout) : "v"(pq), "v"(po) : "memory");
		}
		const int gen_next = gen == 2 ? 0 : gen + 1;
		const unsigned char *bbase_n = reinterpret_cast<const unsigned char *>(Xs) + (int64_t)__builtin_amdgcn_readfirstlane(dn.v0) * ldx * 2;
		uint32_t aoff_n[4] = {0u, 0u, 0u, 0u}, boff_n[4] = {0u, 0u, 0u, 0u};
		f32x16g acc[2][2];
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int tt = 0; tt < 2; ++tt)
#pragma unroll
				for (int e = 0; e < 16; ++e) acc[m][tt][e] = 0.f;
		for (int kt = 0; kt < nk; kt += 2) {
			{   // stage 0 holds k-tile kt: fetch kt + 1 into stage 1 while it is consumed
				const unsigned char *da = abase + (kt + 1) * 128, *db = bbase + (kt + 1) * 128;
				T128_COMPUTE(0, da, aoff, db, boff, (uint32_t)T128_STAGE_BYTES, true);
				// the rows' loads have landed once all but this phase's eight DMA pieces have (memory operations return in order); published by
				// the barrier below, written again, unchanged, by every later k-tile pair (no first-iteration copy of the loop)
				asm volatile("s_waitcnt vmcnt(%2)" : "+v"(nrow), "+v"(nout) : "n"(T128_ROWS_WAIT));
				if (has_next && tid < T128) {
					rows_lds[gen_next * 256 + tid] = nrow;
					rows_lds[gen_next * 256 + 128 + tid] = nout;
				}
				T128_SYNC();
			}
			// stage 1 holds k-tile kt + 1: fetch kt + 2 into stage 0 -- or the next tile's first k-tile
			const bool last = kt + 2 >= nk;
			if (last && has_next) T128_SOURCES(aoff_n, boff_n, gen_next, dn.v_rows);
			const unsigned char *da = last ? abase : abase + (kt + 2) * 128, *db = last ? bbase_n : bbase + (kt + 2) * 128;
			uint32_t ao[4], bo[4];
#pragma unroll
			for (int i = 0; i < 4; ++i) { ao[i] = last ? aoff_n[i] : aoff[i]; bo[i] = last ? boff_n[i] : boff[i]; }
			const bool den = !last || has_next;
			T128_COMPUTE(1, da, ao, db, bo, 0u, den);
			T128_SYNC();
		}
		// ---- this tile's scores.  C/D layout: col = lane & 31 (vector), row = (e & 3) + 8 (e >> 2) + 4 h (pair) within the 32 x 32 block (m, tt).
		// The rows' output offsets first, all of them, by inline-asm reads: hipcc cannot prove that an LDS read does not alias the DMA in
		// flight and put `s_waitcnt vmcnt(1)` in front of EVERY compiler-generated read -- one store's round trip per store, 64 in a row
		// (the first version: 0.06 of the kernel's 0.20 ms).
		u32x4g ro[2][4];
		{
			const uint32_t oaddr = lds0 + (uint32_t)T128_LDS_BYTES + (uint32_t)(gen * 256 + 128 + wm * 64 + 4 * h) * 4u;
#pragma unroll
			for (int m = 0; m < 2; ++m)
#pragma unroll
				for (int g = 0; g < 4; ++g) t128_read(ro[m][g], oaddr, (m * 32 + 8 * g) * 4);
			asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ro[0][0]), "+v"(ro[0][1]), "+v"(ro[0][2]), "+v"(ro[0][3]), "+v"(ro[1][0]), "+v"(ro[1][1]), "+v"(ro[1][2]), "+v"(ro[1][3]));
		}
#pragma unroll
		for (int m = 0; m < 2; ++m)
#pragma unroll
			for (int tt = 0; tt < 2; ++tt) {
				const int n = wn * 64 + tt * 32 + r;
				float *col = S + (d.c0 + n);
#pragma unroll
				for (int e = 0; e < 16; ++e) {
					const int row = wm * 64 + m * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
#ifdef ANNCUR_V_IVF_NOSTORE   // (ablation build)
					if (row < d.p_rows && n < d.v_rows && acc[m][tt][e] == 1.2345e30f) col[ro[m][e >> 2][e & 3]] = acc[m][tt][e];
#else
					if (row < d.p_rows && n < d.v_rows) col[ro[m][e >> 2][e & 3]] = acc[m][tt][e];
#endif
				}
			}
		if (!has_next) break;
		t = t_next; d = dn; gen = gen_next; bbase = bbase_n;
#pragma unroll
		for (int i = 0; i < 4; ++i) { aoff[i] = aoff_n[i]; boff[i] = boff_n[i]; }
	}
#undef T128_SOURCES
#undef T128_COMPUTE
#undef T128_SYNC
#undef T128_SYNC_V
#undef T128_MFMA
#undef T128_WAIT
#undef T128_LOAD
#undef T128_DMA2
#undef T128_DEN
#undef T128_ROWS_WAIT
#undef T128_PIECE
}

// column of a packed score row -> id of the vector (-1 where the score is the -inf padding of a short row)
__global__ __launch_bounds__(256) void ivf_map_ids_packed_kernel(const int32_t *__restrict__ col, const float *__restrict__ val, int64_t n, int32_t k,
																  const int32_t *__restrict__ probe, int32_t nprobe, const uint32_t *__restrict__ coff,
																  const int32_t *__restrict__ offsets, int32_t nlist, const int32_t *__restrict__ ids,
																  int32_t *__restrict__ out) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const int32_t c = col[i];
	int32_t id = -1;
	if (c >= 0 && val[i] > -INFINITY) {
		const int64_t q = i / k;
		for (int s = 0; s < nprobe; ++s) {
			const uint32_t co = coff[q * nprobe + s];
			if (co == IVF_NO_SLOT) continue;
			const int32_t l = probe[q * nprobe + s];
			const uint32_t sz = (uint32_t)(offsets[l + 1] - offsets[l]);
			if ((uint32_t)c >= co && (uint32_t)c < co + sz) { id = ids[offsets[l] + (int32_t)((uint32_t)c - co)]; break; }
		}
	}
	out[i] = id;
}

}  // namespace

extern "C" int anncur_ivf_group_scores(const float *Xs, int64_t ldx, int32_t dp, const int32_t *offsets, const float *Q, int64_t ldq, int32_t nprobe,
									   const int32_t *pair_ids, const int32_t *pair_offsets, const int32_t *tiles, int32_t n_tiles, int64_t lmax, float *S,
									   void *stream) {
	ANNCUR_REQUIRE(dp >= 1 && ldx >= dp && ldq >= dp && nprobe >= 1 && lmax >= 1 && n_tiles >= 0, ANNCUR_E_INVALID, "ivf_group_scores: bad sizes");
	if (n_tiles == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(Xs && offsets && Q && pair_ids && pair_offsets && tiles && S, ANNCUR_E_INVALID, "ivf_group_scores: null pointer");
	hipLaunchKernelGGL(ivf_group_scores_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, Xs, ldx, dp, offsets, Q, ldq, nprobe, pair_ids,
					   pair_offsets, tiles, 0, lmax, S);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_ivf_group_scores_bf16(const void *Xs, int64_t ldx, int32_t dp, const int32_t *offsets, const void *Q, int64_t ldq, int32_t nprobe,
											const int32_t *pair_ids, const int32_t *pair_offsets, const int32_t *tiles, int32_t n_tiles, int64_t lmax, float *S,
											void *stream) {
	ANNCUR_REQUIRE(dp >= 16 && (dp % 16) == 0 && ldx >= dp && ldq >= dp && (ldx % 8) == 0 && (ldq % 8) == 0 && nprobe >= 1 && lmax >= 1 && n_tiles >= 0,
				   ANNCUR_E_INVALID, "ivf_group_scores_bf16: bad sizes (rows zero-padded to a multiple of 16 elements, 16-byte aligned)");
	if (n_tiles == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(Xs && offsets && Q && pair_ids && pair_offsets && tiles && S && ((uintptr_t)Xs % 16) == 0 && ((uintptr_t)Q % 16) == 0, ANNCUR_E_INVALID,
				   "ivf_group_scores_bf16: null or misaligned pointer");
	hipLaunchKernelGGL(ivf_group_scores_bf16_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)Xs, ldx, dp, offsets,
					   (const uint16_t *)Q, ldq, nprobe, pair_ids, pair_offsets, tiles, 0, lmax, S);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_ivf_group_scores_dev(const void *Xs, int dtype, int64_t ldx, int32_t dp, const int32_t *offsets, int32_t nlist, const void *Q, int64_t ldq,
										   int32_t nprobe, const int32_t *pair_ids, const int32_t *pair_offsets, int32_t *tile_start, int32_t max_tiles, int64_t lmax,
										   float *S, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(dtype) && dp >= 1 && ldx >= dp && ldq >= dp && nprobe >= 1 && lmax >= 1 && nlist >= 1 && max_tiles >= 0, ANNCUR_E_INVALID,
				   "ivf_group_scores_dev: bad sizes");
	ANNCUR_REQUIRE(dtype == ANNCUR_F32 || ((dp % 16) == 0 && (ldx % 8) == 0 && (ldq % 8) == 0 && ((uintptr_t)Xs % 16) == 0 && ((uintptr_t)Q % 16) == 0), ANNCUR_E_INVALID,
				   "ivf_group_scores_dev: bf16 rows must be zero-padded to a multiple of 16 elements and 16-byte aligned");
	ANNCUR_REQUIRE(Xs && offsets && Q && pair_ids && pair_offsets && tile_start && S, ANNCUR_E_INVALID, "ivf_group_scores_dev: null pointer");
	hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL(ivf_tile_starts_kernel, dim3(1), dim3(256), 0, st, pair_offsets, offsets, nlist, tile_start);
	ANNCUR_LAUNCH_OK();
	if (max_tiles == 0) return ANNCUR_OK;
	if (dtype == ANNCUR_F32)
		hipLaunchKernelGGL(ivf_group_scores_kernel, dim3((unsigned)max_tiles), dim3(256), 0, st, (const float *)Xs, ldx, dp, offsets, (const float *)Q, ldq, nprobe,
						   pair_ids, pair_offsets, tile_start, nlist, lmax, S);
	else
		hipLaunchKernelGGL(ivf_group_scores_bf16_kernel, dim3((unsigned)max_tiles), dim3(256), 0, st, (const uint16_t *)Xs, ldx, dp, offsets, (const uint16_t *)Q, ldq,
						   nprobe, pair_ids, pair_offsets, tile_start, nlist, lmax, S);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

// ---- round 5: the whole batched search behind one call (see the kernels above)
struct IvfSearchWs { size_t counts, pair_off, cursor, tile_start, coff, row_len, pair_q, pair_out, col, desc, total; };
static IvfSearchWs ivf_search_ws(int64_t nq, int32_t nprobe, int32_t nlist, int32_t k, int64_t max_tiles) {
	IvfSearchWs w;
	size_t o = 0;
	auto take = [&](size_t words) { const size_t at = o; o += (words * 4 + 255) & ~(size_t)255; return at; };
	w.counts = take((size_t)nlist); w.pair_off = take((size_t)nlist + 1); w.cursor = take((size_t)nlist); w.tile_start = take((size_t)nlist + 1);
	w.coff = take((size_t)nq * nprobe); w.row_len = take((size_t)nq); w.pair_q = take((size_t)nq * nprobe); w.pair_out = take((size_t)nq * nprobe);
	w.col = take((size_t)nq * k);
	w.desc = take((size_t)max_tiles * 8);   // (the 128 x 128-tile kernel's descriptors)
	w.total = o;
	return w;
}
static bool ivf_use_tile128(int dtype, int32_t dp, int64_t ldx, int64_t ldq, int64_t nq) {
	return dtype == ANNCUR_BF16 && (dp % 128) == 0 && nq * ldq * 2 < ((int64_t)1 << 32) && 128 * ldx * 2 < ((int64_t)1 << 32);
}
extern "C" int32_t anncur_ivf_search_tile(int dtype, int32_t dp, int64_t ldx, int64_t ldq, int64_t nq) { return ivf_use_tile128(dtype, dp, ldx, ldq, nq) ? T128 : GT; }
extern "C" size_t anncur_ivf_search_workspace_bytes(int64_t nq, int32_t nprobe, int32_t nlist, int32_t k, int64_t max_tiles) {
	if (nq < 0 || nprobe < 1 || nlist < 1 || k < 1 || max_tiles < 0) return 0;
	return ivf_search_ws(nq, nprobe, nlist, k, max_tiles).total;
}
extern "C" int anncur_ivf_search_grouped(const void *Xs, int dtype, int64_t ldx, int32_t dp, const int32_t *offsets, const int32_t *ids, int32_t nlist,
										 const void *Q, int64_t ldq, int64_t nq, const int32_t *probe, int32_t nprobe, int32_t k, int64_t max_tiles,
										 float *S, int64_t pitch, void *workspace, size_t workspace_bytes, float *out_val, int32_t *out_idx, void *stream) {
	ANNCUR_REQUIRE(dtype_ok(dtype) && dp >= 1 && ldx >= dp && ldq >= dp && nprobe >= 1 && nlist >= 1 && nq >= 0 && max_tiles >= 0 && max_tiles < (int64_t)0x7fffffff,
				   ANNCUR_E_INVALID, "ivf_search_grouped: bad sizes");
	ANNCUR_REQUIRE(dtype == ANNCUR_F32 || ((dp % 16) == 0 && (ldx % 8) == 0 && (ldq % 8) == 0 && ((uintptr_t)Xs % 16) == 0 && ((uintptr_t)Q % 16) == 0), ANNCUR_E_INVALID,
				   "ivf_search_grouped: bf16 rows must be zero-padded to a multiple of 16 elements and 16-byte aligned");
	ANNCUR_REQUIRE(k >= 1 && pitch >= k && (pitch % 4) == 0 && nq * pitch < ((int64_t)1 << 32) && nq * (int64_t)nprobe < (int64_t)0x7fffffff, ANNCUR_E_INVALID,
				   "ivf_search_grouped: need k <= pitch, pitch a multiple of 4, nq * pitch < 2^32 elements (split the queries)");
	if (k > IVF_RAGGED_MAX_K || nlist > IVF_MAX_NLIST_LDS) {
		anncur_set_error("ivf_search_grouped: k = %d (<= %d) / nlist = %d (<= %d) outside this path", k, IVF_RAGGED_MAX_K, nlist, IVF_MAX_NLIST_LDS);
		return ANNCUR_E_UNSUPPORTED;
	}
	if (nq == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(Xs && offsets && ids && Q && probe && S && out_val && out_idx && ((uintptr_t)S % 16) == 0, ANNCUR_E_INVALID, "ivf_search_grouped: null or misaligned pointer");
	const IvfSearchWs w = ivf_search_ws(nq, nprobe, nlist, k, max_tiles);
	ANNCUR_REQUIRE(workspace && workspace_bytes >= w.total && ((uintptr_t)workspace % 256) == 0, ANNCUR_E_WORKSPACE, "ivf_search_grouped: workspace of %zu bytes needed (256-byte aligned)", w.total);
	unsigned char *wb = reinterpret_cast<unsigned char *>(workspace);
	int32_t *counts = reinterpret_cast<int32_t *>(wb + w.counts), *pair_off = reinterpret_cast<int32_t *>(wb + w.pair_off), *cursor = reinterpret_cast<int32_t *>(wb + w.cursor);
	int32_t *tile_start = reinterpret_cast<int32_t *>(wb + w.tile_start), *row_len = reinterpret_cast<int32_t *>(wb + w.row_len), *pair_q = reinterpret_cast<int32_t *>(wb + w.pair_q);
	uint32_t *coff = reinterpret_cast<uint32_t *>(wb + w.coff), *pair_out = reinterpret_cast<uint32_t *>(wb + w.pair_out);
	int32_t *col = reinterpret_cast<int32_t *>(wb + w.col);
	hipStream_t st = (hipStream_t)stream;
	const bool t128 = ivf_use_tile128(dtype, dp, ldx, ldq, nq);
	const unsigned qgrid = (unsigned)ceil_div64(nq, 256);
	ANNCUR_HIP_OK(hipMemsetAsync(counts, 0, (size_t)nlist * 4, st));
	hipLaunchKernelGGL(ivf_layout_kernel, dim3(qgrid), dim3(256), (size_t)nlist * 4, st, probe, nq, nprobe, offsets, nlist, k, pitch, coff, row_len, counts, S);
	hipLaunchKernelGGL(ivf_pair_offsets_kernel, dim3(1), dim3(256), 0, st, counts, offsets, nlist, t128 ? T128 : GT, pair_off, cursor, tile_start);
	hipLaunchKernelGGL(ivf_scatter_kernel, dim3(qgrid), dim3(256), (size_t)nlist * 8, st, probe, nq, nprobe, nlist, pitch, coff, cursor, pair_q, pair_out);
	ANNCUR_LAUNCH_OK();
	if (max_tiles > 0) {
		if (t128) {
			const int rc = anncur_ensure_dyn_lds((const void *)ivf_tile128_kernel, T128_LDS_TOTAL);
			if (rc != ANNCUR_OK) return rc;
			T128Desc *desc = reinterpret_cast<T128Desc *>(wb + w.desc);
			hipLaunchKernelGGL(ivf_tile_desc_kernel, dim3((unsigned)ceil_div64(max_tiles, 256)), dim3(256), 0, st, tile_start, nlist, offsets, pair_off, T128, (int32_t)max_tiles, desc);
			int64_t wgs = 2 * (int64_t)anncur_num_cu();   // two resident workgroups per CU walk the tiles
			wgs = (wgs + 7) & ~(int64_t)7;
			if (wgs > ((max_tiles + 7) & ~(int64_t)7)) wgs = (max_tiles + 7) & ~(int64_t)7;
			hipLaunchKernelGGL(ivf_tile128_kernel, dim3((unsigned)wgs), dim3(256), T128_LDS_TOTAL, st, (const uint16_t *)Xs, ldx, dp, (const uint16_t *)Q, ldq,
							   pair_q, pair_out, tile_start, nlist, (int32_t)max_tiles, desc, S);
		} else if (dtype == ANNCUR_F32)
			hipLaunchKernelGGL(ivf_tile64_packed_kernel<float>, dim3((unsigned)max_tiles), dim3(256), 0, st, (const float *)Xs, ldx, dp, offsets, (const float *)Q, ldq, pair_q,
							   pair_out, pair_off, tile_start, nlist, S);
		else
			hipLaunchKernelGGL(ivf_tile64_packed_kernel<uint16_t>, dim3((unsigned)max_tiles), dim3(256), 0, st, (const uint16_t *)Xs, ldx, dp, offsets, (const uint16_t *)Q, ldq,
							   pair_q, pair_out, pair_off, tile_start, nlist, S);
		ANNCUR_LAUNCH_OK();
	}
	const int rc = anncur_rowwise_topk_ragged(S, ANNCUR_F32, nq, pitch, pitch, row_len, k, out_val, col, stream);
	if (rc != ANNCUR_OK) return rc;
	const int64_t n = nq * (int64_t)k;
	hipLaunchKernelGGL(ivf_map_ids_packed_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, st, col, out_val, n, k, probe, nprobe, coff, offsets, nlist, ids, out_idx);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_ivf_map_ids(const int32_t *col, const float *val, int64_t nq, int32_t k, int64_t lmax, const int32_t *probe, int32_t nprobe,
								  const int32_t *offsets, const int32_t *ids, int32_t *out_idx, void *stream) {
	ANNCUR_REQUIRE(nq >= 0 && k >= 1 && lmax >= 1 && nprobe >= 1, ANNCUR_E_INVALID, "ivf_map_ids: bad sizes");
	if (nq == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(col && val && probe && offsets && ids && out_idx, ANNCUR_E_INVALID, "ivf_map_ids: null pointer");
	const int64_t n = nq * (int64_t)k;
	hipLaunchKernelGGL(ivf_map_ids_kernel, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, (hipStream_t)stream, col, n, k, lmax, probe, nprobe, offsets, ids, val,
					   out_idx);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_ivf_build_lists(const int32_t *assign, int64_t n, int32_t nlist, int32_t *counts, int32_t *offsets, int32_t *ids, void *stream) {
	ANNCUR_REQUIRE(n >= 0 && n < (int64_t)0x7fffffff && nlist >= 1 && nlist <= 65535 * 16, ANNCUR_E_INVALID, "ivf_build_lists: bad sizes");
	ANNCUR_REQUIRE(counts && offsets && (n == 0 || (assign && ids)), ANNCUR_E_INVALID, "ivf_build_lists: null pointer");
	hipStream_t st = (hipStream_t)stream;
	hipLaunchKernelGGL(ivf_count_kernel, dim3((unsigned)nlist), dim3(256), 0, st, assign, n, counts);
	hipLaunchKernelGGL(ivf_scan_counts_kernel, dim3(1), dim3(256), 0, st, counts, nlist, offsets);
	if (n > 0) hipLaunchKernelGGL(ivf_fill_kernel, dim3((unsigned)nlist), dim3(256), 0, st, assign, n, offsets, ids);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_ivf_list_means(const float *Xs, int64_t ldx, int32_t d, const int32_t *offsets, int32_t nlist, float *centroids, int64_t ldc,
									 void *stream) {
	ANNCUR_REQUIRE(nlist >= 1 && d >= 1 && ldx >= d && ldc >= d, ANNCUR_E_INVALID, "ivf_list_means: bad sizes");
	ANNCUR_REQUIRE(Xs && offsets && centroids, ANNCUR_E_INVALID, "ivf_list_means: null pointer");
	hipLaunchKernelGGL(ivf_means_kernel, dim3((unsigned)nlist), dim3(256), 0, (hipStream_t)stream, Xs, ldx, d, offsets, centroids, ldc);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_renorm_rows(float *M, int64_t n_rows, int64_t n_cols, int64_t ld, void *stream) {
	ANNCUR_REQUIRE(n_rows >= 0 && n_cols >= 0 && ld >= n_cols && n_rows <= 0x7fffffff, ANNCUR_E_INVALID, "renorm_rows: bad shape");
	if (n_rows == 0 || n_cols == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(M, ANNCUR_E_INVALID, "renorm_rows: null pointer");
	hipLaunchKernelGGL(renorm_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, M, n_cols, ld);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

extern "C" int anncur_ivf_scan(const float *Xs, int64_t ldx, int32_t dp, const int32_t *offsets, const int32_t *ids, const float *Q, int64_t ldq,
							   int64_t nq, const int32_t *probe, int32_t nprobe, int32_t k, float *out_val, int32_t *out_idx, void *stream) {
	ANNCUR_REQUIRE(dp >= 16 && (dp % 16) == 0 && dp <= 16384 && ldx >= dp && (ldx % 4) == 0 && ldq >= dp && (ldq % 4) == 0, ANNCUR_E_INVALID,
				   "ivf_scan: vectors and queries must be zero-padded to a multiple of 16 floats (dp=%d)", dp);
	ANNCUR_REQUIRE(nq >= 0 && nq < (int64_t)0x7fffffff && nprobe >= 1 && k >= 1 && k <= ANNCUR_MAX_TOPK, ANNCUR_E_INVALID, "ivf_scan: bad sizes");
	if (nq == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(Xs && offsets && ids && Q && probe && out_val && out_idx, ANNCUR_E_INVALID, "ivf_scan: null pointer");
	ANNCUR_REQUIRE(((uintptr_t)Xs % 16) == 0 && ((uintptr_t)Q % 16) == 0, ANNCUR_E_INVALID, "ivf_scan: Xs and Q must be 16-byte aligned");
	hipStream_t st = (hipStream_t)stream;
	int rc;
#define LAUNCH_IVF(KM)                                                                                                       \
	do {                                                                                                                     \
		const size_t lds = SelCfg<KM>::LDS_BYTES + (size_t)dp * 4;                                                           \
		if ((rc = anncur_ensure_dyn_lds((const void *)ivf_scan_kernel<KM>, (int)lds)) != ANNCUR_OK) return rc;               \
		hipLaunchKernelGGL((ivf_scan_kernel<KM>), dim3((unsigned)nq), dim3(SEL_THREADS), lds, st, Xs, ldx, dp, offsets, ids, Q, ldq, probe, \
						   nprobe, (uint32_t)k, out_val, out_idx);                                                            \
	} while (0)
	if (k <= 128) LAUNCH_IVF(128); else if (k <= 512) LAUNCH_IVF(512); else LAUNCH_IVF(2048);
#undef LAUNCH_IVF
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

// ------------------------------------------------------------------ descending-norm buckets (index-build hint of the fused top-k)
// bucket[i] of row i by its squared norm, largest norms first: 0 .. n_buckets-1, linear between the largest and the smallest norm
// of the matrix.  Followed by anncur_ivf_build_lists (a stable counting sort) this yields the coarse descending-norm row order the
// index builder passes to anncur_score_topk_ex -- an ordering that only moves speed, so a coarse one does.
namespace {
__global__ __launch_bounds__(256) void row_sumsq_kernel(const float *__restrict__ A, int64_t n_rows, int64_t n_cols, int64_t lda, float *__restrict__ out) {
	const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	if (r >= n_rows) return;
	const float *row = A + r * lda;
	float s = 0.f;
	for (int64_t c = threadIdx.x & 63; c < n_cols; c += 64) { const float v = row[c]; s = fmaf(v, v, s); }
	for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
	if ((threadIdx.x & 63) == 0) out[r] = s;
}
// Smallest / largest row norm as sortable keys: grid-stride over the norms, wave reduction, one LDS combine per workgroup and ONE
// atomic pair per workgroup (<= 256 of them per call).  Round 2 did the atomic pair per ROW from inside row_sumsq_kernel: 100 000
// same-address atomics serialised at the memory side -- 2.27 ms of a 8.7 ms index build at cfg2, 22.7 ms at I = 10^6.
__global__ __launch_bounds__(256) void norm_minmax_kernel(const float *__restrict__ nrm, int64_t n, uint32_t *__restrict__ minmax) {
	__shared__ uint32_t s_lo[4], s_hi[4];
	uint32_t lo = 0xffffffffu, hi = 0u;
	for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
		const float v = nrm[i];
		const uint32_t key = f32_sortable(v == v ? v : 0.f);
		lo = key < lo ? key : lo;
		hi = key > hi ? key : hi;
	}
	for (int d = 32; d > 0; d >>= 1) {
		const uint32_t l2 = (uint32_t)__shfl_xor((int)lo, d), h2 = (uint32_t)__shfl_xor((int)hi, d);
		lo = l2 < lo ? l2 : lo;
		hi = h2 > hi ? h2 : hi;
	}
	if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < 4; ++w) { lo = s_lo[w] < lo ? s_lo[w] : lo; hi = s_hi[w] > hi ? s_hi[w] : hi; }
		atomicMin(&minmax[0], lo);
		atomicMax(&minmax[1], hi);
	}
}
__global__ __launch_bounds__(256) void norm_bucket_kernel(const float *__restrict__ nrm, int64_t n, const uint32_t *__restrict__ minmax, int32_t n_buckets,
														   int32_t *__restrict__ bucket) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	const float lo = f32_unsortable(minmax[0]), hi = f32_unsortable(minmax[1]);
	const float v = nrm[i] == nrm[i] ? nrm[i] : lo;
	const float span = hi - lo;
	int32_t b = span > 0.f ? (int32_t)((hi - v) / span * (float)n_buckets) : 0;
	bucket[i] = b < 0 ? 0 : (b >= n_buckets ? n_buckets - 1 : b);
}
}  // namespace

extern "C" int anncur_norm_buckets(const float *A, int64_t n_rows, int64_t n_cols, int64_t lda, int32_t n_buckets, float *norms, uint32_t *minmax2,
								   int32_t *bucket, void *stream) {
	ANNCUR_REQUIRE(n_rows >= 0 && n_cols >= 0 && lda >= n_cols && n_buckets >= 1, ANNCUR_E_INVALID, "norm_buckets: bad sizes");
	if (n_rows == 0) return ANNCUR_OK;
	ANNCUR_REQUIRE(A && norms && minmax2 && bucket, ANNCUR_E_INVALID, "norm_buckets: null pointer");
	hipStream_t st = (hipStream_t)stream;
	ANNCUR_HIP_OK(hipMemsetAsync(minmax2, 0xff, 4, st));      // running minimum of the sortable keys
	ANNCUR_HIP_OK(hipMemsetAsync(minmax2 + 1, 0, 4, st));     // running maximum
	hipLaunchKernelGGL(row_sumsq_kernel, dim3((unsigned)ceil_div64(n_rows, 4)), dim3(256), 0, st, A, n_rows, n_cols, lda, norms);
	const int64_t mm_blocks = ceil_div64(n_rows, 256 * 16);
	hipLaunchKernelGGL(norm_minmax_kernel, dim3((unsigned)(mm_blocks > 256 ? 256 : mm_blocks)), dim3(256), 0, st, norms, n_rows, minmax2);
	hipLaunchKernelGGL(norm_bucket_kernel, dim3((unsigned)ceil_div64(n_rows, 256)), dim3(256), 0, st, norms, n_rows, minmax2, n_buckets, bucket);
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}
