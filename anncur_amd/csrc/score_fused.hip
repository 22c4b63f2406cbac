// Fused S_hat = X . E  +  per-query top-k for gfx950; S_hat is never written to HBM.
//
// Orientation: the MFMA computes S_hat^T tiles, D[item][query] = sum_k Et[item][k] * X[query][k]
// (v_mfma_f32_32x32x16_bf16, A operand = 32 item rows of E^T from LDS, B operand = the wave's
// queries, resident in registers for the whole kernel).  In the C/D layout the query sits on the
// LANE (col = lane & 31) and the 16 accumulator registers are 16 items, so a query's threshold is
// one lane-local VGPR and the filter is one v_cmp per output element.
//
// Data layout in HBM
//   X   [Q  x Kp] bf16 row-major   queries' scores against the anchor items (C_q), K zero-padded to Kp
//   Et  [Ip x Kp] bf16 row-major   item embeddings E^T (= (U R)^T), Ip = 32-multiple, zero rows past I
//   one 32-item tile of Et is ONE contiguous 64*Kp-byte block -> 16-byte coalesced loads.
// LDS: the tile is stored row-major with the 16-byte chunk index XOR-swizzled by the row so the
// per-k-step ds_read_b128 (32 rows x one chunk column) is bank-conflict free.
//
// Launches (all on the caller's stream):
//   1. prepass   : group maxima of S_hat over a strided sample of item tiles  -> gmax[Q x G]
//   2. threshold : tau[q] = k-th largest group maximum (wave-per-query radix select) -> a lower bound on the
//                  k-th best score of the query (k distinct groups each hold an element >= tau)
//   3. sweep     : all tiles, in 1-3 stages (plan_fused picks the split); every S_hat[q,i] >= tau[q] goes through the
//                  lane's private LDS ring to the lane's private candidate segment in HBM (no atomics); between stages
//                  a wave-per-query kernel raises tau[q] to the k-th best candidate collected so far
//   4. select    : per query, exact top-k of its candidates (wave-level selector; workgroup-level kernel for k > 128).
//                  A segment that overflowed or whose ring wrapped is repaired by recomputing only the item tiles its
//                  split swept; nothing leaves the call inexact.
// Algorithmic work: 2*Q*Kp*I flops in (3) (+ sample fraction in (1)).
#include <stdlib.h>
#include <math.h>
#include <type_traits>
// scripts/placement_sweep.sh builds the library with ANNCUR_PLACEMENT_PAD = 1..n: that many extra instructions in front of every
// sweep loop move the loops through all alignments relative to the instruction-fetch lines.  A result that changes with the pad is a
// missing wait state somewhere (DESIGN.md 4.1 'A latent hazard'); the shipped build has no pad.
#ifdef ANNCUR_PLACEMENT_PAD
#define ANNCUR_PAD_HERE() asm volatile(".rept %0\n\ts_nop 0\n\t.endr" ::"n"(ANNCUR_PLACEMENT_PAD) : "memory")
#else
#define ANNCUR_PAD_HERE() do { } while (0)
#endif
// Tuning knobs (ANNCUR_DEBUG_* environment variables) exist in the experiments library only (`make experiments`, -DANNCUR_TIMING_EXPERIMENTS):
// the product library reads no environment -- knob() is the constant nullptr there and every `if (const char *dbg = knob(...))` folds away.
#ifdef ANNCUR_TIMING_EXPERIMENTS
static inline const char *knob(const char *name) { return getenv(name); }
#else
static inline const char *knob(const char *) { return nullptr; }
#endif
#ifdef ANNCUR_TIMING_EXPERIMENTS
__device__ unsigned long long *d_sweep_stamps = nullptr;  // diagnostic build: {d s_memtime, d s_memrealtime} of the sweep's tile loop per workgroup
__device__ unsigned long long *d_sel_stamps = nullptr;  // diagnostic build: phase stamps of the wave-level select kernels (4 per workgroup)
#define SEL_STAMP(i) do { if (d_sel_stamps && wave == 0 && lane == 0) d_sel_stamps[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SEL_STAMP(i) do { } while (0)
#endif
#include "select.hpp"
#include "wave_select.hpp"
#include "select_stream.hpp"

using namespace anncur;

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;  // native vector: stays in VGPRs (HIP's uint4 struct did not)

constexpr int TILE_I = 32;
constexpr int CHUNK_TILES = 4;   // dynamic tile schedule: tiles per ticket

template <int KP, int QTV = ((KP <= 256) ? 2 : 1)>
struct FusedCfg {
	static constexpr int KSTEPS = KP / 16;
	static constexpr int QT = QTV;                  // 32-query sub-tiles per wave (default: 2 up to Kp = 256, 1 for Kp = 512)
	static constexpr int BQ = 4 * 32 * QT;          // queries per workgroup (4 waves)
	static constexpr int CPR = KP / 8;              // 16-byte chunks per Et row
	static constexpr int TILE_BYTES = TILE_I * KP * 2;
	static constexpr int PASSES = TILE_BYTES / (256 * 16);
	static constexpr int QDEPTH = 8;                 // lane-private LDS hit queue: entries per (lane, sub-tile)
	static constexpr int QUEUE_BYTES = QT * QDEPTH * 256 * 8;
	static constexpr int QUEUE_OFF = (2 * TILE_BYTES + 16383) / 16384 * 16384;  // rings are 16 KiB aligned (filter_one)
	static constexpr int TICKET_OFF = QUEUE_OFF + QUEUE_BYTES;        // two 32-bit words: the chunk tickets thread 0 hands to its workgroup (dynamic tile schedule)
	// (Kp = 512: two 32 KiB tile buffers + 16 KiB of rings are exactly half a CU's LDS -- 16 bytes more and only ONE workgroup fits per CU
	//  (measured: the sweep of cfg4 20 % slower); that body keeps static tile shares and no ticket words)
	static constexpr int LDS_BYTES = TICKET_OFF + (KP >= 512 ? 0 : 16);  // the prepass kernel uses only the tile part
	static constexpr int NAOFF = KSTEPS < 8 ? KSTEPS : 8;             // fragment address registers of the staggered sweep (see stagger_tile)
};

template <int CPR>
__device__ __forceinline__ int swz(int row, int c) {
	if (CPR >= 16) return c ^ (row & 15);
	return c ^ ((row >> 1) & 7);  // CPR == 8 (Kp = 64): 128-byte rows, two rows per 256-byte bank line
}

struct FusedParams {
	const uint16_t *X; int64_t ldx;
	const uint16_t *Et;
	int64_t Q, I;
	int n_tiles, n_full_tiles;
	int S, tiles_per_split;           // sweep partition of the item tiles (of the current stage)
	int tile_begin, tile_end;         // item-tile range of the current sweep stage
	int tile_step;                    // 1: split s sweeps the contiguous range [tile_begin + s * tiles_per_split, + tiles_per_split);
	                                  // S: split s sweeps tile_begin + s, + S, + 2 S ... (interleaved: see launch_fused)
	int carry;                        // 1: segment counts continue from the previous stage
	int n_st, S0, st_per_split;       // prepass: sample tiles and their partition
	int sample_leading;               // prepass samples the leading n_st tiles instead of a strided sample (ANNCUR_TOPK_LEADING_SAMPLE)
	float *gmax; int n_groups;        // prepass output [Q x n_groups]
	const float *tau; int tau_stride; // threshold per query: tau[q * tau_stride]
	uint2 *cand; uint32_t *seg_cnt; int capg;
	int nseg;                         // candidate segments per query (2 S: 32x32x16 body, lane halves; S: 16x16x32 body)
	int rb_major;                     // work id -> (row block, split): 1 = row-block-major (dynamic tile schedule), 0 = split-major
	int flush_tiles;                  // wave-cooperative queue flush period (tiles)
	int debug_nostore;                // timing experiments only: candidates are counted but not stored
	int debug_stamp;                  // timing experiments only: this launch writes the in-kernel clock stamps
	float tau_bias;                   // 0 in production; ANNCUR_DEBUG_TAU_BIAS (timing experiments only: results become wrong)
	int n_wg;                         // grid size (for the XCD remap)
	// dynamic tile schedule of a sweep stage (chunk_tiles > 0): the workgroups of a query row block draw chunks of `chunk_tiles` consecutive
	// tiles from the row block's ticket counter instead of sweeping a fixed share -- see score_kernel
	int chunk_tiles, n_chunks;
	uint32_t *chunk_ctr;              // [n row blocks], zero at launch
	uint8_t *chunk_owner;             // [n row blocks x n_chunks]: which item split swept the chunk (the repair path's map)
	int sliced, chunks_per_slice;     // XCD-sliced tickets: chunk_ctr is [row block][N_SLICES], slice s = chunks [s * chunks_per_slice, + chunks_per_slice)
	// threshold ladder of score16_kernel (score16.hpp): levels [n_rb x BQ][LADDER_LEVELS], counter words [n_rb x BQ][4] (zero at launch),
	// the cell each wave raises to the threshold it ended with (initialised to tau0 by the threshold kernel), k
	int prio_mode;                    // static wave priority of the sweep's workgroups (score16_kernel): see launch_fused
	int ladder_on; uint32_t ladder_k, ladder_mask;   // ladder_mask = period - 1 (a power of two): tiles between two fetches of a wave's counter words
	const float *ladder; uint32_t *ladder_cnt; float *tau_final;
	uint32_t *nfb;                    // the call's fallback counter (workspace word 0): the ring kernel reports a spin timeout there
	int ring_stagger, ring_spin_sleep; // ring kernel: start delay of waves 4..7 in units of 64 cycles; s_sleep between two polls of a waiting wave
};

// Contiguous work ids per XCD (blocks b and b+8 share an XCD's L2): speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int n) {
	const int q = n >> 3, r = n & 7, x = b & 7, l = b >> 3;
	return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}


// One 32-item tile of E^T (contiguous 64*KP bytes in HBM) -> LDS by direct-to-LDS loads (global_load_lds_dwordx4: 1 KiB per
// wave-instruction, LDS destination = wave-uniform base + lane*16, no staging VGPRs, no ds_write).  The bank-conflict
// swizzle is applied on the per-lane SOURCE address (LDS chunk position p holds source chunk (row, c ^ f(row)));
// readers apply the same involution.
template <int KP>
__device__ __forceinline__ void tile_dma(const uint16_t *__restrict__ Et, int tile, unsigned char *buf, int wave, int lane) {
	constexpr int CPR = FusedCfg<KP>::CPR;
	constexpr int PER_WAVE = FusedCfg<KP>::TILE_BYTES / 1024 / 4;
	const unsigned char *src = reinterpret_cast<const unsigned char *>(Et + (int64_t)tile * TILE_I * KP);
#pragma unroll
	for (int i = 0; i < PER_WAVE; ++i) {
		const int piece = wave * PER_WAVE + i;  // wave-uniform
		const int pch = piece * 64 + lane;
		const int row = pch / CPR, cs = pch % CPR;
		const int c = swz<CPR>(row, cs);
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (row * CPR + c) * 16),
										 (__attribute__((address_space(3))) void *)(buf + piece * 1024), 16, 0, 0);
	}
}

// (Round 3 also issued the next tile's pieces BETWEEN the tile's MFMAs instead of in a burst at the head of the step -- the phase stamps
//  charge the burst with 384 of the 2855 cycles of a step -- : no gain, same box, alternating (0.4802 vs 0.4792 ms per sweep): the partner
//  wave's MFMAs cover the burst already.)
// The same DMA for the staggered sweep, hand-placed: hipcc kept the four per-lane source offsets as 64-bit pairs plus four VGPRs of LDS
// destinations that it then moved to M0 through v_readfirstlane (12 VGPRs and ~20 vector instructions per tile and wave in a kernel
// that sits at the 256-register limit).  Here: one 32-bit offset register per piece (computed once), the tile's base in SGPRs
// (saddr form), M0 from a scalar.  voff[i] = byte offset of this lane's 16 bytes of piece i inside a tile (swizzle on the source).
template <int KP>
__device__ __forceinline__ void tile_dma_offsets(uint32_t (&voff)[FusedCfg<KP>::TILE_BYTES / 4096], int wave_u, int lane) {
	constexpr int CPR = FusedCfg<KP>::CPR, PER_WAVE = FusedCfg<KP>::TILE_BYTES / 4096;
#pragma unroll
	for (int i = 0; i < PER_WAVE; ++i) {
		const int pch = (wave_u * PER_WAVE + i) * 64 + lane;
		const int row = pch / CPR, cs = pch % CPR;
		voff[i] = (uint32_t)(row * CPR + swz<CPR>(row, cs)) * 16u;
	}
}
template <int KP>
__device__ __forceinline__ void tile_dma_s(const uint16_t *__restrict__ Et, int tile, uint32_t lds_buf, int wave_u,
											const uint32_t (&voff)[FusedCfg<KP>::TILE_BYTES / 4096]) {
	constexpr int PER_WAVE = FusedCfg<KP>::TILE_BYTES / 4096;
	const unsigned char *src = reinterpret_cast<const unsigned char *>(Et) + (int64_t)tile * FusedCfg<KP>::TILE_BYTES;  // (uniform)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
	for (int i = 0; i < PER_WAVE; ++i) {
		const uint32_t m0v = lds_buf + (uint32_t)(wave_u * PER_WAVE + i) * 1024u;  // (uniform) LDS destination of the piece: M0 + lane * 16
		asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff[i]), "s"(src) : "memory", "m0");
	}
#endif
}

// Threshold filter of one 32x32 accumulator tile: lane = query, register e = item row (e & 3) + 8 (e >> 2) (+ 4 h in item0).
// Survivors go to the lane's private LDS queue (slot i of lane tid at lq[i * 256]: conflict-free, no atomics); the queues
// are drained to the lane's HBM candidate segment by flush_queue() every FLUSH_TILES tiles with ONE store instruction per
// queue slot for the whole wave (a store per hit made the kernel store-issue bound: ~20 sparse stores per tile per wave).
// The queue's LDS accesses are inline asm: with a direct-to-LDS load in flight hipcc cannot prove that an ordinary LDS
// store does not alias the DMA destination and puts s_waitcnt vmcnt(0) in front of EVERY push (measured: it serialised the
// pushes behind the next tile's DMA and the flush stores).  The queue region is disjoint from the tile buffers; LDS
// executes one wave's instructions in order, so a slot written by ds_write_b64 is seen by the later ds_read_b64.
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
	return (uint32_t)(size_t)(__attribute__((address_space(3))) const char *)p;
}
__device__ __forceinline__ void lds_store_u64(uint32_t addr, uint32_t lo, uint32_t hi) {
#if defined(__HIP_DEVICE_COMPILE__)
	const unsigned long long d = ((unsigned long long)hi << 32) | lo;
	asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(d) : "memory");
#endif
}
__device__ __forceinline__ uint2 lds_load_u64(uint32_t addr) {
	unsigned long long d = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d) : "v"(addr) : "memory");
#endif
	return make_uint2((uint32_t)d, (uint32_t)(d >> 32));
}

__device__ __forceinline__ void lds_store_u32(uint32_t addr, uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ uint32_t lds_load_u32_uniform(uint32_t addr) {  // every lane reads the same word; returned as a scalar
	uint32_t d = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d) : "v"(addr) : "memory");
#endif
	return (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
}
// Ticket draw of the dynamic tile schedule: a RETURNING atomic add whose result is not waited for where it is issued.  (Written with
// atomicAdd() hipcc's atomic optimiser turns the one-lane add into a wave-aggregated one and waits vmcnt(0) for it on the spot --
// directly behind the next tile's DMA.)  The destination is in flight after the asm statement, like the fragment ring's registers:
// ticket_wait() is the wait, and takes the register as an in/out operand so that nothing that reads it can be scheduled above it.
__device__ __forceinline__ void ticket_draw(uint32_t &ticket, uint32_t *ctr) {
#if defined(__HIP_DEVICE_COMPILE__)
	const uint32_t one = 1u;
	asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ticket) : "v"(ctr), "v"(one) : "memory");
#endif
}
__device__ __forceinline__ void ticket_wait(uint32_t &ticket) {
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("s_waitcnt vmcnt(0)" : "+v"(ticket)::"memory");
#endif
}
// Two dwords from two separate VGPRs (no 64-bit register pair has to be assembled in the hit path).
__device__ __forceinline__ void lds_store_2x32(uint32_t addr, uint32_t lo, uint32_t hi) {
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(addr), "v"(lo), "v"(hi) : "memory");
#endif
}

// One accumulator element of the filter (element index e is a compile-time constant after unrolling).  The hit path is kept
// to the bare minimum (slot address, item id, one LDS store, count): the sweep is vector-issue bound, every instruction here is
// paid ~0.3 times per MFMA.  The queue is a ring of D slots; what does not belong in it is sorted out at flush time
// (items past I of the matrix' last, partial tile; a count above D = the ring wrapped = overflow).
// Measured on cfg2 (MI355X): branch version 0.58 ms per sweep, predicated version 0.655 ms independent of the hit rate (its
// compare -> exec -> store chain sits in front of the wave's next MFMA) -> the branch version is the one in use.
// (plan_stages picks the variant per sweep stage from the expected hit rate: predicated above ~0.5 taken branches per compare)
template <int D, bool FILTER_PREDICATED = false, bool INLINE_HIT = false>
__device__ __forceinline__ void filter_one(float v, int e, float tau, uint32_t item0, uint32_t lq, uint32_t &qcnt) {
	static_assert((D & (D - 1)) == 0, "queue depth must be a power of two");
	// qcnt is kept pre-shifted (slot stride 2048 B); lq has bits 11..13 clear (16 KiB-aligned ring), so OR == ADD
	if (FILTER_PREDICATED) {
		// Branch-free: no wave-level "any hit" test, the push runs under the compare's lane mask (a block this short gets no
		// s_cbranch_execz from hipcc).  With lane = query the wave-level branch is taken by ~35 % of the compares at k = 100 and by nearly
		// all of them at k >= 500, and a taken branch (out and back) costs more than these always-issued instructions.
		// The compare is compiler-generated, so hipcc keeps its own MFMA -> VALU distance; the first version of this variant did the
		// compare INSIDE an inline-asm string, 0-6 wait states behind the MFMA whose result it read, and lost a survivor now and then
		// (found by the randomised parity run; scripts/check_mfma_hazards.py flags exactly that string).
		if (v >= tau) {
			lds_store_2x32((qcnt & (uint32_t)((D - 1) << 11)) | lq, __float_as_uint(v), item0 + (uint32_t)((e & 3) + 8 * (e >> 2)));
			qcnt += 2048u;
		}
	} else if (INLINE_HIT) {
		// the push block stays IN LINE behind a short forward skip (taken when no lane hits) instead of out of line behind a far
		// branch out and back (taken when one does)
		if (__builtin_expect(__ballot(v >= tau) != 0ull, 1)) {
			if (v >= tau) {
				lds_store_2x32((qcnt & (uint32_t)((D - 1) << 11)) | lq, __float_as_uint(v), item0 + (uint32_t)((e & 3) + 8 * (e >> 2)));
				qcnt += 2048u;
			}
		}
	} else if (__builtin_expect(__ballot(v >= tau) != 0ull, 0)) {
		if (v >= tau) {
			lds_store_2x32((qcnt & (uint32_t)((D - 1) << 11)) | lq, __float_as_uint(v), item0 + (uint32_t)((e & 3) + 8 * (e >> 2)));
			qcnt += 2048u;
		}
	}
}

template <int D>
__device__ __forceinline__ void filter_queue(const f32x16 &acc, float tau, uint32_t item0, uint32_t lq, uint32_t &qcnt) {
#pragma unroll
	for (int e = 0; e < 16; ++e) filter_one<D>(acc[e], e, tau, item0, lq, qcnt);
}

// Dense splits (norm-ordered rows: the leading splits of the first stage; their rings are drained every tile): a lane that finds
// more than D survivors in ONE tile has wrapped its ring.  At k = 500 the prepass threshold lets about half of the leading tiles'
// elements through and a handful of queries did this in every call -- each one a whole-split repair by the workgroup-level kernel
// (0.26 ms per call for five queries).  The tile's accumulator is still in registers when its filter ends, and the lane's ring is
// 8 slots x 8 bytes = 16 floats: the lane overwrites its ring with the RAW accumulator (mark_wrapped_raw) and flags its count; the
// next flush filters those 16 scores itself.  Exact, no repair, and nothing but the accumulator, the ring address and the count is
// live where it happens (a direct store to the segment from inside the tile function cost the headline kernel its registers: the
// query operand went to scratch).  The ring must hold nothing but that tile's survivors, i.e. the split drains every tile.
constexpr uint32_t RING_RAW = 0x80000000u;
template <int D>
__device__ __forceinline__ void mark_wrapped_raw(const f32x16 &acc, uint32_t lq, uint32_t &qcnt) {
	static_assert(D == 8, "the raw tile (16 floats) fills the ring exactly");
	const bool wrapped = (qcnt >> 11) > (uint32_t)D && qcnt < RING_RAW;
	if (__builtin_expect(__ballot(wrapped) != 0ull, 0)) {
		if (wrapped) {
#pragma unroll
			for (int i = 0; i < D; ++i) lds_store_2x32(lq + (uint32_t)i * 2048u, __float_as_uint(acc[2 * i]), __float_as_uint(acc[2 * i + 1]));
			qcnt = RING_RAW;
		}
	}
}

// Drain the lane's ring to its HBM candidate segment: one store instruction per queue slot for the whole wave.
// raw_item0: first item (+ 4 h) of the tile a RING_RAW ring holds (dense splits only).
template <int D, bool BATCH = true>
__device__ __forceinline__ void flush_queue(uint32_t lq, uint32_t &qcnt, uint2 *__restrict__ seg, uint32_t &ncand, uint32_t capg,
											 uint32_t n_items, float tau, uint32_t raw_item0) {
	if (__builtin_expect(__ballot(qcnt >= RING_RAW) != 0ull, 0)) {
		if (qcnt >= RING_RAW) {
			for (int i = 0; i < D; ++i) {
				const uint2 w = lds_load_u64(lq + (uint32_t)i * 2048u);
#pragma unroll
				for (int half = 0; half < 2; ++half) {
					const int e = 2 * i + half;
					const float v = __uint_as_float(half ? w.y : w.x);
					const uint32_t item = raw_item0 + (uint32_t)((e & 3) + 8 * (e >> 2));
					if (v >= tau && item < n_items) {
						if (ncand < capg) seg[ncand] = make_uint2(__float_as_uint(v), item);
						ncand++;
					}
				}
			}
			qcnt = 0;
		}
	}
	uint32_t n = qcnt >> 11;  // (the filter keeps the count pre-shifted by the slot stride)
	if (__builtin_expect(__ballot(n > (uint32_t)D) != 0ull, 0)) {
		// ring wrapped between two flushes of a multi-tile window (p ~ 1e-9 per window): poison the segment count -> the select
		// kernel recomputes this query exactly
		if (n > (uint32_t)D) { ncand = 0x80000000u; n = D; }
	}
#ifdef ANNCUR_V_SERIAL_FLUSH
	constexpr bool BATCH_ = false;
#else
	constexpr bool BATCH_ = BATCH;
#endif
	if constexpr (!BATCH_) {  // (the three-workgroups-per-CU body of ANNCUR_TOPK_QT1 has no sixteen registers to spare: slot by slot)
		for (uint32_t i = 0; __ballot(i < n) != 0ull; ++i) {
			if (i < n) {
				const uint2 w = lds_load_u64(lq + i * 2048u);
				if (w.y < n_items) {
					if (ncand < capg) seg[ncand] = w;
					ncand++;
				}
			}
		}
		qcnt = 0;
		return;
	}
	// Read the occupied slots back to back, wait ONCE, then store.  (Round 2 read, waited and stored slot by slot: a drain with three
	// entries in its fullest ring paid three exposed LDS round trips -- at k = 500, where every tile is drained and the fullest of the 64
	// rings holds 3-4 entries, that was more wave time than the tile's MFMAs.)  Slot i is read if ANY lane holds more than i entries.
	unsigned long long e[D];
	int n_read = 0;  // (uniform) slots read = the fullest ring's count
#pragma unroll
	for (int i = 0; i < D; ++i) {
		if (__ballot((uint32_t)i < n) == 0ull) break;
#if defined(__HIP_DEVICE_COMPILE__)
		asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(e[i]) : "v"(lq), "n"(i * 2048));
#endif
		n_read = i + 1;
	}
#if defined(__HIP_DEVICE_COMPILE__)
	static_assert(D == 8, "the wait below names eight slots");
	// (the destinations are in flight until here; naming them as in/out operands keeps every use below the wait)
	asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7])::"memory");
#endif
#pragma unroll
	for (int i = 0; i < D; ++i) {
		if (i >= n_read) break;
		if ((uint32_t)i < n) {
			const uint2 w = make_uint2((uint32_t)e[i], (uint32_t)(e[i] >> 32));
			if (w.y < n_items) {               // (items past I exist only in the matrix' last, partial tile)
				if (ncand < capg) seg[ncand] = w;  // (capg = 0 in the no-store timing experiment)
				ncand++;                        // (a poisoned count stays > capg)
			}
		}
	}
	qcnt = 0;
}

// ---- A-fragment ring of the staggered sweep.  The LDS reads are inline asm with explicit, COUNTED s_waitcnt lgkmcnt(n):
// hipcc's own wait insertion fell back to lgkmcnt(0) in this loop (every fourth MFMA waited for a fragment requested one
// MFMA earlier).  The compiler does not see that the destination is still in flight after the asm statement; the wait
// statement takes the fragment as an in/out operand so that its consumer cannot be scheduled above the wait.
template <int OFF>
__device__ __forceinline__ void lds_read_frag(u32x4 &dst, uint32_t addr) {
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
#endif
}
__device__ __forceinline__ void lds_read_frag_at(u32x4 &dst, uint32_t addr, int off) {  // `off` folds to an immediate after unrolling
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
#endif
}
__device__ __forceinline__ void lds_wait_frag(u32x4 &frag, int pending) {  // `pending` folds to a constant after unrolling
#if defined(__HIP_DEVICE_COMPILE__)
	if (pending >= 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(frag));
	else if (pending == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(frag));
	else if (pending == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(frag));
	else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(frag));
#endif
}

// One 32-item tile of the staggered sweep (Kp <= 256): the wave's two 32-query sub-tiles run half a tile apart.  While the
// MFMA chain of one sub-tile executes on the matrix pipe, the threshold filter of the OTHER sub-tile's finished accumulator
// is issued element by element in the MFMA shadow (one accumulator register per k-step at Kp = 256):
//   steps 0..K-1   : accA (sub-tile 0 of this tile)  ||  filter of acc1 = sub-tile 1 of the PREVIOUS tile
//   steps K..2K-1  : acc1 (sub-tile 1 of this tile)  ||  filter of accA
// The A fragments (same K fragments for both halves) stream through one ring of AR registers, AR-1 steps ahead, without a
// break between the halves.  CUR = tile buffer parity (compile-time: an immediate offset of the LDS reads).
// Fragment address of k-step s (tile buffer 0): row r, chunk (2 s + h) ^ f(r).  For Kp = 256 (32 chunks per row, f(r) = r & 15) the XOR
// leaves bit 4 of the chunk alone, so k-steps s and s + 8 are 256 bytes apart: eight address registers plus an immediate offset serve
// the sixteen k-steps (the eight registers this saves are what the dynamic tile schedule's ticket lives in).
template <int KP, int CUR, bool PRED, bool INL = false>
__device__ __forceinline__ void stagger_tile(const uint32_t (&aoff)[FusedCfg<KP>::NAOFF], const bf16x8 (&xb)[2][FusedCfg<KP>::KSTEPS],
											  f32x16 &acc1, float tau0, float tau1_prev, uint32_t item0, uint32_t item0_prev,
											  uint32_t lq0, uint32_t lq1, uint32_t &q0, uint32_t &q1, bool dense) {
	using Cfg = FusedCfg<KP>;
	constexpr int K = Cfg::KSTEPS, AR = 5, DIST = 3, OFF = CUR * Cfg::TILE_BYTES;  // ring slots / prefetch distance in k-steps
	constexpr int NA = Cfg::NAOFF;
	constexpr int EPS = 16 / K > 0 ? 16 / K : 1;  // filter elements per k-step (Kp = 64: 4, 128: 2, 256: 1)
	static_assert(K <= 16 && 2 * K >= DIST && AR >= DIST + 2, "staggered path: 2..16 k-steps; a slot is rewritten two MFMAs after its use");
	static_assert(K <= NA || Cfg::CPR == 32, "k-steps beyond the address registers: + 256 bytes needs 32 chunks per row");
	u32x4 ring[AR];
#define ST_READ(slot, s) lds_read_frag_at(ring[slot], aoff[((s) % K) % NA], OFF + (((s) % K) / NA) * 256)
#pragma unroll
	for (int i = 0; i < DIST; ++i) ST_READ(i, i);
	f32x16 accA = {0}, accB = {0};
#pragma unroll
	for (int g = 0; g < 2 * K; ++g) {
		// the slot the new fragment lands in was consumed by the MFMA of step g-2: the asynchronous LDS return can never meet
		// an MFMA that is still reading its operands
		const int nxt = g + DIST;
		if (nxt < 2 * K) ST_READ(nxt % AR, nxt);
#if defined(__HIP_DEVICE_COMPILE__)
		// (keeps the register allocator from handing that read the registers of the fragment the PREVIOUS MFMA was given)
		if (g >= 1) asm volatile("" ::"v"(ring[(g - 1) % AR]));
#endif
		const int after = 2 * K - 1 - g;
		lds_wait_frag(ring[g % AR], after < DIST ? after : DIST);
		const bf16x8 a = __builtin_bit_cast(bf16x8, ring[g % AR]);
		if (g < K) {
			accA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[0][g], accA, 0, 0, 0);
#pragma unroll
			for (int e = (g == 1 ? 0 : g) * EPS; e < (g == 0 ? 0 : g + 1) * EPS; ++e)
				filter_one<Cfg::QDEPTH, PRED, INL>(acc1[e], e, tau1_prev, item0_prev, lq1, q1);
			if (g == K - 1 && dense) mark_wrapped_raw<Cfg::QDEPTH>(acc1, lq1, q1);  // (uniform) the previous tile's sub-tile 1 is filtered
		} else {
			// no filter in the first step of the half: accA's last MFMA is still in the pipe (and the predicated filter is
			// inline asm, invisible to the compiler's MFMA -> VALU hazard handling); step 1 takes two groups instead
			accB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[1][g - K], accB, 0, 0, 0);
#pragma unroll
			for (int e = (g - K == 1 ? 0 : g - K) * EPS; e < (g == K ? 0 : g - K + 1) * EPS; ++e)
				filter_one<Cfg::QDEPTH, PRED, INL>(accA[e], e, tau0, item0, lq0, q0);
		}
	}
#undef ST_READ
	if (dense) mark_wrapped_raw<Cfg::QDEPTH>(accA, lq0, q0);  // (uniform) this tile's sub-tile 0 is filtered
	acc1 = accB;
}

// Wait states between the last MFMA of an accumulate chain and the first instruction that is not the next MFMA of that chain
// (hipcc's own floor for the 8-pass 32x32x16 bf16 MFMA is 11).  hipcc pads this itself -- except where it does not: in the first
// version of stagger1_tile (one accumulator, `accP = acc` at the end of the tile) it copied the accumulator EIGHT states after the
// last MFMA on one path of the unrolled loop (MFMA, s_cbranch, v_lshl_or, s_nop 5, v_mov ...), so three registers were copied without
// the last k-step's contribution whenever instruction fetch did not happen to supply the missing states -- which depended on where
// the loop sat in memory: the shipped build passed every parity test, a diagnostic build with two more instructions in the prologue
// returned wrong scores for item rows 24-26 / 28-30 of every second tile.  scripts/check_mfma_hazards.py (run by the CPU tests on the
// built library) now walks every kernel for this; the statement below makes the distance explicit where a chain ends.
__device__ __forceinline__ void mfma_chain_done(f32x16 &acc) {
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("s_nop 7\n\ts_nop 2" : "+v"(acc));
#endif
}

// Kp = 512 (one 32-query sub-tile per wave: the query operand alone takes 128 VGPRs): the same idea across TILES -- while the
// 32-MFMA chain of tile j executes into `acc`, the filter of tile j-1's accumulator `accP` is issued in its shadow, one element every
// second k-step; A fragments through the counted-wait register ring as in stagger_tile.  The caller alternates two accumulators
// (no copy: see mfma_chain_done).  Fragment address of k-step s:
// row * CPR * 16 + ((2 s + h) ^ (r & 15)) * 16 = aoff8[s & 7] + (s >> 3) * 256 (the XOR only touches the chunk's low four bits),
// so eight address registers serve the 32 k-steps.
template <int KP, int CUR>
__device__ __forceinline__ void stagger1_tile(const uint32_t (&aoff8)[8], const bf16x8 (&xb)[FusedCfg<KP, 1>::KSTEPS], f32x16 &acc,
											   const f32x16 &accP, float tau, uint32_t item0_prev, uint32_t lq, uint32_t &qcnt, bool dense) {
	using Cfg = FusedCfg<KP, 1>;
	constexpr int K = Cfg::KSTEPS, AR = 5, DIST = 3, OFF = CUR * Cfg::TILE_BYTES;
	static_assert((K == 32 || K == 16 || K == 8) && Cfg::CPR >= 16, "Kp = 128, 256 or 512");
	u32x4 ring[AR];
#define S1_READ(slot, s) lds_read_frag_at(ring[slot], aoff8[(s) & 7], OFF + ((s) >> 3) * 256)
	S1_READ(0, 0); S1_READ(1, 1); S1_READ(2, 2);
#pragma unroll
	for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
	for (int g = 0; g < K; ++g) {
		const int nxt = g + DIST;
		if (nxt < K) S1_READ(nxt % AR, nxt);
#if defined(__HIP_DEVICE_COMPILE__)
		if (g >= 1) asm volatile("" ::"v"(ring[(g - 1) % AR]));
#endif
		const int after = K - 1 - g;
		lds_wait_frag(ring[g % AR], after < DIST ? after : DIST);
		acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[g % AR]), xb[g], acc, 0, 0, 0);
		if constexpr (K == 32) {  // one element every second k-step
			if (g & 1) filter_one<Cfg::QDEPTH>(accP[g >> 1], g >> 1, tau, item0_prev, lq, qcnt);
		} else {                  // 16 / K elements per k-step
#pragma unroll
			for (int e = g * (16 / K); e < (g + 1) * (16 / K); ++e) filter_one<Cfg::QDEPTH>(accP[e], e, tau, item0_prev, lq, qcnt);
		}
	}
#undef S1_READ
	mfma_chain_done(acc);
	if (dense) mark_wrapped_raw<Cfg::QDEPTH>(accP, lq, qcnt);  // (uniform) the previous tile is filtered
}

// MODE 0: prepass (GROUP = 16 or 4 items per group maximum).  MODE 1: filter sweep (PRED: branch-free filter, for stages in
// which most compares find a survivor in some lane -- large k).
template <int KP, int MODE, int GROUP, bool PRED = false, bool INL = false, int QTV = FusedCfg<KP>::QT>
__global__ __launch_bounds__(256, (QTV == 1 && KP <= 256) ? 3 : 2) void score_kernel(const FusedParams p) {
	using Cfg = FusedCfg<KP, QTV>;
	constexpr int KSTEPS = Cfg::KSTEPS, QT = Cfg::QT, CPR = Cfg::CPR;
	constexpr bool FLUSH_BATCH = !(QTV == 1 && KP <= 256);   // (see flush_queue)
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int r = lane & 31, h = lane >> 5;
#ifdef ANNCUR_TIMING_EXPERIMENTS
	unsigned long long st_entry = 0;
	if (MODE == 1) { st_entry = __builtin_amdgcn_s_memrealtime(); asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_entry)::"memory"); }
#endif
	const int wid = xcd_remap(blockIdx.x, p.n_wg);
	const int nsplit = (MODE == 0) ? p.S0 : p.S;  // MODE 2 = MODE 1 without the filter (debug timing)
	const int n_rb_ = (int)((p.Q + Cfg::BQ - 1) / Cfg::BQ);
	const bool rbm = MODE == 1 && p.rb_major;   // (uniform) tickets: row-block-major work ids, see score16.hpp
	const int split = rbm ? wid % p.S : wid / n_rb_;
	const int rb = rbm ? wid / p.S : wid - split * n_rb_;
	(void)nsplit;

	// ---- this wave's queries: B operand fragments, resident for the whole kernel
	bf16x8 xb[QT][KSTEPS];
	int64_t qv[QT];
#pragma unroll
	for (int t = 0; t < QT; ++t) {
		qv[t] = (int64_t)rb * Cfg::BQ + wave * 32 * QT + 32 * t + r;
		const bool ok = qv[t] < p.Q;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(p.X + (ok ? qv[t] : 0) * p.ldx) + h;
#pragma unroll
		for (int s = 0; s < KSTEPS; ++s) {
			const u32x4 zero = {0u, 0u, 0u, 0u};
			const u32x4 w = ok ? src[2 * s] : zero;
			xb[t][s] = __builtin_bit_cast(bf16x8, w);
		}
	}

	// The X fragments must be complete before the tile loop: otherwise hipcc's wait-count merge at the loop header
	// (fragment loads possibly pending + an unknown number of candidate stores on the back edge) makes it wait
	// vmcnt(0) in front of the first MFMA of EVERY tile, i.e. right after issuing the next tile's loads.
	__builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only

	// ---- work range
	int j_begin, j_end;  // tile iterations
	if (MODE == 0) { j_begin = split * p.st_per_split; j_end = min(j_begin + p.st_per_split, p.n_st); }
	else if (MODE == 1 && p.tile_step > 1) { j_begin = p.tile_begin + split; j_end = p.tile_end; }                          // interleaved splits
	else { j_begin = p.tile_begin + split * p.tiles_per_split; j_end = min(j_begin + p.tiles_per_split, p.tile_end); }  // MODE 1 and 2
	const int ts = (MODE == 1 && p.tile_step > 1) ? p.tile_step : 1;  // distance between two tiles of this workgroup
	// Norm-ordered rows: the survivors crowd into the leading tiles (several times the average density: about half of their elements
	// pass the prepass threshold at large k).  The first quarter of the first stage's tiles is swept with the rings drained every
	// tile (a ring then holds nothing but one tile: see mark_wrapped_raw); a step drains when the tile before it was such a tile.
	const int dense_end = (MODE == 1 && p.sample_leading && !p.carry) ? p.tile_begin + (p.tile_end - p.tile_begin + 3) / 4 : p.tile_begin;
	const bool every_tile = p.flush_tiles <= 1;  // (uniform) the whole stage drains every tile (large k): every ring holds one tile
#define tile_of(j) ((MODE == 0 && !p.sample_leading) ? (int)(((int64_t)(j) * p.n_full_tiles) / p.n_st) : (j))

	float tau[QT];
	uint32_t ncand[QT], qcnt[QT];
#pragma unroll
	for (int t = 0; t < QT; ++t) {
		tau[t] = (MODE == 1 && qv[t] < p.Q) ? p.tau[qv[t] * p.tau_stride] + p.tau_bias : INFINITY;
		ncand[t] = (MODE == 1 && p.carry && qv[t] < p.Q) ? p.seg_cnt[qv[t] * p.nseg + h * p.S + split] : 0u;
		qcnt[t] = 0;
	}
	// candidate segment of (query, lane half, split); sub-tile t adds a wave-uniform stride
	uint2 *seg0 = p.cand + (qv[0] * p.nseg + h * p.S + split) * (int64_t)p.capg;
	const int64_t seg_dt = (int64_t)32 * p.nseg * p.capg;
	const uint32_t lq0 = lds_addr(smem + Cfg::QUEUE_OFF) + (uint32_t)tid * 8u;  // slot i of sub-tile t at byte lq0 + (t*QDEPTH + i)*2048
	static_assert(Cfg::QUEUE_OFF % (Cfg::QDEPTH * 2048) == 0 && Cfg::QDEPTH * 2048 == 16384, "ring must be 16 KiB aligned");
	if (MODE == 1 && (lds_addr(smem) & 0x3fffu) != 0u) __builtin_trap();  // filter_one() ORs the slot offset into the address

	uint32_t last_item0 = 0;  // first item (+ 4 h) of the last tile filtered: what a raw ring holds at the final flush
	// ---- tile schedule of the staggered sweep (wave-uniform scalars).  Static: tiles j_begin, j_begin + ts, ... below j_end.
	// Dynamic (p.chunk_tiles > 0): the workgroups of a query row block draw chunks of p.chunk_tiles consecutive tiles from the row
	// block's ticket counter, one chunk ahead of the one they sweep.  Why: the 480 workgroups of a cfg2 stage do not run at one speed --
	// in-kernel stamps (round 2) put the ends of their tile loops between 193 and 271 us (median 234) for equal shares: 32 CUs hold one
	// workgroup instead of two, the others differ by XCD / CU placement -- and a launch lasts as long as its slowest workgroup.  With tickets
	// a fast workgroup simply sweeps more chunks.  Any assignment is exact: a workgroup's survivors go to its own segments whatever tiles
	// it swept; the repair path finds a split's tiles in p.chunk_owner.  Thread 0 draws a ticket when the workgroup starts the chunk
	// BEFORE the one the ticket is for (atomic in flight during a whole tile), hands it over through LDS at that tile's barrier.
	int t_cur = j_begin < j_end ? j_begin : -1, t_cend = j_end, t_step = ts, t_next_chunk = -1, t_prev = -1;
	bool ticket_pending = false;
	constexpr bool STAG = MODE == 1 && (QT == 2 || (KP == 512 && !INL));   // the bodies that take the ticket schedule and the hand-placed DMA
	const bool dyn = STAG && KP < 512 && p.chunk_tiles > 0;
	// two LDS words used in turn: a ticket is published at the end of one step and read at the head of the next, and nothing but program
	// order separates that read from the NEXT publication (possible one step later when a chunk is a single tile)
	uint32_t ticket_slot = lds_addr(smem + Cfg::TICKET_OFF);
	if constexpr (STAG) {
		if (dyn) {
			if (tid == 0) {
				const uint32_t c = atomicAdd(p.chunk_ctr + rb, 2u);   // the first chunk and the look-ahead
				if (p.chunk_owner) {
					if (c < (uint32_t)p.n_chunks) p.chunk_owner[(int64_t)rb * p.n_chunks + c] = (uint8_t)split;
					if (c + 1 < (uint32_t)p.n_chunks) p.chunk_owner[(int64_t)rb * p.n_chunks + c + 1] = (uint8_t)split;
				}
				lds_store_u32(ticket_slot, c);
				__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the word is in LDS before this wave reaches the barrier
			}
			__syncthreads();
			const uint32_t c = lds_load_u32_uniform(ticket_slot);
			t_step = 1;
			t_cur = c < (uint32_t)p.n_chunks ? p.tile_begin + (int)c * p.chunk_tiles : -1;
			t_cend = min(t_cur + p.chunk_tiles, p.tile_end);
			t_next_chunk = c + 1 < (uint32_t)p.n_chunks ? p.tile_begin + (int)(c + 1) * p.chunk_tiles : -1;
		}
	}
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);             // (the compiler does not know tid >> 6 is wave-uniform)
	const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(smem));
	uint32_t dma_off[STAG ? Cfg::TILE_BYTES / 4096 : 1];
	if constexpr (STAG) {
		tile_dma_offsets<KP>(dma_off, wave_u, lane);
#ifdef ANNCUR_V_OLD_DMA
		if (t_cur >= 0) tile_dma<KP>(p.Et, t_cur, smem, wave, lane);
#else
		if (t_cur >= 0) tile_dma_s<KP>(p.Et, t_cur, lds_base, wave_u, dma_off);
#endif
	} else if (j_begin < j_end) tile_dma<KP>(p.Et, tile_of(j_begin), smem, wave, lane);
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__syncthreads();
	ANNCUR_PAD_HERE();
#ifdef ANNCUR_TIMING_EXPERIMENTS
	// In-kernel clock (MI355X guide, 'DVFS give-back' (6)): shader cycles per 100 MHz reference tick around the tile loop.  The
	// stamps go to a buffer nothing else reads; the shipped library contains none of this.
	// (this diagnostic is how the MFMA -> v_mov hazard of the Kp = 512 loop was found: the extra instructions moved the loop by 8 bytes
	//  and the scores changed -- see mfma_chain_done())
	unsigned long long st_c0 = 0, st_r0 = 0;
	if (MODE == 1) {
		st_c0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime();
		asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st_c0), "+s"(st_r0)::"memory");
	}
#endif

	if constexpr (MODE == 1 && QT == 2) {
		// ---- staggered sweep (stagger_tile): tile loop unrolled by two so that the LDS buffer parity is a compile-time offset
		f32x16 acc1;
#pragma unroll
		for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
		float tau1_prev = INFINITY;  // no previous tile yet: the filter of acc1 never fires
		uint32_t item0_prev = 0, item0_pp = 0;   // first item (+ 4 h) of the previous tile and of the one before it
		const uint32_t lq1 = lq0 + Cfg::QDEPTH * 2048;
		uint32_t aoff[Cfg::NAOFF];   // LDS byte address of this lane's A fragment of k-step s (< NAOFF) in tile buffer 0
#pragma unroll
		for (int s = 0; s < Cfg::NAOFF; ++s) aoff[s] = lds_addr(smem) + (uint32_t)(r * CPR + swz<CPR>(r, 2 * s + h)) * 16u;
		int flush_in2 = p.flush_tiles;
		// stagger_tile() counts LDS reads with lgkmcnt(n): no scalar load of the prologue may still be in flight (scalar loads share
		// the counter and return out of order)
		__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) only
#ifdef ANNCUR_V_OLD_DMA   /* A/B variant builds (make variant V=...): one feature of the round-3 sweep switched back */
#define STAGGER_DMA(NX, CUR) tile_dma<KP>(p.Et, (NX), smem + ((CUR) ^ 1) * Cfg::TILE_BYTES, wave, lane)
#else
#define STAGGER_DMA(NX, CUR) tile_dma_s<KP>(p.Et, (NX), lds_base + ((CUR) ^ 1) * Cfg::TILE_BYTES, wave_u, dma_off)
#endif
// (Round 3 tried draining by occupancy as well -- a ballot "some lane's ring holds >= 3 / >= 4 entries" per step, beside or instead of the
//  planned window: both cost 4 % of the sweep at cfg2, same box, alternating -- the check itself, not the drains.  The planned window stays.)
#ifdef ANNCUR_TIMING_EXPERIMENTS
		// phase stamps (diagnostic build): shader cycles this WAVE spent per step in {ticket read + DMA issue, ring drain, the tile's MFMA / filter
		// section, the vmcnt wait, the barrier}; summed over the loop, one record per wave (anncur_debug_sweep_phases, scripts/sweep_phases.py)
		uint32_t ph_acc[5] = {0u, 0u, 0u, 0u, 0u};
		uint32_t ph_t = (uint32_t)__builtin_amdgcn_s_memtime();
#define PH(i) do { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); ph_acc[i] += now_ - ph_t; ph_t = now_; } while (0)
#else
#define PH(i) do { } while (0)
#endif
#define STAGGER_STEP(CUR)                                                                                                       \
		do {                                                                                                                    \
			const int J = t_cur;                                                                                                \
			if (ticket_pending) {  /* (uniform) the ticket thread 0 drew during the previous step */                            \
				const uint32_t c = lds_load_u32_uniform(ticket_slot);                                                           \
				t_next_chunk = c < (uint32_t)p.n_chunks ? p.tile_begin + (int)c * p.chunk_tiles : -1;                           \
				ticket_pending = false;                                                                                         \
				ticket_slot ^= 4u;  /* (TICKET_OFF is 16-byte aligned) */                                                       \
			}                                                                                                                   \
			int nx = J + t_step;                                                                                                \
			bool crossed = false;  /* (uniform) the next tile opens the look-ahead chunk: draw the ticket of the one after it */ \
			if (nx >= t_cend) { nx = t_next_chunk; crossed = dyn && nx >= 0; }                                                  \
			if (nx >= 0) STAGGER_DMA(nx, CUR);                                                                                  \
			uint32_t ticket = 0;                                                                                                \
			if (crossed && tid == 0) ticket_draw(ticket, p.chunk_ctr + rb);  /* in flight until ticket_wait() below */          \
			PH(0);                                                                                                              \
			if (t_prev < dense_end || --flush_in2 <= 0) {                                                                       \
				flush_in2 = p.flush_tiles;                                                                                      \
				/* (a raw ring holds one tile: sub-tile 0 of the previous tile, sub-tile 1 of the one before) */                \
				flush_queue<Cfg::QDEPTH>(lq0, qcnt[0], seg0, ncand[0], (uint32_t)p.capg, (uint32_t)p.I, tau[0], item0_prev);    \
				flush_queue<Cfg::QDEPTH>(lq1, qcnt[1], seg0 + seg_dt, ncand[1], (uint32_t)p.capg, (uint32_t)p.I, tau[1], item0_pp); \
			}                                                                                                                   \
			const uint32_t item0 = (uint32_t)J * TILE_I + 4 * h;                                                                \
			PH(1);                                                                                                              \
			stagger_tile<KP, CUR, PRED, INL>(aoff, xb, acc1, tau[0], tau1_prev, item0, item0_prev, lq0, lq1, qcnt[0], qcnt[1], J < dense_end || every_tile); \
			tau1_prev = tau[1]; item0_pp = item0_prev; item0_prev = item0;                                                      \
			PH(2);                                                                                                              \
			ticket_wait(ticket);  /* vmcnt(0): the DMA of the next tile, the queue stores issued with it and the ticket have landed */ \
			PH(3);                                                                                                              \
			if (crossed) {                                                                                                      \
				if (tid == 0) {                                                                                                 \
					lds_store_u32(ticket_slot, ticket);                                                                         \
					if (p.chunk_owner && ticket < (uint32_t)p.n_chunks) p.chunk_owner[(int64_t)rb * p.n_chunks + ticket] = (uint8_t)split; \
					__builtin_amdgcn_s_waitcnt(0xC07F);                                                                         \
				}                                                                                                               \
				ticket_pending = true;                                                                                          \
				t_cend = min(nx + p.chunk_tiles, p.tile_end);                                                                   \
			}                                                                                                                   \
			__syncthreads();                                                                                                    \
			PH(4);                                                                                                              \
			t_prev = J; t_cur = nx;                                                                                             \
		} while (0)
		while (t_cur >= 0) {
			STAGGER_STEP(0);
			if (t_cur < 0) break;
			STAGGER_STEP(1);
		}
#undef STAGGER_STEP
#ifdef ANNCUR_TIMING_EXPERIMENTS
		if (lane == 0 && d_sweep_stamps && p.debug_stamp && blockIdx.x * 4 + wave < 8192) {
			unsigned long long *ph = d_sweep_stamps + 5 * 8192 + (size_t)(blockIdx.x * 4 + wave) * 8;
			for (int i = 0; i < 5; ++i) ph[i] = ph_acc[i];
		}
#endif
#undef PH
		flush_queue<Cfg::QDEPTH>(lq1, qcnt[1], seg0 + seg_dt, ncand[1], (uint32_t)p.capg, (uint32_t)p.I, tau[1], item0_pp);  // keep one tile's hits per queue window
#pragma unroll
		for (int e = 0; e < 16; ++e)  // drain: sub-tile 1 of the last tile
			filter_one<Cfg::QDEPTH>(acc1[e], e, tau1_prev, item0_prev, lq1, qcnt[1]);
		mark_wrapped_raw<Cfg::QDEPTH>(acc1, lq1, qcnt[1]);  // (its ring was just drained: exact in every split)
		last_item0 = item0_prev;
	} else if constexpr (STAG && QT == 1) {
		// ---- software-pipelined sweep for Kp = 512 (stagger1_tile) on the ticket schedule: tile loop unrolled by two (buffer parity =
		// immediate offset); even steps accumulate into accA and filter accB, odd steps the other way round
		f32x16 accA, accB;
#pragma unroll
		for (int e = 0; e < 16; ++e) { accA[e] = 0.f; accB[e] = 0.f; }
		float tau_prev = INFINITY;   // no previous tile yet: the filter never fires
		uint32_t item0_prev = 0, item0_pp = 0;
		uint32_t aoff8[8];
#pragma unroll
		for (int s = 0; s < 8; ++s) aoff8[s] = lds_addr(smem) + (uint32_t)(r * CPR + ((2 * s + h) ^ (r & 15))) * 16u;
		int flush_in2 = p.flush_tiles;
		__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): stagger1_tile() counts LDS reads
#define STAGGER1_STEP(CUR, ACC, ACCP)                                                                                           \
		do {                                                                                                                    \
			const int J = t_cur;                                                                                                \
			if (ticket_pending) {                                                                                               \
				const uint32_t c = lds_load_u32_uniform(ticket_slot);                                                           \
				t_next_chunk = c < (uint32_t)p.n_chunks ? p.tile_begin + (int)c * p.chunk_tiles : -1;                           \
				ticket_pending = false;                                                                                         \
				ticket_slot ^= 4u;                                                                                              \
			}                                                                                                                   \
			int nx = J + t_step;                                                                                                \
			bool crossed = false;                                                                                               \
			if (nx >= t_cend) { nx = t_next_chunk; crossed = dyn && nx >= 0; }                                                  \
			if (nx >= 0) STAGGER_DMA(nx, CUR);                                                                                  \
			uint32_t ticket = 0;                                                                                                \
			if (crossed && tid == 0) ticket_draw(ticket, p.chunk_ctr + rb);                                                     \
			if (t_prev < dense_end || --flush_in2 <= 0) {                                                                       \
				flush_in2 = p.flush_tiles;                                                                                      \
				/* (a raw ring holds the tile before the previous one, filtered during the previous step) */                    \
				flush_queue<Cfg::QDEPTH, FLUSH_BATCH>(lq0, qcnt[0], seg0, ncand[0], (uint32_t)p.capg, (uint32_t)p.I, tau[0], item0_pp); \
			}                                                                                                                   \
			stagger1_tile<KP, CUR>(aoff8, xb[0], ACC, ACCP, tau_prev, item0_prev, lq0, qcnt[0], J < dense_end || every_tile);   \
			tau_prev = tau[0]; item0_pp = item0_prev; item0_prev = (uint32_t)J * TILE_I + 4 * h;                                \
			ticket_wait(ticket);                                                                                                \
			if (crossed) {                                                                                                      \
				if (tid == 0) {                                                                                                 \
					lds_store_u32(ticket_slot, ticket);                                                                         \
					if (p.chunk_owner && ticket < (uint32_t)p.n_chunks) p.chunk_owner[(int64_t)rb * p.n_chunks + ticket] = (uint8_t)split; \
					__builtin_amdgcn_s_waitcnt(0xC07F);                                                                         \
				}                                                                                                               \
				ticket_pending = true;                                                                                          \
				t_cend = min(nx + p.chunk_tiles, p.tile_end);                                                                   \
			}                                                                                                                   \
			__syncthreads();                                                                                                    \
			t_prev = J; t_cur = nx;                                                                                             \
		} while (0)
		bool last_in_a = false;  // (uniform) which accumulator holds the last tile
		while (t_cur >= 0) {
			STAGGER1_STEP(0, accA, accB);
			last_in_a = true;
			if (t_cur < 0) break;
			STAGGER1_STEP(1, accB, accA);
			last_in_a = false;
		}
#undef STAGGER1_STEP
#undef STAGGER_DMA
		flush_queue<Cfg::QDEPTH, FLUSH_BATCH>(lq0, qcnt[0], seg0, ncand[0], (uint32_t)p.capg, (uint32_t)p.I, tau[0], item0_pp);
		if (last_in_a) {  // drain: the last tile (its ring was just drained: the raw path is exact in every split)
#pragma unroll
			for (int e = 0; e < 16; ++e) filter_one<Cfg::QDEPTH>(accA[e], e, tau_prev, item0_prev, lq0, qcnt[0]);
			mark_wrapped_raw<Cfg::QDEPTH>(accA, lq0, qcnt[0]);
		} else {
#pragma unroll
			for (int e = 0; e < 16; ++e) filter_one<Cfg::QDEPTH>(accB[e], e, tau_prev, item0_prev, lq0, qcnt[0]);
			mark_wrapped_raw<Cfg::QDEPTH>(accB, lq0, qcnt[0]);
		}
		last_item0 = item0_prev;
	} else if constexpr (MODE == 1 && QT == 1 && KP >= 128 && !INL) {  // (INL = the plain loop, kept for A/B in the experiments build)
		// ---- software-pipelined sweep for Kp = 512 (stagger1_tile): tile loop unrolled by two (buffer parity = immediate offset);
		// even steps accumulate into accA and filter accB, odd steps the other way round
		f32x16 accA, accB;
#pragma unroll
		for (int e = 0; e < 16; ++e) { accA[e] = 0.f; accB[e] = 0.f; }
		float tau_prev = INFINITY;   // no previous tile yet: the filter never fires
		uint32_t item0_prev = 0;
		uint32_t aoff8[8];
#pragma unroll
		for (int s = 0; s < 8; ++s) aoff8[s] = lds_addr(smem) + (uint32_t)(r * CPR + ((2 * s + h) ^ (r & 15))) * 16u;
		int flush_in2 = p.flush_tiles;
		__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): stagger1_tile() counts LDS reads
#define STAGGER1_STEP(CUR, J, ACC, ACCP)                                                                                        \
		do {                                                                                                                    \
			if ((J) + ts < j_end) tile_dma<KP>(p.Et, (J) + ts, smem + ((CUR) ^ 1) * Cfg::TILE_BYTES, wave, lane);               \
			if ((J) - ts < dense_end || --flush_in2 <= 0) {                                                                     \
				flush_in2 = p.flush_tiles;                                                                                      \
				/* (a raw ring holds the tile before the previous one, filtered during the previous step) */                    \
				flush_queue<Cfg::QDEPTH, FLUSH_BATCH>(lq0, qcnt[0], seg0, ncand[0], (uint32_t)p.capg, (uint32_t)p.I, tau[0], item0_prev - (uint32_t)ts * TILE_I); \
			}                                                                                                                   \
			stagger1_tile<KP, CUR>(aoff8, xb[0], ACC, ACCP, tau_prev, item0_prev, lq0, qcnt[0], (J) < dense_end || every_tile);               \
			tau_prev = tau[0]; item0_prev = (uint32_t)(J) * TILE_I + 4 * h;                                                     \
			__builtin_amdgcn_s_waitcnt(0x0F70);                                                                                 \
			__syncthreads();                                                                                                    \
		} while (0)
		bool last_in_a = false;  // (uniform) which accumulator holds the last tile
		for (int j = j_begin; j < j_end; j += 2 * ts) {
			STAGGER1_STEP(0, j, accA, accB);
			last_in_a = true;
			if (j + ts < j_end) { STAGGER1_STEP(1, j + ts, accB, accA); last_in_a = false; }
		}
#undef STAGGER1_STEP
		flush_queue<Cfg::QDEPTH, FLUSH_BATCH>(lq0, qcnt[0], seg0, ncand[0], (uint32_t)p.capg, (uint32_t)p.I, tau[0], item0_prev - (uint32_t)ts * TILE_I);
		if (last_in_a) {  // drain: the last tile (its ring was just drained: the raw path is exact in every split)
#pragma unroll
			for (int e = 0; e < 16; ++e) filter_one<Cfg::QDEPTH>(accA[e], e, tau_prev, item0_prev, lq0, qcnt[0]);
			mark_wrapped_raw<Cfg::QDEPTH>(accA, lq0, qcnt[0]);
		} else {
#pragma unroll
			for (int e = 0; e < 16; ++e) filter_one<Cfg::QDEPTH>(accB[e], e, tau_prev, item0_prev, lq0, qcnt[0]);
			mark_wrapped_raw<Cfg::QDEPTH>(accB, lq0, qcnt[0]);
		}
		last_item0 = item0_prev;
	} else {
	int flush_in = p.flush_tiles, cur = 0;
	for (int j = j_begin; j < j_end; j += ts, cur ^= 1) {
		const int tile = tile_of(j);
		const bool more = j + ts < j_end;
		if (more && MODE != 3) tile_dma<KP>(p.Et, tile_of(j + ts), smem + (cur ^ 1) * Cfg::TILE_BYTES, wave, lane);
		bool ring_fresh = false;  // (uniform) the rings were drained at the head of this step: they will hold nothing but this tile
		if (MODE == 1) {
			// drain the hit queues of the previous tiles right behind the DMA: the stores get the whole MFMA phase to retire
			// (always behind a dense tile and at the head of the last step: see mark_wrapped_raw below)
			if (j - ts < dense_end || --flush_in <= 0 || !more) {
				flush_in = p.flush_tiles;
				ring_fresh = true;
#pragma unroll
				for (int t = 0; t < QT; ++t) flush_queue<Cfg::QDEPTH>(lq0 + t * Cfg::QDEPTH * 2048, qcnt[t], seg0 + t * seg_dt, ncand[t], (uint32_t)p.capg, (uint32_t)p.I, tau[t], last_item0);
			}
		}

		f32x16 acc[QT];
#pragma unroll
		for (int t = 0; t < QT; ++t)
#pragma unroll
			for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
		const u32x4 *tb = reinterpret_cast<const u32x4 *>(smem + cur * Cfg::TILE_BYTES);
#pragma unroll
		for (int s = 0; s < KSTEPS; ++s) {
			const u32x4 w = tb[r * CPR + swz<CPR>(r, 2 * s + h)];
			const bf16x8 a = __builtin_bit_cast(bf16x8, w);
#pragma unroll
			for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[t][s], acc[t], 0, 0, 0);
		}

		// ---- epilogue.  C/D layout: query = lane & 31 (col), item row = (e & 3) + 8 * (e >> 2) + 4 * h
		if (MODE == 0) {
#pragma unroll
			for (int t = 0; t < QT; ++t) {
				if (GROUP == 16) {
					float m = acc[t][0];
#pragma unroll
					for (int e = 1; e < 16; ++e) m = fmaxf(m, acc[t][e]);
					if (qv[t] < p.Q) p.gmax[qv[t] * p.n_groups + (int64_t)j * 2 + h] = m;
				} else {
					float4 m;
					m.x = fmaxf(fmaxf(acc[t][0], acc[t][1]), fmaxf(acc[t][2], acc[t][3]));
					m.y = fmaxf(fmaxf(acc[t][4], acc[t][5]), fmaxf(acc[t][6], acc[t][7]));
					m.z = fmaxf(fmaxf(acc[t][8], acc[t][9]), fmaxf(acc[t][10], acc[t][11]));
					m.w = fmaxf(fmaxf(acc[t][12], acc[t][13]), fmaxf(acc[t][14], acc[t][15]));
					if (qv[t] < p.Q) *reinterpret_cast<float4 *>(p.gmax + qv[t] * p.n_groups + ((int64_t)j * 2 + h) * 4) = m;
				}
			}
		} else if (MODE == 2 || MODE == 3) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
			for (int t = 0; t < QT; ++t) asm volatile("" ::"v"(acc[t]));  // timing experiment: GEMM + staging only
#endif
		} else {
			const uint32_t item0 = (uint32_t)tile * TILE_I + 4 * h;
#pragma unroll
			for (int t = 0; t < QT; ++t) {
				filter_queue<Cfg::QDEPTH>(acc[t], tau[t], item0, lq0 + t * Cfg::QDEPTH * 2048, qcnt[t]);
				// (uniform) a raw ring must be drained before the next tile's filter: the next step drains when this tile is dense; the
				// last tile is followed by the final drain
				if (ring_fresh && (j < dense_end || !more || every_tile)) mark_wrapped_raw<Cfg::QDEPTH>(acc[t], lq0 + t * Cfg::QDEPTH * 2048, qcnt[t]);
			}
			last_item0 = item0;
		}
		if (MODE != 3) {
			__builtin_amdgcn_s_waitcnt(0x0F70);  // the DMA of tile j+1 (and the queue stores issued with it) have landed
			__syncthreads();
		}
	}
	}  // plain loop

#undef tile_of
#ifdef ANNCUR_TIMING_EXPERIMENTS
	if (MODE == 1 && tid == 0) {
		unsigned long long *stamps = d_sweep_stamps;
		if (stamps && p.debug_stamp && blockIdx.x < 8192) {
			const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
			stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c0;
			stamps[2 * blockIdx.x + 1] = r1 - st_r0;
			stamps[2 * 8192 + 3 * blockIdx.x] = st_entry; stamps[2 * 8192 + 3 * blockIdx.x + 1] = st_r0; stamps[2 * 8192 + 3 * blockIdx.x + 2] = r1;
		}
	}
#endif
	if (MODE == 1) {
#pragma unroll
		for (int t = 0; t < QT; ++t) {
			flush_queue<Cfg::QDEPTH, FLUSH_BATCH>(lq0 + t * Cfg::QDEPTH * 2048, qcnt[t], seg0 + t * seg_dt, ncand[t], (uint32_t)p.capg, (uint32_t)p.I, tau[t], last_item0);
			if (qv[t] < p.Q) p.seg_cnt[qv[t] * p.nseg + h * p.S + split] = ncand[t];
		}
	}
}

// ---- XCD-sliced tickets (round 4).  Under round 3's schedule a row block's workgroups -- spread over all eight XCDs -- drew their chunks from
// ONE counter per row block: every XCD ended up reading every tile of the stage (cfg4's per-GPU shape: 8.9 GB of L2 -> fabric traffic per
// launch against 1.0 GB of E^T, 5.3 TB/s).  Now the stage's chunks are cut into eight contiguous slices, one per XCD, each with its own
// counter per row block: a workgroup draws from the slice of the XCD it RUNS on (s_getreg XCC_ID -- measured, never assumed: placement is
// speed only) and, once that slice is exhausted, steals from the following ones.  All the row blocks resident on an XCD walk the same
// slice at about the same pace, so a tile crosses the fabric about once (plus the steals) instead of once per XCD, and the schedule
// stays dynamic: a slow workgroup still draws fewer chunks, a fast one moves on to help its neighbours.  Any assignment is exact
// (chunk ids stay global: the owner map and the repair path are unchanged).
constexpr int N_SLICES = 8;
__device__ __forceinline__ uint32_t slice_len(int n_chunks, int cps, int s) {
	const int b = s * cps, e = min(b + cps, n_chunks);
	return e > b ? (uint32_t)(e - b) : 0u;
}
// local ticket `l` drawn from slice `slice` of this row block -> global chunk id, or n_chunks when every slice is exhausted.  Steals (a
// blocking atomic per exhausted slice, at most seven per workgroup and stage) happen here; `tried` = slices this workgroup found exhausted.
__device__ __forceinline__ uint32_t slice_resolve(uint32_t l, uint32_t *ctr_rb, int n_chunks, int cps, int &slice, int &tried) {
	for (;;) {
		if (l < slice_len(n_chunks, cps, slice)) return (uint32_t)(slice * cps) + l;
		if (++tried >= N_SLICES) return (uint32_t)n_chunks;
		slice = (slice + 1) & (N_SLICES - 1);
		l = atomicAdd(ctr_rb + slice, 1u);
	}
}
__device__ __forceinline__ int xcc_id() {
	uint32_t x = 0;
#if defined(__HIP_DEVICE_COMPILE__)
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
#endif
	return (int)(x & (N_SLICES - 1));
}

#include "score16.hpp"
#include "score16r.hpp"
#include "score_q1.hpp"
#include "score_q16.hpp"

// ------------------------------------------------------------------ a11: approximation error on the same MFMA loop
// err_sq[q] += sum_i (S_hat[q,i] - A[q,i])^2, norm_sq[q] += sum_i A[q,i]^2 over this workgroup's item tiles; S_hat is never
// written.  Same orientation as the sweep: lane = query, so both sums are lane-local accumulators and the exact matrix is read
// as 4 consecutive items (8 or 16 bytes) per lane and accumulator register group; the next tile's exact values are in flight
// during the MFMAs.  One atomicAdd pair per lane at the end (2 S per query).
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4n __attribute__((ext_vector_type(4)));
template <typename TA> struct ExactQuad;  // four consecutive items of one row of the exact matrix (native vectors: asm operands)
template <> struct ExactQuad<uint16_t> {
	u32x2 w;
	__device__ __forceinline__ void load(const uint16_t *p) { w = *reinterpret_cast<const u32x2 *>(p); }
	__device__ __forceinline__ float get(int c) const {
		const uint32_t x = (c & 2) ? w[1] : w[0];
		return __uint_as_float((c & 1) ? (x & 0xffff0000u) : (x << 16));
	}
};
template <> struct ExactQuad<float> {
	f32x4n w;
	__device__ __forceinline__ void load(const float *p) { w = *reinterpret_cast<const f32x4n *>(p); }
	__device__ __forceinline__ float get(int c) const { return w[c]; }
};

// The tile's MFMA chains with the A fragments through the counted-wait register ring of the sweep (three k-steps ahead, inline-asm
// ds_read_b128, s_waitcnt lgkmcnt(n)): round 2's loop read two k-steps, waited lgkmcnt(0), issued four MFMAs -- eight exposed LDS
// round trips per tile (0.24 of peak).  Both sub-tiles share the fragment.
template <int KP, int CUR>
__device__ __forceinline__ void error_mfma_tile(const uint32_t (&aoff)[FusedCfg<KP>::NAOFF], const bf16x8 (&xb)[FusedCfg<KP>::QT][FusedCfg<KP>::KSTEPS],
												 f32x16 (&acc)[FusedCfg<KP>::QT]) {
	using Cfg = FusedCfg<KP>;
	constexpr int K = Cfg::KSTEPS, NA = Cfg::NAOFF, QT = Cfg::QT, AR = 5, DIST = 3, OFF = CUR * Cfg::TILE_BYTES;
	static_assert(K > DIST, "at least four k-steps");
	u32x4 ring[AR];
#define ER_READ(slot, s) lds_read_frag_at(ring[slot], aoff[(s) % NA], OFF + ((s) / NA) * 256)
	ER_READ(0, 0); ER_READ(1, 1); ER_READ(2, 2);
#pragma unroll
	for (int t = 0; t < QT; ++t)
#pragma unroll
		for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
	for (int g = 0; g < K; ++g) {
		const int nxt = g + DIST;
		if (nxt < K) ER_READ(nxt % AR, nxt);
#if defined(__HIP_DEVICE_COMPILE__)
		if (g >= 1) asm volatile("" ::"v"(ring[(g - 1) % AR]));
#endif
		const int after = K - 1 - g;
		lds_wait_frag(ring[g % AR], after < DIST ? after : DIST);
		const bf16x8 a = __builtin_bit_cast(bf16x8, ring[g % AR]);
#pragma unroll
		for (int t = 0; t < QT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xb[t][g], acc[t], 0, 0, 0);
	}
#undef ER_READ
#pragma unroll
	for (int t = 0; t < QT; ++t) mfma_chain_done(acc[t]);
}

// a11 with the exact matrix staged through LDS (bf16 exact matrix, rows 16-byte aligned: lda % 8 == 0).  error_kernel reads the exact
// values with the accumulator's orientation -- lane = query -- i.e. 8 bytes per lane from 32 different rows per load instruction:
// 2.5 TB/s of exact matrix at cfg2 size whatever the MFMA loop does (round 3: the counted-wait fragment ring alone moved 0.84 -> 0.81 ms;
// the time did not scale with Kp either: 0.60 ms at Kp = 128).  Here the workgroup's tile of the exact matrix -- BQ rows x 64 bytes --
// comes in by the same direct-to-LDS loads as the item tile (16 bytes per lane, four lanes per row, 16 rows per instruction), double
// buffered, one tile ahead; the lanes then read their four quads per sub-tile from LDS (ds_read_b64; the 16-byte chunk of a row is
// XOR-swizzled with (row >> 2) & 3 on the DMA's source address: two-way bank conflicts instead of eight-way).
// (The sweep's stagger with the sums in the filter's place -- sub-tile 0's sums in the shadow of sub-tile 1's chain -- was built and
//  measured: no gain at Kp = 128 (0.460 vs 0.459 ms), 16 spilled registers at Kp = 256 (1.28 ms): the epilogue is not what bounds it.)
template <int KP>
__global__ __launch_bounds__(256, 2) void error_lds_kernel(const FusedParams p, const uint16_t *__restrict__ Aex, int64_t lda,
															float *__restrict__ err_sq, float *__restrict__ norm_sq) {
	using Cfg = FusedCfg<KP>;
	constexpr int KSTEPS = Cfg::KSTEPS, QT = Cfg::QT, CPR = Cfg::CPR;
	constexpr int ATILE = Cfg::BQ * 64, PA = ATILE / 4096, AOFF = 2 * Cfg::TILE_BYTES;   // exact tile bytes, DMA pieces per wave, LDS offset
	constexpr int NAB = 2;                                                               // exact-tile buffers
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int r = lane & 31, h = lane >> 5;
	const int wid = xcd_remap(blockIdx.x, p.n_wg);
	const int n_rb = (int)((p.Q + Cfg::BQ - 1) / Cfg::BQ);
	const int split = wid / n_rb, rb = wid - split * n_rb;

	bf16x8 xb[QT][KSTEPS];
	int64_t qv[QT];
	uint32_t aread[QT], asw[QT];  // LDS byte address of this lane's row of the exact tile (+ 8 h) in buffer 0; the row's chunk swizzle
#pragma unroll
	for (int t = 0; t < QT; ++t) {
		const int row = wave * 32 * QT + 32 * t + r;
		qv[t] = (int64_t)rb * Cfg::BQ + row;
		const bool ok = qv[t] < p.Q;
		aread[t] = lds_addr(smem) + (uint32_t)(AOFF + row * 64 + 8 * h);
		asw[t] = (uint32_t)((row >> 2) & 3) << 4;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(p.X + (ok ? qv[t] : 0) * p.ldx) + h;
#pragma unroll
		for (int s = 0; s < KSTEPS; ++s) {
			const u32x4 zero = {0u, 0u, 0u, 0u};
			const u32x4 w = ok ? src[2 * s] : zero;
			xb[t][s] = __builtin_bit_cast(bf16x8, w);
		}
	}
	// exact-tile DMA: piece i of this wave fills LDS chunks (wave * PA + i) * 64 + lane: row = chunk >> 2, position = chunk & 3.
	// Source = the row block's base (uniform) + a 32-bit byte offset per piece (a row block spans at most 256 rows of the matrix)
	const unsigned char *abase = reinterpret_cast<const unsigned char *>(Aex + (int64_t)rb * Cfg::BQ * lda);
	uint32_t asrc[PA];
#pragma unroll
	for (int i = 0; i < PA; ++i) {
		const int ch = (wave * PA + i) * 64 + lane, row = ch >> 2, pos = ch & 3;
		const int64_t last = p.Q - 1 - (int64_t)rb * Cfg::BQ;   // rows past Q re-read the last row (their sums are dropped)
		asrc[i] = (uint32_t)(((int64_t)row < last ? (int64_t)row : last) * lda * 2) + (uint32_t)(16 * (pos ^ ((row >> 2) & 3)));
	}
	// both DMAs hand-placed (tile_dma_s: SGPR base, 32-bit lane offsets, M0 from a scalar): the builtin's per-lane 64-bit pointers and LDS
	// destinations cost ~24 VGPRs, which the staggered Kp = 256 body does not have
	const int wave_u = __builtin_amdgcn_readfirstlane(wave);
	const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(smem));
	uint32_t dma_off[Cfg::TILE_BYTES / 4096];
	tile_dma_offsets<KP>(dma_off, wave_u, lane);
	const int j_begin = split * p.tiles_per_split, j_end = min(j_begin + p.tiles_per_split, p.n_tiles);
#ifdef ANNCUR_TIMING_EXPERIMENTS
	// ANNCUR_DEBUG_ERR_MODE (p.ring_stagger here): 1 = every exact-tile DMA re-reads the split's first tile (L2-hot: no HBM latency, same
	// instruction stream), 2 = no sums (DMAs and MFMAs only), 4 = the same bytes from CONTIGUOUS memory (wrong values, timing only)
	const int dbg_mode = p.ring_stagger;
	if (dbg_mode == 4) {
#pragma unroll
		for (int i = 0; i < PA; ++i) {
			const int ch = (wave * PA + i) * 64 + lane, row = ch >> 2, pos = ch & 3;
			asrc[i] = (uint32_t)(row * 64 + 16 * (pos ^ ((row >> 2) & 3)));
		}
	}
#define ERRL_SUMS_ON (dbg_mode != 2)
#else
#define ERRL_SUMS_ON true
#endif
	auto adma = [&](int j, int buf) {
#ifdef ANNCUR_TIMING_EXPERIMENTS
		if (dbg_mode == 1) j = j_begin;
#endif
		const unsigned char *src = abase + (int64_t)j * (TILE_I * 2);   // (uniform)
#ifdef ANNCUR_TIMING_EXPERIMENTS
		if (dbg_mode == 4) src = reinterpret_cast<const unsigned char *>(Aex) + ((int64_t)rb * p.n_tiles + j) * ATILE;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
		for (int i = 0; i < PA; ++i) {
			const uint32_t m0v = lds_base + (uint32_t)(AOFF + buf * ATILE + (wave_u * PA + i) * 1024);
			asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(asrc[i]), "s"(src) : "memory", "m0");
		}
#endif
	};
	__builtin_amdgcn_s_waitcnt(0x0F70);  // see score_kernel: keeps vmcnt(0) out of the tile loop

	float se[QT], sn[QT];
#pragma unroll
	for (int t = 0; t < QT; ++t) { se[t] = 0.f; sn[t] = 0.f; }
	// (Round 4, measured and dropped: the exact tiles TWO tiles ahead in a ring of three buffers -- 80 KB at Kp = 256 -- with a counted vmcnt at
	//  the end of a step.  Same box, warm, round robin: 0.617 vs 0.615 ms at Kp = 256, 0.472 vs 0.468 at Kp = 128.  What the exact tile costs is
	//  not its round trip: with every DMA re-reading one L2-hot tile the kernel takes 0.505 / 0.32 ms, with the same bytes from contiguous
	//  memory 0.59 / 0.40 -- profiles/r04_error_kernel_modes.txt.)
	if (j_begin < j_end) {
		tile_dma_s<KP>(p.Et, j_begin, lds_base, wave_u, dma_off);
		adma(j_begin, 0);
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__syncthreads();
	uint32_t aoff[Cfg::NAOFF];
#pragma unroll
	for (int s = 0; s < Cfg::NAOFF; ++s) aoff[s] = lds_addr(smem) + (uint32_t)(r * CPR + swz<CPR>(r, 2 * s + h)) * 16u;
	__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): error_mfma_tile() counts LDS reads
	static_assert(PA <= 15, "vmcnt immediate");
#define ERRL_STEP(ECUR, ACUR, J)                                                                                                \
	do {                                                                                                                        \
		if ((J) + 1 < j_end) tile_dma_s<KP>(p.Et, (J) + 1, lds_base + ((ECUR) ^ 1) * Cfg::TILE_BYTES, wave_u, dma_off);         \
		const bool ahead = (J) + NAB - 1 < j_end;   /* (uniform) */                                                             \
		if (ahead) adma((J) + NAB - 1, ((ACUR) + NAB - 1) % NAB);                                                               \
		f32x16 acc[QT];                                                                                                         \
		error_mfma_tile<KP, ECUR>(aoff, xb, acc);                                                                               \
		ExactQuad<uint16_t> ex[QT][4];  /* items 32 j + 8 g + 4 h + {0..3}, g = 0..3, of the lane's query */                   \
		_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                          \
			_Pragma("unroll") for (int g = 0; g < 4; ++g)                                                                       \
				asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(ex[t][g].w) : "v"(aread[t] + (((uint32_t)g << 4) ^ asw[t])), "n"((ACUR) * ATILE)); \
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                      \
		_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                          \
			_Pragma("unroll") for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(ex[t][g].w));                                  \
		if (ERRL_SUMS_ON)                                                                                                       \
		_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                          \
			_Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                                    \
				const float x = ex[t][e >> 2].get(e & 3);                                                                       \
				const float d = acc[t][e] - x;                                                                                  \
				se[t] = fmaf(d, d, se[t]);                                                                                      \
				sn[t] = fmaf(x, x, sn[t]);                                                                                      \
			}                                                                                                                   \
		else { _Pragma("unroll") for (int t = 0; t < QT; ++t) { se[t] += acc[t][0]; sn[t] += ex[t][0].get(0); } }              \
		/* this wave's parts of the next item tile and of the next exact tile have landed; the barrier orders LDS only */        \
		if (NAB == 3 && ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA) : "memory");                                        \
		else __builtin_amdgcn_s_waitcnt(0x0F70);                                                                                \
		asm volatile("" ::: "memory");                                                                                          \
		__builtin_amdgcn_s_barrier();                                                                                           \
		asm volatile("" ::: "memory");                                                                                          \
	} while (0)
	for (int j = j_begin; j < j_end; j += 2) {
		ERRL_STEP(0, 0, j);
		if (j + 1 < j_end) ERRL_STEP(1, 1, j + 1);
	}
#undef ERRL_STEP
#undef ERRL_SUMS_ON
#pragma unroll
	for (int t = 0; t < QT; ++t)
		if (qv[t] < p.Q) {
			atomicAdd(&err_sq[qv[t]], se[t]);
			atomicAdd(&norm_sq[qv[t]], sn[t]);
		}
}

// Full 32-item tiles only (p.n_tiles = I / 32): the host adds the last I % 32 columns with the strided kernel of gemm.hip.
template <int KP, typename TA>
__global__ __launch_bounds__(256, 2) void error_kernel(const FusedParams p, const TA *__restrict__ Aex, int64_t lda,
														float *__restrict__ err_sq, float *__restrict__ norm_sq) {
	using Cfg = FusedCfg<KP>;
	constexpr int KSTEPS = Cfg::KSTEPS, QT = Cfg::QT, CPR = Cfg::CPR;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int r = lane & 31, h = lane >> 5;
	const int wid = xcd_remap(blockIdx.x, p.n_wg);
	const int n_rb = (int)((p.Q + Cfg::BQ - 1) / Cfg::BQ);
	const int split = wid / n_rb, rb = wid - split * n_rb;

	bf16x8 xb[QT][KSTEPS];
	int64_t qv[QT];
	const TA *rowp[QT];  // this lane's row of the exact matrix, at its half's first item of a tile
#pragma unroll
	for (int t = 0; t < QT; ++t) {
		qv[t] = (int64_t)rb * Cfg::BQ + wave * 32 * QT + 32 * t + r;
		const bool ok = qv[t] < p.Q;
		rowp[t] = Aex + (ok ? qv[t] : 0) * lda + 4 * h;  // rows past Q read row 0 (their sums are dropped)
		const u32x4 *src = reinterpret_cast<const u32x4 *>(p.X + (ok ? qv[t] : 0) * p.ldx) + h;
#pragma unroll
		for (int s = 0; s < KSTEPS; ++s) {
			const u32x4 zero = {0u, 0u, 0u, 0u};
			const u32x4 w = ok ? src[2 * s] : zero;
			xb[t][s] = __builtin_bit_cast(bf16x8, w);
		}
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);  // see score_kernel: keeps vmcnt(0) out of the tile loop

	const int j_begin = split * p.tiles_per_split, j_end = min(j_begin + p.tiles_per_split, p.n_tiles);
	float se[QT], sn[QT];
	ExactQuad<TA> ex[QT][4];  // exact values of the tile whose MFMAs run next: items 32 j + 8 g + 4 h + {0..3}, g = 0..3
#pragma unroll
	for (int t = 0; t < QT; ++t) { se[t] = 0.f; sn[t] = 0.f; }
	if (j_begin < j_end) {
		tile_dma<KP>(p.Et, j_begin, smem, wave, lane);
#pragma unroll
		for (int t = 0; t < QT; ++t)
#pragma unroll
			for (int g = 0; g < 4; ++g) ex[t][g].load(rowp[t] + (int64_t)j_begin * TILE_I + 8 * g);
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);
	__syncthreads();
	uint32_t aoff[Cfg::NAOFF];   // LDS byte address of this lane's A fragment of k-step s (< NAOFF) in tile buffer 0 (see stagger_tile)
#pragma unroll
	for (int s = 0; s < Cfg::NAOFF; ++s) aoff[s] = lds_addr(smem) + (uint32_t)(r * CPR + swz<CPR>(r, 2 * s + h)) * 16u;
	__builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): error_mfma_tile() counts LDS reads
#define ERR_STEP(CUR, J)                                                                                                        \
	do {                                                                                                                        \
		const bool more = (J) + 1 < j_end;                                                                                      \
		if (more) tile_dma<KP>(p.Et, (J) + 1, smem + ((CUR) ^ 1) * Cfg::TILE_BYTES, wave, lane);                                \
		f32x16 acc[QT];                                                                                                         \
		error_mfma_tile<KP, CUR>(aoff, xb, acc);                                                                                \
		/* the exact values are consumed HERE, behind the MFMA chain: hipcc otherwise hoists their unpacking to the top of the */ \
		/* iteration and waits vmcnt(0) there -- for them and for the tile DMA it has just issued */                            \
		_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                          \
			_Pragma("unroll") for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(ex[t][g].w));                                  \
		_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                          \
			_Pragma("unroll") for (int e = 0; e < 16; ++e) {                                                                    \
				const float x = ex[t][e >> 2].get(e & 3);                                                                       \
				const float d = acc[t][e] - x;                                                                                  \
				se[t] = fmaf(d, d, se[t]);                                                                                      \
				sn[t] = fmaf(x, x, sn[t]);                                                                                      \
			}                                                                                                                   \
		/* the next tile's exact values are requested now and consumed after the next MFMA chain; the wait below covers the */  \
		/* tile DMA only (loads return in order: the QT*4 quads issued after it may still be in flight) */                      \
		if (more) {                                                                                                             \
			/* the quads must be issued AFTER the tile DMA (the counted wait relies on it) and after the sums above have */      \
			/* consumed the previous ones: otherwise hipcc sinks the sums below the loads and renames the destinations */       \
			if (QT == 2) asm volatile("" : "+v"(se[0]), "+v"(sn[0]), "+v"(se[QT - 1]), "+v"(sn[QT - 1])::"memory");             \
			else asm volatile("" : "+v"(se[0]), "+v"(sn[0])::"memory");                                                         \
			_Pragma("unroll") for (int t = 0; t < QT; ++t)                                                                      \
				_Pragma("unroll") for (int g = 0; g < 4; ++g) ex[t][g].load(rowp[t] + (int64_t)((J) + 1) * TILE_I + 8 * g);     \
			__builtin_amdgcn_s_waitcnt(QT == 2 ? 0x0F78 : 0x0F74);  /* vmcnt(8) / vmcnt(4) */                                   \
		}                                                                                                                       \
		/* raw barrier: __syncthreads() carries a release fence for which hipcc waits vmcnt(0), i.e. for the quads just issued. */ \
		/* What the barrier orders here is LDS only: this wave's fragment reads are complete, its part of the DMA has landed. */ \
		asm volatile("" ::: "memory");                                                                                          \
		__builtin_amdgcn_s_barrier();                                                                                           \
		asm volatile("" ::: "memory");                                                                                          \
	} while (0)
	for (int j = j_begin; j < j_end; j += 2) {
		ERR_STEP(0, j);
		if (j + 1 < j_end) ERR_STEP(1, j + 1);
	}
#undef ERR_STEP
#pragma unroll
	for (int t = 0; t < QT; ++t)
		if (qv[t] < p.Q) {
			atomicAdd(&err_sq[qv[t]], se[t]);
			atomicAdd(&norm_sq[qv[t]], sn[t]);
		}
}

#include "score_evalf.hpp"

// ------------------------------------------------------------------ select
// Item-tile ranges of the sweep stages (which tiles item split s swept in stage g): [begin[g] + s*tps[g], + tps[g]) below end[g].
// stride > 1: interleaved instead (split s swept begin[g] + s, + stride, ... below end[g]).
// chunk > 0: dynamic schedule -- stage g's tiles were swept in chunks of `chunk` tiles, chunk c of query row block rb by the item
// split owner[g][rb * n_chunks[g] + c] (row block = q / bq).
struct SweepStages {
	int n, begin[3], end[3], tps[3], stride;
	int chunk, bq, n_chunks[3];
	const uint8_t *owner[3];
};

// One workgroup per query: exact top-k of the query's candidate segments.  A segment that overflowed (or whose LDS ring
// wrapped: poisoned count) is repaired locally: the scores of the item tiles its split swept are recomputed here (fp32 dot
// products of the same bf16 operands) and offered unfiltered, the stored candidates of that split are ignored.  Only a query
// that still ends with fewer than k candidates is recomputed in full.
template <int KMAX>
__global__ __launch_bounds__(SEL_THREADS) void select_candidates_kernel(
	const uint2 *__restrict__ cand, const uint32_t *__restrict__ seg_cnt, int nseg, int S, SweepStages stg, int capg,
	const uint16_t *__restrict__ X, int64_t ldx, const uint16_t *__restrict__ Et, int64_t I, int KP, uint32_t k,
	float *__restrict__ out_val, int32_t *__restrict__ out_idx, uint32_t *__restrict__ n_fallback,
	const int32_t *__restrict__ hard_list, const uint32_t *__restrict__ hard_cnt, const float *__restrict__ tau_in, int tau_stride,
	const int32_t *__restrict__ remap) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const SelState s = sel_carve<KMAX>(smem);
	float *xq = reinterpret_cast<float *>(smem + SelCfg<KMAX>::LDS_BYTES);  // [KP], repair / fallback only
	uint32_t *bad = reinterpret_cast<uint32_t *>(xq + KP);                  // [8]: bitmap of splits to repair (S <= 256)
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	auto score_of = [&](int64_t i) {
		float v = 0.f;
		const uint4 *er = reinterpret_cast<const uint4 *>(Et + i * KP);
		for (int c = 0; c < KP / 8; ++c) {
			const uint4 w = er[c];
			const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
			for (int d = 0; d < 4; ++d) {
				v = fmaf(__uint_as_float(ww[d] << 16), xq[8 * c + 2 * d], v);
				v = fmaf(__uint_as_float(ww[d] & 0xffff0000u), xq[8 * c + 2 * d + 1], v);
			}
		}
		return v;
	};
	// either one query per workgroup (hard_list == nullptr) or a grid-stride walk over the queries the wave kernel deferred
	const uint32_t n_work = hard_list ? *hard_cnt : gridDim.x;
	for (uint32_t wi = blockIdx.x; wi < n_work; wi += gridDim.x) {
	__syncthreads();
	sel_init(s);
	if (tid < 8) bad[tid] = 0u;
	__syncthreads();
	const int64_t q = hard_list ? (int64_t)hard_list[wi] : (int64_t)wi;
	const uint32_t *cnts = seg_cnt + q * nseg;
	for (int sg = tid; sg < nseg; sg += SEL_THREADS) {
		const uint32_t c = cnts[sg];
		if (c > (uint32_t)capg) { const int sp = sg % S; atomicOr(&bad[sp >> 5], 1u << (sp & 31)); }
	}
	__syncthreads();
	const bool repair = (bad[0] | bad[1] | bad[2] | bad[3] | bad[4] | bad[5] | bad[6] | bad[7]) != 0u;
	for (int sg = tid; sg < nseg; sg += SEL_THREADS) {
		const int sp = sg % S;
		const uint32_t cc = ((bad[sp >> 5] >> (sp & 31)) & 1u) ? 0u : cnts[sg];
		atomicAdd(&s.scal[9], cc);
		atomicMax(&s.scal[10], cc);
	}
	__syncthreads();
	const uint32_t total = s.scal[9], maxc = s.scal[10];
	// (the threshold the last sweep stage ran with: earlier stages' candidates below it are dropped at the load)
	float tau = tau_in ? tau_in[q * tau_stride] : -INFINITY;
	uint64_t tau_key = 0;
	bool full = !repair && total < k;  // cannot happen with a valid threshold; kept as the last line of defence
	if (!full) {
		for (int sb = 0; sb < nseg; sb += 4) {
			const int sg = sb + wave;
			const int sp = sg % S;
			const bool use = sg < nseg && !((bad[sp >> 5] >> (sp & 31)) & 1u);
			const uint32_t c = use ? cnts[sg] : 0u;
			const uint2 *sp_ptr = cand + (q * nseg + (sg < nseg ? sg : 0)) * (int64_t)capg;
			for (uint32_t e0 = 0; e0 < maxc; e0 += WAVE) {
				const uint32_t e = e0 + lane;
				const bool in = e < c;
				const uint2 ce = in ? sp_ptr[e] : make_uint2(0, 0);
				sel_offer(s, in, __uint_as_float(ce.x), ce.y, tau, tau_key);
				sel_maybe_compact<KMAX>(s, k, tau, tau_key);
			}
		}
	}
	if (repair || full) {
		if (tid == 0) atomicAdd(n_fallback, 1u);
		for (int c = tid; c < KP; c += SEL_THREADS) xq[c] = bf16_bits_to_f32(X[q * ldx + c]);
		__syncthreads();
	}
	if (repair) {
		for (int sp = 0; sp < S; ++sp) {
			if (!((bad[sp >> 5] >> (sp & 31)) & 1u)) continue;  // uniform
			for (int g = 0; g < stg.n; ++g) {
				if (stg.chunk > 0) {  // (uniform) dynamic schedule: the chunks of this query's row block that split sp drew
					const uint8_t *own = stg.owner[g] + (q / stg.bq) * (int64_t)stg.n_chunks[g];
					int it = 0;
					for (int c = 0; c < stg.n_chunks[g]; ++c) {
						if (own[c] != (uint8_t)sp) continue;  // (uniform: every thread reads the same byte)
						const int64_t i_beg = (int64_t)(stg.begin[g] + c * stg.chunk) * TILE_I;
						const int64_t i_end = (int64_t)min(stg.begin[g] + (c + 1) * stg.chunk, stg.end[g]) * TILE_I;
						for (int64_t i0 = i_beg; i0 < i_end; i0 += SEL_THREADS, ++it) {
							const int64_t i = i0 + tid;
							const bool in = i < i_end && i < I;
							const float v = in ? score_of(i) : 0.f;
							sel_offer(s, in, v, (uint32_t)i, tau, tau_key);
							if ((it & 15) == 15) sel_maybe_compact<KMAX>(s, k, tau, tau_key);
						}
					}
					sel_maybe_compact<KMAX>(s, k, tau, tau_key);
					continue;
				}
				if (stg.stride > 1) {  // (uniform) interleaved splits: tiles begin + sp, + stride, ...; 8 tiles (256 items) per pass
					int it = 0;
					for (int tb = stg.begin[g] + sp; tb < stg.end[g]; tb += 8 * stg.stride, ++it) {
						const int tile = tb + (tid >> 5) * stg.stride;
						const int64_t i = (int64_t)tile * TILE_I + (tid & 31);
						const bool in = tile < stg.end[g] && i < I;
						const float v = in ? score_of(i) : 0.f;
						sel_offer(s, in, v, (uint32_t)i, tau, tau_key);
						if ((it & 15) == 15) sel_maybe_compact<KMAX>(s, k, tau, tau_key);
					}
					sel_maybe_compact<KMAX>(s, k, tau, tau_key);
					continue;
				}
				const int t0 = stg.begin[g] + sp * stg.tps[g];
				const int t1 = min(t0 + stg.tps[g], stg.end[g]);
				int it = 0;
				for (int64_t i0 = (int64_t)t0 * TILE_I; i0 < (int64_t)t1 * TILE_I; i0 += SEL_THREADS, ++it) {
					const int64_t i = i0 + tid;
					const bool in = i < (int64_t)t1 * TILE_I && i < I;
					const float v = in ? score_of(i) : 0.f;
					sel_offer(s, in, v, (uint32_t)i, tau, tau_key);
					if ((it & 15) == 15) sel_maybe_compact<KMAX>(s, k, tau, tau_key);
				}
				sel_maybe_compact<KMAX>(s, k, tau, tau_key);
			}
		}
		__syncthreads();
		full = s.scal[0] < k && tau_key == 0;  // fewer than k even after the repair (and nothing was compacted away)
	}
	if (full) {
		__syncthreads();
		if (tid == 0) s.scal[0] = 0;
		__syncthreads();
		tau = -INFINITY; tau_key = 0;
		int it = 0;
		for (int64_t i0 = 0; i0 < I; i0 += SEL_THREADS, ++it) {
			const int64_t i = i0 + tid;
			const bool in = i < I;
			const float v = in ? score_of(i) : 0.f;
			sel_offer(s, in, v, (uint32_t)i, tau, tau_key);
			if ((it & 15) == 15) sel_maybe_compact<KMAX>(s, k, tau, tau_key);
		}
	}
	sel_finish<KMAX>(s, k, out_val + q * (int64_t)k, out_idx + q * (int64_t)k, remap);
	}
}

// Threshold refinement between sweep stages for k > 128 (workgroup-level selector): tau[q] = max(tau[q], k-th best candidate so far).
// A query with an overflowed segment, or with fewer than k candidates yet, keeps its (still valid) threshold.
template <int KMAX>
__global__ __launch_bounds__(SEL_THREADS) void tau_block_kernel(const uint2 *__restrict__ cand, const uint32_t *__restrict__ seg_cnt, int nseg,
																 int capg, uint32_t k, float *__restrict__ tau_out, int tau_stride, int prefilter) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const SelState s = sel_carve<KMAX>(smem);
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int64_t q = blockIdx.x;
	sel_init(s);
	const uint32_t *cnts = seg_cnt + q * nseg;
	for (int sg = tid; sg < nseg; sg += SEL_THREADS) {
		const uint32_t c = cnts[sg];
		if (c > (uint32_t)capg) atomicOr(&s.scal[8], 1u);
		atomicAdd(&s.scal[9], c > (uint32_t)capg ? 0u : c);
		atomicMax(&s.scal[10], c > (uint32_t)capg ? 0u : c);
	}
	__syncthreads();
	if (s.scal[8] != 0u || s.scal[9] <= k) return;  // (uniform)
	const uint32_t maxc = s.scal[10];
	float tau = prefilter ? tau_out[q * tau_stride] : -INFINITY;  // earlier candidates below the running threshold cannot be the k-th best
	uint64_t tau_key = 0;
	for (int sb = 0; sb < nseg; sb += 4) {
		const int sg = sb + wave;
		const uint32_t c = sg < nseg ? cnts[sg] : 0u;
		const uint2 *sp = cand + (q * nseg + (sg < nseg ? sg : 0)) * (int64_t)capg;
		for (uint32_t e0 = 0; e0 < maxc; e0 += WAVE) {
			const uint32_t e = e0 + lane;
			const bool in = e < c;
			const uint2 ce = in ? sp[e] : make_uint2(0, 0);
			sel_offer(s, in, __uint_as_float(ce.x), ce.y, tau, tau_key);
			sel_maybe_compact<KMAX>(s, k, tau, tau_key);
		}
	}
	__syncthreads();
	if (s.scal[0] > k) {  // (uniform) cut to k: the k-th best key is the new threshold
		tau_key = sel_compact<KMAX>(s, k);
		tau = key_val(tau_key);
	}
	if (tid == 0 && tau_key != 0 && s.scal[0] >= k) tau_out[q * tau_stride] = fmaxf(tau_out[q * tau_stride], tau);
}

// One WAVE per query (k <= 128, <= 64 segments): the query's candidate segments are streamed through the wave-level
// selector of wave_select.hpp (wave-private LDS buffer, ballot/popcount radix select, in-register bitonic sort).
// No workgroup barrier.  Queries whose segments overflowed or that collected fewer than k candidates are appended to
// hard_list for the workgroup-level kernel (which recomputes them exactly).
constexpr int WQ_CAP = 1024;
constexpr int WQ_K2 = 1024;  // largest k of the wave-level candidate select (select_stream.hpp: 16 keys per lane in the final sort)
template <int KW> struct WqCfg { static constexpr int CAP = KW <= 128 ? WQ_CAP : 2048, TRIGGER = CAP - WAVE, E = KW / 64; };

// TAU_ONLY (between sweep stages): no output, the query's threshold is raised to the k-th best candidate collected so far.
// KW = 128 or 512: capacity class of k.
template <bool TAU_ONLY, int KW = 128>
__global__ __launch_bounds__(256) void select_wave_kernel(const uint2 *__restrict__ cand, const uint32_t *__restrict__ seg_cnt, int nseg,
														   int capg, int64_t Q, uint32_t k, float *__restrict__ out_val,
														   int32_t *__restrict__ out_idx, uint32_t *__restrict__ hard_cnt,
														   int32_t *__restrict__ hard_list, float *__restrict__ tau, int tau_stride, int prefilter,
														   const int32_t *__restrict__ remap) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int lane = lane_id(), wave = threadIdx.x >> 6;
	const int64_t q = (int64_t)blockIdx.x * 4 + wave;
	if (q >= Q) return;
	SEL_STAMP(0);
	const uint32_t c = (lane < nseg) ? seg_cnt[q * nseg + lane] : 0u;
	const uint32_t inc = wave_scan_incl<DppAdd>(c);
	const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, WAVE - 1), pre = inc - c;
	if (__ballot(c > (uint32_t)capg) != 0ull || total < k) {
		if (!TAU_ONLY && lane == 0) hard_list[atomicAdd(hard_cnt, 1u)] = (int32_t)q;
		return;  // TAU_ONLY: keep the old (still valid) threshold
	}
	constexpr int CAP = WqCfg<KW>::CAP, TRIGGER = WqCfg<KW>::TRIGGER;
	WaveSel w = wsel_init<CAP>(smem + wave * WaveSelLayout<CAP>::BYTES);
	// the threshold the last sweep stage ran with is a valid lower bound on the k-th best: candidates of earlier stages below it
	// are dropped at the load (at least k candidates are >= it by construction)
	if (tau && prefilter) w.tau = tau[q * tau_stride];
	const uint2 *qc = cand + q * nseg * (int64_t)capg;
	// The candidates are fetched LOAD_U chunks of 64 at a time: the batch's flat-index -> (segment, entry) searches advance in
	// LOCKSTEP (one round = LOAD_U independent ds_bpermute, one wait), the loads are unconditional (clamped address) and issued back
	// to back.  Written chunk by chunk hipcc emitted LOAD_U serial chains of seven LDS round trips and one exposed HBM latency per chunk.
	// (A search without LDS -- the segment ends read into scalars with v_readlane, every lane counting the ends at or below its index --
	//  was slower: 9.5 k instead of 5 k cycles per wave for the 23 ends x 8 chunks.)
	SEL_STAMP(1);
	constexpr int LOAD_U = 8;
	for (uint32_t j0 = 0; j0 < total; j0 += LOAD_U * WAVE) {
		int sg[LOAD_U];  // last segment whose exclusive prefix is <= j (skips empty segments)
#pragma unroll
		for (int u = 0; u < LOAD_U; ++u) sg[u] = 0;
#pragma unroll
		for (int step = 32; step >= 1; step >>= 1) {
			if (step < nseg) {  // (uniform: segments >= nseg do not exist)
				uint32_t pv[LOAD_U];
#pragma unroll
				for (int u = 0; u < LOAD_U; ++u) pv[u] = __shfl(pre, (sg[u] + step) & 63);
#pragma unroll
				for (int u = 0; u < LOAD_U; ++u)
					if (sg[u] + step < nseg && pv[u] <= j0 + (uint32_t)(u * WAVE + lane)) sg[u] += step;
			}
		}
		uint32_t ps[LOAD_U];
#pragma unroll
		for (int u = 0; u < LOAD_U; ++u) ps[u] = __shfl(pre, sg[u]);
		uint2 e[LOAD_U];
		if (j0 == 0) SEL_STAMP(4);
#pragma unroll
		for (int u = 0; u < LOAD_U; ++u) {
			const uint32_t j = j0 + (uint32_t)(u * WAVE + lane);
			e[u] = qc[j < total ? (int64_t)sg[u] * capg + (j - ps[u]) : 0];
		}
#ifdef ANNCUR_TIMING_EXPERIMENTS
		if (j0 == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SEL_STAMP(5); }
#endif
#pragma unroll
		for (int u = 0; u < LOAD_U; ++u) {
			if (j0 + (uint32_t)(u * WAVE) < total) {  // (uniform)
				wsel_offer(w, j0 + (uint32_t)(u * WAVE + lane) < total, __uint_as_float(e[u].x), e[u].y);
				if (w.cnt > (uint32_t)TRIGGER) wsel_compact<4, false, true>(w, k);
			}
		}
	}
	SEL_STAMP(2);
	if (TAU_ONLY) {
		if (w.cnt > k) wsel_compact<4, false, true>(w, k);
		if (lane == 0 && w.cnt >= k) tau[q * tau_stride] = fmaxf(tau[q * tau_stride], w.tau);
		SEL_STAMP(3);
		return;
	}
	wsel_finish<0, 4, WqCfg<KW>::E, true>(w, k, out_val + q * (int64_t)k, out_idx + q * (int64_t)k, remap);
	SEL_STAMP(3);
}

}  // namespace
namespace {
#include "score_wide.hpp"
}
namespace {

// ------------------------------------------------------------------ host-side plan
struct FusedPlan {
	bool ok;
	int QT, BQ, n_rb, n_tiles, n_full, S, tiles_per_split, group, n_st, S0, st_per_split, n_groups, capg, kmax, flush_tiles;
	int n_stages, stage_end[3], stage_tps[3], stage_flush[3], stage_pred[3];
	int leading;
	int lg;   // candidate segments per query and item split: 2 (32x32x16 sweep: lane halves), 1 (16x16x32 sweep: wave-level queue)
	bool body16;  // the sweep stages run score16_kernel
	bool bodyef;  // the sweep stages run evalf_kernel (anncur_eval_fused: candidates + error sums in one pass; 32x32x16, wave queue, static shares)
	bool wg8;     // ... score16_kernel<Kp, 8>: the same body in 8-wave workgroups of BQ_s = 512 queries, one per CU (half the DMA pieces per wave)
	bool ring16;  // ... score16r_kernel: 8-wave workgroups of BQ_s = 512 queries, flag-synchronised tile ring (score16r.hpp)
	int BQ_s, n_rb_s;   // query rows per sweep workgroup and the sweep's row blocks (the prepass keeps BQ / n_rb)
	bool bodyq1;  // the sweep stages run scoreq1_kernel (Kp = 512)
	bool bodyq16; // ... scoreq16_kernel: the same body on 16x16x32 MFMAs
	int chunk;  // dynamic tile schedule of the sweep stages: tiles per ticket (0: static shares)
	bool ladder;  // score16_kernel moves its thresholds up a ladder of levels inside ONE sweep launch (score16.hpp): no stages, no refinement launches
	int ladder_k2; // rank (in the prepass sample's group maxima) of the ladder's top level
	size_t off_lcnt, off_lvl, off_tau2;
	size_t off_gmax, off_tval, off_tidx, off_segcnt, off_cand, off_tau, off_hard, off_ctr, off_owner, total;
};

int num_cu() { return anncur_num_cu(); }

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
int next_pow2(int x) { int p = 1; while (p < x) p <<= 1; return p; }

// Sweep stages: between stages the threshold is raised to the k-th best candidate seen so far, which cuts the survivors of the
// remaining tiles from H0 to ~1.2 k / (fraction seen).  Any split is exact; the split is picked by a small cost model with
// constants measured on MI355X (cfg2, round 1): ~2.4e-11 s of chip time per survivor, ~25 us + 3 ns per query per extra stage
// (threshold kernel + launch ramp) with the wave-level refinement kernel (k <= 128, <= 64 segments), ~1.4e-11 s per candidate
// read with the workgroup-level one.  `unit_items` = items per tile unit of P.n_tiles (32, or 256 for the wide kernel).
void plan_stages(FusedPlan &P, int64_t Q, int k, double exp_hits, bool staged, double fmin, int unit_items) {
	double frac[3] = {1.0, 1.0, 1.0};
	P.n_stages = 1;
	if (staged) {
		const double H0 = exp_hits, c_hit = 2.4e-11 * (double)Q;
		const double c_stage = k <= WSEL_K ? 25e-6 + 3e-9 * (double)Q : (k <= WQ_K2 ? 25e-6 + 1.2e-8 * (double)Q : 25e-6 + 1.4e-11 * (double)Q * 0.3 * H0);
		auto later = [&](double f) { const double h = 1.2 * k / f; return h < H0 ? h : H0; };
		double best = H0 * c_hit;
		static const double grid[] = {0.02, 0.03, 0.04, 0.06, 0.08, 0.10, 0.12, 0.15, 0.18, 0.22, 0.26, 0.30, 0.35, 0.40, 0.50};
		const int ng = (int)(sizeof(grid) / sizeof(grid[0]));
		for (int i = 0; i < ng; ++i) {
			const double f1 = grid[i];
			if (f1 < fmin) continue;
			const double c2 = (f1 * H0 + (1 - f1) * later(f1)) * c_hit + c_stage;
			if (c2 < best) { best = c2; P.n_stages = 2; frac[0] = f1; frac[1] = 1.0; }
			for (int j = i + 1; j < ng; ++j) {
				const double f2 = grid[j];
				if (f2 - f1 < fmin) continue;
				const double c3 = (f1 * H0 + (f2 - f1) * later(f1) + (1 - f2) * later(f2)) * c_hit + 2 * c_stage;
				if (c3 < best) { best = c3; P.n_stages = 3; frac[0] = f1; frac[1] = f2; }
			}
		}
	}
	// The hit-count model above knows nothing of what a first stage costs beyond its survivors: it runs against the loosest threshold
	// (dense rings, exec-mask filter) and every candidate it collects is read again by the refinement.  Measured at cfg2 size on MI355X
	// (round 3, one box, alternating): k = 100 -- the model's 0.35 stays (0.245: + 2 % sweep time on the bench matrices, level on random
	// operands); k = 500 -- the model's 0.35: 1.18 ms per call, 0.30 1.12, 0.22 1.11, 0.15 1.15 -> above k = 128 a two-stage plan
	// takes 0.6 of the model's first fraction.  The 16x16x32 body (k <= 128) gains more from a tight threshold than it loses to a
	// loose one (higher clock on sparse tiles, costlier pushes on dense ones): cfg2 sweep launches 0.458 ms at 0.35, 0.455 at 0.30,
	// 0.453 at 0.26, 0.450 at 0.22, 0.453 at 0.18, 0.461 at 0.14 -> 0.63 of the model's fraction.
	if (P.n_stages == 2) {
		double shrink = P.body16 ? 0.63 : (k <= WSEL_K ? 1.0 : 0.6);
		if (const char *dbg = knob("ANNCUR_DEBUG_F1_SHRINK")) shrink = atof(dbg);
		const double f1 = frac[0] * shrink;
		frac[0] = f1 > fmin ? f1 : (fmin < frac[0] ? fmin : frac[0]);
	}
#ifdef ANNCUR_TIMING_EXPERIMENTS
	if (const char *dbg = getenv("ANNCUR_DEBUG_STAGES")) {  // tuning knob "f1,f2" or "f1" (>= 1: single stage)
		double f1 = 0, f2 = 0;
		const int n = sscanf(dbg, "%lf,%lf", &f1, &f2);
		if (staged && n == 2 && f1 > 0 && f2 > f1 && f2 < 1) { frac[0] = f1; frac[1] = f2; P.n_stages = 3; }
		else if (staged && n == 1 && f1 > 0 && f1 < 1) { frac[0] = f1; frac[1] = 1.0; P.n_stages = 2; }
		else if (n == 1 && f1 >= 1) P.n_stages = 1;
	}
#endif
	double rate = exp_hits / ((double)P.n_tiles * 2.0);  // expected hits per (query half, tile) in the first stage
	int prev = 0;
	for (int i = 0; i < P.n_stages; ++i) {
		int end = (int)(frac[i] * P.n_tiles + 0.5);
		if (i == P.n_stages - 1 || end > P.n_tiles) end = P.n_tiles;
		if (end <= prev) end = prev + 1 < P.n_tiles ? prev + 1 : P.n_tiles;
		P.stage_end[i] = end;
		P.stage_tps[i] = (end - prev + P.S - 1) / P.S;
		// ring window: expected hits per (lane, sub-tile) ring and window <= 0.35, so that 8 slots wrap with p ~ 1.5e-10 per window
		// (cfg2: 3.1e7 windows per call -> one repaired query per ~200 calls; at 0.5 it was one per ~10 calls)
		int ft = (int)(0.35 / (rate > 1e-9 ? rate : 1e-9));
		P.stage_flush[i] = ft < 1 ? 1 : (ft > 8 ? 8 : ft);
		// Filter variant of the stage: `rate` survivors per (lane, sub-tile, tile) = rate / 16 per element, so a compare finds a survivor
		// in one of its 64 lanes with P = 1 - exp(-4 rate).  The ballot variant pays for that with a wave-level branch out and back, the
		// exec variant (filter_one<.., true>) with two more scalar instructions on EVERY compare and a shorter survivor block.  Measured
		// (MI355X, one process, alternating): cfg2 k = 100 (P ~ 0.3-0.5) exec 0.535 vs ballot 0.551 ms per sweep; k = 500 (P ~ 0.9)
		// 0.84 vs 0.88; I = 10^6, Kp = 256 (P ~ 0.2 over most of the sweep) 2.80 vs 2.76 -> exec above P = 0.25.
		// (staggered Kp <= 256 loop only: launch_fused ignores it elsewhere)
		P.stage_pred[i] = (1.0 - exp(-4.0 * rate)) > 0.25 ? 1 : 0;
		if (const char *dbg = knob("ANNCUR_DEBUG_ALL_PRED")) P.stage_pred[i] = atoi(dbg) != 0;
		// next stage: threshold = k-th best of the fraction seen so far
		rate = (double)k / ((double)end * unit_items) * 16.0 * 1.2;
		prev = end;
	}
}

FusedPlan plan_fused(int64_t Q, int64_t I, int KP, int k, bool leading = false, bool mfma16 = false, bool qt1 = false, bool mfma32 = false, bool ring = false, bool evalf = false,
					 bool no_ladder = false) {
	FusedPlan P{};
	P.ok = false;
	P.leading = leading ? 1 : 0;
	if (!(KP == 64 || KP == 128 || KP == 256 || KP == 512)) return P;
	if (k < 1 || k > ANNCUR_MAX_TOPK || Q < 1 || I < 1 || I >= (int64_t)0x7fffffff - 64 || k > I) return P;
	// qt1 (ANNCUR_TOPK_QT1, Kp = 128 / 256): one 32-query sub-tile per wave with the cross-tile pipeline of the Kp = 512 sweep,
	// 3 workgroups per CU (<= 168 VGPRs) instead of two sub-tiles staggered inside a wave at 2 workgroups per CU
	if (evalf && !(KP <= 256 && I < (int64_t)(1 << 26))) return P;   // anncur_eval_fused: Kp <= 256 (two workgroups per CU), queue entries carry the query
	if (evalf) { qt1 = false; mfma16 = false; mfma32 = false; ring = false; leading = false; P.leading = 0; }
	P.bodyef = evalf;
	P.QT = (KP <= 256 && !(qt1 && KP >= 128)) ? 2 : 1;
	P.BQ = 128 * P.QT;
	P.n_rb = (int)ceil_div64(Q, P.BQ);
	P.n_tiles = (int)ceil_div64(I, TILE_I);
	P.n_full = (int)(I / TILE_I);
	// prepass sample: enough groups that the k-th largest group maximum is a tight bound
	int target = (4 * k > 512) ? 4 * k : 512;
	if (const char *dbg = knob("ANNCUR_DEBUG_SAMPLE_GROUPS")) target = atoi(dbg);   // prepass sample size in groups (tuning knob)
	int n_st16 = (target + 1) / 2;
	if ((int64_t)n_st16 * 8 <= P.n_full) { P.group = 16; P.n_st = n_st16; }
	else { P.group = 4; P.n_st = (target + 7) / 8; }
	if ((int64_t)P.n_st * 4 > P.n_full) return P;  // problem too small for the fused path: use dense GEMM + scan
	P.n_groups = P.n_st * (P.group == 16 ? 2 : 8);
	if (P.n_groups < k) return P;
	int slots = ((P.QT == 1 && KP <= 256) ? 3 : 2) * num_cu();
	if (knob("ANNCUR_DEBUG_ONE_WG")) slots = num_cu();  // one sweep workgroup per CU (co-residence experiment)
	// Dynamic tile schedule (staggered 32x32x16 sweep, Kp <= 256): tickets of CHUNK_TILES tiles per query row block instead of fixed shares
	// (score_kernel).  Workgroups per row block: enough to fill every slot (rounded UP -- a workgroup that finds no ticket left ends at
	// once), at most 32 so that the 2 S segments of a query fit the wave-level select.
	// Kp = 512: the body with the wave-level queue (score_q1.hpp: queue + counters + ticket words fit the 16 KB the rings took) unless
	// ANNCUR_TOPK_MFMA32 asks for the per-lane-ring body (static shares: no LDS left for its ticket words) or I >= 2^26
	P.bodyq1 = KP == 512 && !mfma32 && I < (int64_t)(1 << 26);
	if (knob("ANNCUR_DEBUG_NO_Q1")) P.bodyq1 = false;
	P.bodyq16 = P.bodyq1;   // (16x16x32 MFMAs: cfg4 per-GPU shape, one box, alternating: sweep launches 5.21 -> 5.02 ms, step 7.19 -> 7.00; scoreq1_kernel stays for A/B)
	if (const char *dbg = knob("ANNCUR_DEBUG_Q16")) P.bodyq16 = P.bodyq1 && atoi(dbg) != 0;
	const bool ticketed = (P.QT == 2 || P.bodyq1) && !evalf;   // the bodies with the ticket schedule
	P.chunk = ticketed ? CHUNK_TILES : 0;
	if (const char *dbg = knob("ANNCUR_DEBUG_CHUNK")) P.chunk = ticketed ? atoi(dbg) : 0;
	// Body of the sweep stages, decided here because the ring body changes the decomposition (512-query workgroups, one per CU)
	const bool can16 = KP <= 256 && P.QT == 2 && I < (int64_t)(1 << 26);
	// (default up to k = 384 since late round 4: same process, warm, round robin at cfg2 size -- k = 150: 0.733 vs 0.738 ms for the 32x32x16 body,
	//  200: 0.779 vs 0.827, 256: 0.829 vs 0.878, 300: 0.909 vs 0.942, 384: 0.951 vs 0.968; level from 500 on: 1.04 / 1.04, 1000: 1.50 / 1.50)
	// Round 5: with the in-flight threshold ladder (score16.hpp; needs the wave-level threshold kernel, i.e. <= 4096 group maxima: k <= 1024) the
	// 16x16x32 body wins up to k = 1024 -- one process, alternating, cfg2 size: k = 500 0.984 ms against 1.055-1.072 for the staged 32x32x16 body
	// (1970 against 2606 candidates per query, select 0.144 against 0.19), k = 1000 1.38 against 1.53; without the ladder (ANNCUR_TOPK_STAGED) the
	// round-4 limit stands
	constexpr int BODY16_MAX_K = 384, BODY16_MAX_K_LADDER = 1024;
	const bool ladder_possible = !no_ladder && ticketed && P.n_groups <= 4096;
	P.body16 = can16 && !mfma32 && !evalf && (mfma16 || k <= (ladder_possible ? BODY16_MAX_K_LADDER : BODY16_MAX_K));
	if (knob("ANNCUR_DEBUG_MFMA16") && !evalf) P.body16 = can16 && atoi(knob("ANNCUR_DEBUG_MFMA16")) != 0;
#ifdef ANNCUR_TIMING_EXPERIMENTS
	P.ring16 = P.body16 && k <= WSEL_K && KP >= 128 && P.chunk == CHUNK_TILES && ring;   // opt-in (ANNCUR_TOPK_RING): measured slower than the barrier body, see score16r.hpp
#else
	P.ring16 = false;   // (the tile-ring body is compiled into the experiments library only; the product refuses the flag: score_topk_impl)
	(void)ring;
#endif
	if (const char *dbg = knob("ANNCUR_DEBUG_RING16")) P.ring16 = P.body16 && KP >= 128 && P.chunk == CHUNK_TILES && atoi(dbg) != 0;
	P.wg8 = false;
	if (const char *dbg = knob("ANNCUR_DEBUG_WG8")) P.wg8 = P.body16 && !P.ring16 && KP >= 128 && atoi(dbg) != 0;
	P.BQ_s = (P.ring16 || P.wg8) ? 512 : P.BQ;
	P.n_rb_s = (int)ceil_div64(Q, P.BQ_s);
	const int slots_s = (P.ring16 || P.wg8) ? num_cu() : slots;   // sweep workgroups resident at once
	int S = P.chunk > 0 ? (slots_s + P.n_rb_s - 1) / P.n_rb_s : slots_s / P.n_rb_s;
	if (P.chunk > 0 && k <= WQ_K2 && S > WAVE / 2) S = WAVE / 2;
	if (S < 1) S = 1;
	if (S > 255) S = P.chunk > 0 ? 255 : (S > 256 ? 256 : S);   // (the owner map holds a split in a byte, 255 = none)
	// Body of the sweep stages (Kp <= 256, two sub-tiles per wave): 16x16x32 MFMAs with one candidate queue per wave (score16.hpp; its queue
	// entries carry the query beside the item: I < 2^26) for k <= 128, 32x32x16 with per-lane rings above (at k = 500 half of the leading
	// tiles' elements pass the first threshold: the rings' raw-tile hand-over is built for that).  ANNCUR_TOPK_MFMA16 / _MFMA32 force one.
	// Measured at cfg2, MI355X, same box, alternating (round 3): sweep launches 0.457 ms (16x16x32) vs 0.479 (32x32x16) at equal stage
	// split, 0.450 with the split below; a mixed plan (first stage 32x32x16, later stages 16x16x32) was level with 16x16x32 throughout.
	if (S > P.n_tiles) S = P.n_tiles;
	P.tiles_per_split = (P.n_tiles + S - 1) / S;
	P.S = (P.n_tiles + P.tiles_per_split - 1) / P.tiles_per_split;
	int S0 = slots / P.n_rb;
	if (S0 < 1) S0 = 1;
	if (S0 > P.n_st) S0 = P.n_st;
	P.st_per_split = (P.n_st + S0 - 1) / S0;
	P.S0 = (P.n_st + P.st_per_split - 1) / P.st_per_split;
	// candidate segments per (query, item split): two (lane halves) in the 32x32x16 sweep, ONE in the 16x16x32 sweep (wave-level queue)
	P.lg = (P.body16 || P.bodyq1 || P.bodyef) ? 1 : 2;
	// expected survivors per query ~ 1.3 k * (tiles / sample tiles), spread over lg S lane segments
	// (segment capacity -- hence the workspace size -- is planned for the strided sample whatever the hint; with item rows ordered
	//  by descending norm the leading sample's threshold lets ~40 % fewer elements through: measured on the synthetic protocol)
	const double exp_hits_cap = 1.3 * k * ((double)P.n_tiles / P.n_st);
	const double exp_hits = (leading ? 0.65 : 1.0) * exp_hits_cap;
	const double per_seg = exp_hits_cap / ((double)P.lg * P.S);
	int capg = next_pow2((int)(4.0 * per_seg) + 32);
	if (capg < 64) capg = 64;
	if (capg > 16384) capg = 16384;
	if (const char *dbg = knob("ANNCUR_DEBUG_CAPG")) capg = atoi(dbg);
	P.capg = capg;
	// queue window: keep the expected hits per (lane, sub-tile) window near 0.5 so that 8 slots overflow with p ~ 1e-9
	const double per_lane_tile = exp_hits / ((double)P.n_tiles * 2.0);  // hits per query-half per tile
	int ft = (int)(0.35 / (per_lane_tile > 1e-9 ? per_lane_tile : 1e-9));
	P.flush_tiles = ft < 1 ? 1 : (ft > 8 ? 8 : ft);
	// Threshold ladder (score16.hpp, round 5): the default 16x16x32 body raises its thresholds inside ONE launch from counts of what it keeps, so it
	// runs unstaged.  The ladder's top level = the sample's group maximum of rank k2 ~ where the k-th best of ALL items is expected to fall among
	// the sample's (k x sample items / I; the norm-ordered leading sample holds about twice its share of the high scorers), at least 3, at most k / 2.
	P.ladder = P.body16 && !P.ring16 && !P.wg8 && !no_ladder && P.n_groups <= 4096 && P.chunk > 0;
	if (const char *dbg = knob("ANNCUR_DEBUG_LADDER")) P.ladder = P.ladder && atoi(dbg) != 0;
	{
		double r = (double)k * ((double)P.n_st * TILE_I / (double)I) * (leading ? 2.0 : 1.25);
		if (const char *dbg = knob("ANNCUR_DEBUG_LADDER_K2")) r = atof(dbg);
		int k2 = (int)(r + 0.5);
		if (k2 > k / 2) k2 = k / 2;
		if (k2 < 3) k2 = 3;
		if (k2 > k) k2 = k;
		P.ladder_k2 = k2;
	}
	plan_stages(P, Q, k, exp_hits, !P.ladder && (k <= WQ_K2 ? P.lg * P.S <= WAVE : true) && P.n_tiles >= 24 * P.S, 4.0 * P.S / P.n_tiles, TILE_I);
	P.kmax = k <= 128 ? 128 : (k <= 512 ? 512 : 2048);
	size_t off = 256;
	P.off_ctr = off;    off = align256(off + (size_t)P.n_rb * 3 * 4 * N_SLICES);   // ticket counters [stage][row block][slice]: zeroed with the header, one memset
	P.off_lcnt = off;   off = align256(off + (P.ladder ? (size_t)P.n_rb_s * P.BQ_s * 16 : 0));   // ladder counter words: zeroed with the header too
	P.off_gmax = off;   off = align256(off + (size_t)Q * P.n_groups * 4);
	P.off_tval = off;   off = align256(off + (size_t)Q * k * 4);
	P.off_tidx = off;   off = align256(off + (size_t)Q * k * 4);
	P.off_segcnt = off; off = align256(off + (size_t)Q * P.lg * P.S * 4);
	P.off_tau = off;    off = align256(off + (size_t)Q * 4);
	P.off_hard = off;   off = align256(off + (size_t)Q * 4);
	P.off_owner = off;  off = align256(off + (size_t)3 * P.n_rb * (size_t)(P.n_tiles / (P.chunk > 0 ? P.chunk : P.n_tiles) + 2));   // chunk owners
	P.off_lvl = off;    off = align256(off + (P.ladder ? (size_t)P.n_rb_s * P.BQ_s * 4 * LADDER_LEVELS : 0));
	P.off_tau2 = off;   off = align256(off + (P.ladder ? (size_t)Q * 4 : 0));
	P.off_cand = off;   off = align256(off + (size_t)Q * P.lg * P.S * (size_t)P.capg * 8);
	P.total = off;
	P.ok = true;
	return P;
}

#define EV(i) do { if (ev) ANNCUR_HIP_OK(hipEventRecord(ev[i], st)); } while (0)

// contiguous item ranges per split instead of interleaved tiles (timing experiment)
bool contiguous_splits() {
	if (const char *dbg = knob("ANNCUR_DEBUG_CONTIG")) return atoi(dbg) != 0;
	return false;
}

// k <= 128: which wave-level candidate select runs (the buffer-and-compact one or the streaming one)
bool stream_select_small() {
	if (const char *dbg = knob("ANNCUR_DEBUG_STREAM128")) return atoi(dbg) != 0;
	return false;
}

// Threshold refinement between two sweep stages: tau[q] = max(tau[q], k-th best candidate collected so far).
int launch_tau_refine(const uint2 *cand, const uint32_t *seg_cnt, int nseg, int capg, int64_t Q, int k, int kmax, float *tau, int tau_stride,
					  int prefilter, hipStream_t st) {
	int rc;
	if (k <= WQ_K2 && nseg <= WAVE) {
#define LAUNCH_WTAU(KW)                                                                                                           \
		do {                                                                                                                      \
			constexpr int lds = 4 * WaveSelLayout<WqCfg<KW>::CAP>::BYTES;                                                         \
			if ((rc = anncur_ensure_dyn_lds((const void *)select_wave_kernel<true, KW>, lds)) != ANNCUR_OK) return rc;            \
			hipLaunchKernelGGL((select_wave_kernel<true, KW>), dim3((unsigned)ceil_div64(Q, 4)), dim3(256), lds, st, cand, seg_cnt, nseg, capg, Q, \
							   (uint32_t)k, (float *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr, tau, tau_stride, prefilter, (const int32_t *)nullptr); \
		} while (0)
		if (k <= WSEL_K && !stream_select_small()) LAUNCH_WTAU(128);
		else {  // streaming radix select (select_stream.hpp): 1 KB of LDS per wave
			constexpr int lds = 4 * StreamSelLayout::BYTES;
			hipLaunchKernelGGL((select_stream_kernel<true, 2>), dim3((unsigned)ceil_div64(Q, 4)), dim3(256), lds, st, cand, seg_cnt, nseg, capg, Q,
							   (uint32_t)k, (float *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr, (int32_t *)nullptr, tau, tau_stride, prefilter, (const int32_t *)nullptr);
		}
#undef LAUNCH_WTAU
	} else {
#define LAUNCH_TAU(KM)                                                                                                        \
		do {                                                                                                                  \
			if ((rc = anncur_ensure_dyn_lds((const void *)tau_block_kernel<KM>, (int)SelCfg<KM>::LDS_BYTES)) != ANNCUR_OK) return rc; \
			hipLaunchKernelGGL((tau_block_kernel<KM>), dim3((unsigned)Q), dim3(SEL_THREADS), SelCfg<KM>::LDS_BYTES, st, cand, seg_cnt, \
							   nseg, capg, (uint32_t)k, tau, tau_stride, prefilter);                                           \
		} while (0)
		if (kmax == 128) LAUNCH_TAU(128); else if (kmax == 512) LAUNCH_TAU(512); else LAUNCH_TAU(2048);
#undef LAUNCH_TAU
	}
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

// Per-query exact top-k of the collected candidates.  Fast path: one wave per query (k <= 128, <= 64 segments); what it cannot
// take (overflowed segment, fewer than k candidates) lands in hard_list for the workgroup-level kernel, which repairs it exactly.
int launch_select(const FusedPlan &P, int nseg, const SweepStages &stages, const uint2 *cand, const uint32_t *seg_cnt, const uint16_t *X, int64_t ldx,
				  const uint16_t *Et, int64_t Q, int64_t I, int KP, int k, float *out_val, int32_t *out_idx, unsigned char *ws, const float *tau,
				  int tau_stride, hipStream_t st, const int32_t *remap = nullptr) {
	int rc;
#define LAUNCH_SELECT(KM)                                                                                              \
	do {                                                                                                               \
		const size_t lds = SelCfg<KM>::LDS_BYTES + (size_t)KP * 4 + 32;                                                \
		if ((rc = anncur_ensure_dyn_lds((const void *)select_candidates_kernel<KM>, (int)lds)) != ANNCUR_OK) return rc; \
		hipLaunchKernelGGL((select_candidates_kernel<KM>), dim3(sel_grid), dim3(SEL_THREADS), lds, st, cand, seg_cnt, nseg, \
						   P.S, stages, P.capg, X, ldx, Et, I, KP, (uint32_t)k, out_val, out_idx, (uint32_t *)ws, hard_list, hard_cnt,      \
						   (P.n_stages > 1 || P.ladder) ? tau : (const float *)nullptr, tau_stride, remap);                              \
	} while (0)
	const int32_t *hard_list = nullptr;
	const uint32_t *hard_cnt = nullptr;
	unsigned sel_grid = (unsigned)Q;
	if (k <= WQ_K2 && nseg <= WAVE) {
		int32_t *hl = (int32_t *)(ws + P.off_hard);
		uint32_t *hc = (uint32_t *)(ws + 4);
#define LAUNCH_WSEL(KW)                                                                                                           \
		do {                                                                                                                      \
			constexpr int lds = 4 * WaveSelLayout<WqCfg<KW>::CAP>::BYTES;                                                         \
			if ((rc = anncur_ensure_dyn_lds((const void *)select_wave_kernel<false, KW>, lds)) != ANNCUR_OK) return rc;           \
			hipLaunchKernelGGL((select_wave_kernel<false, KW>), dim3((unsigned)ceil_div64(Q, 4)), dim3(256), lds, st, cand, seg_cnt, nseg, P.capg, Q, \
							   (uint32_t)k, out_val, out_idx, hc, hl, const_cast<float *>(tau), tau_stride, (P.n_stages > 1 || P.ladder) ? 1 : 0, remap); \
		} while (0)
#define LAUNCH_SSEL(EE)                                                                                                           \
		do {                                                                                                                      \
			constexpr int lds = 4 * StreamSelLayout::BYTES;                                                                       \
			hipLaunchKernelGGL((select_stream_kernel<false, EE>), dim3((unsigned)ceil_div64(Q, 4)), dim3(256), lds, st, cand, seg_cnt, nseg, P.capg, Q, \
							   (uint32_t)k, out_val, out_idx, hc, hl, const_cast<float *>(tau), tau_stride, (P.n_stages > 1 || P.ladder) ? 1 : 0, remap); \
		} while (0)
		if (k <= WSEL_K) { if (stream_select_small()) LAUNCH_SSEL(2); else LAUNCH_WSEL(128); }
		else if (k <= 256) LAUNCH_SSEL(4);
		else if (k <= 512) LAUNCH_SSEL(8);
		else LAUNCH_SSEL(16);
#undef LAUNCH_SSEL
#undef LAUNCH_WSEL
		ANNCUR_LAUNCH_OK();
		hard_list = hl; hard_cnt = hc;
		sel_grid = (unsigned)(Q < 1024 ? Q : 1024);
	}
	if (P.kmax == 128) LAUNCH_SELECT(128); else if (P.kmax == 512) LAUNCH_SELECT(512); else LAUNCH_SELECT(2048);
#undef LAUNCH_SELECT
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

// tau = k-th largest group maximum of the prepass (a valid lower bound on the query's k-th best score)
int launch_threshold(const FusedPlan &P, const float *gmax, int64_t Q, int k, unsigned char *ws, const float *&tau, int &tau_stride, hipStream_t st) {
	if (P.n_groups <= 4096) {  // one wave per query, keys in LDS (kth_value_wave_kernel)
		float *t = (float *)(ws + P.off_tau);
		const int rc = P.ladder ? anncur_internal_kth_value(gmax, Q, P.n_groups, P.n_groups, k, t, 1, st, /*coarse=*/1, (float *)(ws + P.off_lvl), P.ladder_k2, (float *)(ws + P.off_tau2))
								: anncur_internal_kth_value(gmax, Q, P.n_groups, P.n_groups, k, t, 1, st, /*coarse=*/1);   // (any lower bound on the k-th best will do)
		tau = t; tau_stride = 1;
		return rc;
	}
	float *tval = (float *)(ws + P.off_tval);
	tau = tval + (k - 1); tau_stride = k;
	return anncur_rowwise_topk(gmax, ANNCUR_F32, Q, P.n_groups, P.n_groups, k, tval, (int32_t *)(ws + P.off_tidx), st);
}

#ifdef ANNCUR_TIMING_EXPERIMENTS
unsigned long long *g_stamps = nullptr;  // diagnostic build: the last sweep launch's per-workgroup {cycles, 100 MHz ticks}
#endif

// ------------------------------------------------------------------ a8: the exact scan co-scheduled with the retrieval (anncur_eval_topk)
// The retrieval is a chain of MFMA-bound sweep launches with latency-bound launches in between (threshold, refinement, select): one
// wave per query on a few KB each, most of the chip idle.  The exact top-k scan of A is HBM-bound and independent of all of them.
// Both kinds want every CU to themselves (the sweep fills the register file and the LDS; a scan that arrives first occupies the CUs
// the sweep is waiting for), so they are not left to race on two streams: the scan is cut into row chunks, and chunk i is launched on
// the auxiliary stream exactly beside the i-th latency-bound launch -- fork after the launch in front of it, join before the next
// sweep -- where it finds the chip free and hides that launch.  Chunks are whole rounds of the scan (rows in flight on the chip) so
// that none ends on a mostly empty round; the last takes the remainder.
struct CoScan {
	hipStream_t aux;
	hipEvent_t *ev;            // [2 * slots]: fork / join pairs
	const void *A; int a_dtype; int64_t lda, Q, I; int32_t k;
	float *val; int32_t *idx;
	int slots, next;           // chunks planned / launched so far
	int64_t row_end[8];        // chunk i = rows [row_end[i-1], row_end[i])
};

int co_plan(CoScan &co, int slots) {
	if (slots > 8) slots = 8;
	if (slots < 1) slots = 1;
	const int64_t round = anncur_internal_scan_rows_in_flight(co.a_dtype);
	const int64_t rounds = co.Q / round;                      // whole rounds available
	int64_t per = rounds / slots;                             // whole rounds per chunk (the last chunk takes what is left)
	if (per < 1) {                                            // fewer rounds than slots: one round per chunk while they last
		per = 1;
		slots = rounds >= 1 ? (int)rounds : 1;
	}
	int64_t r = 0;
	for (int i = 0; i < slots; ++i) { r = (i == slots - 1) ? co.Q : r + per * round; co.row_end[i] = r < co.Q ? r : co.Q; }
	co.slots = slots;
	co.next = 0;
	return anncur_event_pool(&co.ev, 2 * slots);
}
// launch the next chunk on the auxiliary stream behind everything enqueued on `st` so far
int co_fork(CoScan *co, hipStream_t st) {
	if (!co || co->next >= co->slots) return ANNCUR_OK;
	const int i = co->next++;
	const int64_t r0 = i ? co->row_end[i - 1] : 0, r1 = co->row_end[i];
	if (r1 <= r0) return ANNCUR_OK;
	ANNCUR_HIP_OK(hipEventRecord(co->ev[2 * i], st));
	ANNCUR_HIP_OK(hipStreamWaitEvent(co->aux, co->ev[2 * i], 0));
	const char *a = (const char *)co->A + (size_t)r0 * (size_t)co->lda * dtype_size(co->a_dtype);
	const int rc = anncur_rowwise_topk(a, co->a_dtype, r1 - r0, co->I, co->lda, co->k, co->val + r0 * co->k, co->idx + r0 * co->k, co->aux);
	if (rc != ANNCUR_OK) return rc;
	ANNCUR_HIP_OK(hipEventRecord(co->ev[2 * i + 1], co->aux));
	return ANNCUR_OK;
}
// `st` waits for the chunks launched so far
int co_join(CoScan *co, hipStream_t st) {
	if (!co || co->next < 1) return ANNCUR_OK;
	const int i = co->next - 1;
	if (co->row_end[i] > (i ? co->row_end[i - 1] : 0)) ANNCUR_HIP_OK(hipStreamWaitEvent(st, co->ev[2 * i + 1], 0));
	return ANNCUR_OK;
}
// whatever the plan left unlaunched (fewer latency-bound launches than chunks), then the final join
int co_finish(CoScan *co, hipStream_t st) {
	if (!co) return ANNCUR_OK;
	int rc;
	while (co->next < co->slots) {
		if ((rc = co_fork(co, st)) != ANNCUR_OK) return rc;
		if ((rc = co_join(co, st)) != ANNCUR_OK) return rc;
	}
	return co_join(co, st);
}

// A launch failed after chunks of the exact scan were forked onto the auxiliary stream: the launch stream still waits for every chunk
// issued so far, so that no work is left running into the caller's buffers behind its back and a stream capture stays joined.  (The
// exact result is then incomplete: outputs are undefined whenever a call returns an error.)
void co_abort(CoScan *co, hipStream_t st) {
	if (!co) return;
	for (int i = 0; i < co->next; ++i)
		if (co->row_end[i] > (i ? co->row_end[i - 1] : 0)) (void)hipStreamWaitEvent(st, co->ev[2 * i + 1], 0);
}

// error_lds_kernel / evalf_kernel address the workgroup's tile of the exact matrix as a uniform 64-bit base + a 32-bit byte offset per lane,
// (row within the row block) x pitch x 2 + chunk: the routes that use them are taken only while the last row's offset stays below 2^32
// (bq = 256 rows: pitch < 8 421 504 elements).  ADVICE r4: unchecked, a longer pitch read the wrong rows silently.
bool exact_tile_offsets_fit(int64_t lda, int bq) { return lda >= 0 && (uint64_t)(bq - 1) * (uint64_t)lda * 2u + 64u < ((uint64_t)1 << 32); }

// anncur_eval_fused: the exact matrix and the two per-row sums the sweep stages also produce (evalf_kernel)
struct EvalArgs { const uint16_t *A; int64_t lda; float *err_sq, *norm_sq; const uint16_t *Et_hint; };   // Et_hint: see anncur_eval_fused_ex (may be null)

template <int KP, int QTV = FusedCfg<KP>::QT>
int launch_fused(const FusedPlan &P, const void *X, int64_t ldx, const void *Et, int64_t Q, int64_t I, int k, float *out_val,
				 int32_t *out_idx, unsigned char *ws, hipStream_t st, hipEvent_t *ev, CoScan *co = nullptr, const int32_t *item_ids = nullptr,
				 const EvalArgs *ea = nullptr) {
	using Cfg = FusedCfg<KP, QTV>;
	FusedParams p{};
	p.X = (const uint16_t *)X; p.ldx = ldx; p.Et = (const uint16_t *)Et; p.Q = Q; p.I = I;
	p.n_tiles = P.n_tiles; p.n_full_tiles = P.n_full; p.S = P.S; p.tiles_per_split = P.tiles_per_split;
	p.tile_begin = 0; p.tile_end = P.n_tiles; p.carry = 0; p.tile_step = 1;
	p.n_st = P.n_st; p.S0 = P.S0; p.st_per_split = P.st_per_split; p.sample_leading = P.leading;
	p.gmax = (float *)(ws + P.off_gmax); p.n_groups = P.n_groups;
	p.cand = (uint2 *)(ws + P.off_cand); p.seg_cnt = (uint32_t *)(ws + P.off_segcnt); p.capg = P.capg; p.flush_tiles = P.flush_tiles;
	p.tau_bias = 0.f;
	p.chunk_tiles = 0; p.n_chunks = 0; p.chunk_ctr = nullptr; p.chunk_owner = nullptr;
	p.nfb = (uint32_t *)ws;
	p.ring_stagger = 0; p.ring_spin_sleep = 1;
	if (const char *dbg = knob("ANNCUR_DEBUG_RING_STAGGER")) p.ring_stagger = atoi(dbg);
	if (const char *dbg = knob("ANNCUR_DEBUG_RING_SLEEP")) p.ring_spin_sleep = atoi(dbg);
	p.nseg = P.lg * P.S;
	p.prio_mode = 0;
	if (const char *dbg = knob("ANNCUR_DEBUG_PRIO")) p.prio_mode = atoi(dbg);
	p.ladder_on = P.ladder ? 1 : 0; p.ladder_k = (uint32_t)k; p.ladder_mask = (uint32_t)(LADDER_PERIOD - 1);
#ifdef ANNCUR_TIMING_EXPERIMENTS
	if (const char *dbg = getenv("ANNCUR_DEBUG_LADDER_PERIOD")) { int v = atoi(dbg); if (v >= 1 && (v & (v - 1)) == 0) p.ladder_mask = (uint32_t)(v - 1); }
#endif
	p.ladder = (const float *)(ws + P.off_lvl); p.ladder_cnt = (uint32_t *)(ws + P.off_lcnt); p.tau_final = (float *)(ws + P.off_tau2);
#ifdef ANNCUR_TIMING_EXPERIMENTS
	if (getenv("ANNCUR_DEBUG_STAMPS")) {
		if (!g_stamps) {
			ANNCUR_HIP_OK(hipMalloc((void **)&g_stamps, 13 * 8192 * sizeof(unsigned long long)));
			ANNCUR_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(d_sweep_stamps), &g_stamps, sizeof(g_stamps)));
		}
		ANNCUR_HIP_OK(hipMemsetAsync(g_stamps, 0, 13 * 8192 * sizeof(unsigned long long), st));
	}
#endif
#ifdef ANNCUR_TIMING_EXPERIMENTS  // (the -DANNCUR_TIMING_EXPERIMENTS build of scripts/fused_microbench.py only: results become wrong)
	{ const char *dbg = getenv("ANNCUR_DEBUG_TAU_BIAS"); p.tau_bias = dbg ? (float)atof(dbg) : 0.f; }
	if (getenv("ANNCUR_DEBUG_NOSTORE")) { p.capg = 0; item_ids = nullptr; }  // every candidate is dropped at the store (the select then reads slots nobody wrote: no id map through them)
#endif

	const int chunk = (P.chunk > 0 && (Cfg::QT == 2 || P.bodyq1)) ? P.chunk : 0;   // (the one-sub-tile bodies with per-lane rings keep static shares)
	const int owner_stride = P.n_rb * (P.n_tiles / (chunk > 0 ? chunk : P.n_tiles) + 2);
	ANNCUR_HIP_OK(hipMemsetAsync(ws, 0, (chunk > 0 || P.ladder) ? P.off_gmax : 256, st));   // header (+ the stages' ticket counters, the ladder's counter words)
	EV(0);
	// 1. prepass
	p.n_wg = P.n_rb * P.S0;
	{
		// anncur_eval_fused_ex's hint: the prepass samples the LEADING tiles of the norm-ordered copy (the likeliest high scorers: a tighter first
		// threshold), the sweep keeps the item order the exact tiles need.  Any subset of the items gives a valid threshold.
		FusedParams pp = p;
		if (ea && ea->Et_hint) { pp.Et = ea->Et_hint; pp.sample_leading = 1; }
		if (P.group == 16)
			hipLaunchKernelGGL((score_kernel<KP, 0, 16, false, false, QTV>), dim3(p.n_wg), dim3(256), 2 * Cfg::TILE_BYTES, st, pp);
		else
			hipLaunchKernelGGL((score_kernel<KP, 0, 4, false, false, QTV>), dim3(p.n_wg), dim3(256), 2 * Cfg::TILE_BYTES, st, pp);
	}
	ANNCUR_LAUNCH_OK();
	EV(1);
	// 2. tau = k-th largest group maximum (beside it: the first chunk of the exact scan, anncur_eval_topk only)
	int rc;
	if ((rc = co_fork(co, st)) != ANNCUR_OK) return rc;
	rc = launch_threshold(P, p.gmax, Q, k, ws, p.tau, p.tau_stride, st);
	if (rc != ANNCUR_OK) return rc;
	if ((rc = co_join(co, st)) != ANNCUR_OK) return rc;
	EV(2);
	// 3. sweep, in stages; between stages the thresholds are raised from the candidates collected so far.
	// Item split s of a stage sweeps the tiles begin + s, begin + s + S, ... (interleaved), not a contiguous range: with norm-ordered rows
	// the survivors crowd into the leading tiles, all the workgroups of a stage run at once, and with contiguous ranges the stage took
	// as long as its FIRST split (cfg2: the first stage, 22 % of the tiles, 0.236 ms against 0.306 ms for the other 78 %).
	// (score16_kernel keeps contiguous ranges)
	const int tile_step = (!P.body16 && !P.bodyq1 && !P.bodyef && P.S > 1 && chunk == 0 && !contiguous_splits()) ? P.S : 1;
	p.n_wg = P.n_rb_s * P.S;
	if ((rc = anncur_ensure_dyn_lds((const void *)score_kernel<KP, 1, 16, false, false, QTV>, Cfg::LDS_BYTES)) != ANNCUR_OK) return rc;
	for (int stg = 0, prev = 0; stg < P.n_stages; prev = P.stage_end[stg], ++stg) {
		EV(5 + 2 * stg);
		p.tile_begin = prev; p.tile_end = P.stage_end[stg]; p.tiles_per_split = P.stage_tps[stg];
		p.flush_tiles = P.stage_flush[stg]; p.carry = stg > 0;
		p.tile_step = tile_step;
		if (chunk > 0) {
			// (row-block-major work ids -- the workgroups of a row block on ONE XCD -- were measured for the ticket schedule, round 3, one box:
			//  cfg4 shape sweep 5.13 -> 6.28 ms and L2-miss traffic 9.0 -> 13.0 GB per launch, cfg2 0.451 -> 0.480 ms: split-major stays)
			p.rb_major = 0;
			if (const char *dbg = knob("ANNCUR_DEBUG_RB_MAJOR")) p.rb_major = atoi(dbg);
			p.chunk_tiles = chunk; p.n_chunks = (p.tile_end - p.tile_begin + chunk - 1) / chunk;
			// XCD-sliced tickets: the wave-queue bodies (Kp = 512: the shape whose sweep was close to the fabric's bandwidth; Kp <= 256: traffic only); [row block][slice] counters
			// (Kp <= 256: measured at cfg2, one process, interleaved -- sweep launches 0.5030 ms sliced vs 0.4914 unsliced, bare loops level:
			//  there the fabric is nowhere near its limit and the steals' blocking atomics cost more than the traffic they save (328 -> 180 MB
			//  per launch); the score16 body keeps the code path (ANNCUR_DEBUG_SLICED=2 in the experiments build) but runs unsliced)
			p.sliced = (P.bodyq1 || P.bodyq16) ? 1 : 0;
			if (const char *dbg = knob("ANNCUR_DEBUG_SLICED")) p.sliced = (atoi(dbg) == 2 && P.body16 && !P.ring16) ? 1 : (p.sliced && atoi(dbg) != 0);
			p.chunks_per_slice = (p.n_chunks + N_SLICES - 1) / N_SLICES;
			p.chunk_ctr = (uint32_t *)(ws + P.off_ctr) + (size_t)stg * P.n_rb * N_SLICES;
			p.chunk_owner = (uint8_t *)(ws + P.off_owner) + (size_t)stg * owner_stride;
		}
		bool launched = false;
#ifdef ANNCUR_TIMING_EXPERIMENTS
		{ const char *dbg = getenv("ANNCUR_DEBUG_STAMP_STAGE"); p.debug_stamp = dbg ? (atoi(dbg) == stg) : (stg == P.n_stages - 1); }  // which launch leaves its stamps
		{ const char *dbg = getenv("ANNCUR_DEBUG_FLUSH_TILES"); if (dbg) p.flush_tiles = atoi(dbg); }
		if (getenv("ANNCUR_DEBUG_GEMM_NOSYNC")) {  // MFMA + LDS fragment reads, no staging, no barriers
			if ((rc = anncur_ensure_dyn_lds((const void *)score_kernel<KP, 3, 16, false, false, QTV>, Cfg::LDS_BYTES)) != ANNCUR_OK) return rc;
			hipLaunchKernelGGL((score_kernel<KP, 3, 16, false, false, QTV>), dim3(p.n_wg), dim3(256), Cfg::LDS_BYTES, st, p);
			launched = true;
		} else if (getenv("ANNCUR_DEBUG_GEMM_ONLY")) {  // no candidates are produced
			if ((rc = anncur_ensure_dyn_lds((const void *)score_kernel<KP, 2, 16, false, false, QTV>, Cfg::LDS_BYTES)) != ANNCUR_OK) return rc;
			hipLaunchKernelGGL((score_kernel<KP, 2, 16, false, false, QTV>), dim3(p.n_wg), dim3(256), Cfg::LDS_BYTES, st, p);
			launched = true;
		}
#endif
		if constexpr (KP <= 256 && QTV == 2) {  // anncur_eval_fused: candidates + error sums in one pass (score_evalf.hpp)
			if (!launched && P.bodyef) {
				if (!ea) { anncur_set_error("launch_fused: evalf plan without the exact matrix"); return ANNCUR_E_INVALID; }
				if ((rc = anncur_ensure_dyn_lds((const void *)evalf_kernel<KP>, EvalFCfg<KP>::LDS_BYTES)) != ANNCUR_OK) return rc;
				hipLaunchKernelGGL((evalf_kernel<KP>), dim3(p.n_wg), dim3(256), EvalFCfg<KP>::LDS_BYTES, st, p, ea->A, ea->lda, ea->err_sq, ea->norm_sq);
				launched = true;
			}
		}
		if constexpr (KP == 512) {  // Kp = 512 with the wave-level queue and tickets (score_q1.hpp)
			if (!launched && P.bodyq16) {
				if ((rc = anncur_ensure_dyn_lds((const void *)scoreq16_kernel<KP>, FusedQ1Cfg<KP>::LDS_BYTES)) != ANNCUR_OK) return rc;
				hipLaunchKernelGGL((scoreq16_kernel<KP>), dim3(p.n_wg), dim3(256), FusedQ1Cfg<KP>::LDS_BYTES, st, p);
				launched = true;
			}
#ifdef ANNCUR_TIMING_EXPERIMENTS
			if (!launched && P.bodyq1) {
				if ((rc = anncur_ensure_dyn_lds((const void *)scoreq1_kernel<KP>, FusedQ1Cfg<KP>::LDS_BYTES)) != ANNCUR_OK) return rc;
				hipLaunchKernelGGL((scoreq1_kernel<KP>), dim3(p.n_wg), dim3(256), FusedQ1Cfg<KP>::LDS_BYTES, st, p);
				launched = true;
			}
#endif
		}
#ifdef ANNCUR_TIMING_EXPERIMENTS
		if constexpr (KP >= 128 && KP <= 256 && QTV == 2) {  // 16x16x32 sweep, 8-wave workgroups with the flag-synchronised tile ring (score16r.hpp)
			if (!launched && P.ring16) {
				if ((rc = anncur_ensure_dyn_lds((const void *)score16r_kernel<KP>, Ring16Cfg<KP>::LDS_BYTES)) != ANNCUR_OK) return rc;
				hipLaunchKernelGGL((score16r_kernel<KP>), dim3(p.n_wg), dim3(512), Ring16Cfg<KP>::LDS_BYTES, st, p);
				launched = true;
			}
		}
#endif
#ifdef ANNCUR_TIMING_EXPERIMENTS
		// score16_kernel<Kp, 8>: the barrier body in 8-wave workgroups (512 queries, one per CU: half the DMA pieces per wave).  Measured (round 4,
		// one process): sweep launches 0.5125 vs 0.4874 ms, bare 0.436 vs 0.395 -- with both waves of a SIMD in ONE workgroup the partners run in
		// lockstep through DMA issue, barrier and MFMA section; two independent 4-wave workgroups per CU are 10 % faster.  Experiments build only.
		if constexpr (KP >= 128 && KP <= 256 && QTV == 2) {
			if (!launched && P.body16 && P.wg8) {
				constexpr int lds8 = Fused16Cfg<KP, 8>::LDS_BYTES;
				if ((rc = anncur_ensure_dyn_lds((const void *)score16_kernel<KP, 8>, lds8)) != ANNCUR_OK) return rc;
				hipLaunchKernelGGL((score16_kernel<KP, 8>), dim3(p.n_wg), dim3(512), lds8, st, p);
				launched = true;
			}
		}
#endif
		if constexpr (KP <= 256 && QTV == 2) {  // 16x16x32 sweep (score16.hpp): one segment per (query, item split)
			if (!launched && P.body16) {
				if ((rc = anncur_ensure_dyn_lds((const void *)score16_kernel<KP>, Fused16Cfg<KP>::LDS_BYTES)) != ANNCUR_OK) return rc;
				hipLaunchKernelGGL((score16_kernel<KP>), dim3(p.n_wg), dim3(256), Fused16Cfg<KP>::LDS_BYTES, st, p);
				launched = true;
			}
		}
		if constexpr (Cfg::QT == 2) {  // (the branch-free filter lives in the staggered path)
			if (!launched && P.stage_pred[stg]) {
				if ((rc = anncur_ensure_dyn_lds((const void *)score_kernel<KP, 1, 16, true, false, QTV>, Cfg::LDS_BYTES)) != ANNCUR_OK) return rc;
				hipLaunchKernelGGL((score_kernel<KP, 1, 16, true, false, QTV>), dim3(p.n_wg), dim3(256), Cfg::LDS_BYTES, st, p);
				launched = true;
			}
		}
#ifdef ANNCUR_TIMING_EXPERIMENTS
		if constexpr (KP == 512) {
			if (!launched && getenv("ANNCUR_DEBUG_PLAIN512")) {  // the sweep without the cross-tile software pipeline
				if ((rc = anncur_ensure_dyn_lds((const void *)score_kernel<KP, 1, 16, false, true, QTV>, Cfg::LDS_BYTES)) != ANNCUR_OK) return rc;
				hipLaunchKernelGGL((score_kernel<KP, 1, 16, false, true, QTV>), dim3(p.n_wg), dim3(256), Cfg::LDS_BYTES, st, p);
				launched = true;
			}
		}
		if constexpr (Cfg::QT == 2) {
			if (!launched && getenv("ANNCUR_DEBUG_INLINE_HIT")) {
				if ((rc = anncur_ensure_dyn_lds((const void *)score_kernel<KP, 1, 16, false, true, QTV>, Cfg::LDS_BYTES)) != ANNCUR_OK) return rc;
				hipLaunchKernelGGL((score_kernel<KP, 1, 16, false, true, QTV>), dim3(p.n_wg), dim3(256), Cfg::LDS_BYTES, st, p);
				launched = true;
			}
		}
		if (!launched && getenv("ANNCUR_DEBUG_ONE_WG")) {  // padded LDS request: a second workgroup does not fit on the CU
			if ((rc = anncur_ensure_dyn_lds((const void *)score_kernel<KP, 1, 16, false, false, QTV>, 84 * 1024)) != ANNCUR_OK) return rc;
			hipLaunchKernelGGL((score_kernel<KP, 1, 16, false, false, QTV>), dim3(p.n_wg), dim3(256), 84 * 1024, st, p);
			launched = true;
		}
#endif
		if (!launched) hipLaunchKernelGGL((score_kernel<KP, 1, 16, false, false, QTV>), dim3(p.n_wg), dim3(256), Cfg::LDS_BYTES, st, p);
		ANNCUR_LAUNCH_OK();
		EV(6 + 2 * stg);
		if (stg + 1 < P.n_stages) {
			if ((rc = co_fork(co, st)) != ANNCUR_OK) return rc;
			if ((rc = launch_tau_refine(p.cand, p.seg_cnt, P.lg * P.S, P.capg, Q, k, P.kmax, const_cast<float *>(p.tau), p.tau_stride, stg > 0 ? 1 : 0, st)) != ANNCUR_OK)
				return rc;
			if ((rc = co_join(co, st)) != ANNCUR_OK) return rc;
		}
	}
	EV(3);
	// 4. select (beside it: the last chunk of the exact scan; the caller joins)
	if ((rc = co_fork(co, st)) != ANNCUR_OK) return rc;
	SweepStages stages{};
	stages.n = P.n_stages;
	stages.stride = tile_step;
	stages.chunk = chunk; stages.bq = P.BQ_s;
	for (int g = 0, prev = 0; g < P.n_stages; prev = P.stage_end[g], ++g) {
		stages.begin[g] = prev; stages.end[g] = P.stage_end[g]; stages.tps[g] = P.stage_tps[g];
		stages.n_chunks[g] = chunk > 0 ? (P.stage_end[g] - prev + chunk - 1) / chunk : 0;
		stages.owner[g] = (const uint8_t *)(ws + P.off_owner) + (size_t)g * owner_stride;
	}
	// (ladder: the prefilter is the highest threshold a workgroup ended the sweep with -- k candidates at or above it were counted)
	if ((rc = launch_select(P, P.lg * P.S, stages, p.cand, p.seg_cnt, p.X, ldx, p.Et, Q, I, KP, k, out_val, out_idx, ws, P.ladder ? p.tau_final : p.tau,
							P.ladder ? 1 : p.tau_stride, st, item_ids)) != ANNCUR_OK) return rc;
	EV(4);
	return ANNCUR_OK;
}

// ------------------------------------------------------------------ wide inner dimension (Kp > 512): score_wide.hpp
constexpr int WIDE_KP_MAX = 4096;
bool wide_kp(int KP) { return KP > 512 && KP <= WIDE_KP_MAX && (KP % 128) == 0; }

// Same plan structure in units of 256-item block tiles (P.n_tiles, stage_end, tiles_per_split); candidate segments per query:
// 4 S = lane half x wave item half x item split.
FusedPlan plan_wide(int64_t Q, int64_t I, int KP, int k, bool leading = false) {
	FusedPlan P{};
	P.ok = false;
	P.leading = leading ? 1 : 0;
	if (!wide_kp(KP)) return P;
	if (k < 1 || k > ANNCUR_MAX_TOPK || Q < 1 || I < 1 || I >= (int64_t)0x7fffffff - 512 || k > I) return P;
	P.QT = 2;
	P.BQ = WBN;
	P.n_rb = (int)ceil_div64(Q, WBN);
	P.n_tiles = (int)ceil_div64(I, WBM);
	P.n_full = (int)(I / WBM);
	const int target = (4 * k > 512) ? 4 * k : 512;  // groups: enough that the k-th largest group maximum is a tight bound
	const int n16 = (target + 15) / 16;               // 16 groups of 16 items per sampled block tile
	if ((int64_t)n16 * 8 <= P.n_full) { P.group = 16; P.n_st = n16; }
	else { P.group = 4; P.n_st = (target + 63) / 64; }
	if ((int64_t)P.n_st * 2 > P.n_full) return P;     // problem too small for a sampled threshold: dense GEMM + scan
	P.n_groups = P.n_st * (P.group == 16 ? 16 : 64);
	if (P.n_groups < k) return P;
	const int slots = num_cu();                        // one 512-thread workgroup (128 KiB of LDS) per CU
	int S = slots / P.n_rb;
	if (S < 1) S = 1;
	if (S > 64) S = 64;
	if (k <= WQ_K2 && S > 16) S = 16;                  // 4 S <= 64 segments: wave-level refinement / select kernels
	if (S > P.n_tiles) S = P.n_tiles;
	P.tiles_per_split = (P.n_tiles + S - 1) / S;
	P.S = (P.n_tiles + P.tiles_per_split - 1) / P.tiles_per_split;
	int S0 = slots / P.n_rb;
	if (S0 < 1) S0 = 1;
	if (S0 > P.n_st) S0 = P.n_st;
	P.st_per_split = (P.n_st + S0 - 1) / S0;
	P.S0 = (P.n_st + P.st_per_split - 1) / P.st_per_split;
	const double exp_hits_cap = 1.3 * k * ((double)P.n_tiles / P.n_st);
	const double exp_hits = (leading ? 0.65 : 1.0) * exp_hits_cap;
	const double per_seg = exp_hits_cap / (4.0 * P.S);
	int capg = next_pow2((int)(4.0 * per_seg) + 32);
	if (capg < 64) capg = 64;
	if (capg > 16384) capg = 16384;
	while (capg > 64 && (int64_t)capg * 4 * P.S >= (1 << 21)) capg >>= 1;  // a workgroup's 256 x 4 S segments span < 4 GiB (32-bit store offsets)
	P.capg = capg;
	P.flush_tiles = 1;
	plan_stages(P, Q, k, exp_hits, (k <= WQ_K2 ? 4 * P.S <= WAVE : true) && P.n_tiles >= 4 * P.S, 1.0 * P.S / P.n_tiles, WBM);
	P.kmax = k <= 128 ? 128 : (k <= 512 ? 512 : 2048);
	size_t off = 256;
	P.off_gmax = off;   off = align256(off + (size_t)Q * P.n_groups * 4);
	P.off_tval = off;   off = align256(off + (size_t)Q * k * 4);
	P.off_tidx = off;   off = align256(off + (size_t)Q * k * 4);
	P.off_segcnt = off; off = align256(off + (size_t)Q * 4 * P.S * 4);
	P.off_tau = off;    off = align256(off + (size_t)Q * 4);
	P.off_hard = off;   off = align256(off + (size_t)Q * 4);
	P.off_cand = off;   off = align256(off + (size_t)Q * 4 * P.S * (size_t)P.capg * 8);
	P.total = off;
	P.ok = true;
	return P;
}

int launch_wide(const FusedPlan &P, const void *X, int64_t ldx, const void *Et, int64_t Q, int64_t I, int KP, int k, float *out_val,
				int32_t *out_idx, unsigned char *ws, hipStream_t st, hipEvent_t *ev, CoScan *co = nullptr, const int32_t *item_ids = nullptr) {
	WideParams p{};
	p.X = (const uint16_t *)X; p.ldx = ldx; p.Et = (const uint16_t *)Et; p.et_rows = ceil_div64(I, TILE_I) * TILE_I; p.Q = Q; p.I = I; p.Kp = KP;
	p.n_rb = P.n_rb; p.S = P.S;
	p.n_st = P.n_st; p.S0 = P.S0; p.st_per_split = P.st_per_split; p.sample_leading = P.leading; p.n_bt_full = P.n_full;
	p.gmax = (float *)(ws + P.off_gmax); p.n_groups = P.n_groups;
	p.cand = (uint2 *)(ws + P.off_cand); p.seg_cnt = (uint32_t *)(ws + P.off_segcnt); p.capg = P.capg;
	p.tau_bias = 0.f;
#ifdef ANNCUR_TIMING_EXPERIMENTS
	{ const char *dbg = getenv("ANNCUR_DEBUG_TAU_BIAS"); p.tau_bias = dbg ? (float)atof(dbg) : 0.f; }
#endif
	int rc;
	ANNCUR_HIP_OK(hipMemsetAsync(ws, 0, 256, st));
	EV(0);
	// 1. prepass over the sampled block tiles
	p.n_wg = P.n_rb * P.S0;
	if (P.group == 16) {
		if ((rc = anncur_ensure_dyn_lds((const void *)wide_kernel<0, 16>, W_LDS_TOTAL)) != ANNCUR_OK) return rc;
		hipLaunchKernelGGL((wide_kernel<0, 16>), dim3(p.n_wg), dim3(512), W_LDS_TOTAL, st, p);
	} else {
		if ((rc = anncur_ensure_dyn_lds((const void *)wide_kernel<0, 4>, W_LDS_TOTAL)) != ANNCUR_OK) return rc;
		hipLaunchKernelGGL((wide_kernel<0, 4>), dim3(p.n_wg), dim3(512), W_LDS_TOTAL, st, p);
	}
	ANNCUR_LAUNCH_OK();
	EV(1);
	// 2. threshold
	if ((rc = co_fork(co, st)) != ANNCUR_OK) return rc;
	if ((rc = launch_threshold(P, p.gmax, Q, k, ws, p.tau, p.tau_stride, st)) != ANNCUR_OK) return rc;
	if ((rc = co_join(co, st)) != ANNCUR_OK) return rc;
	EV(2);
	// 3. sweep in stages
	p.n_wg = P.n_rb * P.S;
	if ((rc = anncur_ensure_dyn_lds((const void *)wide_kernel<1, 16>, W_LDS_TOTAL)) != ANNCUR_OK) return rc;
	for (int stg = 0, prev = 0; stg < P.n_stages; prev = P.stage_end[stg], ++stg) {
		EV(5 + 2 * stg);
		p.bt_begin = prev; p.bt_end = P.stage_end[stg]; p.bt_per_split = P.stage_tps[stg]; p.carry = stg > 0;
		hipLaunchKernelGGL((wide_kernel<1, 16>), dim3(p.n_wg), dim3(512), W_LDS_TOTAL, st, p);
		ANNCUR_LAUNCH_OK();
		EV(6 + 2 * stg);
		if (stg + 1 < P.n_stages) {
			if ((rc = co_fork(co, st)) != ANNCUR_OK) return rc;
			if ((rc = launch_tau_refine(p.cand, p.seg_cnt, 4 * P.S, P.capg, Q, k, P.kmax, const_cast<float *>(p.tau), p.tau_stride, stg > 0 ? 1 : 0, st)) != ANNCUR_OK)
				return rc;
			if ((rc = co_join(co, st)) != ANNCUR_OK) return rc;
		}
	}
	EV(3);
	// 4. select (stage ranges in 32-item tiles for the repair path)
	if ((rc = co_fork(co, st)) != ANNCUR_OK) return rc;
	SweepStages stages{};
	stages.n = P.n_stages;
	const int n_tiles32 = (int)ceil_div64(I, TILE_I), u = WBM / TILE_I;
	for (int g = 0, prev = 0; g < P.n_stages; prev = P.stage_end[g], ++g) {
		stages.begin[g] = prev * u;
		stages.end[g] = P.stage_end[g] * u < n_tiles32 ? P.stage_end[g] * u : n_tiles32;
		stages.tps[g] = P.stage_tps[g] * u;
	}
#if defined(ANNCUR_V_WIDE_NOREAD) || defined(ANNCUR_V_WIDE_MFMAONLY) || defined(ANNCUR_V_WIDE_NODMA_NOREAD) || defined(ANNCUR_V_WIDE_NODMA) || defined(ANNCUR_V_WIDE_NOBAR) || defined(ANNCUR_V_WIDE_NOFILTER) || defined(ANNCUR_V_WIDE_BASE)
	// (ablation builds of wide_kernel: the candidates are garbage, the select -- which would repair every query from scratch -- is skipped)
	(void)item_ids; (void)out_val; (void)out_idx;
#else
	if ((rc = launch_select(P, 4 * P.S, stages, p.cand, p.seg_cnt, p.X, ldx, p.Et, Q, I, KP, k, out_val, out_idx, ws, p.tau, p.tau_stride, st, item_ids)) != ANNCUR_OK) return rc;
#endif
	EV(4);
	return ANNCUR_OK;
}
#undef EV

FusedPlan plan_any(int64_t Q, int64_t I, int KP, int k, int flags = 0) {
	const bool leading = (flags & ANNCUR_TOPK_LEADING_SAMPLE) != 0, mfma16 = (flags & ANNCUR_TOPK_MFMA16) != 0, qt1 = (flags & ANNCUR_TOPK_QT1) != 0;
	return wide_kp(KP) ? plan_wide(Q, I, KP, k, leading) : plan_fused(Q, I, KP, k, leading, mfma16 && !qt1, qt1, (flags & ANNCUR_TOPK_MFMA32) != 0, (flags & ANNCUR_TOPK_RING) != 0,
																	  false, (flags & ANNCUR_TOPK_STAGED) != 0);
}
bool ring_flag_ok(int flags) {
#ifdef ANNCUR_TIMING_EXPERIMENTS
	(void)flags; return true;
#else
	return (flags & ANNCUR_TOPK_RING) == 0;
#endif
}
constexpr int TOPK_FLAGS = ANNCUR_TOPK_LEADING_SAMPLE | ANNCUR_TOPK_MFMA16 | ANNCUR_TOPK_QT1 | ANNCUR_TOPK_MFMA32 | ANNCUR_TOPK_RING | ANNCUR_TOPK_STAGED;

}  // namespace

extern "C" size_t anncur_score_topk_workspace_bytes(int64_t Q, int64_t I, int32_t Kp, int32_t k) {
	const FusedPlan P = plan_any(Q, I, Kp, k);
	if (!P.ok) return 0;
	size_t t = P.total;
	for (int flags : {ANNCUR_TOPK_MFMA16, ANNCUR_TOPK_MFMA32, ANNCUR_TOPK_QT1, ANNCUR_TOPK_RING, ANNCUR_TOPK_STAGED}) {  // (whatever variant flag the call will carry)
		const FusedPlan V = plan_any(Q, I, Kp, k, flags);
		if (V.ok && V.total > t) t = V.total;
	}
	return t;
}

extern "C" int anncur_score_topk_supported(int64_t Q, int64_t I, int32_t Kp, int32_t k) {
	return plan_any(Q, I, Kp, k).ok ? 1 : 0;
}

// (item_ids: the select kernels report item_ids[row of Et] -- the map is applied where the indices are written, not by a launch of its own;
//  a row index is always < I there: it comes out of a candidate the sweep wrote or a recomputed item)
static int score_topk_impl(const void *X, int64_t ldx, const void *Et, int64_t lde, int64_t Q, int64_t I, int32_t Kp,
						   int32_t k, float *out_val, int32_t *out_idx, void *workspace, size_t workspace_bytes,
						   void *stream, hipEvent_t *ev, int32_t flags = 0, const int32_t *item_ids = nullptr, CoScan *co = nullptr, const EvalArgs *ea = nullptr) {
	ANNCUR_REQUIRE((flags & ~TOPK_FLAGS) == 0, ANNCUR_E_INVALID, "score_topk: unknown flags 0x%x", flags);
	ANNCUR_REQUIRE(ring_flag_ok(flags), ANNCUR_E_UNSUPPORTED, "score_topk: ANNCUR_TOPK_RING (the tile-ring sweep body) is compiled into the experiments library only "
				   "(make -C anncur_amd/csrc experiments; ANNCUR_LIB=.../libanncur_hip_exp.so): measured slower than the default body");
	const FusedPlan P = ea ? plan_fused(Q, I, Kp, k, false, false, false, false, false, true) : plan_any(Q, I, Kp, k, flags);
	ANNCUR_REQUIRE(P.ok, ANNCUR_E_UNSUPPORTED,
				   "score_topk: (Q=%lld, I=%lld, Kp=%d, k=%d) is outside the fused path (Kp in {64,128,256,512} or a multiple of 128 up to %d, "
				   "1<=k<=%d, I large enough for a sampled threshold); use anncur_gemm + anncur_rowwise_topk",
				   (long long)Q, (long long)I, Kp, k, WIDE_KP_MAX, ANNCUR_MAX_TOPK);
	ANNCUR_REQUIRE(X && Et && out_val && out_idx, ANNCUR_E_INVALID, "score_topk: null pointer");
	ANNCUR_REQUIRE(lde == Kp, ANNCUR_E_INVALID, "score_topk: Et must be packed (lde == Kp), got lde=%lld", (long long)lde);
	ANNCUR_REQUIRE(ldx >= Kp && (ldx % 8) == 0, ANNCUR_E_INVALID, "score_topk: ldx must be >= Kp and a multiple of 8");
	ANNCUR_REQUIRE(((uintptr_t)X % 16) == 0 && ((uintptr_t)Et % 16) == 0, ANNCUR_E_INVALID, "score_topk: X and Et must be 16-byte aligned");
	ANNCUR_REQUIRE(workspace && workspace_bytes >= P.total && ((uintptr_t)workspace % 256) == 0, ANNCUR_E_WORKSPACE,
				   "score_topk: workspace of %zu bytes (256-byte aligned) required, got %zu", P.total, workspace_bytes);
	hipStream_t st = (hipStream_t)stream;
	unsigned char *ws = (unsigned char *)workspace;
	int rc;
	if (co && (rc = co_plan(*co, P.n_stages + 1)) != ANNCUR_OK) return rc;   // one chunk per latency-bound launch: threshold, refinements, select
	switch (Kp) {
		case 64: rc = launch_fused<64>(P, X, ldx, Et, Q, I, k, out_val, out_idx, ws, st, ev, co, item_ids, ea); break;
		case 128: rc = P.QT == 1 ? launch_fused<128, 1>(P, X, ldx, Et, Q, I, k, out_val, out_idx, ws, st, ev, co, item_ids)
								  : launch_fused<128>(P, X, ldx, Et, Q, I, k, out_val, out_idx, ws, st, ev, co, item_ids, ea); break;
		case 256: rc = P.QT == 1 ? launch_fused<256, 1>(P, X, ldx, Et, Q, I, k, out_val, out_idx, ws, st, ev, co, item_ids)
								  : launch_fused<256>(P, X, ldx, Et, Q, I, k, out_val, out_idx, ws, st, ev, co, item_ids, ea); break;
		case 512: rc = launch_fused<512>(P, X, ldx, Et, Q, I, k, out_val, out_idx, ws, st, ev, co, item_ids); break;
		default: rc = launch_wide(P, X, ldx, Et, Q, I, Kp, k, out_val, out_idx, ws, st, ev, co, item_ids); break;
	}
	if (rc == ANNCUR_OK) rc = co_finish(co, st);
	if (rc != ANNCUR_OK) co_abort(co, st);   // (never return with forked chunks unjoined)
	return rc;
}

extern "C" int anncur_score_topk_ex(const void *X, int64_t ldx, const void *Et, int64_t lde, int64_t Q, int64_t I, int32_t Kp,
									int32_t k, float *out_val, int32_t *out_idx, void *workspace, size_t workspace_bytes,
									int32_t flags, const int32_t *item_ids, void *stream) {
	return score_topk_impl(X, ldx, Et, lde, Q, I, Kp, k, out_val, out_idx, workspace, workspace_bytes, stream, nullptr, flags, item_ids);
}

extern "C" int anncur_score_topk(const void *X, int64_t ldx, const void *Et, int64_t lde, int64_t Q, int64_t I, int32_t Kp,
								 int32_t k, float *out_val, int32_t *out_idx, void *workspace, size_t workspace_bytes,
								 void *stream) {
	return score_topk_impl(X, ldx, Et, lde, Q, I, Kp, k, out_val, out_idx, workspace, workspace_bytes, stream, nullptr);
}

/* a8: exact top-k of the stored scores AND the fused approximate retrieval in one call, the scan's row chunks co-scheduled with the
 * retrieval's latency-bound launches (CoScan above).  Results are those of anncur_rowwise_topk + anncur_score_topk_ex. */
extern "C" int anncur_eval_topk(const void *A, int a_dtype, int64_t lda, int32_t k_exact, float *exact_val, int32_t *exact_idx,
								const void *X, int64_t ldx, const void *Et, int64_t lde, int64_t Q, int64_t I, int32_t Kp, int32_t k_retvr,
								float *approx_val, int32_t *approx_idx, void *workspace, size_t workspace_bytes, int32_t flags,
								const int32_t *item_ids, void *stream, void *aux_stream) {
	ANNCUR_REQUIRE(dtype_ok(a_dtype) && A && exact_val && exact_idx && lda >= I && k_exact >= 1 && k_exact <= I && k_exact <= ANNCUR_MAX_TOPK,
				   ANNCUR_E_INVALID, "eval_topk: bad exact-scan arguments");
	if (Q == 0) return ANNCUR_OK;
	if (!aux_stream || aux_stream == stream) {   // no second stream: the two parts one after the other
		const int rc = anncur_rowwise_topk(A, a_dtype, Q, I, lda, k_exact, exact_val, exact_idx, stream);
		if (rc != ANNCUR_OK) return rc;
		return score_topk_impl(X, ldx, Et, lde, Q, I, Kp, k_retvr, approx_val, approx_idx, workspace, workspace_bytes, stream, nullptr, flags, item_ids);
	}
	CoScan co{};
	co.aux = (hipStream_t)aux_stream;
	co.A = A; co.a_dtype = a_dtype; co.lda = lda; co.Q = Q; co.I = I; co.k = k_exact; co.val = exact_val; co.idx = exact_idx;
	return score_topk_impl(X, ldx, Et, lde, Q, I, Kp, k_retvr, approx_val, approx_idx, workspace, workspace_bytes, stream, nullptr, flags, item_ids, &co);
}

/* a8 + a11 of one grid cell of entry point A in ONE sweep: top-k_retvr of S_hat = X . E AND err_sq[q] = sum_i (S_hat - A)^2,
 * norm_sq[q] = sum_i A^2 (score_evalf.hpp).  Operands as anncur_score_topk (Et in ITEM order) + the exact matrix as
 * anncur_approx_error_packed's lds route takes it (bf16, 16-byte aligned rows). */
extern "C" size_t anncur_eval_fused_workspace_bytes(int64_t Q, int64_t I, int32_t Kp, int32_t k) {
	const FusedPlan P = plan_fused(Q, I, Kp, k, false, false, false, false, false, true);
	return P.ok ? P.total : 0;
}

extern "C" int anncur_eval_fused_ex(const void *X, int64_t ldx, const void *Et, int64_t lde, const void *Et_hint, const void *A, int a_dtype, int64_t lda,
									int64_t Q, int64_t I, int32_t Kp, int32_t k, float *out_val, int32_t *out_idx, float *err_sq, float *norm_sq,
									void *workspace, size_t workspace_bytes, void *stream);
extern "C" int anncur_eval_fused(const void *X, int64_t ldx, const void *Et, int64_t lde, const void *A, int a_dtype, int64_t lda,
								 int64_t Q, int64_t I, int32_t Kp, int32_t k, float *out_val, int32_t *out_idx, float *err_sq, float *norm_sq,
								 void *workspace, size_t workspace_bytes, void *stream) {
	return anncur_eval_fused_ex(X, ldx, Et, lde, nullptr, A, a_dtype, lda, Q, I, Kp, k, out_val, out_idx, err_sq, norm_sq, workspace, workspace_bytes, stream);
}
extern "C" int anncur_eval_fused_ex(const void *X, int64_t ldx, const void *Et, int64_t lde, const void *Et_hint, const void *A, int a_dtype, int64_t lda,
									int64_t Q, int64_t I, int32_t Kp, int32_t k, float *out_val, int32_t *out_idx, float *err_sq, float *norm_sq,
									void *workspace, size_t workspace_bytes, void *stream) {
	ANNCUR_REQUIRE(Kp == 64 || Kp == 128 || Kp == 256, ANNCUR_E_UNSUPPORTED, "eval_fused: Kp must be 64, 128 or 256 (got %d): use anncur_score_topk + anncur_approx_error_packed", Kp);
	ANNCUR_REQUIRE(a_dtype == ANNCUR_BF16 && A && (lda % 8) == 0 && ((uintptr_t)A % 16) == 0 && lda >= I, ANNCUR_E_UNSUPPORTED,
				   "eval_fused: the exact matrix must be bf16 with 16-byte aligned rows (lda a multiple of 8): use anncur_score_topk + anncur_approx_error(_packed) otherwise");
	ANNCUR_REQUIRE(err_sq && norm_sq, ANNCUR_E_INVALID, "eval_fused: null pointer");
	ANNCUR_REQUIRE(exact_tile_offsets_fit(lda, 256), ANNCUR_E_UNSUPPORTED,
				   "eval_fused: row pitch %lld of the exact matrix is too long for the kernel's 32-bit tile offsets (255 rows x pitch x 2 bytes must stay below 2^32): use anncur_score_topk + anncur_approx_error_packed",
				   (long long)lda);
	ANNCUR_REQUIRE(anncur_eval_fused_workspace_bytes(Q, I, Kp, k) > 0, ANNCUR_E_UNSUPPORTED, "eval_fused: shape (Q=%lld, I=%lld, Kp=%d, k=%d) is outside the fused path",
				   (long long)Q, (long long)I, Kp, k);
	if (Q == 0) return ANNCUR_OK;
	hipStream_t st = (hipStream_t)stream;
	ANNCUR_HIP_OK(hipMemsetAsync(err_sq, 0, (size_t)Q * 4, st));
	ANNCUR_HIP_OK(hipMemsetAsync(norm_sq, 0, (size_t)Q * 4, st));
	const int64_t I_full = I / TILE_I * TILE_I;
	if (I_full < I) {  // the last I % 32 columns' error terms: strided kernel of gemm.hip, accumulating into the same sums
		const int rc = anncur_internal_approx_error_acc((const uint16_t *)X, ANNCUR_BF16, ldx, (const uint16_t *)Et + I_full * lde, ANNCUR_BF16, lde,
														(const void *)((const uint16_t *)A + I_full), a_dtype, lda, Q, I - I_full, Kp, err_sq, norm_sq, stream);
		if (rc != ANNCUR_OK) return rc;
	}
	const EvalArgs ea{(const uint16_t *)A, lda, err_sq, norm_sq, (const uint16_t *)Et_hint};
	return score_topk_impl(X, ldx, Et, lde, Q, I, Kp, k, out_val, out_idx, workspace, workspace_bytes, stream, nullptr, 0, nullptr, nullptr, &ea);
}

extern "C" int anncur_score_topk_timed(const void *X, int64_t ldx, const void *Et, int64_t lde, int64_t Q, int64_t I, int32_t Kp,
									   int32_t k, float *out_val, int32_t *out_idx, void *workspace, size_t workspace_bytes,
									   int32_t flags, const int32_t *item_ids, void *stream, float *stage_ms) {
	ANNCUR_REQUIRE(stage_ms, ANNCUR_E_INVALID, "score_topk_timed: stage_ms is null");
	constexpr int NEV = 11;  // 0..4 stage boundaries, 5..10 begin/end of up to three sweep launches
	hipEvent_t ev[NEV];
	for (int i = 0; i < NEV; ++i) ANNCUR_HIP_OK(hipEventCreate(&ev[i]));
	const FusedPlan P = plan_any(Q, I, Kp, k, flags);
	int rc = score_topk_impl(X, ldx, Et, lde, Q, I, Kp, k, out_val, out_idx, workspace, workspace_bytes, stream, ev, flags, item_ids);
	if (rc == ANNCUR_OK) {
		hipError_t e = hipEventSynchronize(ev[4]);
		if (e != hipSuccess) { anncur_set_error("hipEventSynchronize: %s", hipGetErrorString(e)); rc = ANNCUR_E_HIP; }
		for (int i = 0; i < 4 && rc == ANNCUR_OK; ++i)
			if (hipEventElapsedTime(&stage_ms[i], ev[i], ev[i + 1]) != hipSuccess) { anncur_set_error("hipEventElapsedTime failed"); rc = ANNCUR_E_HIP; }
		stage_ms[4] = 0.f;
		stage_ms[6] = stage_ms[7] = stage_ms[8] = 0.f;
		for (int g = 0; g < P.n_stages && rc == ANNCUR_OK; ++g) {
			float ms = 0.f;
			if (hipEventElapsedTime(&ms, ev[5 + 2 * g], ev[6 + 2 * g]) != hipSuccess) { anncur_set_error("hipEventElapsedTime failed"); rc = ANNCUR_E_HIP; }
			stage_ms[4] += ms;
			stage_ms[6 + g] = ms;
		}
		stage_ms[5] = (float)P.n_stages;
	}
	for (int i = 0; i < NEV; ++i) (void)hipEventDestroy(ev[i]);
	return rc;
}

#ifdef ANNCUR_TIMING_EXPERIMENTS
/* experiments build only (scripts/r4/timeline_probe.py): the same launches with the CALLER's 11 events recorded at the stage boundaries
 * (0 start, 1 after the prepass, 2 after the threshold, 5 + 2 g / 6 + 2 g around sweep launch g, 3 after the last stage, 4 after the
 * select) and no synchronisation -- for a timeline of several calls in flight on several streams against one base event */
extern "C" int anncur_score_topk_events(const void *X, int64_t ldx, const void *Et, int64_t lde, int64_t Q, int64_t I, int32_t Kp,
										int32_t k, float *out_val, int32_t *out_idx, void *workspace, size_t workspace_bytes,
										int32_t flags, const int32_t *item_ids, void *stream, void *const *events11) {
	ANNCUR_REQUIRE(events11, ANNCUR_E_INVALID, "score_topk_events: events is null");
	hipEvent_t ev[11];
	for (int i = 0; i < 11; ++i) ev[i] = (hipEvent_t)events11[i];
	return score_topk_impl(X, ldx, Et, lde, Q, I, Kp, k, out_val, out_idx, workspace, workspace_bytes, stream, ev, flags, item_ids);
}
#endif

/* plan introspection for benchmarks / DESIGN.md: n_sample_tiles, n_tiles, S, capg, group (tiles of 32 items, or of 256 for Kp > 512) */
extern "C" int anncur_score_topk_plan(int64_t Q, int64_t I, int32_t Kp, int32_t k, int32_t *out5) {
	const FusedPlan P = plan_any(Q, I, Kp, k);
	ANNCUR_REQUIRE(P.ok && out5, ANNCUR_E_UNSUPPORTED, "score_topk_plan: unsupported shape");
	out5[0] = P.n_st; out5[1] = P.n_tiles; out5[2] = P.S; out5[3] = P.capg; out5[4] = P.group;
	return ANNCUR_OK;
}

/* the same for the flags of anncur_score_topk_ex: out[0 .. n_out) = {sample tiles, item tiles, S, segment capacity, group, segments per
 * query and item split (2: 32x32x16 sweep, 1: 16x16x32 sweep, 4: wide kernel), 32-query sub-tiles per wave, sweep stages,
 * stage_end[3], stage body[3] (0: 32x32x16 with the ballot filter, 1: with the exec-mask filter, 2: 16x16x32, 3 / 4: Kp = 512 with the
 * wave-level queue on 32x32x16 / 16x16x32 MFMAs, 5: 16x16x32 in 8-wave workgroups with the flag-synchronised tile ring), ring drain period[3],
 * threshold ladder (1: the sweep raises its thresholds in-launch, score16.hpp; 0: staged), rank of the ladder's top level} --
 * what a test needs to see that a variant flag was honoured */
extern "C" int anncur_score_topk_plan_ex(int64_t Q, int64_t I, int32_t Kp, int32_t k, int32_t flags, int32_t *out, int32_t n_out) {
	ANNCUR_REQUIRE((flags & ~TOPK_FLAGS) == 0, ANNCUR_E_INVALID, "score_topk_plan_ex: unknown flags 0x%x", flags);
	ANNCUR_REQUIRE(ring_flag_ok(flags), ANNCUR_E_UNSUPPORTED, "score_topk_plan_ex: ANNCUR_TOPK_RING is compiled into the experiments library only");
	const FusedPlan P = plan_any(Q, I, Kp, k, flags);
	ANNCUR_REQUIRE(P.ok && out && n_out >= 0, ANNCUR_E_UNSUPPORTED, "score_topk_plan_ex: unsupported shape");
	const bool wide = wide_kp(Kp);
	int32_t v[19] = {P.n_st, P.n_tiles, P.S, P.capg, P.group, wide ? 4 : P.lg, P.QT, P.n_stages};
	v[17] = (!wide && P.ladder) ? 1 : 0; v[18] = (!wide && P.ladder) ? P.ladder_k2 : 0;
	for (int g = 0; g < 3; ++g) {
		const bool on = g < P.n_stages;
		v[8 + g] = on ? P.stage_end[g] : 0; v[11 + g] = on ? (!wide && P.bodyef ? 6 : !wide && P.ring16 ? 5 : !wide && P.body16 ? 2 : (!wide && P.bodyq16 ? 4 : (!wide && P.bodyq1 ? 3 : P.stage_pred[g]))) : 0; v[14 + g] = on ? P.stage_flush[g] : 0;
	}
	for (int i = 0; i < n_out && i < 19; ++i) out[i] = v[i];
	return ANNCUR_OK;
}

/* diagnostics: candidates the sweep of the LAST call on this workspace kept, per query on average (the segment counts it left behind;
 * synchronises the stream).  What a threshold plan is judged by: k ln(I / k) is what an online threshold can reach. */
extern "C" int anncur_score_topk_survivors(const void *workspace, int64_t Q, int64_t I, int32_t Kp, int32_t k, int32_t flags, double *mean_per_query,
										   void *stream) {
	ANNCUR_REQUIRE((flags & ~TOPK_FLAGS) == 0 && workspace && mean_per_query, ANNCUR_E_INVALID, "score_topk_survivors: bad arguments");
	const FusedPlan P = plan_any(Q, I, Kp, k, flags);
	ANNCUR_REQUIRE(P.ok, ANNCUR_E_UNSUPPORTED, "score_topk_survivors: unsupported shape");
	const size_t n = (size_t)Q * (wide_kp(Kp) ? 4 : P.lg) * P.S;
	std::vector<uint32_t> h(n);
	ANNCUR_HIP_OK(hipMemcpyAsync(h.data(), (const unsigned char *)workspace + P.off_segcnt, n * 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
	ANNCUR_HIP_OK(hipStreamSynchronize((hipStream_t)stream));
	double tot = 0.0;
	for (size_t i = 0; i < n; ++i) tot += (h[i] & 0x80000000u) ? 0.0 : (double)h[i];   // (a poisoned count marks a repaired split)
	*mean_per_query = tot / (double)Q;
	return ANNCUR_OK;
}

/* a11 on packed bf16 operands (the layout of anncur_score_topk): err_sq[q] = sum_i (X[q,:].Et[i,:] - A[q,i])^2, norm_sq[q] = sum_i A[q,i]^2 */
extern "C" int anncur_approx_error_packed(const void *X, int64_t ldx, const void *Et, int64_t lde, const void *A, int a_dtype, int64_t lda,
										  int64_t Q, int64_t I, int32_t Kp, float *err_sq, float *norm_sq, void *stream) {
	ANNCUR_REQUIRE(Kp == 64 || Kp == 128 || Kp == 256 || Kp == 512, ANNCUR_E_UNSUPPORTED, "approx_error_packed: Kp must be 64, 128, 256 or 512 (got %d)", Kp);
	ANNCUR_REQUIRE(dtype_ok(a_dtype), ANNCUR_E_INVALID, "approx_error_packed: bad dtype");
	ANNCUR_REQUIRE(Q >= 0 && I >= 1 && I < (int64_t)0x7fffffff - 64 && lda >= I, ANNCUR_E_INVALID, "approx_error_packed: bad shape");
	ANNCUR_REQUIRE(X && Et && A && err_sq && norm_sq, ANNCUR_E_INVALID, "approx_error_packed: null pointer");
	ANNCUR_REQUIRE(lde == Kp && ldx >= Kp && (ldx % 8) == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Et % 16) == 0, ANNCUR_E_INVALID,
				   "approx_error_packed: X / Et must be packed bf16 (lde == Kp, ldx multiple of 8, 16-byte aligned)");
	ANNCUR_REQUIRE((lda % 4) == 0 && ((uintptr_t)A % 16) == 0, ANNCUR_E_UNSUPPORTED,
				   "approx_error_packed: the exact matrix must be 16-byte aligned with lda a multiple of 4 (use anncur_approx_error otherwise)");
	if (Q == 0) return ANNCUR_OK;
	hipStream_t st = (hipStream_t)stream;
	ANNCUR_HIP_OK(hipMemsetAsync(err_sq, 0, (size_t)Q * 4, st));
	ANNCUR_HIP_OK(hipMemsetAsync(norm_sq, 0, (size_t)Q * 4, st));
	const int64_t I_full = I / TILE_I * TILE_I;
	if (I_full < I) {  // the last I % 32 columns: strided kernel of gemm.hip, accumulating into the same sums
		const int rc = anncur_internal_approx_error_acc((const uint16_t *)X, ANNCUR_BF16, ldx, (const uint16_t *)Et + I_full * lde, ANNCUR_BF16, lde,
														a_dtype == ANNCUR_F32 ? (const void *)((const float *)A + I_full) : (const void *)((const uint16_t *)A + I_full),
														a_dtype, lda, Q, I - I_full, Kp, err_sq, norm_sq, stream);
		if (rc != ANNCUR_OK) return rc;
	}
	if (I_full == 0) return ANNCUR_OK;
	FusedParams p{};
	p.X = (const uint16_t *)X; p.ldx = ldx; p.Et = (const uint16_t *)Et; p.Q = Q; p.I = I_full;
	const int QT = (Kp <= 256) ? 2 : 1, BQ = 128 * QT;
	const int n_rb = (int)ceil_div64(Q, BQ);
	p.n_tiles = (int)(I_full / TILE_I);
	int S = 2 * num_cu() / n_rb;
	if (S < 1) S = 1;
	if (S > p.n_tiles) S = p.n_tiles;
	p.tiles_per_split = (p.n_tiles + S - 1) / S;
	S = (p.n_tiles + p.tiles_per_split - 1) / p.tiles_per_split;
	p.n_wg = n_rb * S;
	if (const char *dbg = knob("ANNCUR_DEBUG_ERR_MODE")) p.ring_stagger = atoi(dbg);
#define LAUNCH_ERR(KPV, TA)                                                                                                   \
	hipLaunchKernelGGL((error_kernel<KPV, TA>), dim3(p.n_wg), dim3(256), 2 * FusedCfg<KPV>::TILE_BYTES, st, p, (const TA *)A, lda, err_sq, norm_sq)
#define LAUNCH_ERR_K(TA)                                                                                                      \
	do {                                                                                                                      \
		switch (Kp) {                                                                                                         \
			case 64: LAUNCH_ERR(64, TA); break;                                                                               \
			case 128: LAUNCH_ERR(128, TA); break;                                                                             \
			case 256: LAUNCH_ERR(256, TA); break;                                                                             \
			default: LAUNCH_ERR(512, TA); break;                                                                              \
		}                                                                                                                     \
	} while (0)
	// bf16 exact matrix with 16-byte aligned rows: the kernel that stages the exact tile through LDS
	// (and a row pitch whose 32-bit tile offsets cannot wrap: exact_tile_offsets_fit; longer pitches take error_kernel's 64-bit row pointers)
	const bool lds_exact = a_dtype == ANNCUR_BF16 && (lda % 8) == 0 && ((uintptr_t)A % 16) == 0 && exact_tile_offsets_fit(lda, Kp <= 256 ? 256 : 128);
	if (lds_exact) {
#define LAUNCH_ERRL(KPV)                                                                                                      \
		do {                                                                                                                  \
			const int lds_ = 2 * FusedCfg<KPV>::TILE_BYTES + 2 * FusedCfg<KPV>::BQ * 64;                                      \
			{ const int rc_ = anncur_ensure_dyn_lds((const void *)error_lds_kernel<KPV>, lds_); if (rc_ != ANNCUR_OK) return rc_; } \
			hipLaunchKernelGGL((error_lds_kernel<KPV>), dim3(p.n_wg), dim3(256), lds_, st, p, (const uint16_t *)A, lda, err_sq, norm_sq); \
		} while (0)
		switch (Kp) {
			case 64: LAUNCH_ERRL(64); break;
			case 128: LAUNCH_ERRL(128); break;
			case 256: LAUNCH_ERRL(256); break;
			default: LAUNCH_ERRL(512); break;
		}
#undef LAUNCH_ERRL
	} else if (a_dtype == ANNCUR_F32) LAUNCH_ERR_K(float); else LAUNCH_ERR_K(uint16_t);
#undef LAUNCH_ERR_K
#undef LAUNCH_ERR
	ANNCUR_LAUNCH_OK();
	return ANNCUR_OK;
}

#ifdef ANNCUR_TIMING_EXPERIMENTS
/* diagnostic build only (not in include/anncur_hip.h): median in-kernel clock in GHz over the workgroups of the last sweep launch
 * that ran with ANNCUR_DEBUG_STAMPS set (s_memtime ticks per s_memrealtime tick x 100 MHz), and the median loop duration in us. */
extern "C" int anncur_debug_read_stamps(double *clock_ghz, double *loop_us, int *n_wg) {
	if (!g_stamps) return ANNCUR_E_INVALID;
	static unsigned long long h[2 * 8192];
	if (hipMemcpy(h, g_stamps, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return ANNCUR_E_HIP;
	double r[8192], u[8192];
	int n = 0;
	for (int i = 0; i < 8192; ++i)
		if (h[2 * i + 1] > 0) { r[n] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; u[n] = (double)h[2 * i + 1] / 100.0; ++n; }
	if (n == 0) return ANNCUR_E_INVALID;
	for (int i = 1; i < n; ++i) { double x = r[i], y = u[i]; int j = i - 1; while (j >= 0 && r[j] > x) { r[j + 1] = r[j]; --j; } r[j + 1] = x; j = i - 1; while (j >= 0 && u[j] > y) { u[j + 1] = u[j]; --j; } u[j + 1] = y; }
	*clock_ghz = r[n / 2]; *loop_us = u[n / 2]; *n_wg = n;
	return ANNCUR_OK;
}
#endif

#ifdef ANNCUR_TIMING_EXPERIMENTS
/* diagnostic build only: arm (n_wg > 0) or read the phase stamps of select_wave_kernel.  out[3] = median cycles of the phases
 * (prologue, candidate load loop, final compaction / sort) over the workgroups' first waves. */
extern "C" int anncur_debug_sel_stamps(int arm, double *out) {
	static unsigned long long *buf = nullptr;
	const int N = 4096;
	if (!buf) { if (hipMalloc((void **)&buf, 8 * N * sizeof(unsigned long long)) != hipSuccess) return ANNCUR_E_HIP; }
	if (arm) {
		if (hipMemset(buf, 0, 8 * N * sizeof(unsigned long long)) != hipSuccess) return ANNCUR_E_HIP;
		unsigned long long *v = arm > 0 ? buf : nullptr;
		return hipMemcpyToSymbol(HIP_SYMBOL(d_sel_stamps), &v, sizeof(v)) == hipSuccess ? ANNCUR_OK : ANNCUR_E_HIP;
	}
	static unsigned long long h[8 * 4096];
	if (hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return ANNCUR_E_HIP;
	// out[0..2]: prologue, load loop, finish; out[3..4] (select_wave_kernel only): first batch's segment search, its loads' latency
	static const int from[5] = {0, 1, 2, 1, 4}, to[5] = {1, 2, 3, 4, 5};
	static double ph[4096];
	for (int j = 0; j < 5; ++j) {
		int n = 0;
		for (int i = 0; i < N; ++i)
			if (h[8 * i + 3] > h[8 * i] && h[8 * i + to[j]] > h[8 * i + from[j]]) ph[n++] = (double)(h[8 * i + to[j]] - h[8 * i + from[j]]);
		for (int i = 1; i < n; ++i) { double x = ph[i]; int t = i - 1; while (t >= 0 && ph[t] > x) { ph[t + 1] = ph[t]; --t; } ph[t + 1] = x; }
		out[j] = n ? ph[n / 2] : 0.0;
	}
	return ANNCUR_OK;
}
#endif

#ifdef ANNCUR_TIMING_EXPERIMENTS
/* diagnostic build only: timeline of the last sweep launch in us relative to the first workgroup's entry: out = {entry p50, entry max,
 * loop start p50, loop start max, loop end p50, loop end max} */
extern "C" int anncur_debug_sweep_timeline(double *out) {
	if (!g_stamps) return ANNCUR_E_INVALID;
	static unsigned long long h[5 * 8192];
	if (hipMemcpy(h, g_stamps, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return ANNCUR_E_HIP;
	static double v[3][8192];
	int n = 0; unsigned long long t0 = ~0ull;
	for (int i = 0; i < 8192; ++i) if (h[2 * 8192 + 3 * i + 2] > 0 && h[2 * 8192 + 3 * i] < t0) t0 = h[2 * 8192 + 3 * i];
	for (int i = 0; i < 8192; ++i)
		if (h[2 * 8192 + 3 * i + 2] > 0) { for (int j = 0; j < 3; ++j) v[j][n] = (double)(h[2 * 8192 + 3 * i + j] - t0) / 100.0; ++n; }
	if (!n) return ANNCUR_E_INVALID;
	for (int j = 0; j < 3; ++j) {
		for (int i = 1; i < n; ++i) { double x = v[j][i]; int t = i - 1; while (t >= 0 && v[j][t] > x) { v[j][t + 1] = v[j][t]; --t; } v[j][t + 1] = x; }
		out[2 * j] = v[j][n / 2]; out[2 * j + 1] = v[j][n - 1];
	}
	return ANNCUR_OK;
}
#endif

#ifdef ANNCUR_TIMING_EXPERIMENTS
/* diagnostic build only: per-wave phase cycles of the stamped sweep launch (staggered Kp <= 256 body): out[8192 x 8], words 0..4 = cycles in
 * {ticket + DMA issue, ring drain, MFMA / filter section, vmcnt wait, barrier} */
extern "C" int anncur_debug_sweep_phases(unsigned long long *out) {
	if (!g_stamps) return ANNCUR_E_INVALID;
	return hipMemcpy(out, g_stamps + 5 * 8192, 8 * 8192 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? ANNCUR_OK : ANNCUR_E_HIP;
}
#endif

#ifdef ANNCUR_TIMING_EXPERIMENTS
/* diagnostic build only: raw stamps of the last sweep launch, 5 x 8192 words ({cycles, ticks} x 8192, then {entry, loop start, loop end} x 8192) */
extern "C" int anncur_debug_sweep_raw(unsigned long long *out) {
	if (!g_stamps) return ANNCUR_E_INVALID;
	return hipMemcpy(out, g_stamps, 5 * 8192 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? ANNCUR_OK : ANNCUR_E_HIP;
}
#endif
