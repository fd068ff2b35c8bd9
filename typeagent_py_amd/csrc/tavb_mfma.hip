// Batched lookup on fp16 corpora: S = X . Q^T as a dense (rows x D) . (D x queries)
// contraction on the matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulate), with the
// score map, threshold and per-query top-k selection fused into the epilogue so that
// the [queries x rows] score matrix (41 GB at 1024 x 10M) never exists.
//
// This is the batch the reference leaves as a TODO (storage/sqlite/reltermsindex.py:259-271);
// its semantics are Q independent `fuzzy_lookup_embedding` calls (vectorbase.py:163-190).
// Products of two fp16 values are exact in fp32, so against an oracle fed the same
// fp16-rounded values only the accumulation order differs (fp32 noise ~5e-8).
//
// Decomposition (256-query tile, `mfma_scan_kernel`)
//   * operand roles: A = corpus tile (M = 320 rows), B = query tile (N = 256 queries).  With this orientation the MFMA
//     result layout puts ONE query in each lane (col = lane & 31) and 16 corpus rows in its 16 accumulator registers,
//     so the epilogue's "does this score beat the query's current k-th best" test needs one threshold register per
//     lane and one max3 chain + compare per 32 x 32 block.
//   * workgroup = 4 waves (2 along rows x 2 along queries), one per SIMD with the whole 512-register budget: a
//     160 x 128 sub-tile = 5 x 4 MFMA tiles each.  K advances in steps of 64 halves (whole 128-byte lines); both operand
//     slabs are staged into a two-slot LDS ring by LDS-DMA (`buffer_load ... lds`, 16 B per lane) that runs ahead of the
//     MFMAs across tile boundaries, with raw `s_barrier` and explicit `s_waitcnt`.  Details in the kernel's header.
//   * a workgroup owns one query tile and one contiguous range of corpus rows and walks that range tile by tile
//     (persistent); the workgroups that share a row range (one per query tile) get block ids congruent mod 8 so they
//     run on the same XCD at the same time and the corpus tile is fetched from HBM once and re-read from that XCD's L2.
//   * selection: per (workgroup, query) a candidate buffer of CAPW keys in global memory plus, in LDS, its fill count
//     and the current admission threshold.  A score that beats the threshold is clipped, packed into a key and
//     appended (one LDS atomic per lane per block).  A buffer that could overflow on the next tile is compacted to its
//     BAND (`compact_to_band`: the k-th best score by bisection on the score bits, no sort; everything within 2 delta_q below it
//     stays) and the threshold rises to the band's cut.  At the end of a launch the buffers are left as they are;
//     `select_band_kernel` (one workgroup per QUERY) picks the band over all row ranges and derives the next admission threshold.
//   * the host scans the corpus in phases of growing size (threshold ladder, tavb_abi.hip): the k-th best score after a
//     phase seeds the admission thresholds of the next (`thr_in`).
//
// Two kernel families live here: the 256-query fp16 tile described above and, at the end of the file, a 32/64-query tile
// (`skinny_scan_kernel`) for fp32 and fp16 corpora that carries small batches -- and every batch on the reference's fp32
// layout -- at HBM speed.  On fp16 corpora the 256-query tile multiplies fp16-ROUNDED queries: it is used as an exact
// filter, its candidates are rescored with the fp32 queries (tavb_rescore.hip).

#include <hip/hip_runtime.h>

#include <type_traits>

#include "tavb_device.h"
#include "tavb_internal.h"

namespace tavb {

namespace {

constexpr int BM = 256;   // corpus rows per tile of the 32/64-query kernel
constexpr int BN = 256;   // queries per tile of the 256-query kernel
constexpr int BK = 64;    // dim must be a multiple of this
constexpr int CAP = 512;    // candidate keys per (workgroup, query) of the 32/64-query tile; must be >= BM + max k
constexpr int CAPW = 1024;  // ... of the 256-query tile: three 320-row tiles fit before the first compaction, so a short first
                            // ladder phase (<= 2 tiles per workgroup) never compacts at all

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_void;

struct MfmaDeviceParams {
  const _Float16* corpus;
  const _Float16* queries;  // [nq_padded, dim]
  u64* cand;                // [blocks][queries per tile][CAP or CAPW]
  int* counts;              // 256-query tile: [blocks][BN] keys left in each candidate buffer when the launch ends
  u64* lists;               // [nq][n_splits][k]
  int64_t rows;
  int64_t rows_per_split;   // multiple of BM
  int32_t dim;
  int32_t nq;
  int32_t n_qtiles;
  int32_t n_splits;
  int32_t list_stride;      // lists per query in `lists` (>= n_splits; extra slots belong to the caller)
  int32_t k;
  uint32_t index_base;
  float min_score;
  const float* thr_in;  // optional [nq_padded] admission thresholds from a sample pass (exclusive bound)
  const int* active;    // optional: number of live queries, read on the device; query tiles past it return at once
  int32_t active_min;   // ... and the launch as a whole returns at once unless active_min < *active <= active_max (two fallbacks share one work list)
  int32_t active_max;
  int64_t split_plane;  // 256-query tile, SPLIT form: bytes from the high to the low plane of the queries ([2][nq_padded][dim] fp16); 0 otherwise
  const int* gate;      // optional device-side counter: the whole launch returns at once when *gate > gate_max (a filter phase of a batch already known to need
  int32_t gate_max;     // the exact form: tavb_abi.hip::run_tile_ladder)
  const float* band;    // 128/256-query tile, optional [nq_padded]: keep every key within band[q] below the k-th best (band selection)
  unsigned* lost;       // ... [nq_padded]: atomicMax of the score bits below which a query LOST band rows (a band that did not fit a buffer)
};

// Pin a wave-uniform pointer into SGPRs.  Without this the compiler strength-reduces the eight
// LDS-DMA source addresses of a K step into eight 64-bit VGPR induction variables (16 VGPRs, spilled
// in this kernel); with it each address is "SGPR base + 32-bit VGPR offset" (the saddr form).
__device__ __forceinline__ const char* sgpr_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// LDS read-modify-write / store that the compiler cannot see as LDS traffic.  hipcc's wait-count pass orders every LDS
// write or atomic behind all in-flight LDS-DMA (`s_waitcnt vmcnt(0)`): it cannot tell that the fill counters and flags
// never alias the operand rings.  In these kernels that wait sits in the admission slow path and drains up to twenty
// 1 KiB loads (1-2 us) every time a 32 x 32 block admits a row.
__device__ __forceinline__ int lds_add_rtn(int* counter, int v) {
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)counter;
  int old;
  asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(v) : "memory");
  return old;
}
__device__ __forceinline__ void lds_store_i32(__attribute__((address_space(3))) volatile int* flag, int v) {
  const uint32_t addr = (uint32_t)(uintptr_t)flag;
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_store_i32(volatile int* flag, int v) {
  lds_store_i32((__attribute__((address_space(3))) volatile int*)flag, v);
}

// Reduce one query's candidate buffer (n unsorted keys) to its best 64, sorted best-first and
// spread over the lanes (rank r in lane r).  One wave; wave-uniform arguments.
__device__ __forceinline__ WaveTopK<1> best_of_buffer(const u64* buf, int n, int lane) {
  WaveTopK<1> best;
  best.clear();
  for (int off = 0; off < n; off += 64) {
    const u64 key = (off + lane < n) ? buf[off + lane] : 0ull;
    WaveTopK<1> chunk;
    chunk.key[0] = sort64_ascending(key, lane);  // ascending == "reversed best-first"
    best.merge_reversed(chunk, lane);
  }
  return best;
}

// Compaction of one query's candidate buffer (n <= CAPACITY unsorted keys) to the keys that can still make the top k:
// everything at or above the k-th best key.  No sort: the k-th best SCORE is found by bisection on its bit pattern
// (scores are in [0, 1]: the patterns order like the floats) -- one ballot per 64 keys per bit, on the bits below the
// highest bit in which the buffer's scores differ (~20 of them) -- and the survivors are packed to the front with ballot
// prefix sums.  When more than k + 32 keys tie at that score (duplicate rows), the same bisection on the ordinal half of
// the key cuts the ties exactly (smaller ordinal wins), so a buffer always shrinks to about k and cannot overflow.
// ~5x cheaper than sorting 64-key chunks and merging them (the cold start of a launch compacts every buffer of the
// workgroup after its first tile).  One wave; wave-uniform arguments; returns the number of keys kept.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v, int lane) {
  v = max(v, (uint32_t)xor_lane_i32<1>((int)v, lane));
  v = max(v, (uint32_t)xor_lane_i32<2>((int)v, lane));
  v = max(v, (uint32_t)xor_lane_i32<4>((int)v, lane));
  v = max(v, (uint32_t)xor_lane_i32<8>((int)v, lane));
  v = max(v, (uint32_t)xor_lane_i32<16>((int)v, lane));
  v = max(v, (uint32_t)xor_lane_i32<32>((int)v, lane));
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

template <int CAPACITY>
__device__ __forceinline__ int compact_to_kth(u64* buf, int n, int k, int lane, float* kth_score) {
  constexpr int PER = CAPACITY / 64;
  u64 key[PER];
  uint32_t sc[PER];
  uint32_t mx = 0u, mn_inv = 0u;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int idx = j * 64 + lane;
    key[j] = (idx < n) ? buf[idx] : 0ull;
    sc[j] = (uint32_t)(key[j] >> 32);
    mx = max(mx, sc[j]);
    if (key[j] != 0ull) mn_inv = max(mn_inv, ~sc[j]);
  }
  if (n <= k) {
    *kth_score = -1.0f;
    return n;
  }
  mx = wave_max_u32(mx, lane);
  const uint32_t mn = ~wave_max_u32(mn_inv, lane);
  auto count_ge = [&](uint32_t t) {
    int c = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) c += __popcll(__builtin_amdgcn_ballot_w64(key[j] != 0ull && sc[j] >= t));
    return c;
  };
  uint32_t t = mn;  // every key is >= mn: count = n > k
  if (mx != mn) {
    const int top = 31 - __builtin_clz(mx ^ mn);
    t = (top == 31) ? 0u : (mx & ~((2u << top) - 1u));  // the common leading bits
    for (int b = top; b >= 0; --b) {
      const uint32_t trial = t | (1u << b);
      if (count_ge(trial) >= k) t = trial;
    }
  }
  // t = the k-th best score.  Ties at t beyond the slack are cut by ordinal (the low word: bigger = smaller ordinal).
  int above = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) above += __popcll(__builtin_amdgcn_ballot_w64(key[j] != 0ull && sc[j] > t));
  uint32_t t_lo = 0u;
  if (count_ge(t) > k + 32) {
    const int need = k - above;  // >= 1 of the tied keys are still needed
    for (int b = 31; b >= 0; --b) {
      const uint32_t trial = t_lo | (1u << b);
      int c = 0;
#pragma unroll
      for (int j = 0; j < PER; ++j) c += __popcll(__builtin_amdgcn_ballot_w64(sc[j] == t && key[j] != 0ull && (uint32_t)key[j] >= trial));
      if (c >= need) t_lo = trial;
    }
  }
  int base = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const bool keep = key[j] != 0ull && (sc[j] > t || (sc[j] == t && (uint32_t)key[j] >= t_lo));
    const u64 m = __builtin_amdgcn_ballot_w64(keep);
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) buf[pos] = key[j];
    base += __popcll(m);
  }
  *kth_score = __uint_as_float(t);
  return base;
}

// Score bits of the BAND cut that goes with a k-th best score `t_bits`: a key is kept when its score bits are >= the result.
// cut = t - band, rounded DOWN one more ulp (the subtraction rounds to nearest); 0 = keep everything (band wider than the score,
// or infinite: nothing can be ruled out).
__device__ __forceinline__ uint32_t band_cut_bits(uint32_t t_bits, float band) {
  const float c = __uint_as_float(t_bits) - band;
  if (!(c > 0.0f)) return 0u;
  const uint32_t b = __float_as_uint(c);
  return b > 0u ? b - 1u : 0u;
}

// Band compaction of one query's candidate buffer (the 128/256-query tile as an exact FILTER, tavb_rescore.hip): keep every key
// whose score is within `band` (= 2 delta_q, the filter's rigorous error bound both ways) of the k-th best score of the buffer --
// all of them, not a fixed number -- because exactly those rows can still be among the query's exact top k.  On ordinary data
// that is k + a handful; on clustered data (near-duplicate rows around rank k) it is the cluster, whatever its size, as long as
// it fits: when more than `limit` keys would stay, the buffer is cut to its strict best k (by key: smaller ordinal wins ties)
// and *lost_bits = the score bits of that k-th best: rows scoring <= it were (and, through the raised threshold, will be) dropped
// although they may lie inside the band.  That only matters if the query's FINAL band reaches down to that level -- the select
// kernel compares (a big cluster of near-duplicates inside one row range overflows the buffers of every query whose local k-th
// best is below the cluster's score, but it is irrelevant to all those whose final k-th best is far above it).
// Returns the number of keys kept; *thr_excl = the exclusive admission bound that goes with the cut (score > *thr_excl), or
// -inf when everything qualifies; *lost_bits = 0 when nothing was lost.  One wave; wave-uniform arguments.
template <int CAPACITY>
__device__ __forceinline__ int compact_to_band(u64* buf, int n, int k, int lane, float band, int limit, float* thr_excl, uint32_t* lost_bits) {
  constexpr int PER = CAPACITY / 64;
  u64 key[PER];
  uint32_t sc[PER];
  uint32_t mx = 0u, mn_inv = 0u;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int idx = j * 64 + lane;
    key[j] = (idx < n) ? buf[idx] : 0ull;
    sc[j] = (uint32_t)(key[j] >> 32);
    mx = max(mx, sc[j]);
    if (key[j] != 0ull) mn_inv = max(mn_inv, ~sc[j]);
  }
  *lost_bits = 0u;
  *thr_excl = -__builtin_inff();
  if (n <= k) return n;
  mx = wave_max_u32(mx, lane);
  const uint32_t mn = ~wave_max_u32(mn_inv, lane);
  auto count_ge = [&](uint32_t t) {
    int c = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) c += __popcll(__builtin_amdgcn_ballot_w64(key[j] != 0ull && sc[j] >= t));
    return c;
  };
  uint32_t t = mn;  // every key is >= mn: count = n > k
  if (mx != mn) {
    const int top = 31 - __builtin_clz(mx ^ mn);
    t = (top == 31) ? 0u : (mx & ~((2u << top) - 1u));  // the common leading bits
    for (int b = top; b >= 0; --b) {
      const uint32_t trial = t | (1u << b);
      if (count_ge(trial) >= k) t = trial;
    }
  }
  // t = the k-th best score
  uint32_t cut = band_cut_bits(t, band);
  uint32_t t_lo = 0u;
  bool strict = false;
  if (count_ge(cut) > limit) {  // the band does not fit: strict best k (ties at t cut by ordinal, the low word: bigger = smaller ordinal)
    strict = true;
    *lost_bits = t > 0u ? t : 1u;
    cut = t;
    int above = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) above += __popcll(__builtin_amdgcn_ballot_w64(key[j] != 0ull && sc[j] > t));
    if (count_ge(t) > k + 32) {
      const int need = k - above;  // >= 1 of the tied keys are still needed
      for (int b = 31; b >= 0; --b) {
        const uint32_t trial = t_lo | (1u << b);
        int c = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) c += __popcll(__builtin_amdgcn_ballot_w64(sc[j] == t && key[j] != 0ull && (uint32_t)key[j] >= trial));
        if (c >= need) t_lo = trial;
      }
    }
  }
  int base = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const bool keep = key[j] != 0ull && (sc[j] > cut || (sc[j] == cut && (uint32_t)key[j] >= t_lo));
    const u64 m = __builtin_amdgcn_ballot_w64(keep);
    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) buf[pos] = key[j];
    base += __popcll(m);
  }
  // rows that come later in this row range have bigger ordinals than everything kept: at a strict cut a tie at t loses (score > t);
  // at a band cut every score >= cut stays welcome (score > the float just below cut)
  *thr_excl = strict ? __uint_as_float(t) : (cut > 0u ? __uint_as_float(cut - 1u) : -__builtin_inff());
  return base;
}

#define TAVB_SB() __builtin_amdgcn_sched_barrier(0)
#define TAVB_BARRIER()            \
  do {                            \
    TAVB_SB();                    \
    __builtin_amdgcn_s_barrier(); \
    TAVB_SB();                    \
  } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------------------------
// The 256-query tile: four waves with the whole register file each, K steps of whole cache lines.
// (Round 1 shipped an 8-wave 256 x 256 tile with K steps of 32 halves -- "variant 3" -- and a 4-wave 384 x 256 one --
// "variant 5"; this kernel, "variant 6" in the profiles, replaced both: profiles/r02_cfg3_ablation.md.)
//
// Why four waves: an 8-wave 256 x 256 tile sits on the machine balance between the L2 -> CU operand path and the
// matrix pipe (32 KiB of operands per 256 x 256 x 32 step).  Fewer operand bytes per flop needs a bigger tile per CU,
// and the biggest one the register file allows is held by FOUR waves (2 x 2), one per SIMD, each with the full
// 512-register budget.  hipcc picks the AGPR or the VGPR form of an MFMA builtin per FUNCTION, so > 256 accumulators
// cannot be split over the two files through the builtin (hundreds of spills); the MFMAs are therefore inline asm with
// explicit register classes ("+a" / "+v").  A volatile asm is ordered against memory operations, so the program order
// -- MFMA, LDS read, MFMA, ..., MFMA, LDS-DMA -- IS the schedule (no sched_group_barrier).  The asm MFMAs are invisible
// to the compiler's hazard recognizer: the epilogue opens with the wait states an MFMA result needs.
//
// Why whole lines: with K steps of 32 halves a staging piece is sixteen 64-byte HALF lines; the CU's
// texture-address path serves a 1 KiB piece of that shape in ~14 ns from L2 against ~7.7 ns for eight whole 128-byte
// lines (tools/microbench/load_paths.hip, all CUs pulling; profiles/r02_operand_path.md).  A round-1 four-wave kernel
// with 32-half steps kept that path ~90 % busy and its waves stalled at the ISSUE of their staging loads.  Here a K
// step is 64 halves = one 128-byte line per row:
//   * tile = 320 corpus rows x 256 queries; a wave owns 160 x 128 = 5 x 4 MFMA tiles (320 accumulator registers: 15
//     tiles in AGPRs, 5 in VGPRs; the spare AGPRs are where the register allocator parks VGPR values during the
//     epilogue -- with all 256 taken it parks them in scratch, and a scratch reload is a VMEM load queued behind the
//     whole in-flight LDS-DMA).  (384 rows at K = 64 need 2 x 80 KiB of LDS: all 160 KiB, nothing left for the
//     selection state.)
//   * LDS: two slots per operand (A 40 KiB, B 32 KiB each) = 144 KiB.  Rows are 128 bytes; 16-byte slot j of row r
//     sits at physical slot j ^ ((r >> 1) & 7) (applied to the global SOURCE address of the staging loads, because
//     LDS-DMA writes lane-linear, and to the fragment reads), which spreads the 16 lanes of every ds_read_b128 group
//     over the 16 bank slots: SQ_LDS_BANK_CONFLICT = 0 measured.
//   * staging goes through buffer descriptors (`buffer_load_dwordx4 ... lds`): the per-lane part of an address is one
//     of two persistent 32-bit VGPR offsets (even / odd piece: the swizzle term has a piece-parity bit), the piece and the
//     K step are the scalar offset, the tile is the descriptor base, and rows past the end of the corpus are cut off by
//     the descriptor's size (they read as zero; the epilogue masks them anyway).  A piece = 8 rows x 128 bytes; per
//     step 40 A + 32 B pieces = 18 per wave (the 32-half form needed 40 per wave for the same K range).
//   * a step is four quarters (k16 slices) of 20 MFMAs.  Quarter q multiplies fragment set q & 1 while the 9
//     fragment reads of the next quarter fill the other set; one barrier per step, in front of quarter 3:
//       q0, q1, q2: multiply slices 0-2 of slot P; fetch slices 1-3 of slot P
//       ---- vmcnt(0): this wave's pieces of step S+1 landed; lgkmcnt(0): slot P read out; s_barrier ----
//       q3: multiply slice 3; fetch slice 0 of slot P^1 (step S+1, across tile boundaries too)
//     Slot P is then free: the pieces of step S+2 are issued behind the MFMAs of q3 (N3 of them), of the next q0 (N0)
//     and q1 (N1) -- corpus pieces first, they have the longest way -- and have until the next barrier to land.
//   * the first quarter of a tile multiplies into a ZERO C operand instead of clearing 320 registers.
//   * NI = 2 (the 128-query width, WideGeom<2>): a wave owns 160 x 64 = 10 accumulator blocks, all in AGPRs; half the MFMAs per
//     step no longer cover the loaded HBM latency with one corpus slab in flight, and this width has the LDS for a THIRD
//     corpus slot (3 x 40 + 2 x 16 KiB): a round then stages query slab S + 1 first and corpus slab S + 2 behind it, and the
//     wait in front of quarter 3 is counted (`vmcnt(10)`: everything but the ten corpus pieces of slab S + 2 has landed).
//     HBM-bound: 5.8 TB/s at 128 queries (profiles/r02_mid_batch.md).
// Measured and rejected (profiles/r02_cfg3_ablation.md): touching the corpus lines of the step 1 / 2 / 4 steps ahead
// into L2 with one 4-byte load per line (-3 .. -6 %); other piece-per-quarter schedules (no difference).
// ---------------------------------------------------------------------------------------------
constexpr int BM6 = 320;
constexpr int NT6 = 256;
constexpr int SLOT_A6 = BM6 * 128;  // 40 KiB
constexpr int PIECES_A6 = BM6 / 8 / 4;  // per wave per step: 10

// NI = 32-query MFMA blocks per wave along the query axis: 4 (256-query workgroup tile) or 2 (128-query tile, for batches of
// 65 .. 128 queries: half the MFMAs and half the query-operand traffic per corpus byte -- HBM-bound instead of padding-bound)
template <int NI>
struct WideGeom {
  static constexpr int QT = 64 * NI;           // queries per workgroup tile
  static constexpr int WQ = 32 * NI;           // ... per wave
  static constexpr int NT = 5 * NI;            // 32 x 32 accumulator blocks per wave: 20 or 10
  static constexpr int NA = NI == 4 ? 15 : NT; // blocks 0 .. NA-1 accumulate in AGPRs, the rest in VGPRs (the spare AGPRs are where the allocator parks VGPR values in the epilogue: no scratch)
  static constexpr int SLOT_B = QT * 128;      // 32 or 16 KiB
  // corpus ring: the 128-query tile is HBM-bound and has the LDS for a third slot (3 x 40 + 2 x 16 KiB): the corpus slab of
  // step S + 2 is in flight while step S is multiplied (one slab in flight left the tile latency-bound at 5.2 TB/s)
  static constexpr int RA = NI == 2 ? 3 : 2;
  static constexpr int B_RING = RA * SLOT_A6;  // the query ring (always two slots) sits behind the corpus ring
  static constexpr int CTRL = B_RING + 2 * SLOT_B;
  static constexpr int LDS = CTRL + QT * 8 + 16;
  static constexpr int PIECES_B = QT / 8 / 4;  // per wave per step: 8 or 4
  static constexpr int PIECES = PIECES_A6 + PIECES_B;
};

// index of the staging piece issued behind MFMA `i` of quarter `q` (-1: none): n pieces spread evenly over the NT MFMAs
template <int NT, int N3, int N0, int N1>
constexpr int staging_piece_at(int q, int i) {
  const int n = q == 3 ? N3 : q == 0 ? N0 : q == 1 ? N1 : 0;
  const int base = q == 3 ? 0 : q == 0 ? N3 : N3 + N0;
  for (int j = 0; j < n; ++j)
    if ((j * NT + NT / 2) / n == i) return base + j;
  return -1;
}

// SPLIT: the queries arrive as TWO fp16 planes (q = hi + lo to 2^-22) and a tile runs its K loop twice over the corpus rows, once per plane, into
// the same accumulators: fp32 query x fp16 row like the 64-query exact tile, at the wide tile's rate -- the bounded fallback for batches in which
// MANY queries have more near-duplicates than a band holds (tavb_rescore.hip).  Twice the MFMAs; only this instantiation pays for it.
// BD ("B direct"): the query operand does not go through LDS at all.  The library lays the fp16 queries out in MFMA-FRAGMENT-MAJOR order (1 KiB per
// (K step, k16 slice, 32-query block): lane l = query l & 31, halves 8 (l >> 5) .. + 7 of the slice -- query_prepare_kernel), so a fragment is ONE
// coalesced 16 B-per-lane load out of L2 straight into the registers the MFMA reads; four register sets rotate, the loads run three quarters
// (~1.2 us) ahead.  Per K step the LDS then moves 120 KiB instead of 216 (no query slab written, no query fragments read); the price is that
// both row halves of the workgroup load the same fragments (L2 -> CU traffic 104 KiB per step instead of 72).
template <int ABL, int NI, int N3, int N0, int N1, bool SPLIT = false, bool BD = false>
__global__ void __launch_bounds__(NT6) mfma_scan_kernel(const MfmaDeviceParams p) {
  using G = WideGeom<NI>;
  constexpr int BN = G::QT, NT = G::NT, SLOT_B6 = G::SLOT_B, PIECES_B6 = G::PIECES_B, B_RING6 = G::B_RING;
  // BD: the 64 KiB the query ring occupied pay for a THIRD corpus slot -- a corpus piece then has more than a whole K step (~2 us) to land instead
  // of 0.4 .. 1 step (the last pieces of a slab are issued in quarter 1 and needed behind quarter 2: an HBM round trip does not fit)
  constexpr int RA = BD ? 3 : G::RA;
  constexpr int CTRL6 = BD ? 3 * SLOT_A6 : G::CTRL;
  constexpr int PIECES6 = BD ? PIECES_A6 : G::PIECES;
  static_assert(!BD || (NI == 4 && !SPLIT), "the direct query operand is built for the 256-query tile");
  static_assert(N3 + N0 + N1 == PIECES6, "every piece of a step is issued exactly once");
  static_assert(N3 <= NT && N0 <= NT && N1 <= NT && NI + 5 <= NT, "one piece / one fragment read behind an MFMA at most");
  extern __shared__ __align__(16) unsigned char smem[];
  float* thr_lds = reinterpret_cast<float*>(smem + CTRL6);
  int* cnt_lds = reinterpret_cast<int*>(smem + CTRL6 + BN * 4);
  typedef __attribute__((address_space(3))) volatile int lds_flag;
  lds_flag* need_compact = (lds_flag*)(smem + CTRL6 + BN * 8);

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1;  // rows wm * 160 ..
  const int wn = wave & 1;   // queries wn * 32 * NI ..

  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int t = b >> 3;
  const int qtile = t % p.n_qtiles;
  const int split = (t / p.n_qtiles) * 8 + xcd;
  if (split >= p.n_splits) return;
  if (p.gate != nullptr && *p.gate > p.gate_max) return;  // most of the batch is going to the exact form anyway: this filter phase would be wasted work
  int live_q = p.nq;  // queries that exist: the batch, or -- for a fixed-shape launch over a device-side work list (tavb_rescore.hip) -- the slots in use
  if (p.active != nullptr) {  // nothing to do, or not this kernel's share
    const int live = *p.active;
    if (live <= p.active_min || live > p.active_max || qtile * BN >= live) return;
    live_q = live < live_q ? live : live_q;
  }
  const int64_t r_begin = (int64_t)split * p.rows_per_split;
  const int64_t r_end = (r_begin + p.rows_per_split < p.rows) ? r_begin + p.rows_per_split : p.rows;
  const int logical_block = split * p.n_qtiles + qtile;
  u64* my_cand = p.cand + (size_t)logical_block * BN * CAPW;
  int* my_counts = p.counts + (size_t)logical_block * BN;

  const float thr0 = (p.min_score > 0.0f) ? __uint_as_float(__float_as_uint(p.min_score) - 1u) : -__builtin_inff();
  for (int i = tid; i < BN; i += NT6) {
    // NaN threshold admits nothing; neither does one above 1 (scores are clipped to [0, 1]) -- with that, `score > thr` alone implies
    // `clip(score) >= min_score` (thr >= the float below min_score), and the epilogue needs no second test per row
    float t0 = (p.min_score != p.min_score || p.min_score > 1.0f) ? __builtin_inff() : thr0;
    const int qg0 = qtile * BN + i;
    // padding queries -- and the unused slots of the last live tile of a work list: zero queries, every row scores 0.5, and from the second
    // ladder phase on their thr_in is NaN (the select kernel skips them), so without this they would admit every row of the big phases --
    // admit nothing
    if (qg0 >= live_q) t0 = __builtin_inff();
    else if (p.thr_in && p.thr_in[qg0] > t0) t0 = p.thr_in[qg0];  // k-th best so far: a valid lower bound
    thr_lds[i] = t0;
    cnt_lds[i] = 0;
  }
  if (tid == 0) *need_compact = 0;

  const int D = p.dim;
  const int steps_per_plane = D / 64;
  const int steps_per_tile = SPLIT ? 2 * steps_per_plane : steps_per_plane;
  const uint32_t row_bytes = (uint32_t)D * 2u;
  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const char* qbase = reinterpret_cast<const char*>(p.queries) + (size_t)qtile * BN * row_bytes;
  const int n_tiles = (r_end > r_begin) ? (int)((r_end - r_begin + BM6 - 1) / BM6) : 0;
  if (n_tiles == 0) {
    for (int i = tid; i < BN; i += NT6) my_counts[i] = 0;  // empty row range: empty buffers
    return;
  }

  // ---- per-lane constants of the K loop: two staging offsets (even / odd piece), three fragment-address terms.
  //      Staging: lane l = row l >> 3 of an 8-row piece, PHYSICAL 16-byte slot l & 7, which holds logical slot
  //      (l & 7) ^ ((row >> 1) & 7); with row = 8 * piece + (l >> 3) that is (l & 7) ^ (4 * (piece & 1) + (l >> 4)).
  int st_even, st_odd;
  uint32_t frag_x, a_lane, b_lane;
  {
    int zero = 0;
    asm volatile("" : "+v"(zero));
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero));
    const uint32_t lane_row = (uint32_t)(ln >> 3);
    st_even = (int)(lane_row * row_bytes + (uint32_t)(((ln & 7) ^ (ln >> 4)) * 16));
    st_odd = (int)(lane_row * row_bytes + (uint32_t)(((ln & 7) ^ (4 + (ln >> 4))) * 16));
    const int frag_row = ln & 31;
    frag_x = (uint32_t)(((ln >> 5) ^ ((frag_row >> 1) & 7)) << 4);  // byte (k16 << 5) ^ frag_x within the 128-byte row
    a_lane = (uint32_t)((wm * 160 + frag_row) * 128);              // + mi * 4096
    b_lane = (uint32_t)(B_RING6 + (wn * G::WQ + frag_row) * 128);    // + ni * 4096
  }
  const __amdgpu_buffer_rsrc_t rsrc_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sgpr_ptr(qbase)), 0, (int)(BN * row_bytes) + (SPLIT ? (int)p.split_plane : 0), 0x00020000);
  const int plane_jump = SPLIT ? (int)p.split_plane - steps_per_plane * 128 : 0;  // K step s >= steps_per_plane reads step s - steps_per_plane of the low plane

  // ---- stager.  A "round" is what one K step issues: two slots deep (256-query tile) round S = corpus slab S + 1 then query
  //      slab S + 1; three corpus slots deep (128-query tile) round S = query slab S + 1 FIRST, then corpus slab S + 2, so that
  //      the counted wait of step S ("everything but the newest PIECES_A6 loads has landed") covers query slab S + 1 and corpus
  //      slab S + 1 while corpus slab S + 2 stays in flight.  Piece IDX of a round: its position in that order.
  int sa_kt = 0, sa_tile = 0, sa_slot = 0;  // corpus slab being staged
  int sb_kt = 0, sb_slot = 0;               // query slab being staged
  auto stage_a = [&](auto j_tag) {
    constexpr int J = decltype(j_tag)::value;
    const int tile = sa_tile < n_tiles ? sa_tile : n_tiles - 1;  // past the end: harmless reloads of the last tile
    const int64_t row0 = (ABL & 4) ? 0 : r_begin + (int64_t)tile * BM6;
    const int64_t left = p.rows - row0;
    const int valid = (int)(left < BM6 ? left : BM6);  // rows past the end of the corpus read as zero (masked in the epilogue)
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(sgpr_ptr(corpus + (size_t)row0 * row_bytes)), 0, __builtin_amdgcn_readfirstlane(valid * (int)row_bytes), 0x00020000);
    const int pc = wave * PIECES_A6 + J;
    unsigned char* la = smem + sa_slot * SLOT_A6 + pc * 1024;
    const int a_kt = (SPLIT && sa_kt >= steps_per_plane) ? sa_kt - steps_per_plane : sa_kt;  // second plane: the same corpus columns again
    const int soff = __builtin_amdgcn_readfirstlane(a_kt * 128 + pc * 8 * (int)row_bytes);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)la, 16, (J & 1) ? st_odd : st_even, soff, 0, 0);
    if constexpr (J == PIECES_A6 - 1) {
      sa_slot = (sa_slot + 1 == RA) ? 0 : sa_slot + 1;
      const bool wrap = (sa_kt + 1 == steps_per_tile);
      sa_kt = wrap ? 0 : sa_kt + 1;
      sa_tile += wrap ? 1 : 0;
    }
  };
  auto stage_b = [&](auto j_tag) {
    constexpr int BJ = decltype(j_tag)::value;
    const int pc = wave * PIECES_B6 + BJ;
    unsigned char* lb = smem + B_RING6 + sb_slot * SLOT_B6 + pc * 1024;
    const int soff = __builtin_amdgcn_readfirstlane(((ABL & 8) ? 0 : sb_kt * 128 + ((SPLIT && sb_kt >= steps_per_plane) ? plane_jump : 0)) +
                                                    pc * 8 * (int)row_bytes);  // ablation 8: the query operand's K step 0 every time (cache resident)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void*)lb, 16, (BJ & 1) ? st_odd : st_even, soff, 0, 0);
    if constexpr (BJ == PIECES_B6 - 1) {
      sb_slot ^= 1;
      sb_kt = (sb_kt + 1 == steps_per_tile) ? 0 : sb_kt + 1;
    }
  };
  auto stage_piece = [&](auto idx_tag) {
    constexpr int IDX = decltype(idx_tag)::value;
    if constexpr (BD) {
      stage_a(std::integral_constant<int, IDX>{});
    } else if constexpr (RA == 2) {
      if constexpr (IDX < PIECES_A6) stage_a(std::integral_constant<int, IDX>{});
      else stage_b(std::integral_constant<int, IDX - PIECES_A6>{});
    } else {
      if constexpr (IDX < PIECES_B6) stage_b(std::integral_constant<int, IDX>{});
      else stage_a(std::integral_constant<int, IDX - PIECES_B6>{});
    }
  };
  auto stage_range = [&]<int... I>(std::integer_sequence<int, I...>) { (stage_piece(std::integral_constant<int, I>{}), ...); };
  auto stage_a_all = [&]<int... I>(std::integer_sequence<int, I...>) { (stage_a(std::integral_constant<int, I>{}), ...); };

  // ---- prologue: step 0 whole (three slots: and corpus slab 1), then the first N3 pieces of round 0 (what quarter 3 of a
  //      step "-1" would have issued)
  if constexpr (RA == 2) {
    stage_range(std::make_integer_sequence<int, PIECES6>{});
    stage_range(std::make_integer_sequence<int, N3>{});
    wait_vmcnt<N3>();
  } else {
    stage_a_all(std::make_integer_sequence<int, PIECES_A6>{});  // corpus slab 0
    stage_range(std::make_integer_sequence<int, PIECES6>{});    // "round -1": query slab 0, corpus slab 1
    stage_range(std::make_integer_sequence<int, N3>{});
    wait_vmcnt<PIECES_A6 + N3>();
  }
  __syncthreads();  // step 0 landed everywhere, thresholds initialised (the waits above are counted: nothing is drained)

  typedef int i32x4 __attribute__((ext_vector_type(4)));
  constexpr int NA_TILES = G::NA;
  f32x16 acc_a[NA_TILES];
  f32x16 acc_v[NT - NA_TILES > 0 ? NT - NA_TILES : 1];
#define TAVB_MFMA6_A(ACC, A, B) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(__builtin_bit_cast(i32x4, A)), "v"(__builtin_bit_cast(i32x4, B)))
#define TAVB_MFMA6_V(ACC, A, B) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(__builtin_bit_cast(i32x4, A)), "v"(__builtin_bit_cast(i32x4, B)))
#define TAVB_MFMA6_A0(ACC, A, B) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(ACC) : "v"(__builtin_bit_cast(i32x4, A)), "v"(__builtin_bit_cast(i32x4, B)))
#define TAVB_MFMA6_V0(ACC, A, B) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(ACC) : "v"(__builtin_bit_cast(i32x4, A)), "v"(__builtin_bit_cast(i32x4, B)))

  f16x8 a0[5], b0[NI], a1[5], b1[NI];
  f16x8 b2[BD ? NI : 1], b3[BD ? NI : 1];  // BD: four rotating sets of query fragments (quarter q multiplies set q, the loads for quarter q + 3 fill set (q + 3) & 3)
  constexpr int BQ_SLICE = (BN / 32) * 1024;  // bytes of one k16 slice of the tile's queries in fragment-major order
  int bq_soff = 0;                            // BD: byte offset (from the tile's queries) of the slice to load next; wraps with the tile
  const int bq_tile_bytes = BN * (int)row_bytes;
  int bq_voff = 0;
  auto bq_load = [&](f16x8(&dst)[BD ? NI : 1]) {  // one slice: this wave's NI fragments (its half of the tile's query blocks)
    if constexpr (BD) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        dst[ni] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, bq_voff + ni * 1024, bq_soff, 0));
      bq_soff = (bq_soff + BQ_SLICE == bq_tile_bytes) ? 0 : bq_soff + BQ_SLICE;
    }
  };
  {
    const unsigned char* abase = smem + (a_lane + frag_x);
    const unsigned char* bbase = smem + (b_lane + frag_x);
    if constexpr (BD) {
      int zero_b = 0;
      asm volatile("" : "+v"(zero_b));
      const int ln_b = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero_b));
      bq_voff = ln_b * 16 + wn * NI * 1024;
      bq_load(b0);
      bq_load(b1);
      bq_load(b2);
    } else {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b0[ni] = *reinterpret_cast<const f16x8*>(bbase + ni * 4096);
    }
#pragma unroll
    for (int mi = 0; mi < 5; ++mi) a0[mi] = *reinterpret_cast<const f16x8*>(abase + mi * 4096);
  }
  int rd = 0, rd_a = 0;  // ring slots (query, corpus) of the step being multiplied

  // One quarter: the NT MFMAs of one k16 slice on (fa, fb); behind them, in program order, the NI + 5 fragment reads of the
  // next quarter (slot `nslot`, slice NKK) into (na, nb) and the staging pieces the schedule puts into quarter Q.
  auto quarter = [&](auto q_tag, auto first_tag, f16x8(&fa)[5], f16x8(&fb)[NI], f16x8(&na)[5], f16x8(&nb)[NI], int nslot_a, int nslot, auto nkk_tag) {
    constexpr int Q = decltype(q_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;  // first quarter of a tile: C = 0
    constexpr int NKK = decltype(nkk_tag)::value;
    const uint32_t kx = (uint32_t)(NKK << 5) ^ frag_x;
    const unsigned char* abase = smem + nslot_a * SLOT_A6 + (a_lane + kx);
    const unsigned char* bbase = smem + nslot * SLOT_B6 + (b_lane + kx);
    auto mfma_at = [&](auto i_tag) {
      constexpr int I = decltype(i_tag)::value;
      // issue order of the NT MFMAs of a slice (measurement, profiles/r06_mfma_power.md): corpus fragment outermost (ships), query fragment
      // outermost (ABL & 1024: the operand whose bits the board's power follows more closely stays put for five MFMAs), and either walked
      // boustrophedon (ABL & 2048: exactly one operand changes between any two consecutive MFMAs)
      constexpr bool BMAJ = (ABL & 1024) != 0, SERP = (ABL & 2048) != 0;
      constexpr int outer = BMAJ ? I / 5 : I / NI;
      constexpr int inner0 = BMAJ ? I % 5 : I % NI;
      constexpr int inner = (SERP && (outer & 1)) ? (BMAJ ? 4 : NI - 1) - inner0 : inner0;
      constexpr int mi = BMAJ ? inner : outer, ni = BMAJ ? outer : inner;
      constexpr int J = mi * NI + ni;  // the accumulator block
      if constexpr ((ABL & 1) == 0) {
        if constexpr (FIRST) {
          if constexpr (J < NA_TILES)
            TAVB_MFMA6_A0(acc_a[J], fa[mi], fb[ni]);
          else
            TAVB_MFMA6_V0(acc_v[J - NA_TILES], fa[mi], fb[ni]);
        } else {
          if constexpr (J < NA_TILES)
            TAVB_MFMA6_A(acc_a[J], fa[mi], fb[ni]);
          else
            TAVB_MFMA6_V(acc_v[J - NA_TILES], fa[mi], fb[ni]);
        }
      }
      if constexpr ((ABL & 32) == 0) {
        if constexpr (BD) {  // the slice three quarters ahead, straight from L2 into the set the previous quarter has just finished with
          if constexpr (I < NI) nb[I] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, bq_voff + I * 1024, bq_soff, 0));
        } else {
          if constexpr (I < NI) nb[I] = *reinterpret_cast<const f16x8*>(bbase + I * 4096);
        }
        if constexpr (I >= NI && I < NI + 5) na[I - NI] = *reinterpret_cast<const f16x8*>(abase + (I - NI) * 4096);
      }
      if constexpr ((ABL & 2) == 0) {
        constexpr int PC = staging_piece_at<NT, N3, N0, N1>(Q, I);
        if constexpr (PC >= 0) stage_piece(std::integral_constant<int, PC>{});
      }
    };
    [&]<int... I>(std::integer_sequence<int, I...>) { (mfma_at(std::integral_constant<int, I>{}), ...); }
    (std::make_integer_sequence<int, NT>{});
    if constexpr (BD) bq_soff = (bq_soff + BQ_SLICE == bq_tile_bytes) ? 0 : bq_soff + BQ_SLICE;
    if constexpr ((ABL & 1) != 0) asm volatile("" ::"v"(fa[0]), "v"(fa[4]), "v"(fb[0]), "v"(fb[NI - 1]));
  };
  using Q0 = std::integral_constant<int, 0>;
  using Q1 = std::integral_constant<int, 1>;
  using Q2 = std::integral_constant<int, 2>;
  using Q3 = std::integral_constant<int, 3>;
  auto step = [&](auto first_tag) {
    if constexpr (BD) {
      quarter(Q0{}, first_tag, a0, b0, a1, b3, rd_a, rd, Q1{});
      quarter(Q1{}, std::false_type{}, a1, b1, a0, b0, rd_a, rd, Q2{});
      quarter(Q2{}, std::false_type{}, a0, b2, a1, b1, rd_a, rd, Q3{});
      // corpus slab S+1 has landed in this wave: behind its last piece (quarter 1 of the PREVIOUS step) came five quarters' query-fragment loads
      // (5 NI) and the ten pieces of slab S+2, which may all still be in flight (loads return in order: the count is exact)
      if constexpr ((ABL & 2) == 0) wait_vmcnt<5 * NI + PIECES_A6>();
      __builtin_amdgcn_s_waitcnt(0xC07F);
      TAVB_BARRIER();
      const int nxt_a = rd_a + 1 == RA ? 0 : rd_a + 1;
      quarter(Q3{}, std::false_type{}, a1, b3, a0, b2, nxt_a, rd, Q0{});
      rd_a = nxt_a;
    } else {
      quarter(Q0{}, first_tag, a0, b0, a1, b1, rd_a, rd, Q1{});
      quarter(Q1{}, std::false_type{}, a1, b1, a0, b0, rd_a, rd, Q2{});
      quarter(Q2{}, std::false_type{}, a0, b0, a1, b1, rd_a, rd, Q3{});
      // ---- step S+1 has landed in this wave (two slots: nothing newer is in flight; three: only corpus slab S+2 is); the
      //      slots of step S are read out; meet
      if constexpr ((ABL & 2) == 0) wait_vmcnt<(RA == 2 ? 0 : PIECES_A6)>();
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) through the builtin: visible to the compiler's wait-count pass
      TAVB_BARRIER();
      const int nxt_a = (RA == 2) ? (rd_a ^ 1) : (rd_a + 1 == RA ? 0 : rd_a + 1);
      quarter(Q3{}, std::false_type{}, a1, b1, a0, b0, nxt_a, rd ^ 1, Q0{});
      rd ^= 1;
      rd_a = nxt_a;
    }
  };

  for (int tile = 0; tile < n_tiles; ++tile) {
    const int64_t row0 = r_begin + (int64_t)tile * BM6;
    const bool tile_full = row0 + BM6 <= r_end;  // wave-uniform: every row of this tile belongs to the row range
    if constexpr ((ABL & 1) != 0) {  // MFMAs ablated: give the accumulators a value
#pragma unroll
      for (int i = 0; i < NA_TILES; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_a[i][r] = 0.f;
#pragma unroll
      for (int i = 0; i < NT - NA_TILES; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_v[i][r] = 0.f;
    }
    step(std::true_type{});
#pragma unroll 1
    for (int kt = 1; kt < steps_per_tile; ++kt) step(std::false_type{});

    // ---- epilogue: admission test on the raw dot products, append .  The asm MFMAs are invisible
    //      to the compiler's hazard recognizer: a 32x32x16 MFMA needs 18 wait states before its result may be read.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int zero_e = 0;
    asm volatile("" : "+v"(zero_e));
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero_e));
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int ql = wn * G::WQ + ni * 32 + (lane_e & 31);
        const float thr = thr_lds[ql];
        const float thr_pre = fmaf(thr, 2.0f, -1.0f) - 4.8e-7f;  // score > thr implies dot > thr_pre: fma(dot, 0.5, 0.5) is monotone, the margin covers both roundings
#pragma unroll
        for (int mi = 0; mi < 5; ++mi) {
          constexpr int VT = NT - NA_TILES > 0 ? NT - NA_TILES : 1;
          if ((mi * NI + ni >= NA_TILES) != (pass == 0)) continue;  // pass 0: VGPR tiles, pass 1: AGPR tiles
          const f32x16 dots = (mi * NI + ni < NA_TILES) ? acc_a[mi * NI + ni] : acc_v[(mi * NI + ni - NA_TILES + VT) % VT];
          float top = dots[0];
#pragma unroll
          for (int r = 1; r < 16; ++r) top = __builtin_fmaxf(top, dots[r]);
          TAVB_SB();  // one block at a time
          const bool any = ((ABL & ~3072) == 0 || ABL == 512) && (top > thr_pre);  // (bits 1024 / 2048: MFMA issue order, not an ablation)
          if constexpr (ABL == 512) asm volatile("" ::"s"(__builtin_amdgcn_ballot_w64(any)));
          if constexpr ((ABL & ~3072) != 0 && ABL != 512) asm volatile("" ::"v"(top));
          if (ABL != 512 && __builtin_amdgcn_ballot_w64(any) != 0ull) {
            // (rare: ~1 % of the blocks once the ladder's thresholds are in -- but each costs the workgroup ~0.3 us, and a batch has a few hundred
            //  thousand of them.)  One compare per row whose result is a WAVE mask in scalar registers (v_cmp into an SGPR pair: no per-lane
            //  bit twiddling); a row nobody admits -- 15 of 16 in the usual case -- costs one scalar test.  An admitted row takes its slot
            //  with one LDS atomic per admitting lane.  Rows past the end of the row range exist only in a range's last tile (wave-uniform).
            const int64_t row_base = row0 + wm * 160 + mi * 32 + 4 * (lane_e >> 5);
            // rows of this block that belong to the row range, seen from this lane's first row (>= 32: all of them)
            const int64_t left64 = r_end - row_base;
            const int rows_left = tile_full ? 64 : (int)(left64 < 64 ? left64 : 64);
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // four rows at a time: their masks stay in scalar registers
              float sc[4];
              u64 m[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                sc[j] = fmaf(dots[4 * g + j], 0.5f, 0.5f);
                asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m[j]) : "v"(sc[j]), "v"(thr));  // (the builtin ballot goes through a 0/1 VGPR and back)
              }
              if ((m[0] | m[1] | m[2] | m[3]) == 0ull) continue;  // wave-uniform: nobody admits any of the four
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int r_off = j + 8 * g;  // row 4 g + j of the accumulator = tile row (r & 3) + 8 (r >> 2) from row_base
                if (m[j] == 0ull) continue;
                if (((m[j] >> lane_e) & 1ull) != 0ull && r_off < rows_left) {
                  const int pos = lds_add_rtn(&cnt_lds[ql], 1);
                  if (pos + 1 > CAPW - BM6) lds_store_i32(need_compact, 1);  // this buffer could overflow on the next tile
                  float s1 = (sc[j] > 0.0f) ? sc[j] : 0.0f;
                  s1 = (s1 > 1.0f) ? 1.0f : s1;
                  if (pos < CAPW) my_cand[(size_t)ql * CAPW + pos] = make_key(s1, (uint32_t)(row_base + r_off) + p.index_base);
                }
              }
            }
          }
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TAVB_BARRIER();
    if (*need_compact != 0) {  // workgroup-uniform: read after the barrier
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TAVB_BARRIER();
      for (int q = wave; q < BN; q += NT6 / 64) {
        const int n = cnt_lds[q];
        if (n > CAPW - BM6) {
          u64* buf = my_cand + (size_t)q * CAPW;
          const int qg = qtile * BN + q;  // (a padding query admits nothing: never here)
          const float band = p.band ? p.band[qg] : 0.0f;
          float thr_excl;
          uint32_t lost;
          const int kept = compact_to_band<CAPW>(buf, n < CAPW ? n : CAPW, p.k, lane_e, band, CAPW - BM6 - 64, &thr_excl, &lost);
          if (lane_e == 0) {
            cnt_lds[q] = kept;
            if (thr_excl > thr_lds[q]) thr_lds[q] = thr_excl;
            if (lost != 0u && p.lost) atomicMax(&p.lost[qg], lost);  // (n > CAPW cannot happen: a tile appends at most BM6 keys)
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TAVB_BARRIER();
      if (tid == 0) *need_compact = 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TAVB_BARRIER();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead LDS-DMA before the block retires
  __syncthreads();

  // the buffers stay unsorted: tavb::select_band_kernel picks the band over all workgroups' buffers of a query
  for (int i = tid; i < BN; i += NT6) my_counts[i] = cnt_lds[i] < CAPW ? cnt_lds[i] : CAPW;
}


// ---------------------------------------------------------------------------------------------
// SKINNY kernel: 3 .. 32 queries per pass at HBM speed (32-query tile), 33+ on 64-query tiles; fp32 AND fp16 corpora.
//
// The streaming tiers keep the queries in LDS and every lane re-reads them for every row, so beyond four
// queries they are LDS-bound (6.9 TB/s of corpus at one query, 4.8 at eight, then one more pass per eight
// queries); the 256-query MFMA tile above wastes 7/8 of its operand traffic on padding at 32 queries and
// only exists for fp16.  This kernel is the piece in between -- and the only matrix-core path for the
// reference's own dtype, fp32: `v_mfma_f32_32x32x2_f32` multiplies fp32 exactly and accumulates in fp32
// (157 TFLOP/s peak, enough to keep up with HBM at 32 queries: 1M x 1536 x 32 x 2 = 98 GFLOP per 6.1 GB pass).
//   * tile = 256 corpus rows x 32 (or 64) queries, 4 waves, each wave owns 64 rows (two, or four, 32 x 32 MFMA tiles).
//   * K advances STEP bytes per row per step for either dtype: 128 (a whole cache line per row and staging lane group:
//     the texture-address path serves eight 128-byte lines twice as fast as sixteen 64-byte half lines,
//     profiles/r02_operand_path.md) whenever a row is a multiple of 128 bytes, else 64.  A wave stages the 1 KiB pieces of
//     ITS OWN 64 rows by LDS-DMA, so the corpus operand needs no cross-wave synchronisation; the waves also share out
//     the pieces of the query operand, which all of them read: one barrier per step.  Ring of 2 .. 4 slots, counted vmcnt;
//     72 KiB configurations run two workgroups per CU, the others one.
//   * LDS rows are STEP bytes; 16-byte slot j of row r sits at physical slot j ^ ((r >> 1) & 7) (128-byte rows) or
//     j ^ ((r >> 2) & 3) (64-byte rows) -- on the global SOURCE address of the staging loads and on the fragment reads --
//     so the 16 lanes of a ds_read_b128 group hit 16 different bank slots.  For fp32 a lane's 16-byte fragment is four
//     consecutive k of its row -- lanes 0-31 take k = 8g .. 8g+3, lanes 32-63 k = 8g+4 .. 8g+7 -- and feeds four
//     MFMAs: MFMA e multiplies k = 8g+e (lower half-wave) and 8g+4+e (upper), the same pairing on both operands, which
//     is all a dot product needs.
//   * fp16 corpora: the fp32 queries are split into an fp16 high and an fp16 low plane (q = hi + lo to 2^-22; both are
//     multiplied -- the kernel is load-bound, the second MFMA is free), so a lookup on an fp16 corpus means the same
//     thing here as in the streaming tiers (fp32 query x fp16 rows).
//   * NI = 2 (64 queries per tile): twice the MFMAs per operand byte -- for batches of 33+ queries, which would
//     otherwise stream the corpus once per 32 queries (fp32) or pay for a 256-query tile (fp16, 33 .. 64 queries).
//   * epilogue / candidate buffers / compaction as in the 256-query tile (32 or 64 queries per block); at the end every
//     buffer is sorted into a list, tavb::merge_kernel merges the lists of the row ranges.
// ---------------------------------------------------------------------------------------------
constexpr int SQ32 = 32;            // queries per 32 x 32 MFMA block; a tile is NI of them (32 or 64 queries)
constexpr int S_THREADS = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4 global_f32x4;
template <typename T, int NI, int STEP, bool DEEP = false, int DR = 0, bool HALF = false>
struct SkinnyGeom {
  // HALF: 128-row tiles (a wave owns 32 rows = one 32 x 32 block per 32 queries) -- half the corpus slot, so that a ring of three fits twice
  // into a CU's LDS and TWO workgroups share a CU (one waits at its barrier, the other multiplies), as the fp32 32-query tile always did
  static constexpr int BMT = HALF ? BM / 2 : BM;   // corpus rows per tile
  static constexpr int RW = BMT / 4;               // ... per wave
  static constexpr int MI = RW / 32;               // 32-row blocks per wave
  static constexpr bool F32 = sizeof(T) == 4;
  static constexpr int SQ = NI * SQ32;
  static constexpr int PLANES = F32 ? 1 : 2;
  static constexpr int LPR = STEP / 16;            // lanes (16-byte slots) per row of a staging piece
  static constexpr int RPP = 64 / LPR;             // rows per 1 KiB piece: 16 or 8
  static constexpr int NPA = RW / RPP;             // corpus pieces per wave per step: 4 or 8 (2 or 4 with half tiles)
  static constexpr int SH = STEP == 64 ? 2 : 1;    // swizzle term = (row >> SH) & (LPR - 1)
  static constexpr int NG = STEP / 32;             // 32-byte k slices per step: 2 or 4
  static constexpr int SLOT_A = BMT * STEP;        // 16 or 32 KiB (half tiles: 8 or 16)
  static constexpr int PLANE_B = SQ * STEP;        // one query operand plane, one step
  static constexpr int SLOT_B = PLANES * PLANE_B;
  // ring depth: measured (profiles/r02_mid_batch.md) -- for 32 fp32 queries two workgroups per CU with two slots each beat one
  // workgroup with three or four slots (the depth in flight is not what limits this tile)
  static constexpr int RING = STEP == 64 ? ((F32 && NI == 1) ? 4 : 3) : ((F32 && NI == 1) ? 2 : 3);
  // DEEP (32-query tile, whole-line steps): the corpus ring one slot deeper than the query ring -- 4 x 32 KiB of corpus + 3 query slots
  // (152 KiB fp16, 140 KiB fp32), one workgroup per CU.  The tile is HBM-bound and what it lacks is bytes in flight: with a ring of three,
  // 32 .. 64 KiB of corpus per CU are on their way at any time (8 TB/s x ~2 us of loaded latency / 256 CUs = 62 KiB: the edge); with four
  // corpus slots 64 .. 96 KiB.  The query slabs come out of L2 and need no deeper ring.
  // DR > 0 (whole-line steps): REGISTER staging.  Every byte that is on its way from HBM needs somewhere to land; with LDS-DMA that is an LDS
  // slot, and 160 KiB of LDS hold 80 .. 110 KiB in flight however the ring is cut -- the edge of what 8 TB/s x ~2 us / 256 CUs asks for.  The
  // register file is three times the LDS: the loads of the next DR K steps (corpus AND query pieces, the same coalesced 1 KiB pieces) land in
  // DR x (NPA + PLANES * NPB) x 4 VGPRs per lane and are written to LDS (ds_write_b128, the layout the LDS-DMA would have produced) when
  // their step comes up.  LDS then holds ONE corpus slot (each wave's 64 rows are its own: DS operations of a wave execute in order, nothing
  // to synchronise) and two query slots; the compiler counts the waits (plain register dependencies).
  static constexpr int RA = DR > 0 ? 1 : DEEP ? 4 : RING;       // corpus slots
  static constexpr int RB = DR > 0 ? 2 : DEEP ? 3 : RING;       // query slots (RB <= RA for the LDS-DMA rings)
  static constexpr int B_RING = RA * SLOT_A;
  static constexpr int CTRL = RA * SLOT_A + RB * SLOT_B;
  static constexpr int LDS = CTRL + SQ * 8 + 16;
  static constexpr int WG_PER_CU = LDS <= 76 * 1024 ? 2 : 1;
  static constexpr int B_PIECES = SQ / RPP;        // query pieces per plane per step: 2 .. 8
  static constexpr int NPB = (B_PIECES + 3) / 4;   // ... per wave (waves >= B_PIECES stage none when there are fewer than 4)
  static_assert((DR > 0 || RB <= RA) && RB >= 2 && LDS <= 160 * 1024, "ring does not fit");
  static_assert(DR == 0 || STEP == 128, "register staging is built for whole-line steps");
};

template <typename T, int NI, int STEP, int ABL, bool DEEP = false, int DR = 0, bool HALF = false>
__global__ void __launch_bounds__(S_THREADS, ((DR >= 4 || (DR == 3 && sizeof(T) == 2)) ? 1 : 2)) skinny_scan_kernel(const MfmaDeviceParams p) {
  using G = SkinnyGeom<T, NI, STEP, DEEP, DR, HALF>;
  constexpr int BMT = G::BMT, RW = G::RW, MI = G::MI;
  constexpr int SQ = G::SQ;
  constexpr bool F32 = G::F32;
  constexpr int RA = G::RA, RB = G::RB, S_SLOT_A = G::SLOT_A, S_SLOT_B = G::SLOT_B, S_B_RING = G::B_RING, S_CTRL = G::CTRL;
  constexpr int NPA = G::NPA, NPB = G::NPB, RPP = G::RPP, LPR = G::LPR, NG = G::NG;
  extern __shared__ __align__(16) unsigned char smem[];
  float* thr_lds = reinterpret_cast<float*>(smem + S_CTRL);
  int* cnt_lds = reinterpret_cast<int*>(smem + S_CTRL + SQ * 4);
  // (an LDS-typed pointer: through a generic one the per-tile read below is a FLAT load, which counts in vmcnt AND lgkmcnt and cannot be waited
  // for by count -- the compiler drained the whole staging queue behind it once per tile, and with a flat access pending anywhere in the loop
  // it turns the first counted wait of every K-loop iteration into vmcnt(0) as well)
  typedef __attribute__((address_space(3))) volatile int lds_flag_t;
  lds_flag_t* need_compact = (lds_flag_t*)(smem + S_CTRL + SQ * 8);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // rows wave * 64 .. of the tile

  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int t = b >> 3;
  const int qtile = t % p.n_qtiles;
  const int split = (t / p.n_qtiles) * 8 + xcd;
  if (split >= p.n_splits) return;
  int live_q = p.nq;
  if (p.active != nullptr) {  // fixed-shape launch over a device-side work list (tavb_rescore.hip)
    const int live = *p.active;
    if (live <= p.active_min || live > p.active_max || qtile * SQ >= live) return;
    live_q = live < live_q ? live : live_q;
  }
  const int64_t r_begin = (int64_t)split * p.rows_per_split;
  const int64_t r_end = (r_begin + p.rows_per_split < p.rows) ? r_begin + p.rows_per_split : p.rows;
  const int logical_block = split * p.n_qtiles + qtile;
  u64* my_cand = p.cand + (size_t)logical_block * SQ * CAP;

  const float thr0 = (p.min_score > 0.0f) ? __uint_as_float(__float_as_uint(p.min_score) - 1u) : -__builtin_inff();
  if (tid < SQ) {  // SQ <= 64 < S_THREADS
    float t0 = (p.min_score != p.min_score || p.min_score > 1.0f) ? __builtin_inff() : thr0;  // NaN threshold admits nothing, nor one above 1 (as in the 256-query tile)
    const int qg0 = qtile * SQ + tid;
    if (qg0 >= live_q) t0 = __builtin_inff();  // padding queries (and unused work-list slots) admit nothing
    else if (p.thr_in && p.thr_in[qg0] > t0) t0 = p.thr_in[qg0];  // k-th best so far: a valid lower bound
    thr_lds[tid] = t0;
    cnt_lds[tid] = 0;
  }
  if (tid == 0) *need_compact = 0;

  const size_t row_bytes = (size_t)p.dim * sizeof(T);
  const int steps_per_tile = (int)(row_bytes / STEP);
  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const char* qbase = reinterpret_cast<const char*>(p.queries) + (size_t)qtile * SQ * row_bytes;
  const size_t plane_bytes = (size_t)p.n_qtiles * SQ * row_bytes;  // fp16: the low plane follows the high plane
  const int n_tiles = (r_end > r_begin) ? (int)((r_end - r_begin + BMT - 1) / BMT) : 0;
  if (n_tiles == 0) {
    for (int q = wave; q < SQ; q += S_THREADS / 64) {
      const int qg = qtile * SQ + q;
      if (qg < p.nq && lane < p.k) p.lists[((size_t)qg * p.list_stride + split) * (size_t)p.k + lane] = 0ull;
    }
    return;
  }

  // ---- stager: piece j of this wave = corpus rows wave * 64 + j * RPP .. of the tile (its own rows); lane l = row
  //      l / LPR, PHYSICAL 16-byte slot l % LPR, which holds the logical slot (l % LPR) ^ ((row >> SH) & (LPR - 1)).
  //      Query piece pb = wave + 4 i covers query rows pb * RPP ..
  const int st_row_in_piece = lane / LPR;
  uint32_t st_off[NPA];
  auto set_offsets = [&](int64_t row0) {
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
      const int row = wave * RW + j * RPP + st_row_in_piece;  // row of the tile: fixes the swizzle term
      int64_t r = row;
      if (row0 + r >= p.rows) r = p.rows - 1 - row0;  // stay in bounds; masked in the epilogue
      st_off[j] = (uint32_t)r * (uint32_t)row_bytes + (uint32_t)((((lane % LPR) ^ ((row >> G::SH) & (LPR - 1)))) * 16);
    }
  };
  uint32_t st_off_b[NPB];
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    const int row = (wave + 4 * i) * RPP + st_row_in_piece;
    st_off_b[i] = (uint32_t)row * (uint32_t)row_bytes + (uint32_t)((((lane % LPR) ^ ((row >> G::SH) & (LPR - 1)))) * 16);
  }
  const bool stages_b = G::B_PIECES >= 4 || wave < G::B_PIECES;  // (with fewer than four query pieces the last waves stage none)
  // Two rings: corpus slabs run RA - 1 steps ahead of the multiply, query slabs RB - 1 (RB <= RA).  A "round" = what one K step issues:
  // the query slab of step S + RB - 1 FIRST, then the corpus slab of step S + RA - 1 -- loads return in order, so with that order the
  // counted wait below leaves the newest corpus slabs in flight.
  int st_tile = 0, st_kt = 0, st_slot = 0;  // corpus slab being staged
  int sb_kt = 0, sb_slot = 0;               // query slab being staged
  set_offsets(r_begin);

  auto stage_b = [&]() {
    if (stages_b) {
      const char* gb = sgpr_ptr(qbase + (size_t)sb_kt * STEP);
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        unsigned char* lb = smem + S_B_RING + sb_slot * S_SLOT_B + (wave + 4 * i) * 1024;
        __builtin_amdgcn_global_load_lds((global_void*)(gb + (size_t)st_off_b[i]), (lds_void*)lb, 16, 0, 0);
        if constexpr (!F32)  // the low plane of the split queries
          __builtin_amdgcn_global_load_lds((global_void*)(gb + plane_bytes + (size_t)st_off_b[i]), (lds_void*)(lb + G::PLANE_B), 16, 0, 0);
      }
    }
    if (++sb_slot == RB) sb_slot = 0;
    if (++sb_kt == steps_per_tile) sb_kt = 0;
  };
  auto stage_a = [&]() {
    const int tile = st_tile < n_tiles ? st_tile : n_tiles - 1;  // past the end: harmless reloads of the last tile
    const int64_t src_row0 = (ABL & 4) ? 0 : r_begin + (int64_t)tile * BMT;
    const char* ga = sgpr_ptr(corpus + (size_t)src_row0 * row_bytes + (size_t)st_kt * STEP);
    unsigned char* la = smem + st_slot * S_SLOT_A + wave * (RW * STEP);
#pragma unroll
    for (int j = 0; j < NPA; ++j)
      __builtin_amdgcn_global_load_lds((global_void*)(ga + (size_t)st_off[j]), (lds_void*)(la + j * 1024), 16, 0, 0);  // (a non-temporal policy here measured 30 % slower)
    if (++st_slot == RA) st_slot = 0;
    if (++st_kt == steps_per_tile) {
      st_kt = 0;
      ++st_tile;
      if (st_tile < n_tiles) set_offsets(r_begin + (int64_t)st_tile * BMT);
    }
  };
  auto stage_next = [&]() {
    stage_b();
    stage_a();
  };
  // this wave's loads of step S have landed.  Issue order per round: NB query loads, then NA corpus loads.  Query slab S is the first thing
  // of round S - RB + 1: behind it come that round's corpus slab and RB - 2 whole rounds; corpus slab S is the last thing of round
  // S - RA + 1, with RA - 2 whole rounds behind it.  Whatever is younger than BOTH may still be in flight.
  auto wait_landed = [&]() {
    constexpr int NB = G::PLANES * NPB;
    if (stages_b)
      wait_vmcnt<((RA - 2) * (NPA + NB) < NPA + (RB - 2) * (NPA + NB)) ? (RA - 2) * (NPA + NB) : NPA + (RB - 2) * (NPA + NB)>();
    else
      wait_vmcnt<((RA - 2) < (RB - 1) ? (RA - 2) : (RB - 1)) * NPA>();
  };

  // ---- register staging (DR > 0): slot u of the register ring holds the pieces of the K steps congruent to u mod DR
  constexpr int DRN = DR > 0 ? DR : 1;
  constexpr int NBR = G::PLANES * NPB;
  f32x4 areg[DRN][NPA];
  f32x4 breg[DRN][NBR];
  auto load_regs = [&](auto u_tag) {  // the pieces of the next un-issued K step -> register slot U
    constexpr int U = decltype(u_tag)::value;
    const int tile = st_tile < n_tiles ? st_tile : n_tiles - 1;  // past the end: harmless reloads of the last tile
    const char* ga = sgpr_ptr(corpus + (size_t)(r_begin + (int64_t)tile * BMT) * row_bytes + (size_t)st_kt * STEP);
#pragma unroll
    for (int j = 0; j < NPA; ++j) areg[U][j] = *(global_f32x4*)(ga + (size_t)st_off[j]);  // (global, not flat: a flat load counts in lgkmcnt too and cannot be waited for by count)
    if (stages_b) {
      const char* gb = sgpr_ptr(qbase + (size_t)st_kt * STEP);
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        breg[U][G::PLANES * i] = *(global_f32x4*)(gb + (size_t)st_off_b[i]);
        if constexpr (!F32) breg[U][G::PLANES * i + 1] = *(global_f32x4*)(gb + plane_bytes + (size_t)st_off_b[i]);
      }
    }
    if (++st_kt == steps_per_tile) {
      st_kt = 0;
      ++st_tile;
      if (st_tile < n_tiles) set_offsets(r_begin + (int64_t)st_tile * BMT);
    }
  };
  auto commit_regs = [&](auto u_tag, int bslot) {  // register slot U -> LDS, where the LDS-DMA of the other variants would have put it
    constexpr int U = decltype(u_tag)::value;
    unsigned char* la = smem + wave * (RW * STEP) + lane * 16;
#pragma unroll
    for (int j = 0; j < NPA; ++j) *reinterpret_cast<f32x4*>(la + j * 1024) = areg[U][j];
    if (stages_b) {
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        unsigned char* lb = smem + S_B_RING + bslot * S_SLOT_B + (wave + 4 * i) * 1024 + lane * 16;
        *reinterpret_cast<f32x4*>(lb) = breg[U][G::PLANES * i];
        if constexpr (!F32) *reinterpret_cast<f32x4*>(lb + G::PLANE_B) = breg[U][G::PLANES * i + 1];
      }
    }
  };

  // ---- fragment addresses: row (lane & 31) of a 32-row block, logical 16-byte slot 2 * g + (lane >> 5)
  const int frag_row = lane & 31;
  const uint32_t frag_x = (uint32_t)(((lane >> 5) ^ ((frag_row >> G::SH) & (LPR - 1))) << 4);
  const uint32_t a_lane = (uint32_t)((wave * RW + frag_row) * STEP);  // + mi * 32 * STEP
  const uint32_t b_lane = (uint32_t)(S_B_RING + frag_row * STEP);    // + ni * 32 * STEP

  // ---- prologue: RA - 1 corpus slabs and RB - 1 query slabs in flight, in the order of the rounds that would have issued them
  //      (register staging: the first DR steps, one per register slot)
  if constexpr (DR > 0) {
    [&]<int... U>(std::integer_sequence<int, U...>) { (load_regs(std::integral_constant<int, U>{}), ...); }
    (std::make_integer_sequence<int, DRN>{});
  } else {
#pragma unroll
    for (int i = 0; i < RA - 1; ++i) {
      if (i >= RA - RB) stage_b();
      stage_a();
    }
  }

  int rd = 0, rd_b = 0;
  for (int tile = 0; tile < n_tiles; ++tile) {
    const int64_t row0 = r_begin + (int64_t)tile * BMT;
    const bool tile_full = row0 + BMT <= r_end;  // wave-uniform: every row of this tile belongs to the row range
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    auto k_step = [&](auto u_tag) {  // one K step; U = register slot (register staging only)
      if constexpr (DR > 0) {
        // this step's pieces out of the registers (the compiler waits for exactly these loads: whatever was issued after them -- the next
        // DR - 1 steps -- stays in flight), the registers refilled with the step DR ahead; the query slot written here was last read two steps
        // ago, with a barrier in between
        commit_regs(u_tag, rd_b);
        load_regs(u_tag);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TAVB_BARRIER();  // the query pieces of step S are visible
      } else {
        if constexpr ((ABL & 2) == 0) wait_landed();  // this wave's share of step S is in LDS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TAVB_BARRIER();  // the query pieces of step S are visible; everybody is done with the slot of step S - 1
        if constexpr ((ABL & 2) == 0) stage_next();  // query slab S + RB - 1, corpus slab S + RA - 1 -> the slots of step S - 1
      }
      const unsigned char* abase = smem + rd * S_SLOT_A;
      const unsigned char* bbase = smem + rd_b * S_SLOT_B;
#pragma unroll
      for (int gh = 0; gh < NG / 2; ++gh) {  // two 32-byte k slices at a time
        f32x4 af[2][MI], bf[2][NI], bl[2][NI];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const uint32_t kx = (uint32_t)((gh * 2 + g) << 5) ^ frag_x;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            bf[g][ni] = *reinterpret_cast<const f32x4*>(bbase + (b_lane + kx) + ni * 32 * STEP);
            if constexpr (!F32) bl[g][ni] = *reinterpret_cast<const f32x4*>(bbase + (b_lane + kx) + ni * 32 * STEP + G::PLANE_B);
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) af[g][mi] = *reinterpret_cast<const f32x4*>(abase + (a_lane + kx) + mi * 32 * STEP);
        }
        if constexpr ((ABL & 1) == 0) {
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            if constexpr (F32) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                  for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mi][e], bf[g][ni][e], acc[mi][ni], 0, 0, 0);
            } else {
#pragma unroll
              for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[g][mi]), __builtin_bit_cast(f16x8, bl[g][ni]),
                                                                       acc[mi][ni], 0, 0, 0);
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[g][mi]), __builtin_bit_cast(f16x8, bf[g][ni]),
                                                                       acc[mi][ni], 0, 0, 0);
                }
            }
          }
        } else {
          asm volatile("" ::"v"(af[0][0]), "v"(af[1][MI - 1]), "v"(bf[0][0]), "v"(bf[1][NI - 1]));
          if constexpr (!F32) asm volatile("" ::"v"(bl[0][0]), "v"(bl[1][NI - 1]));
        }
      }
      if (++rd == RA) rd = 0;
      if (++rd_b == RB) rd_b = 0;
    };
    if constexpr (DR > 0) {  // (the launcher checks steps_per_tile % DR == 0: a tile starts on register slot 0)
#pragma unroll 1
      for (int kt = 0; kt < steps_per_tile; kt += DR)
        [&]<int... U>(std::integer_sequence<int, U...>) { (k_step(std::integral_constant<int, U>{}), ...); }
      (std::make_integer_sequence<int, DRN>{});
    } else {
#pragma unroll 1
      for (int kt = 0; kt < steps_per_tile; ++kt) k_step(std::integral_constant<int, 0>{});
    }

    // ---- epilogue: admission test on the raw dot products, append (as in the 256-query tile)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int ql = ni * 32 + (lane & 31);
      const float thr = thr_lds[ql];
      const float thr_pre = fmaf(thr, 2.0f, -1.0f) - 4.8e-7f;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        float top = acc[mi][ni][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) top = __builtin_fmaxf(top, acc[mi][ni][r]);
        const bool any = (ABL == 0) && (top > thr_pre);
        if constexpr (ABL != 0) asm volatile("" ::"v"(acc[mi][ni]));
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
          // (the admission path of the 256-query tile: wave masks in scalar registers, four rows at a time, one LDS atomic per admitted row)
          const int64_t row_base = row0 + wave * RW + mi * 32 + 4 * (lane >> 5);
          const int64_t left64 = r_end - row_base;
          const int rows_left = tile_full ? 64 : (int)(left64 < 64 ? left64 : 64);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float sc[4];
            u64 m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              sc[j] = fmaf(acc[mi][ni][4 * g + j], 0.5f, 0.5f);
              asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m[j]) : "v"(sc[j]), "v"(thr));
            }
            if ((m[0] | m[1] | m[2] | m[3]) == 0ull) continue;  // wave-uniform
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r_off = j + 8 * g;
              if (m[j] == 0ull) continue;
              if (((m[j] >> lane) & 1ull) != 0ull && r_off < rows_left) {
                const int pos = lds_add_rtn(&cnt_lds[ql], 1);
                if (pos + 1 > CAP - BMT) lds_store_i32(need_compact, 1);  // this buffer could overflow on the next tile
                float s1 = (sc[j] > 0.0f) ? sc[j] : 0.0f;
                s1 = (s1 > 1.0f) ? 1.0f : s1;
                if (pos < CAP) {
                  const u64 key = make_key(s1, (uint32_t)(row_base + r_off) + p.index_base);
                  // issued behind the compiler's back: a store it can see among the pending staging loads makes its wait-count pass drain the
                  // whole queue at the next loop header (on gfx9 loads and stores share vmcnt and are not ordered against each other) -- once
                  // per tile in the LDS-DMA variants, at every K-loop iteration with register staging, whose waits the compiler counts.  An
                  // extra entry in the queue only makes a counted wait wait longer.
                  u64* dst = my_cand + (size_t)ql * CAP + pos;
                  asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(key) : "memory");
                }
              }
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TAVB_BARRIER();
    if (*need_compact != 0) {  // workgroup-uniform: read after the barrier
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), through the builtin: the compiler's wait-count pass sees the queue empty from here on
      TAVB_BARRIER();
      for (int q = wave; q < SQ; q += S_THREADS / 64) {
        const int n = cnt_lds[q];
        if (n > CAP - BMT) {
          u64* buf = my_cand + (size_t)q * CAP;
          float kth_score;
          const int kept = compact_to_kth<CAP>(buf, n < CAP ? n : CAP, p.k, lane, &kth_score);
          if (lane == 0) {
            cnt_lds[q] = kept;
            if (kept >= p.k && kth_score > thr_lds[q]) thr_lds[q] = kth_score;
          }
          __builtin_amdgcn_s_waitcnt(0x0F70);
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);  // (whatever the compaction left pending: the loop headers see staging loads only)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TAVB_BARRIER();
      if (tid == 0) *need_compact = 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TAVB_BARRIER();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead LDS-DMA before the block retires
  __syncthreads();

  for (int q = wave; q < SQ; q += S_THREADS / 64) {
    const int qg = qtile * SQ + q;
    if (qg >= p.nq) continue;
    const int n = cnt_lds[q];
    const WaveTopK<1> best = best_of_buffer(my_cand + (size_t)q * CAP, n < CAP ? n : CAP, lane);
    u64* out = p.lists + ((size_t)qg * p.list_stride + split) * (size_t)p.k;
    if (lane < p.k) out[lane] = best.key[0];
  }
}

// ---------------------------------------------------------------------------------------------
// BAND of ONE query over the (unsorted) candidate buffers that the workgroups of all row ranges left behind, plus the band
// carried over from the earlier ladder phases -- and, from it, the admission threshold of the next phase.
// The band = every key whose score is within band[q] (= 2 delta_q, tavb_rescore.hip) of the query's k-th best score: exactly
// the rows that can still be in the exact top k once the candidates are rescored with the fp32 query.  Its size is whatever the
// data makes it (k + a few on isotropic data, a whole cluster of near-duplicates on clustered data), up to kc_max.  Where a
// band did not fit on the way (a candidate buffer in the tile kernel, the cache here) the keys were cut to the strict best k and
// lost[q] holds the highest score level at which rows were dropped; with `verdict` (the last phase) the query is declared
// incomplete -- verdict[q] = 1, the caller re-runs it on the exact tile -- when the final band does not fit kc_max or reaches
// down to that level.
// One workgroup per query, so the selection work of a launch is spread over 1024 workgroups x 256 threads instead of
// being the serial tail of 256 workgroups.
//   * the keys of the query (a few hundred after a selective phase; every row of the phase after the cold first one)
//     stream ONCE through an LDS cache of SEL_CACHE keys.  Whenever the cache is nearly full it is cut down to its band,
//     and that cut -- a valid lower bound on the final one -- filters the keys that follow (expected survivors on data in
//     random order: k * remaining / seen), so the exact selection always runs on a few thousand keys held in registers.
//   * exact selection of the k-th best = bisection on the bit pattern of the score (scores are in [0, 1]: the patterns order
//     like the floats), block-wide counts per bit, only on the bits in which the keys differ; in the strict (overflow) form
//     ties at the k-th best score are cut the same way on the ordinal half (smaller ordinal wins): exactly min(k, total) keys.
//   * output: the band's keys, UNSORTED, + their count; thr = just below the band cut (or the caller's floor).
// ---------------------------------------------------------------------------------------------
constexpr int SEL_CACHE = 4096;  // keys of a query held in LDS (32 KiB: four workgroups per CU, so the 1024 queries of a batch are all resident at
                                 // once -- the streaming is latency-bound; 8192 keys / two workgroups per CU: 0.53 ms per cfg3 batch instead of 0.29)
constexpr int SEL_PER = SEL_CACHE / 256;

__global__ void __launch_bounds__(256) select_band_kernel(const u64* __restrict__ cand, const int* __restrict__ counts, int n_splits, int nq_padded, int k,
                                                          int kc_max, const u64* __restrict__ carried, const int* __restrict__ carried_cnt,
                                                          const float* __restrict__ floor, const float* __restrict__ band, u64* __restrict__ out,
                                                          int* __restrict__ out_cnt, float* __restrict__ thr_out, unsigned* __restrict__ lost,
                                                          int* __restrict__ verdict, const int* __restrict__ active, int active_min, int active_max,
                                                          const int* __restrict__ gate, int gate_max, int* __restrict__ doomed, int doom_limit) {
  if (gate != nullptr && *gate > gate_max) return;  // the tile launch in front of this one was skipped too (see MfmaDeviceParams::gate)
  if (active != nullptr) {  // fixed-shape launch over a device-side work list: slots past it (or a list that is not this fallback's share) have no buffers
    const int live = *active;
    if (live <= active_min || live > active_max || (int)blockIdx.x >= live) return;
  }
  extern __shared__ __align__(16) unsigned char sel_smem[];
  u64* cache = reinterpret_cast<u64*>(sel_smem);  // [SEL_CACHE]
  __shared__ int off[260];  // exclusive prefix of the per-split counts (+ the carried band as one more "split")
  __shared__ float red[4];
  __shared__ int n_picked, n_cached;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int q = blockIdx.x;
  const int n_src = n_splits + (carried != nullptr ? 1 : 0);  // <= 257
  const float band_q = band ? band[q] : 0.0f;

  auto block_sum = [&](int v) -> int {  // exact: counts stay far below 2^24
    const float w = wave_sum((float)v);
    if (lane == 0) red[wave] = w;
    __syncthreads();
    const int total = (int)(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    return total;
  };
  // ---- flat order of the query's keys: source s holds count(s) keys at flat positions off[s] .. off[s+1)
  __shared__ int cnt_of[260];
  for (int sp = tid; sp < n_src; sp += 256)  // (n_splits <= 256: one load per thread, all in flight at once)
    cnt_of[sp] = (sp < n_splits) ? counts[(size_t)sp * nq_padded + q] : carried_cnt[q];  // buffers are laid out [row range][padded query] whatever the tile width
  if (tid == 0) n_picked = 0;
  __syncthreads();
  for (int sp = tid; sp <= n_src; sp += 256) {
    int run = 0;
    for (int j = 0; j < sp; ++j) run += cnt_of[j];
    off[sp] = run;
  }
  __syncthreads();
  const int total = off[n_src];
  // the source of flat position `flat` = the last one with off[s] <= flat.  A thread's positions only grow, so on a long stream (the all-admitted
  // first ladder phase: thousands of keys per source) it walks forward from the source of its previous key -- one LDS read per key instead of a
  // binary search's seven or eight dependent ones, which is what that selection's time was made of; a short stream (a few keys per source, one
  // round) keeps the binary search.
  const bool walk = total > 2 * SEL_CACHE;
  int src_at = 0;
  auto key_at = [&](int flat) -> u64 {
    int lo = src_at;
    if (walk) {
      while (lo + 1 < n_src && off[lo + 1] <= flat) ++lo;
      src_at = lo;
    } else {
      int hi = n_src - 1;
      lo = 0;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= flat) lo = mid; else hi = mid - 1;
      }
    }
    const int i = flat - off[lo];
    if (lo == n_splits) return carried[(size_t)q * kc_max + i];
    return cand[((size_t)lo * nq_padded + q) * (size_t)CAPW + i];
  };

  u64 key[SEL_PER];
  int n_keys = 0;  // keys in the cache (block-uniform)
  auto count = [&](auto&& pred) {
    int c = 0;
#pragma unroll
    for (int j = 0; j < SEL_PER; ++j) c += (key[j] != 0ull && pred(key[j])) ? 1 : 0;
    return block_sum(c);
  };
  // loads cache[0 .. n) into key[] and finds the exact (t_hi, t_lo): the need-th best key among them is the smallest key with score > t_hi or
  // (score == t_hi and low >= t_lo) [t_lo only when `strict`]; returns false when n < need (everything is wanted)
  auto kth_of_cache = [&](int n, int need, bool strict, uint32_t* t_hi_out, uint32_t* t_lo_out) -> bool {
#pragma unroll
    for (int j = 0; j < SEL_PER; ++j) {
      const int i = tid + 256 * j;
      key[j] = (i < n) ? cache[i] : 0ull;
    }
    if (n < need) return false;
    uint32_t mx = 0u, mn_inv = 0u;
#pragma unroll
    for (int j = 0; j < SEL_PER; ++j) {
      const uint32_t sc = (uint32_t)(key[j] >> 32);
      mx = max(mx, sc);
      if (key[j] != 0ull) mn_inv = max(mn_inv, ~sc);
    }
    mx = wave_max_u32(mx, lane);
    mn_inv = wave_max_u32(mn_inv, lane);
    if (lane == 0) red[wave] = __uint_as_float(mx);
    __syncthreads();
    mx = max(max(__float_as_uint(red[0]), __float_as_uint(red[1])), max(__float_as_uint(red[2]), __float_as_uint(red[3])));
    __syncthreads();
    if (lane == 0) red[wave] = __uint_as_float(mn_inv);
    __syncthreads();
    const uint32_t mn = ~max(max(__float_as_uint(red[0]), __float_as_uint(red[1])), max(__float_as_uint(red[2]), __float_as_uint(red[3])));
    __syncthreads();
    uint32_t t = mn;
    if (mx != mn) {
      const int top = 31 - __builtin_clz(mx ^ mn);
      t = (top == 31) ? 0u : (mx & ~((2u << top) - 1u));
      for (int b = top; b >= 0; --b) {
        const uint32_t trial = t | (1u << b);
        if (count([&](u64 kk) { return (uint32_t)(kk >> 32) >= trial; }) >= need) t = trial;
      }
    }
    uint32_t t_lo = 0u;
    if (strict) {
      const int above = count([&](u64 kk) { return (uint32_t)(kk >> 32) > t; });
      const int ties = count([&](u64 kk) { return (uint32_t)(kk >> 32) == t; });
      if (above + ties > need) {  // cut the ties by ordinal (low word: bigger = smaller ordinal)
        const int need_ties = need - above;
        for (int b = 31; b >= 0; --b) {
          const uint32_t trial = t_lo | (1u << b);
          if (count([&](u64 kk) { return (uint32_t)(kk >> 32) == t && (uint32_t)kk >= trial; }) >= need_ties) t_lo = trial;
        }
      }
    }
    *t_hi_out = t;
    *t_lo_out = t_lo;
    return true;
  };
  // the cut that goes with the k-th best of the n cached keys (which kth_of_cache leaves in key[]): the band below it, or -- when the
  // band holds more than `room` keys -- the strict best k (*was_strict; the level below which keys were dropped goes to lost_here);
  // (f_hi, f_lo): keep a key iff score > f_hi or (score == f_hi and low >= f_lo)
  uint32_t lost_here = 0u;
  auto cut_of_cache = [&](int n, int room, uint32_t* f_hi, uint32_t* f_lo, bool* was_strict) -> bool {
    uint32_t t_hi = 0u, t_lo = 0u;
    *was_strict = false;
    if (!kth_of_cache(n, k, false, &t_hi, &t_lo)) return false;
    const uint32_t cut = band_cut_bits(t_hi, band_q);
    if (count([&](u64 kk) { return (uint32_t)(kk >> 32) >= cut; }) <= room) {
      *f_hi = cut;
      *f_lo = 0u;
      return true;
    }
    *was_strict = true;
    kth_of_cache(n, k, true, &t_hi, &t_lo);
    lost_here = max(lost_here, t_hi > 0u ? t_hi : 1u);
    *f_hi = t_hi;
    *f_lo = t_lo;
    return true;
  };

  // ---- stream the keys through the cache: whenever it is nearly full, it is cut down to its band, and that cut -- a valid
  //      lower bound on the final one -- filters what comes next.  On data in random order the first cut is the only one
  //      (the filter then passes k * remaining / seen keys); adversarial orders just cut more often.
  uint32_t f_hi = 0u, f_lo = 0u;
  bool have_filter = false;
  if (tid == 0) n_cached = 0;
  __syncthreads();
  constexpr int UNR = 8;  // keys per thread per round: their loads are all in flight together (the loop is latency-bound otherwise)
  for (int base = 0; base < total; base += 256 * UNR) {
    if (n_cached > SEL_CACHE - 256 * UNR) {  // block-uniform (read after a barrier)
      const int n = n_cached;
      __syncthreads();
      bool mid_strict;
      // (room for the band a cut keeps: a quarter of the cache, or the whole band buffer when that is wider -- what is left of the cache still
      //  takes the 2048 keys of the next round)
      have_filter = cut_of_cache(n, kc_max > SEL_CACHE / 4 ? kc_max : SEL_CACHE / 4, &f_hi, &f_lo, &mid_strict);  // n >= k here
      if (tid == 0) n_cached = 0;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < SEL_PER; ++j) {  // keep the band (the keys sit in this thread's registers)
        const u64 kk = key[j];
        const uint32_t hi = (uint32_t)(kk >> 32);
        if (kk != 0ull && (hi > f_hi || (hi == f_hi && (uint32_t)kk >= f_lo))) cache[atomicAdd(&n_cached, 1)] = kk;
      }
      __syncthreads();
    }
    u64 kk[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int i = base + u * 256 + tid;
      kk[u] = (i < total) ? key_at(i) : 0ull;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const uint32_t hi = (uint32_t)(kk[u] >> 32);
      const bool keep = kk[u] != 0ull && (!have_filter || hi > f_hi || (hi == f_hi && (uint32_t)kk[u] >= f_lo));
      const u64 m = __builtin_amdgcn_ballot_w64(keep);
      int wbase = 0;
      if (lane == 0 && m != 0ull) wbase = atomicAdd(&n_cached, __popcll(m));
      wbase = __builtin_amdgcn_readfirstlane(wbase);
      if (keep) cache[wbase + __popcll(m & ((1ull << lane) - 1ull))] = kk[u];
    }
    __syncthreads();
  }
  n_keys = n_cached;
  __syncthreads();
  uint32_t t_hi = 0u, t_lo = 0u;
  bool strict = false;
  const bool enough = cut_of_cache(n_keys, kc_max, &t_hi, &t_lo, &strict);  // (fewer than k keys: all of them are the band; key[] is loaded either way)
  // ---- write the band (unsorted) and its size
#pragma unroll
  for (int j = 0; j < SEL_PER; ++j) {
    const u64 kv = key[j];
    const uint32_t hi = (uint32_t)(kv >> 32);
    const bool take = kv != 0ull && (!enough || hi > t_hi || (hi == t_hi && (uint32_t)kv >= t_lo));
    if (take) {
      const int idx = atomicAdd(&n_picked, 1);
      if (idx < kc_max) out[(size_t)q * kc_max + idx] = kv;
    }
  }
  __syncthreads();
  if (tid == 0) {
    out_cnt[q] = n_picked < kc_max ? n_picked : kc_max;
    // a band over the rows seen so far that is already this full will not fit at the end (the caller extrapolates: doom_limit): counted, and the
    // launches of the last filter phase gate themselves on the count
    if (doomed != nullptr && (n_picked > doom_limit || strict)) atomicAdd(doomed, 1);  // (strict: it does not even fit now)
    uint32_t lost_all = lost_here;
    if (lost != nullptr) {
      if (lost_here != 0u) atomicMax(&lost[q], lost_here);
      lost_all = max(lost_all, lost[q]);  // (the tile kernels of this and the earlier phases are done: plain read)
    }
    // incomplete: the final band itself did not fit, or rows were dropped somewhere at a level the final band reaches (its cut is t_hi when
    // there are k keys, 0 -- everything counts -- when there are fewer)
    if (verdict != nullptr) verdict[q] = (strict || (lost_all != 0u && lost_all >= (enough ? t_hi : 0u))) ? 1 : 0;
    if (thr_out != nullptr) {
      float t = -__builtin_inff();
      if (enough && t_hi > 0u) t = __uint_as_float(t_hi - (strict ? 0u : 1u));  // strict: later rows tie-lose (score > t); band: score >= cut
      if (floor != nullptr && floor[q] > t) t = floor[q];
      thr_out[q] = t;
    }
  }
}

}  // namespace

// thr[q] = the largest float below the k-th best score of the sample pass (so that `score > thr` admits
// every row scoring >= that k-th best), or -inf when the sample did not yield k hits.
// `floor` (optional, [nq]): per-query thresholds that hold from the start (the relaxed min_score of the filter pass).
__global__ void sample_threshold_kernel(const u64* __restrict__ keys, int nq, int k, const float* __restrict__ floor, float* __restrict__ thr) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const u64 kth = keys[(size_t)q * k + (k - 1)];
  float t = -__builtin_inff();
  if (kth != 0ull) {
    const uint32_t bits = (uint32_t)(kth >> 32);
    t = bits ? __uint_as_float(bits - 1u) : -__builtin_inff();
  }
  if (floor != nullptr && floor[q] > t) t = floor[q];
  thr[q] = t;
}

hipError_t launch_sample_thresholds(const unsigned long long* keys, int nq, int k, const float* floor, float* thr, hipStream_t stream) {
  hipLaunchKernelGGL(sample_threshold_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, keys, nq, k, floor, thr);
  return hipGetLastError();
}

// 128-query tiles when they leave less padding than 256-query tiles (up to 128 queries: one HBM-bound pass instead of a
// half-empty 256-query tile; 257 .. 384 and 513 .. 640: measured 6 % and 3.5 % faster, profiles/r02_mid_batch.md)
int mfma_query_tile(int nq) {
  const int n128 = (nq + 127) / 128;
  return ((n128 & 1) && n128 <= 5) ? 128 : BN;
}

// ... and 128-query tiles on a corpus so small that 256-query tiles would give a CU fewer than four tiles to walk (query tiles x corpus
// tiles <= 4 n_cu): twice the workgroups or half the tile, half the all-admitted epilogue of a workgroup's first tile -- 256 queries over
// 1000 / 5000 / 20000 / 50000 fp16 rows 0.131 / 0.157 / 0.181 / 0.238 -> 0.090 / 0.116 / 0.141 / 0.197 ms (257 queries took three 128-query
// tiles all along and were faster than 256), 1024 queries over 30720 / 50000 rows 0.317 / 0.380 -> 0.264 / 0.342 ms, 512 over 120000 rows
// 0.392 -> 0.367; beyond that the wider tile's operand reuse wins (1024 queries over 250k rows: 0.910 against 0.927 ms).  tools/regime_sweep.py,
// profiles/r06_regime_sweep.md
int mfma_query_tile_for(int nq, int64_t rows, int n_cu) {
  const int qt = mfma_query_tile(nq);
  if (qt == 128) return qt;
  const int64_t corpus_tiles = (rows + BM6 - 1) / BM6;
  const int64_t wgs = (int64_t)((nq + BN - 1) / BN) * corpus_tiles;
  return wgs <= 4 * (int64_t)n_cu ? 128 : qt;
}

// k: the band selection holds any k the fused selections serve (a band of k + its 2-delta neighbourhood has to fit the 640 keys a candidate
// buffer keeps between compactions and the 1024 of the select kernel: k = 256 leaves the same slack as k = 32 on isotropic data)
bool mfma_supported(int dim, int k) { return dim % BK == 0 && dim >= BK && dim <= 16384 && k >= 1 && k <= TAVB_MAX_FUSED_K; }

int mfma_pick_splits(int64_t rows, int nq_padded, int tile, int n_cu) {
  // One workgroup per CU and all of them resident at once: the grid is (groups of 8 row ranges) x query tiles x 8, so the
  // number of row ranges is a multiple of 8 with groups * n_qtiles * 8 <= n_cu.  (85 ranges for 3 query tiles made 264
  // workgroups on 256 CUs: a second scheduling round for the last 8, and a 768-query batch slower than a 1024-query one.)
  const int n_qtiles = nq_padded / tile > 0 ? nq_padded / tile : 1;
  int splits = (n_cu / (8 * n_qtiles)) * 8;
  if (splits < 8) splits = 8;
  const int64_t tiles = (rows + BM - 1) / BM;
  if (splits > tiles) splits = (int)tiles;
  return splits;
}

// candidate buffers of a launch: nq_padded = tiles x queries per tile; the 256-query tile has the deeper buffers
size_t mfma_workspace_bytes(int n_splits, int nq_padded, bool wide) {
  return (size_t)n_splits * (size_t)nq_padded * (wide ? CAPW : CAP) * sizeof(u64);
}

hipError_t launch_select_band(const unsigned long long* cand, const int* counts, int n_splits, int nq, int nq_padded, int k, int kc_max,
                              const unsigned long long* carried, const int* carried_cnt, const float* floor, const float* band, unsigned long long* out,
                              int* out_cnt, float* thr_out, unsigned* lost, int* verdict, hipStream_t stream, const int* active, int active_min,
                              int active_max, const int* gate, int gate_max, int* doomed, int doom_limit) {
  if (nq < 1 || k < 1 || k > TAVB_MAX_FUSED_K || n_splits < 1 || n_splits > 256 || nq_padded < nq || kc_max < k || kc_max > SEL_CACHE / 2 || kc_max > kBandMax) return hipErrorInvalidValue;
  constexpr int lds = SEL_CACHE * (int)sizeof(u64);
  hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(select_band_kernel), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(select_band_kernel, dim3(nq), dim3(256), lds, stream, cand, counts, n_splits, nq_padded, k, kc_max, carried, carried_cnt, floor, band,
                     out, out_cnt, thr_out, lost, verdict, active, active_min, active_max, gate, gate_max, doomed, doom_limit);
  return hipGetLastError();
}

hipError_t launch_mfma_scan(const MfmaParams& p, hipStream_t stream) {
  const int tile = p.wide_tile == 128 ? 128 : BN;
  if (!mfma_supported(p.dim, p.k) || p.nq_padded % tile != 0 || p.n_splits < 1) return hipErrorInvalidValue;
  MfmaDeviceParams d{};
  d.corpus = reinterpret_cast<const _Float16*>(p.corpus);
  d.queries = reinterpret_cast<const _Float16*>(p.queries);
  d.lists = p.lists;
  d.rows = p.rows;
  d.dim = p.dim;
  d.nq = p.nq;
  d.n_qtiles = p.nq_padded / tile;
  d.n_splits = p.n_splits;
  d.list_stride = p.list_stride > p.n_splits ? p.list_stride : p.n_splits;
  d.k = p.k;
  d.index_base = p.index_base;
  d.min_score = p.min_score;
  d.thr_in = p.thr_in;
  d.band = p.band;
  d.lost = p.lost;
  d.active = p.active;
  d.active_min = p.active_min;
  d.active_max = p.active_max > 0 ? p.active_max : 0x7fffffff;
  d.split_plane = p.split_plane;
  d.gate = p.gate;
  d.gate_max = p.gate_max;
  const int64_t per = (p.rows + p.n_splits - 1) / p.n_splits;
  const int bm = BM6;
  d.rows_per_split = ((per + bm - 1) / bm) * bm;
  if (!p.workspace || !p.counts) return hipErrorInvalidValue;
  d.cand = p.workspace;
  d.counts = p.counts;
  // grid: groups of 8 consecutive block ids = 8 different row ranges (one per XCD)
  const int groups = (p.n_splits + 7) / 8;
  const int grid = groups * d.n_qtiles * 8;
  auto go = [&](auto kern, int threads, int lds) -> hipError_t {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, d);
    return hipGetLastError();
  };
  // `ablate` modes exist to time parts of a kernel (results are garbage): see profiles/r02_cfg3_ablation.md
  {
    if (p.dim % 64 != 0) return hipErrorInvalidValue;
    constexpr int LDS256 = WideGeom<4>::LDS, LDS128 = WideGeom<2>::LDS;
    if (p.split_plane > 0) {
      if (tile != BN) return hipErrorInvalidValue;
      return go(mfma_scan_kernel<0, 4, 8, 6, 4, true>, NT6, LDS256);
    }
    if (tile == 128) {
      switch (p.ablate) {  // (measurement: what bounds the 128-query width, profiles/r05_mid_batch.md)
        case 1: return go(mfma_scan_kernel<1, 2, 6, 4, 4>, NT6, LDS128);      // no MFMAs
        case 256: return go(mfma_scan_kernel<256, 2, 6, 4, 4>, NT6, LDS128);  // no admissions
        case 257: return go(mfma_scan_kernel<257, 2, 6, 4, 4>, NT6, LDS128);  // neither: staging, fragment reads, barriers
        case 264: return go(mfma_scan_kernel<264, 2, 6, 4, 4>, NT6, LDS128);  // no admissions, query operand cache resident
        default: return go(mfma_scan_kernel<0, 2, 6, 4, 4>, NT6, LDS128);
      }
    }
    if (p.bdirect && p.ablate == 0) return go(mfma_scan_kernel<0, 4, 4, 3, 3, false, true>, NT6, 3 * SLOT_A6 + BN * 8 + 16);  // queries in fragment-major order (query_prepare_kernel)
    switch (p.ablate) {
      case 256: return go(mfma_scan_kernel<256, 4, 8, 6, 4>, NT6, LDS256);  // everything except admissions
      case 260: return go(mfma_scan_kernel<260, 4, 8, 6, 4>, NT6, LDS256);  // same, corpus tile 0 re-read by every block (L2 resident)
      case 264: return go(mfma_scan_kernel<264, 4, 8, 6, 4>, NT6, LDS256);  // same as 256, query operand K step 0 re-read (cache resident)
      case 268: return go(mfma_scan_kernel<268, 4, 8, 6, 4>, NT6, LDS256);  // both operands cache resident
      case 258: return go(mfma_scan_kernel<258, 4, 8, 6, 4>, NT6, LDS256);  // no LDS-DMA, no admissions
      case 1282: return go(mfma_scan_kernel<1282, 4, 8, 6, 4>, NT6, LDS256);  // the same in the other MFMA issue orders (sched 3 / 4 / 5)
      case 2306: return go(mfma_scan_kernel<2306, 4, 8, 6, 4>, NT6, LDS256);
      case 3330: return go(mfma_scan_kernel<3330, 4, 8, 6, 4>, NT6, LDS256);
      default: break;
    }
    switch (p.sched) {  // staging pieces per quarter (q3, q0, q1), MFMA issue order: measurement
      case 1: return go(mfma_scan_kernel<0, 4, 10, 8, 0>, NT6, LDS256);
      case 2: return go(mfma_scan_kernel<0, 4, 6, 6, 6>, NT6, LDS256);
      case 3: return go(mfma_scan_kernel<1024, 4, 8, 6, 4>, NT6, LDS256);  // query fragment outermost
      case 4: return go(mfma_scan_kernel<2048, 4, 8, 6, 4>, NT6, LDS256);  // corpus fragment outermost, boustrophedon
      case 5: return go(mfma_scan_kernel<3072, 4, 8, 6, 4>, NT6, LDS256);  // query fragment outermost, boustrophedon
      default: return go(mfma_scan_kernel<0, 4, 8, 6, 4>, NT6, LDS256);
    }
  }
  return hipErrorInvalidValue;
}

int skinny_query_tile(int nq) { return nq > SQ32 ? 2 * SQ32 : SQ32; }  // 64-query tiles for batches of 33 and more

bool skinny_supported(int dim, int k, bool f32) {
  return (dim * (f32 ? 4 : 2)) % 64 == 0 && dim > 0 && k >= 1 && k <= 64;
}

// Staging variant of the 32/64-query tile (whole-line steps only): 0 = LDS-DMA ring (ships), 1 = deep corpus ring (4 + 3 slots), 4 = register
// staging four K steps deep, 2 = half tiles (128 rows, two or three workgroups per CU).  The others are measurement variants behind option mfma_sched (8 / 5 / 6; 7 = ring, explicitly): with the per-tile
// drains gone (see the kernel) the ring, the deep ring and register staging 3 / 4 / 6 steps deep all stream 10M x 1536 fp16 rows under 32 queries
// at 6.4 .. 6.5 TB/s -- 40 .. 240 KiB in flight per CU make no difference, the tile is not short of bytes in flight (profiles/r05_mid_batch.md).
constexpr int kSkinnyVariantDefault[2] = {0, 0};  // {fp16, fp32}
// K step of the tile: whole 128-byte lines whenever a row is a multiple of that
static bool skinny_line_steps(int dim, bool f32) { return (dim * (f32 ? 4 : 2)) % 128 == 0; }

static int skinny_variant(int dim, bool f32, int tile, int sched) {
  if (!skinny_line_steps(dim, f32) || sched == 9 || sched == 7) return 0;
  int v = sched == 8 ? 1 : sched == 5 ? 4 : sched == 6 ? 2 : kSkinnyVariantDefault[f32 ? 1 : 0];
  if (tile != 32 && v != 0) v = 0;  // (the 64-query tile keeps the ring)
  const int steps = dim * (f32 ? 4 : 2) / 128;
  if (v >= 3 && steps % v != 0) v = 0;  // a tile starts on register slot 0
  return v;
}

static int skinny_wg_per_cu(int dim, bool f32, int tile, int sched) {
  const bool line = skinny_line_steps(dim, f32);
  const int v = skinny_variant(dim, f32, tile, sched);
  if (v == 1 || v >= 4) return 1;
  if (v == 2) return f32 ? 3 : 2;  // half tiles: 72 KiB (fp16) / 40 KiB (fp32) of LDS per workgroup
  if (f32) {
    if (tile == 64) return line ? SkinnyGeom<float, 2, 128>::WG_PER_CU : SkinnyGeom<float, 2, 64>::WG_PER_CU;
    return line ? SkinnyGeom<float, 1, 128>::WG_PER_CU : SkinnyGeom<float, 1, 64>::WG_PER_CU;
  }
  if (tile == 64) return line ? SkinnyGeom<_Float16, 2, 128>::WG_PER_CU : SkinnyGeom<_Float16, 2, 64>::WG_PER_CU;
  return line ? SkinnyGeom<_Float16, 1, 128>::WG_PER_CU : SkinnyGeom<_Float16, 1, 64>::WG_PER_CU;
}

int skinny_pick_splits(int64_t rows, int nq_padded, int tile, int n_cu, int dim, bool f32, int sched) {
  const int n_qtiles = nq_padded / tile;
  int splits = (skinny_wg_per_cu(dim, f32, tile, sched) * n_cu) / (n_qtiles > 0 ? n_qtiles : 1);  // every workgroup resident at once
  splits = (splits / 8) * 8;                                                               // whole groups of 8 (one row range per XCD)
  if (splits < 8) splits = 8;
  const int64_t tiles = (rows + BM - 1) / BM;
  if (splits > tiles) splits = (int)tiles;
  return splits;
}

// Same contract as launch_mfma_scan, except that the tile writes sorted lists (p.lists) itself.  p.queries: fp32 corpus ->
// [nq_padded, dim] fp32; fp16 corpus -> [2, nq_padded, dim] fp16, the high and the low plane of the split fp32 queries
// (launch_f32_split_f16).  nq_padded is a multiple of the tile (p.skinny_tile = 32 or 64 queries).
hipError_t launch_skinny_scan(const MfmaParams& p, hipStream_t stream) {
  const bool f32 = p.f32 != 0;
  const int tile = p.skinny_tile == 64 ? 64 : 32;
  if (!skinny_supported(p.dim, p.k, f32) || p.nq_padded % tile != 0 || p.n_splits < 1 || !p.workspace) return hipErrorInvalidValue;
  MfmaDeviceParams d{};
  d.corpus = reinterpret_cast<const _Float16*>(p.corpus);
  d.queries = reinterpret_cast<const _Float16*>(p.queries);
  d.lists = p.lists;
  d.rows = p.rows;
  d.dim = p.dim;
  d.nq = p.nq;
  d.n_qtiles = p.nq_padded / tile;
  d.n_splits = p.n_splits;
  d.list_stride = p.list_stride > p.n_splits ? p.list_stride : p.n_splits;
  d.k = p.k;
  d.index_base = p.index_base;
  d.min_score = p.min_score;
  d.thr_in = p.thr_in;
  d.active = p.active;
  d.active_min = p.active_min;
  d.active_max = p.active_max > 0 ? p.active_max : 0x7fffffff;
  d.cand = p.workspace;
  const int64_t per = (p.rows + p.n_splits - 1) / p.n_splits;
  d.rows_per_split = ((per + BM - 1) / BM) * BM;
  const int groups = (p.n_splits + 7) / 8;
  const int grid = groups * d.n_qtiles * 8;
  auto go = [&](auto kern, int lds) -> hipError_t {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S_THREADS), lds, stream, d);
    return hipGetLastError();
  };
  const bool line = skinny_line_steps(p.dim, f32) && p.sched != 9;  // (sched 9: force the 64-byte steps, measurement)
  switch (skinny_variant(p.dim, f32, tile, p.sched)) {
    case 1:
      return f32 ? go(skinny_scan_kernel<float, 1, 128, 0, true>, SkinnyGeom<float, 1, 128, true>::LDS)
                 : go(skinny_scan_kernel<_Float16, 1, 128, 0, true>, SkinnyGeom<_Float16, 1, 128, true>::LDS);
    case 2:
      return f32 ? go(skinny_scan_kernel<float, 1, 128, 0, false, 0, true>, SkinnyGeom<float, 1, 128, false, 0, true>::LDS)
                 : go(skinny_scan_kernel<_Float16, 1, 128, 0, false, 0, true>, SkinnyGeom<_Float16, 1, 128, false, 0, true>::LDS);
    case 4:
      return f32 ? go(skinny_scan_kernel<float, 1, 128, 0, false, 4>, SkinnyGeom<float, 1, 128, false, 4>::LDS)
                 : go(skinny_scan_kernel<_Float16, 1, 128, 0, false, 4>, SkinnyGeom<_Float16, 1, 128, false, 4>::LDS);
    default: break;
  }
  if (f32) {
    if (tile == 64) return line ? go(skinny_scan_kernel<float, 2, 128, 0>, SkinnyGeom<float, 2, 128>::LDS) : go(skinny_scan_kernel<float, 2, 64, 0>, SkinnyGeom<float, 2, 64>::LDS);
    return line ? go(skinny_scan_kernel<float, 1, 128, 0>, SkinnyGeom<float, 1, 128>::LDS) : go(skinny_scan_kernel<float, 1, 64, 0>, SkinnyGeom<float, 1, 64>::LDS);
  }
  if (tile == 64)
    return line ? go(skinny_scan_kernel<_Float16, 2, 128, 0>, SkinnyGeom<_Float16, 2, 128>::LDS) : go(skinny_scan_kernel<_Float16, 2, 64, 0>, SkinnyGeom<_Float16, 2, 64>::LDS);
  return line ? go(skinny_scan_kernel<_Float16, 1, 128, 0>, SkinnyGeom<_Float16, 1, 128>::LDS) : go(skinny_scan_kernel<_Float16, 1, 64, 0>, SkinnyGeom<_Float16, 1, 64>::LDS);
}

}  // namespace tavb
