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
// Decomposition
//   * operand roles: A = corpus tile (M = 256 rows), B = query tile (N = 256 queries).
//     With this orientation the MFMA result layout puts ONE query in each lane
//     (col = lane & 31) and 16 corpus rows in its 16 accumulator registers, so the
//     epilogue's "does this score beat the query's current k-th best" test needs one
//     threshold register per lane and one v_cmp per score.
//   * workgroup = 8 waves (2 along rows x 4 along queries), each wave a 128 x 64
//     sub-tile = 4 x 2 MFMA tiles of 32 x 32 (128 accumulator registers).
//   * K loop in steps of BK = 64 halves: both operand slabs (32 KiB each) are staged
//     into LDS with LDS-DMA (global_load_lds, 16 B per lane), double buffered: slab
//     t+1 is in flight while slab t feeds the MFMAs.  The 16-byte slots of each 128-byte
//     LDS row are XOR-swizzled with ((row >> 1) & 7) -- applied to the per-lane global
//     SOURCE address, because LDS-DMA writes lane-linear -- so that the ds_read_b128
//     fragment reads (16 lanes of a group read 16 different rows at one k-slot) hit 16
//     different bank slots instead of two.
//   * a workgroup owns one query tile and one contiguous range of corpus rows and walks
//     that range tile by tile (persistent); the workgroups that share a row range (one
//     per query tile) get block ids congruent mod 8 so they run on the same XCD at the
//     same time and the corpus tile is fetched from HBM once and re-read from that
//     XCD's L2.
//   * selection: per (workgroup, query) a candidate buffer of CAP keys in global memory
//     plus, in LDS, its fill count and the current admission threshold.  A score that
//     beats the threshold is clipped, packed into a key and appended (LDS atomic for the
//     slot).  When a buffer could overflow on the next tile it is compacted to its best k
//     (wave-wide bitonic sort + merge) and the threshold rises to its k-th score; the
//     expected number of compactions per query is O(log(rows / CAP)).  At the end every
//     buffer is compacted and written as a sorted list; tavb_merge merges the lists of
//     the row ranges.
//   * the host scans the corpus in phases of growing size (threshold ladder, tavb_abi.hip): the
//     k-th best score after a phase seeds the admission thresholds of the next (`thr_in`).
//
// Two kernel families live here: the 256-query fp16 tile described above (variants 1-4 of its K
// loop; 3 is the default) and, at the end of the file, a 32-query tile for fp32 and fp16 corpora
// that carries small batches -- and every batch on the reference's fp32 layout -- at HBM speed.

#include <hip/hip_runtime.h>

#include <type_traits>

#include "tavb_device.h"
#include "tavb_internal.h"

namespace tavb {

namespace {

constexpr int BM = 256;   // corpus rows per tile
constexpr int BN = 256;   // queries per tile
constexpr int BK = 64;    // halves per K step (128 bytes per row)
constexpr int NTHREADS = 512;
constexpr int CAP = 512;  // candidate keys per (workgroup, query); must be >= BM + max k
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 64 KiB
constexpr int A_BYTES = BM * BK * 2;             // 32 KiB
constexpr int LDS_BYTES = 2 * STAGE_BYTES + BN * 8;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void global_void;

struct MfmaDeviceParams {
  const _Float16* corpus;
  const _Float16* queries;  // [nq_padded, dim]
  u64* cand;                // [blocks][BN][CAP]
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
  int32_t group_sel;
  int32_t a_tiled;  // corpus given as the K-blocked image of pack_tiled_kernel
  const float* thr_in;  // optional [nq_padded] admission thresholds from a sample pass (exclusive bound)
  int* sync;            // optional [n_splits] tile rendezvous counters (zeroed before the launch), variant 3
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

// ascending bitonic sort of one key per lane (lane 63 ends up with the largest)
template <int SIZE, int STRIDE>
__device__ __forceinline__ u64 sort_stage(u64 key, int lane) {
  const u64 other = xor_lane_u64<STRIDE>(key, lane);
  const bool asc_block = (lane & SIZE) == 0 || SIZE == 64;
  const bool lower = (lane & STRIDE) == 0;
  const bool keep_min = (lower == asc_block);
  const bool mine_small = key < other;
  return (keep_min == mine_small) ? key : other;
}

__device__ __forceinline__ u64 sort64_ascending(u64 k, int lane) {
  k = sort_stage<2, 1>(k, lane);
  k = sort_stage<4, 2>(k, lane);
  k = sort_stage<4, 1>(k, lane);
  k = sort_stage<8, 4>(k, lane);
  k = sort_stage<8, 2>(k, lane);
  k = sort_stage<8, 1>(k, lane);
  k = sort_stage<16, 8>(k, lane);
  k = sort_stage<16, 4>(k, lane);
  k = sort_stage<16, 2>(k, lane);
  k = sort_stage<16, 1>(k, lane);
  k = sort_stage<32, 16>(k, lane);
  k = sort_stage<32, 8>(k, lane);
  k = sort_stage<32, 4>(k, lane);
  k = sort_stage<32, 2>(k, lane);
  k = sort_stage<32, 1>(k, lane);
  k = sort_stage<64, 32>(k, lane);
  k = sort_stage<64, 16>(k, lane);
  k = sort_stage<64, 8>(k, lane);
  k = sort_stage<64, 4>(k, lane);
  k = sort_stage<64, 2>(k, lane);
  k = sort_stage<64, 1>(k, lane);
  return k;
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

// VARIANT 1: every wave alternates {read fragments, 8 MFMAs} in lock step, one barrier per K step.
// VARIANT 2: the two wave groups (rows 0-127 / 128-255 of the tile = waves 0-3 / 4-7, one wave of
//            each group per SIMD) run half a phase apart: while one group issues its 16 MFMAs of a
//            half K step, the other group reads its next fragments from LDS and issues the LDS-DMA
//            for the next slab.  Raw s_barrier (no implied vmcnt drain), counted waits placed by hand.
#define TAVB_SB() __builtin_amdgcn_sched_barrier(0)
#define TAVB_BARRIER()            \
  do {                            \
    TAVB_SB();                    \
    __builtin_amdgcn_s_barrier(); \
    TAVB_SB();                    \
  } while (0)

// ABLATE (measurement only, results are garbage): 1 = no MFMAs, 2 = no LDS-DMA after the first slab,
// 3 = neither (barriers + fragment reads only).
// PRIO: 0 = no s_setprio, 1 = MFMA phase at priority 1, 2 = LOAD phase at priority 1.
template <int VARIANT, int ABLATE, int PRIO>
__global__ void __launch_bounds__(NTHREADS) mfma_scan_kernel(const MfmaDeviceParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* thr_lds = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES);      // [BN] admission threshold (exclusive)
  int* cnt_lds = reinterpret_cast<int*>(smem + 2 * STAGE_BYTES + BN * 4);  // [BN] buffer fill

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;  // 0..1 : which 128 rows of the tile
  const int wn = wave & 3;   // 0..3 : which 64 queries of the tile

  // block -> (row range, query tile); ranges sharing rows are congruent mod 8 (same XCD)
  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int t = b >> 3;
  const int qtile = t % p.n_qtiles;
  const int split = (t / p.n_qtiles) * 8 + xcd;
  if (split >= p.n_splits) return;
  const int64_t r_begin = (int64_t)split * p.rows_per_split;
  const int64_t r_end = (r_begin + p.rows_per_split < p.rows) ? r_begin + p.rows_per_split : p.rows;
  const int logical_block = split * p.n_qtiles + qtile;
  u64* my_cand = p.cand + (size_t)logical_block * BN * CAP;

  // admission is `score > thr`: start just below min_score (or at -inf when everything qualifies)
  const float thr0 = (p.min_score > 0.0f) ? __uint_as_float(__float_as_uint(p.min_score) - 1u) : -__builtin_inff();
  for (int i = tid; i < BN; i += NTHREADS) {
    float t0 = (p.min_score != p.min_score) ? __builtin_inff() : thr0;  // NaN threshold admits nothing
    const int qg0 = qtile * BN + i;
    if (qg0 >= p.nq) t0 = __builtin_inff();  // padding queries admit nothing
    else if (p.thr_in && p.thr_in[qg0] > t0) t0 = p.thr_in[qg0];  // k-th best of the sample pass: a valid lower bound
    thr_lds[i] = t0;
    cnt_lds[i] = 0;
  }

  const int D = p.dim;
  const int n_ksteps = D / BK;
  const size_t row_bytes = (size_t)D * 2;
  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const char* qbase = reinterpret_cast<const char*>(p.queries) + (size_t)qtile * BN * row_bytes;

  // --- staging: each wave issues 4 LDS-DMA instructions per operand per K step; instruction i
  //     covers tile rows 8i .. 8i+7 (8 lanes x 16 B per 128-byte row).  Addresses are a wave-uniform
  //     base (SGPRs) plus a 32-bit per-lane offset that is loop invariant.
  const int st_row_in_inst = lane >> 3;
  const int st_slot = lane & 7;
  uint32_t st_off_b[4];  // query operand: constant for the whole kernel
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + st_row_in_inst;
    st_off_b[j] = (uint32_t)row * (uint32_t)row_bytes + (uint32_t)((st_slot ^ ((row >> 1) & 7)) * 16);
  }
  uint32_t st_off_a[4];  // corpus operand: per tile (rows past the corpus end are clamped)

  auto set_tile_offsets = [&](int64_t row0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (wave * 4 + j) * 8 + st_row_in_inst;
      int64_t grow = row0 + row;
      if (grow >= p.rows) grow = p.rows - 1;  // stay in bounds; masked in the epilogue
      st_off_a[j] = (uint32_t)(grow - row0) * (uint32_t)row_bytes + (uint32_t)((st_slot ^ ((row >> 1) & 7)) * 16);
    }
  };

  auto stage = [&](int buf, int64_t row0, int kt) {
    unsigned char* abase = smem + buf * STAGE_BYTES + wave * 4096;
    unsigned char* bbase = abase + A_BYTES;
    const char* ga = sgpr_ptr(corpus + (size_t)row0 * row_bytes + (size_t)kt * (BK * 2));  // wave-uniform
    const char* gb = sgpr_ptr(qbase + (size_t)kt * (BK * 2));                               // wave-uniform
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((global_void*)(ga + (size_t)st_off_a[j]), (lds_void*)(abase + j * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((global_void*)(gb + (size_t)st_off_b[j]), (lds_void*)(bbase + j * 1024), 16, 0, 0);
    }
  };

  // fragment reads: lane l reads row (l & 31) of a 32-row block at logical 16-byte slot
  // 2*k16 + (l >> 5); the physical slot is that XOR ((row >> 1) & 7), and because the block bases
  // are multiples of 16 rows the XOR term depends on the lane only: offset = (k16 << 5) ^ frag_x.
  const int frag_row = lane & 31;
  const uint32_t frag_x = (uint32_t)(((lane >> 5) ^ ((frag_row >> 1) & 7)) << 4);
  const uint32_t a_lane = (uint32_t)((wm * 128 + frag_row) * 128);            // + mi * 4096
  const uint32_t b_lane = (uint32_t)(A_BYTES + (wn * 64 + frag_row) * 128);   // + ni * 4096

  __syncthreads();  // thresholds / counters initialised

  for (int64_t row0 = r_begin; row0 < r_end; row0 += BM) {
    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    if constexpr (VARIANT == 1) {
      set_tile_offsets(row0);
      stage(0, row0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int cur = 0;
      for (int kt = 0; kt < n_ksteps; ++kt) {
        if (kt + 1 < n_ksteps) stage(cur ^ 1, row0, kt + 1);
        const unsigned char* sbase = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int k16 = 0; k16 < 4; ++k16) {
          const uint32_t kx = (uint32_t)(k16 << 5) ^ frag_x;
          f16x8 af[4], bf[2];
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            af[mi] = *reinterpret_cast<const f16x8*>(sbase + (a_lane + kx) + mi * 4096);
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            bf[ni] = *reinterpret_cast<const f16x8*>(sbase + (b_lane + kx) + ni * 4096);
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
      }
    } else {
      // slab 0 of this tile: issued here for the first tile, under the previous epilogue otherwise
      if (row0 == r_begin) {
        set_tile_offsets(row0);
        stage(0, row0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      // which waves form a ping-pong group: the two groups must hold one wave per SIMD each
      const int group = (p.group_sel == 0) ? (wave >> 2) : (p.group_sel == 1) ? (wave & 1) : ((wave >> 1) & 1);  // wave-uniform
      if (group == 1) TAVB_BARRIER();  // group 1 runs one barrier interval behind group 0
      for (int kt = 0; kt < n_ksteps; ++kt) {
        const unsigned char* sbase = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          // ---- LOAD phase: fragments of two k16 sub-steps, next slab's LDS-DMA
          if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(1);
          f16x8 af[2][4], bf[2][2];
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const uint32_t kx = (uint32_t)((half * 2 + kk) << 5) ^ frag_x;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
              af[kk][mi] = *reinterpret_cast<const f16x8*>(sbase + (a_lane + kx) + mi * 4096);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              bf[kk][ni] = *reinterpret_cast<const f16x8*>(sbase + (b_lane + kx) + ni * 4096);
          }
          if ((ABLATE & 2) == 0 && half == 0 && kt + 1 < n_ksteps) stage((kt + 1) & 1, row0, kt + 1);
          // the slab issued during this K step must have landed before the barrier that precedes group
          // 0's first read of it (group 1 waits here, group 0 after its MFMAs below)
          if (half == 1 && group == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the buffer can be restaged
          if constexpr (PRIO == 2) __builtin_amdgcn_s_setprio(0);
          TAVB_BARRIER();
          // ---- MFMA phase
          if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni) {
                if constexpr ((ABLATE & 1) == 0)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk][mi], bf[kk][ni], acc[mi][ni], 0, 0, 0);
                else
                  asm volatile("" ::"v"(af[kk][mi]), "v"(bf[kk][ni]));
              }
          if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
          if (half == 1 && group == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          TAVB_BARRIER();
        }
      }
      if (group == 0) TAVB_BARRIER();  // re-align the groups
      if ((ABLATE & 2) == 0 && row0 + BM < r_end) {  // next tile's first slab flies under the epilogue
        set_tile_offsets(row0 + BM);
        stage(0, row0 + BM, 0);
      }
    }

    // ---- epilogue: score, admission test, append ------------------------------------
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int ql = wn * 64 + ni * 32 + (lane & 31);  // this lane's query within the tile
      const float thr = thr_lds[ql];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        bool any = false;
        float sc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[r] = fmaf(acc[mi][ni][r], 0.5f, 0.5f);  // == (dot + 1) / 2 rounded once
          any = any || (sc[r] > thr);
        }
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
          if (any) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (sc[r] > thr) {
                const int64_t row = row0 + wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float s = sc[r];
                s = (s > 0.0f) ? s : 0.0f;
                s = (s > 1.0f) ? 1.0f : s;
                if (row < r_end && s >= p.min_score) {
                  const int pos = atomicAdd(&cnt_lds[ql], 1);
                  if (pos < CAP) my_cand[(size_t)ql * CAP + pos] = make_key(s, (uint32_t)row + p.index_base);
                }
              }
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
    __syncthreads();

    // ---- compaction of buffers that could overflow on the next tile ---------------------
    for (int q = wave; q < BN; q += NTHREADS / 64) {
      const int n = cnt_lds[q];  // wave-uniform (same address)
      if (n > CAP - BM) {
        u64* buf = my_cand + (size_t)q * CAP;
        const WaveTopK<1> best = best_of_buffer(buf, n < CAP ? n : CAP, lane);
        if (lane < p.k) buf[lane] = best.key[0];  // keep the best k at the front
        const int kept = __popcll(__ballot(best.key[0] != 0ull && lane < p.k));
        const u64 kth = best.at(p.k - 1);
        if (lane == 0) {
          cnt_lds[q] = kept;
          const float kth_score = __uint_as_float((uint32_t)(kth >> 32));
          if (kth != 0ull && kth_score > thr_lds[q]) thr_lds[q] = kth_score;
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
    __syncthreads();
  }

  // ---- final: every buffer -> sorted list of k keys ---------------------------------------
  for (int q = wave; q < BN; q += NTHREADS / 64) {
    const int qg = qtile * BN + q;
    if (qg >= p.nq) continue;
    const int n = cnt_lds[q];
    const WaveTopK<1> best = best_of_buffer(my_cand + (size_t)q * CAP, n < CAP ? n : CAP, lane);
    u64* out = p.lists + ((size_t)qg * p.list_stride + split) * (size_t)p.k;
    if (lane < p.k) out[lane] = best.key[0];
  }
}


// ---------------------------------------------------------------------------------------------
// VARIANT 3: same tile shape and ping-pong wave groups as variant 2, but the operand stream is
// decoupled from the tile loop:
//   * K advances in steps of 32 halves (64 bytes per row, 16 KiB per operand per step);
//   * the corpus operand (A) and the query operand (B) have separate LDS rings -- NA slots for A
//     (long latency: HBM / Infinity Cache), NB slots for B (short latency: the query tile stays in
//     L2) -- and separate stagers: the four waves of group 0 issue A's LDS-DMA, the four waves of
//     group 1 issue B's, so each wave's vmcnt queue holds one operand's loads only and a counted
//     `s_waitcnt vmcnt(4 * (depth - 1))` retires exactly the step that is needed next while
//     depth - 1 steps stay in flight;
//   * the step stream runs straight across tile boundaries (the loads for the next tile's first
//     steps are already in flight while the current tile's epilogue runs);
//   * 16-byte slots of the 64-byte LDS rows are XOR-swizzled with (row >> 2) & 3.
// Everything else (admission / append / compaction, result lists) is as in variants 1 and 2.
// ---------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NA, int NB, int ABL, int A_AUX = 0>
__global__ void __launch_bounds__(NTHREADS) mfma_scan_kernel_v3(const MfmaDeviceParams p) {
  constexpr int KS = 32;                 // halves per step
  constexpr int SLOT = 256 * KS * 2;     // 16 KiB: one operand, one step
  constexpr int DA = NA - 1, DB = NB - 1;  // steps in flight
  constexpr int B_RING = NA * SLOT;
  constexpr int CTRL = (NA + NB) * SLOT;
  extern __shared__ __align__(16) unsigned char smem[];
  float* thr_lds = reinterpret_cast<float*>(smem + CTRL);
  int* cnt_lds = reinterpret_cast<int*>(smem + CTRL + BN * 4);
  volatile int* need_compact = reinterpret_cast<volatile int*>(smem + CTRL + BN * 8);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;
  const int wn = wave & 3;
  const int group = wm;  // group 0 = waves 0-3 (tile rows 0-127, stages A); group 1 = waves 4-7 (stages B)
  const int lw = wave & 3;

  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int t = b >> 3;
  const int qtile = t % p.n_qtiles;
  const int split = (t / p.n_qtiles) * 8 + xcd;
  if (split >= p.n_splits) return;
  const int64_t r_begin = (int64_t)split * p.rows_per_split;
  const int64_t r_end = (r_begin + p.rows_per_split < p.rows) ? r_begin + p.rows_per_split : p.rows;
  const int logical_block = split * p.n_qtiles + qtile;
  u64* my_cand = p.cand + (size_t)logical_block * BN * CAP;

  const float thr0 = (p.min_score > 0.0f) ? __uint_as_float(__float_as_uint(p.min_score) - 1u) : -__builtin_inff();
  for (int i = tid; i < BN; i += NTHREADS) {
    float t0 = (p.min_score != p.min_score) ? __builtin_inff() : thr0;  // NaN threshold admits nothing
    const int qg0 = qtile * BN + i;
    if (qg0 >= p.nq) t0 = __builtin_inff();  // padding queries admit nothing
    else if (p.thr_in && p.thr_in[qg0] > t0) t0 = p.thr_in[qg0];  // k-th best of the sample pass: a valid lower bound
    thr_lds[i] = t0;
    cnt_lds[i] = 0;
  }
  if (tid == 0) *need_compact = 0;

  const int D = p.dim;
  const int steps_per_tile = D / KS;
  const size_t row_bytes = (size_t)D * 2;
  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const char* qbase = reinterpret_cast<const char*>(p.queries) + (size_t)qtile * BN * row_bytes;
  const int n_tiles = (r_end > r_begin) ? (int)((r_end - r_begin + BM - 1) / BM) : 0;
  if (n_tiles == 0) {
    // empty row range: emit empty lists
    for (int q = wave; q < BN; q += NTHREADS / 64) {
      const int qg = qtile * BN + q;
      if (qg < p.nq && lane < p.k) p.lists[((size_t)qg * p.list_stride + split) * (size_t)p.k + lane] = 0ull;
    }
    return;
  }

  // ---- stager state: instruction j of this wave covers operand rows (lw*4 + j)*16 .. +15, four
  //      lanes (16-byte slots) per 64-byte row
  const int st_row_in_inst = lane >> 2;
  const int st_slot = lane & 3;
  uint32_t st_off[4];
  auto set_offsets = [&](int64_t row0, bool clamp) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (lw * 4 + j) * 16 + st_row_in_inst;
      int64_t r = row;
      if (clamp && row0 + r >= p.rows) r = p.rows - 1 - row0;  // stay in bounds; masked in the epilogue
      st_off[j] = (uint32_t)r * (uint32_t)row_bytes + (uint32_t)((st_slot ^ ((row >> 2) & 3)) * 16);
      if (clamp && p.a_tiled) st_off[j] = (uint32_t)((lw * 4 + j) * 1024 + lane * 16);  // the stored image is the LDS image
    }
  };
  int st_tile = 0;   // tile of the next step this wave stages (group 0 only; group 1's operand has no tiles)
  int st_kt = 0;     // K step within the tile
  int st_slot_idx = 0;  // ring slot it goes to
  set_offsets(r_begin, group == 0);

  // Tile rendezvous.  The n_qtiles workgroups of a row range stream the same corpus rows and sit on the same XCD
  // (same L2).  Left alone they drift apart (their admission work differs) until each re-read of a corpus slice
  // misses L2 and comes from the Infinity Cache at a third of the rate.  So the corpus stagers meet at every tile
  // start: one lane counts the workgroup in, the stager waves poll the counter with a scalar load (lgkmcnt: the
  // counted vmcnt queue of the LDS-DMA stream is left alone) until all n_qtiles workgroups have arrived.  The wait is
  // bounded -- a peer that is not resident (fewer free CUs than workgroups) only costs the first time-out, after
  // which this workgroup stops waiting -- so the result never depends on it.
  int* const sync_ctr = (p.sync != nullptr && p.n_qtiles > 1) ? p.sync + split : nullptr;
  bool sync_on = sync_ctr != nullptr;
  auto rendezvous = [&](int tile) {
    if (lw == 0 && lane == 0) __hip_atomic_fetch_add(sync_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!sync_on) return;
    const int want = p.n_qtiles * (tile + 1);
    for (int spin = 0;; ++spin) {
      int seen;
      asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(seen) : "s"(sync_ctr) : "memory");
      if (seen >= want) break;
      if (spin >= 256) {  // ~100 us: give up for good
        sync_on = false;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  };

  auto stage_next = [&]() {
    if (group == 0) {
      if (sync_ctr != nullptr && st_kt == 0 && st_tile < n_tiles) rendezvous(st_tile);
      const int tile = st_tile < n_tiles ? st_tile : n_tiles - 1;  // past the end: harmless reloads of the last tile
      const int64_t src_row0 = (ABL & 4) ? 0 : r_begin + (int64_t)tile * BM;  // ablation: every block re-reads tile 0 (L2 resident)
      const char* g = p.a_tiled ? sgpr_ptr(corpus + (size_t)src_row0 * row_bytes + (size_t)st_kt * SLOT)  // block (tile, step): 16 KiB
                                : sgpr_ptr(corpus + (size_t)src_row0 * row_bytes + (size_t)st_kt * (KS * 2));
      unsigned char* l = smem + st_slot_idx * SLOT + lw * 4096;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((global_void*)(g + (size_t)st_off[j]), (lds_void*)(l + j * 1024), 16, 0, A_AUX);
      if (++st_slot_idx == NA) st_slot_idx = 0;
      if (++st_kt == steps_per_tile) {
        st_kt = 0;
        ++st_tile;
        if (st_tile < n_tiles) set_offsets(r_begin + (int64_t)st_tile * BM, true);
      }
    } else {
      const char* g = sgpr_ptr(qbase + (size_t)st_kt * (KS * 2));
      unsigned char* l = smem + B_RING + st_slot_idx * SLOT + lw * 4096;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_global_load_lds((global_void*)(g + (size_t)st_off[j]), (lds_void*)(l + j * 1024), 16, 0, 0);
      if (++st_slot_idx == NB) st_slot_idx = 0;
      if (++st_kt == steps_per_tile) st_kt = 0;
    }
  };

  // ---- fragment read addresses: row (lane & 31) of a 32-row block, logical slot 2*k16 + (lane >> 5),
  //      physical slot = logical ^ ((row >> 2) & 3)  ->  byte (k16 << 5) ^ frag_x within the 64-byte row
  const int frag_row = lane & 31;
  const uint32_t frag_x = (uint32_t)(((lane >> 5) ^ ((frag_row >> 2) & 3)) << 4);
  const uint32_t a_lane = (uint32_t)((wm * 128 + frag_row) * 64);           // + mi * 2048
  const uint32_t b_lane = (uint32_t)(B_RING + (wn * 64 + frag_row) * 64);   // + ni * 2048

  // ---- prologue: fill the pipelines, wait for step 0
  if (group == 0) {
#pragma unroll 1
    for (int i = 0; i < DA; ++i) stage_next();
    wait_vmcnt<4 * (DA - 1)>();
  } else {
#pragma unroll 1
    for (int i = 0; i < DB; ++i) stage_next();
    wait_vmcnt<4 * (DB - 1)>();
  }
  __syncthreads();  // ring step 0 landed, thresholds initialised (no LDS-DMA is drained: the waits above are counted)
  if (group == 1) TAVB_BARRIER();  // group 1 runs one barrier interval behind group 0

  int rd_a = 0, rd_b = 0;  // ring slots of the step being consumed
  for (int tile = 0; tile < n_tiles; ++tile) {
    const int64_t row0 = r_begin + (int64_t)tile * BM;
    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

#pragma unroll 1
    for (int kt = 0; kt < steps_per_tile; ++kt) {
      // ---- LOAD phase
      const unsigned char* abase = smem + rd_a * SLOT;
      const unsigned char* bbase = smem + rd_b * SLOT;
      f16x8 af[2][4], bf[2][2];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const uint32_t kx = (uint32_t)(kk << 5) ^ frag_x;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) af[kk][mi] = *reinterpret_cast<const f16x8*>(abase + (a_lane + kx) + mi * 2048);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) bf[kk][ni] = *reinterpret_cast<const f16x8*>(bbase + (b_lane + kx) + ni * 2048);
      }
      if constexpr ((ABL & 2) == 0) {
        stage_next();  // the slot being refilled was last read one step ago (two barriers back)
        if (group == 1) wait_vmcnt<4 * (DB - 1)>();  // B of the next step has landed
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TAVB_BARRIER();
      // ---- MFMA phase
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            if constexpr ((ABL & 1) == 0)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk][mi], bf[kk][ni], acc[mi][ni], 0, 0, 0);
            else
              asm volatile("" ::"v"(af[kk][mi]), "v"(bf[kk][ni]));
          }
      __builtin_amdgcn_s_setprio(0);
      if constexpr ((ABL & 2) == 0) {
        if (group == 0) wait_vmcnt<4 * (DA - 1)>();  // A of the next step has landed
      }
      TAVB_BARRIER();
      if (++rd_a == NA) rd_a = 0;
      if (++rd_b == NB) rd_b = 0;
    }
    if (group == 0) TAVB_BARRIER();  // re-align the groups for the epilogue

    // ---- epilogue: score, admission test, append
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int ql = wn * 64 + ni * 32 + (lane & 31);
      const float thr = thr_lds[ql];
      // Fast test on the raw dot products: score = fma(dot, 0.5, 0.5) is monotone in dot, so `score > thr` implies
      // `dot > 2 thr - 1 - 2^-21` (the margin covers the roundings of both fmas with room to spare).  One max3
      // chain + one compare per 32x32 block instead of 16 fmas + 16 compares; the exact test is in the slow path.
      const float thr_pre = fmaf(thr, 2.0f, -1.0f) - 4.8e-7f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        float top = acc[mi][ni][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) top = __builtin_fmaxf(top, acc[mi][ni][r]);
        const bool any = (ABL == 0) && (top > thr_pre);
        if constexpr (ABL != 0) asm volatile("" ::"v"(acc[mi][ni]));  // keep the MFMAs alive when admissions are ablated
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
          float sc[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[r] = fmaf(acc[mi][ni][r], 0.5f, 0.5f);
          // slow path, taken by the whole wave when any lane admits something: every lane builds the
          // bit mask of its admitted rows, reserves that many buffer slots with ONE LDS atomic (the
          // latency of the returning atomic is paid once per 32x32 block, not once per key), then
          // stores its keys with predicated stores.
          const int64_t row_base = row0 + wm * 128 + mi * 32 + 4 * (lane >> 5);
          unsigned admit = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float s = sc[r];
            s = (s > 0.0f) ? s : 0.0f;
            s = (s > 1.0f) ? 1.0f : s;
            const bool ok = (sc[r] > thr) && (row_base + (r & 3) + 8 * (r >> 2) < r_end) && (s >= p.min_score);
            admit |= ok ? (1u << r) : 0u;
          }
          const int n_adm = __popc(admit);
          int pos = 0;
          if (n_adm > 0) {
            pos = lds_add_rtn(&cnt_lds[ql], n_adm);
            if (pos + n_adm > CAP - BM) lds_store_i32(need_compact, 1);  // this buffer could overflow on the next tile
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if ((admit >> r) & 1u) {
              float s = sc[r];
              s = (s > 0.0f) ? s : 0.0f;
              s = (s > 1.0f) ? 1.0f : s;
              if (pos < CAP)
                my_cand[(size_t)ql * CAP + pos] = make_key(s, (uint32_t)(row_base + (r & 3) + 8 * (r >> 2)) + p.index_base);
              ++pos;
            }
          }
        }
      }
    }
    // Compaction is rare (O(log rows) times per query).  Only then do the appended keys have to be in
    // memory for another wave to read, so only then does the workgroup pay a drain of its (otherwise
    // still flying) LDS-DMA queues; normally the epilogue ends at this barrier.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TAVB_BARRIER();
    if (*need_compact != 0) {  // workgroup-uniform: read after the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TAVB_BARRIER();
    for (int q = wave; q < BN; q += NTHREADS / 64) {
      const int n = cnt_lds[q];
      if (n > CAP - BM) {
        u64* buf = my_cand + (size_t)q * CAP;
        const WaveTopK<1> best = best_of_buffer(buf, n < CAP ? n : CAP, lane);
        if (lane < p.k) buf[lane] = best.key[0];
        const int kept = __popcll(__ballot(best.key[0] != 0ull && lane < p.k));
        const u64 kth = best.at(p.k - 1);
        if (lane == 0) {
          cnt_lds[q] = kept;
          const float kth_score = __uint_as_float((uint32_t)(kth >> 32));
          if (kth != 0ull && kth_score > thr_lds[q]) thr_lds[q] = kth_score;
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
    if (group == 1) TAVB_BARRIER();  // stagger again
  }
  if (group == 0) TAVB_BARRIER();  // pairs with group 1's last stagger barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the run-ahead loads before the block retires
  __syncthreads();

  for (int q = wave; q < BN; q += NTHREADS / 64) {
    const int qg = qtile * BN + q;
    if (qg >= p.nq) continue;
    const int n = cnt_lds[q];
    const WaveTopK<1> best = best_of_buffer(my_cand + (size_t)q * CAP, n < CAP ? n : CAP, lane);
    u64* out = p.lists + ((size_t)qg * p.list_stride + split) * (size_t)p.k;
    if (lane < p.k) out[lane] = best.key[0];
  }
}


// ---------------------------------------------------------------------------------------------
// VARIANT 4: no ping-pong.  Every wave software-pipelines itself: while its 16 MFMAs of step S
// issue (one every 32 cycles), the 12 fragment reads of step S+1 (second register set) and its 4
// LDS-DMA instructions for step S+N are slotted into the gaps between them
// (sched_group_barrier: 1 MFMA : 1 DS read / 1 VMEM).  One barrier per K step.  Rings, swizzle,
// decoupled stagers (waves 0-3 stage A, waves 4-7 stage B) and counted vmcnt waits as in variant 3.
//   step S:  [stager: vmcnt(4*(N-2)) -> step S+1 landed]  lgkmcnt(0)  s_barrier
//            { MFMAs(S) on frag set S&1 | ds_reads of ring slot S+1 into frag set (S+1)&1 |
//              LDS-DMA of step S+N into ring slot S (everyone finished reading it before the barrier) }
// ---------------------------------------------------------------------------------------------
template <int NA, int NB, int ABL>
__global__ void __launch_bounds__(NTHREADS) mfma_scan_kernel_v4(const MfmaDeviceParams p) {
  constexpr int KS = 32;
  constexpr int SLOT = 256 * KS * 2;
  constexpr int B_RING = NA * SLOT;
  constexpr int CTRL = (NA + NB) * SLOT;
  extern __shared__ __align__(16) unsigned char smem[];
  float* thr_lds = reinterpret_cast<float*>(smem + CTRL);
  int* cnt_lds = reinterpret_cast<int*>(smem + CTRL + BN * 4);
  volatile int* need_compact = reinterpret_cast<volatile int*>(smem + CTRL + BN * 8);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;
  const int wn = wave & 3;
  const bool is_a = wave < 4;  // waves 0-3 stage the corpus operand, waves 4-7 the query operand
  const int lw = wave & 3;

  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int t = b >> 3;
  const int qtile = t % p.n_qtiles;
  const int split = (t / p.n_qtiles) * 8 + xcd;
  if (split >= p.n_splits) return;
  const int64_t r_begin = (int64_t)split * p.rows_per_split;
  const int64_t r_end = (r_begin + p.rows_per_split < p.rows) ? r_begin + p.rows_per_split : p.rows;
  const int logical_block = split * p.n_qtiles + qtile;
  u64* my_cand = p.cand + (size_t)logical_block * BN * CAP;

  const float thr0 = (p.min_score > 0.0f) ? __uint_as_float(__float_as_uint(p.min_score) - 1u) : -__builtin_inff();
  for (int i = tid; i < BN; i += NTHREADS) {
    float t0 = (p.min_score != p.min_score) ? __builtin_inff() : thr0;  // NaN threshold admits nothing
    const int qg0 = qtile * BN + i;
    if (qg0 >= p.nq) t0 = __builtin_inff();  // padding queries admit nothing
    else if (p.thr_in && p.thr_in[qg0] > t0) t0 = p.thr_in[qg0];  // k-th best of the sample pass: a valid lower bound
    thr_lds[i] = t0;
    cnt_lds[i] = 0;
  }
  if (tid == 0) *need_compact = 0;

  const int D = p.dim;
  const int steps_per_tile = D / KS;  // even: D is a multiple of 64
  const uint32_t row_bytes = (uint32_t)D * 2u;
  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const char* qbase = reinterpret_cast<const char*>(p.queries) + (size_t)qtile * BN * row_bytes;
  const int n_tiles = (r_end > r_begin) ? (int)((r_end - r_begin + BM - 1) / BM) : 0;
  if (n_tiles == 0) {
    for (int q = wave; q < BN; q += NTHREADS / 64) {
      const int qg = qtile * BN + q;
      if (qg < p.nq && lane < p.k) p.lists[((size_t)qg * p.list_stride + split) * (size_t)p.k + lane] = 0ull;
    }
    return;
  }

  // ---- stager: instruction j covers operand rows (lw*4 + j)*16 .. +15, four 16-byte slots per row.
  //      Branch-free: everything that differs between the A and B stagers is a scalar select.
  const bool lin = is_a && p.a_tiled;  // K-blocked corpus image: 16 KiB per (tile, step), already in LDS order
  uint32_t st_rowoff[4], st_slotoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (lw * 4 + j) * 16 + (lane >> 2);
    st_rowoff[j] = lin ? 0u : (uint32_t)row * row_bytes;
    st_slotoff[j] = lin ? (uint32_t)((lw * 4 + j) * 1024 + lane * 16) : (uint32_t)(((lane & 3) ^ ((row >> 2) & 3)) * 16);
  }
  const int ring_n = is_a ? NA : NB;
  const uint32_t ring_base = (is_a ? 0u : (uint32_t)B_RING) + (uint32_t)lw * 4096u;
  // running, wave-uniform source pointer of the next step to stage: +64 bytes per K step (+16 KiB in the
  // K-blocked image); at the end of a tile's K range the A stager jumps to the next tile (or, past the last
  // tile, back to the start of the last one: harmless reloads that keep the vmcnt bookkeeping uniform), the
  // B stager back to k = 0.
  const int64_t k_step = lin ? (int64_t)SLOT : (int64_t)(KS * 2);
  const int64_t k_rewind = -(int64_t)(steps_per_tile - 1) * k_step;
  const int64_t tile_jump = is_a ? (lin ? k_step : k_rewind + (int64_t)BM * row_bytes) : k_rewind;
  const int64_t wrap_delta = k_rewind - k_step;        // added when the K range of a tile ends
  const int64_t advance_delta = tile_jump - k_rewind;  // added on top when the stager moves on to the next tile
  const char* st_ptr = is_a ? corpus + (size_t)((ABL & 4) ? 0 : r_begin) * row_bytes : qbase;
  int64_t st_last_row = is_a ? (p.rows - 1 - r_begin) : 255;  // last valid row of the staged tile, relative to its row 0
  int st_tiles_left = is_a ? n_tiles - 1 : 0;
  int st_kt = 0, st_slot = 0;

  auto stage_next = [&]() {
    const char* g = sgpr_ptr(st_ptr);
    // rows past the end of the corpus are clamped to its last row (they are masked in the epilogue)
    const uint32_t max_rowoff = (uint32_t)(st_last_row < 255 ? st_last_row : 255) * row_bytes;
    unsigned char* l = smem + ring_base + (uint32_t)st_slot * SLOT;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t off = (st_rowoff[j] < max_rowoff ? st_rowoff[j] : max_rowoff) + st_slotoff[j];
      __builtin_amdgcn_global_load_lds((global_void*)(g + (size_t)off), (lds_void*)(l + j * 1024), 16, 0, 0);
    }
    st_slot = (st_slot + 1 == ring_n) ? 0 : st_slot + 1;
    const bool wrap = (st_kt + 1 == steps_per_tile);
    const bool advance = wrap && st_tiles_left > 0 && (ABL & 4) == 0;
    st_kt = wrap ? 0 : st_kt + 1;
    // arithmetic instead of a nested select: the compiler turns a select tree over run-time 64-bit values
    // into a scratch-resident lookup table, which drags the whole stager state into scratch memory
    st_ptr += k_step + (int64_t)wrap * wrap_delta + (int64_t)advance * advance_delta;
    st_last_row -= advance ? BM : 0;
    st_tiles_left -= advance ? 1 : 0;
  };

  const int frag_row = lane & 31;
  const uint32_t frag_x = (uint32_t)(((lane >> 5) ^ ((frag_row >> 2) & 3)) << 4);
  const uint32_t a_lane = (uint32_t)((wm * 128 + frag_row) * 64);
  const uint32_t b_lane = (uint32_t)(B_RING + (wn * 64 + frag_row) * 64);

  auto read_frags = [&](f16x8(&af)[2][4], f16x8(&bf)[2][2], int slot_a, int slot_b) {
    const unsigned char* abase = smem + slot_a * SLOT;
    const unsigned char* bbase = smem + slot_b * SLOT;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const uint32_t kx = (uint32_t)(kk << 5) ^ frag_x;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) af[kk][mi] = *reinterpret_cast<const f16x8*>(abase + (a_lane + kx) + mi * 2048);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) bf[kk][ni] = *reinterpret_cast<const f16x8*>(bbase + (b_lane + kx) + ni * 2048);
    }
  };

  // ---- prologue: fill both rings completely, wait for step 0, read its fragments
  if (is_a) {
#pragma unroll 1
    for (int i = 0; i < NA; ++i) stage_next();
    wait_vmcnt<4 * (NA - 1)>();
  } else {
#pragma unroll 1
    for (int i = 0; i < NB; ++i) stage_next();
    wait_vmcnt<4 * (NB - 1)>();
  }
  TAVB_BARRIER();
  f16x8 af0[2][4], bf0[2][2], af1[2][4], bf1[2][2];
  read_frags(af0, bf0, 0, 0);
  int rd_a = 1, rd_b = 1;  // ring slots of the step whose fragments are read next

  f32x16 acc[4][2];

  // one K step: MFMAs on (fu_a, fu_b), prefetch the next step's fragments into (fl_a, fl_b)
  auto step = [&](f16x8(&fu_a)[2][4], f16x8(&fu_b)[2][2], f16x8(&fl_a)[2][4], f16x8(&fl_b)[2][2], bool sync) {
    if (sync) {
    if (is_a)
      wait_vmcnt<4 * (NA - 2)>();  // the next step's A slab has landed
    else
      wait_vmcnt<4 * (NB - 2)>();  // the next step's B slab has landed
    // this step's fragments are in registers, so its ring slot is free.  The builtin (not inline asm) so
    // that the compiler's own wait-count bookkeeping sees it and does not put a second lgkmcnt(0) -- one
    // that would also drain the reads issued below -- in front of the first MFMA.
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt/expcnt untouched
    TAVB_BARRIER();
    }
    if constexpr ((ABL & 32) == 0) read_frags(fl_a, fl_b, rd_a, rd_b);
    if constexpr ((ABL & 2) == 0) stage_next();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          if constexpr ((ABL & 1) == 0)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fu_a[kk][mi], fu_b[kk][ni], acc[mi][ni], 0, 0, 0);
          else
            asm volatile("" ::"v"(fu_a[kk][mi]), "v"(fu_b[kk][ni]));
        }
    // interleave: one LDS read after each of the first 12 MFMAs, one LDS-DMA after each of the last 4
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    rd_a = (rd_a + 1 == NA) ? 0 : rd_a + 1;
    rd_b = (rd_b + 1 == NB) ? 0 : rd_b + 1;
  };

  for (int tile = 0; tile < n_tiles; ++tile) {
    const int64_t row0 = r_begin + (int64_t)tile * BM;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

#pragma unroll 1
    for (int kt = 0; kt < steps_per_tile; kt += 2) {
      step(af0, bf0, af1, bf1, (ABL & 16) ? (kt & 3) == 0 : true);
      step(af1, bf1, af0, bf0, (ABL & 24) ? false : true);
    }

    // ---- epilogue: score, admission test, append
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int ql = wn * 64 + ni * 32 + (lane & 31);
      const float thr = thr_lds[ql];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        bool any = false;
        float sc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[r] = fmaf(acc[mi][ni][r], 0.5f, 0.5f);
          any = any || ((ABL == 0 || ABL == 512) && sc[r] > thr);
        }
        if constexpr (ABL != 0) asm volatile("" ::"v"(acc[mi][ni]));
        if constexpr (ABL == 512) asm volatile("" ::"s"(__builtin_amdgcn_ballot_w64(any)));
        if (ABL != 512 && __builtin_amdgcn_ballot_w64(any) != 0ull) {
          // slow path, taken by the whole wave when any lane admits something: every lane builds the
          // bit mask of its admitted rows, reserves that many buffer slots with ONE LDS atomic (the
          // latency of the returning atomic is paid once per 32x32 block, not once per key), then
          // stores its keys with predicated stores.
          const int64_t row_base = row0 + wm * 128 + mi * 32 + 4 * (lane >> 5);
          unsigned admit = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float s = sc[r];
            s = (s > 0.0f) ? s : 0.0f;
            s = (s > 1.0f) ? 1.0f : s;
            const bool ok = (sc[r] > thr) && (row_base + (r & 3) + 8 * (r >> 2) < r_end) && (s >= p.min_score);
            admit |= ok ? (1u << r) : 0u;
          }
          const int n_adm = __popc(admit);
          int pos = 0;
          if (n_adm > 0) {
            pos = lds_add_rtn(&cnt_lds[ql], n_adm);
            if (pos + n_adm > CAP - BM) lds_store_i32(need_compact, 1);  // this buffer could overflow on the next tile
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if ((admit >> r) & 1u) {
              float s = sc[r];
              s = (s > 0.0f) ? s : 0.0f;
              s = (s > 1.0f) ? 1.0f : s;
              if (pos < CAP)
                my_cand[(size_t)ql * CAP + pos] = make_key(s, (uint32_t)(row_base + (r & 3) + 8 * (r >> 2)) + p.index_base);
              ++pos;
            }
          }
        }
      }
    }
    // Compaction is rare (O(log rows) times per query).  Only then do the appended keys have to be in
    // memory for another wave to read, so only then does the workgroup pay a drain of its (otherwise
    // still flying) LDS-DMA queues; normally the epilogue ends at this barrier.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TAVB_BARRIER();
    if (*need_compact != 0) {  // workgroup-uniform: read after the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TAVB_BARRIER();
    for (int q = wave; q < BN; q += NTHREADS / 64) {
      const int n = cnt_lds[q];
      if (n > CAP - BM) {
        u64* buf = my_cand + (size_t)q * CAP;
        const WaveTopK<1> best = best_of_buffer(buf, n < CAP ? n : CAP, lane);
        if (lane < p.k) buf[lane] = best.key[0];
        const int kept = __popcll(__ballot(best.key[0] != 0ull && lane < p.k));
        const u64 kth = best.at(p.k - 1);
        if (lane == 0) {
          cnt_lds[q] = kept;
          const float kth_score = __uint_as_float((uint32_t)(kth >> 32));
          if (kth != 0ull && kth_score > thr_lds[q]) thr_lds[q] = kth_score;
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

  for (int q = wave; q < BN; q += NTHREADS / 64) {
    const int qg = qtile * BN + q;
    if (qg >= p.nq) continue;
    const int n = cnt_lds[q];
    const WaveTopK<1> best = best_of_buffer(my_cand + (size_t)q * CAP, n < CAP ? n : CAP, lane);
    u64* out = p.lists + ((size_t)qg * p.list_stride + split) * (size_t)p.k;
    if (lane < p.k) out[lane] = best.key[0];
  }
}

 
// ---------------------------------------------------------------------------------------------
// VARIANT 5: fewer operand bytes per flop.  Variants 1-4 sit on the machine balance between the
// L2 -> CU fabric and the matrix pipe (profiles/r01_cfg3_operand_path.md): a 256 x 256 tile pulls
// 32 KiB through L2 per 1024 MFMA cycles.  Here the tile is 384 corpus rows x 256 queries held by
// FOUR waves (2 x 2), one per SIMD, each with the whole 512-register budget: a 192 x 128 sub-tile =
// 6 x 4 MFMA tiles = 384 accumulator registers + two 40-register fragment sets.  hipcc selects the AGPR or the
// VGPR form of an MFMA builtin per FUNCTION, so 384 accumulators cannot be split over the two files through
// the builtin (600 spills); the MFMAs are therefore inline asm with explicit register classes: 16 tiles
// accumulate in AGPRs ("+a"), 8 in VGPRs ("+v").  A volatile asm is ordered against memory operations, so
// the program order below -- MFMA, LDS read, MFMA, ..., MFMA, LDS-DMA -- IS the schedule (no
// sched_group_barrier needed).  Per flop
// that is 17 % fewer L2 -> LDS bytes and LDS-DMA instructions, 44 % fewer LDS fragment-read bytes
// and a third fewer barriers than the 8-wave tile.
//   * K advances in steps of 32 halves; a step is two half-steps (k16 slices) of 24 MFMAs each.
//   * one ring of 3 slots per operand (A: 24 KiB per slot, B: 16 KiB); every wave stages
//     6 A pieces + 4 B pieces (1 KiB each) per step, five per half-step, so its vmcnt queue has
//     the same shape in every wave and one counted wait serves both operands.
//   * software pipeline of one wave (no partner wave on the SIMD to hide anything):
//       (S,0): MFMAs on frag set 0 (step S, k 0-15)  | read set 1 <- slot S, k 16-31   | stage 2nd half of step S+2
//       ---- vmcnt(10): step S+1 landed in this wave; lgkmcnt(0); s_barrier (the only one per step) ----
//       (S,1): MFMAs on frag set 1                   | read set 0 <- slot S+1, k 0-15  | stage 1st half of step S+3
//     Slot S is last read in (S,0), so after the barrier it takes step S+3; slot S+1 is first read
//     after the barrier that follows the wait for its loads; loads have >= 1.5 steps to land.
//   * the step stream runs across tile boundaries; the epilogue (admission test on the raw dot
//     products, appends, rare compaction) is as in variant 3.
// ---------------------------------------------------------------------------------------------
constexpr int BM5 = 384;
constexpr int NT5 = 256;
constexpr int SLOT_A5 = BM5 * 64;   // 24 KiB
constexpr int SLOT_B5 = BN * 64;    // 16 KiB
constexpr int RING_A5 = 3;  // (a fourth A slot with the corpus stream one step further ahead measured no gain)
constexpr int RING_B5 = 3;
constexpr int B_RING5 = RING_A5 * SLOT_A5;
constexpr int CTRL5 = RING_A5 * SLOT_A5 + RING_B5 * SLOT_B5;
constexpr int LDS5 = CTRL5 + BN * 8 + 16;

template <int ABL>
__global__ void __launch_bounds__(NT5) mfma_scan_kernel_v5(const MfmaDeviceParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* thr_lds = reinterpret_cast<float*>(smem + CTRL5);
  int* cnt_lds = reinterpret_cast<int*>(smem + CTRL5 + BN * 4);
  typedef __attribute__((address_space(3))) volatile int lds_flag;
  lds_flag* need_compact = (lds_flag*)(smem + CTRL5 + BN * 8);  // explicit LDS pointer: the generic-pointer form miscompiles in this kernel

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1;  // rows wm * 192 ..
  const int wn = wave & 1;   // queries wn * 128 ..

  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int t = b >> 3;
  const int qtile = t % p.n_qtiles;
  const int split = (t / p.n_qtiles) * 8 + xcd;
  if (split >= p.n_splits) return;
  const int64_t r_begin = (int64_t)split * p.rows_per_split;
  const int64_t r_end = (r_begin + p.rows_per_split < p.rows) ? r_begin + p.rows_per_split : p.rows;
  const int logical_block = split * p.n_qtiles + qtile;
  u64* my_cand = p.cand + (size_t)logical_block * BN * CAP;

  const float thr0 = (p.min_score > 0.0f) ? __uint_as_float(__float_as_uint(p.min_score) - 1u) : -__builtin_inff();
  for (int i = tid; i < BN; i += NT5) {
    float t0 = (p.min_score != p.min_score) ? __builtin_inff() : thr0;  // NaN threshold admits nothing
    const int qg0 = qtile * BN + i;
    if (qg0 >= p.nq) t0 = __builtin_inff();  // padding queries admit nothing
    else if (p.thr_in && p.thr_in[qg0] > t0) t0 = p.thr_in[qg0];  // k-th best so far: a valid lower bound
    thr_lds[i] = t0;
    cnt_lds[i] = 0;
  }
  if (tid == 0) *need_compact = 0;

  const int D = p.dim;
  const int steps_per_tile = D / 32;
  const uint32_t row_bytes = (uint32_t)D * 2u;
  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const char* qbase = reinterpret_cast<const char*>(p.queries) + (size_t)qtile * BN * row_bytes;
  const int n_tiles = (r_end > r_begin) ? (int)((r_end - r_begin + BM5 - 1) / BM5) : 0;
  if (n_tiles == 0) {
    for (int q = wave; q < BN; q += NT5 / 64) {
      const int qg = qtile * BN + q;
      if (qg < p.nq && lane < p.k) p.lists[((size_t)qg * p.list_stride + split) * (size_t)p.k + lane] = 0ull;
    }
    return;
  }

  // ---- stager.  Piece = 16 operand rows x 64 bytes -> 1 KiB of LDS, lane l = row l >> 2, 16-byte slot l & 3 of that
  //      row, fetched from the XOR-swizzled source slot.  Wave w stages A pieces 6w .. 6w+5 and B pieces 4w .. 4w+3.
  // The five per-lane constants of the K loop: two staging offsets, three fragment-address terms.
  int a_off0, b_off0;
  uint32_t frag_x, a_lane, b_lane;
  auto set_lane_constants = [&]() {
    int zero = 0;
    asm volatile("" : "+v"(zero));
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero));
    const uint32_t lane_row = (uint32_t)(ln >> 2);
    const uint32_t lane_slot = (uint32_t)(((ln & 3) ^ ((ln >> 4) & 3)) * 16);
    a_off0 = (int)(((uint32_t)wave * 96u + lane_row) * row_bytes + lane_slot);
    b_off0 = (int)(((uint32_t)wave * 64u + lane_row) * row_bytes + lane_slot);
    const int frag_row = ln & 31;
    frag_x = (uint32_t)(((ln >> 5) ^ ((frag_row >> 2) & 3)) << 4);
    a_lane = (uint32_t)((wm * 192 + frag_row) * 64);            // + mi * 2048
    b_lane = (uint32_t)(B_RING5 + (wn * 128 + frag_row) * 64);  // + ni * 2048
  };
  set_lane_constants();
  // Staging goes through buffer descriptors (`buffer_load_dwordx4 ... lds`): the per-lane part of an address is ONE
  // persistent 32-bit VGPR offset per operand, the piece (16 rows apart) and the K step are the scalar offset, the tile is
  // the descriptor base, and rows past the end of the corpus are cut off by the descriptor's size (they read as zero; the
  // epilogue masks them anyway).  A per-piece VGPR address temp -- what the flat form needs once a clamp is involved --
  // would be overwritten while its LDS-DMA is in flight, which hipcc guards with `s_waitcnt vmcnt(0)` inside the K loop.
  const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sgpr_ptr(qbase)), 0, (int)(BN * row_bytes), 0x00020000);
  // Separate A / B stager state (the two streams could run at different depths; they run at the same one).
  int sa_kt = 0, sa_tile = 0, sa_slot = 0;  // next A step to stage: K step, tile, ring slot (of RING_A5)
  int sb_kt = 0, sb_slot = 0;               // next B step to stage
  auto stage_a_piece = [&](auto aj_tag) {
    constexpr int AJ = decltype(aj_tag)::value;
    const int tile = sa_tile < n_tiles ? sa_tile : n_tiles - 1;  // past the end: harmless reloads of the last tile
    const int64_t row0 = (ABL & 4) ? 0 : r_begin + (int64_t)tile * BM5;
    const int64_t left = p.rows - row0;
    const int valid = (int)(left < BM5 ? left : BM5);
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(sgpr_ptr(corpus + (size_t)row0 * row_bytes)), 0, __builtin_amdgcn_readfirstlane(valid * (int)row_bytes), 0x00020000);
    unsigned char* la = smem + sa_slot * SLOT_A5 + wave * 6144;
    const int soff = __builtin_amdgcn_readfirstlane(sa_kt * 64 + AJ * 16 * (int)row_bytes);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_void*)(la + AJ * 1024), 16, a_off0, soff, 0, 0);
    if constexpr (AJ == 5) {
      sa_slot = (sa_slot + 1 == RING_A5) ? 0 : sa_slot + 1;
      const bool wrap = (sa_kt + 1 == steps_per_tile);
      sa_kt = wrap ? 0 : sa_kt + 1;
      sa_tile += wrap ? 1 : 0;
    }
  };
  auto stage_b_piece = [&](auto bj_tag) {
    constexpr int BJ = decltype(bj_tag)::value;
    unsigned char* lb = smem + B_RING5 + sb_slot * SLOT_B5 + wave * 4096;
    const int soff = __builtin_amdgcn_readfirstlane(sb_kt * 64 + BJ * 16 * (int)row_bytes);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_void*)(lb + BJ * 1024), 16, b_off0, soff, 0, 0);
    if constexpr (BJ == 3) {
      sb_slot = (sb_slot + 1 == RING_B5) ? 0 : sb_slot + 1;
      sb_kt = (sb_kt + 1 == steps_per_tile) ? 0 : sb_kt + 1;
    }
  };
  // piece J of stage half HALF: HALF 0 = A pieces 0-4; HALF 1 = A piece 5, then B pieces 0-3
  auto stage_piece = [&](auto half_tag, auto j_tag) {
    constexpr int HALF = decltype(half_tag)::value;
    constexpr int J = decltype(j_tag)::value;
    if constexpr (HALF == 0)
      stage_a_piece(std::integral_constant<int, J>{});
    else if constexpr (J == 0)
      stage_a_piece(std::integral_constant<int, 5>{});
    else
      stage_b_piece(std::integral_constant<int, J - 1>{});
  };
  using J0 = std::integral_constant<int, 0>;
  using J1 = std::integral_constant<int, 1>;
  using J2 = std::integral_constant<int, 2>;
  using J3 = std::integral_constant<int, 3>;
  using J4 = std::integral_constant<int, 4>;
  auto stage_half = [&](auto half_tag) {  // prologue only: a whole half at once
    stage_piece(half_tag, J0{});
    stage_piece(half_tag, J1{});
    stage_piece(half_tag, J2{});
    stage_piece(half_tag, J3{});
    stage_piece(half_tag, J4{});
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;

  // ---- fragment reads: row (lane & 31) of a 32-row block, logical 16-byte slot 2 * kk + (lane >> 5)
  auto read_frags = [&](f16x8(&af)[6], f16x8(&bf)[4], int slot_a, int slot_b, int kk) {
    const uint32_t kx = (uint32_t)(kk << 5) ^ frag_x;
    const unsigned char* abase = smem + slot_a * SLOT_A5 + (a_lane + kx);
    const unsigned char* bbase = smem + slot_b * SLOT_B5 + (b_lane + kx);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) bf[ni] = *reinterpret_cast<const f16x8*>(bbase + ni * 2048);
#pragma unroll
    for (int mi = 0; mi < 6; ++mi) af[mi] = *reinterpret_cast<const f16x8*>(abase + mi * 2048);
  };

  // ---- prologue: steps 0 and 1 and the first half of step 2 in flight; wait for step 0 (everything but the last 15
  //      pieces); read the first fragments
  stage_half(H0{}); stage_half(H1{});
  stage_half(H0{}); stage_half(H1{});
  stage_half(H0{});
  wait_vmcnt<15>();
  __syncthreads();  // step 0 landed everywhere, thresholds initialised (the wait above is counted: nothing is drained)
  f16x8 a0[6], b0[4], a1[6], b1[4];
  read_frags(a0, b0, 0, 0, 0);
  int rd_a = 0, rd_b = 0;  // ring slots of the step being multiplied

  // tile t = mi * 4 + ni.  Fifteen tiles accumulate in AGPRs, nine in VGPRs: the sixteen AGPRs left over are where the
  // register allocator parks VGPR values during the epilogue (v_accvgpr_write / read, no memory involved); with all 256
  // AGPRs taken it parks them in scratch instead, and a scratch reload is a VMEM load behind the LDS-DMA queue.
  constexpr int NA_TILES = 15;
  f32x16 acc_a[NA_TILES];       // tiles 0 .. 14
  f32x16 acc_v[24 - NA_TILES];  // tiles 15 .. 23
  typedef int i32x4 __attribute__((ext_vector_type(4)));  // an <8 x half> asm operand gets repacked with v_perm; 4 x i32 does not
#define TAVB_MFMA_A(ACC, A, B) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(__builtin_bit_cast(i32x4, A)), "v"(__builtin_bit_cast(i32x4, B)))
#define TAVB_MFMA_V(ACC, A, B) \
  asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(__builtin_bit_cast(i32x4, A)), "v"(__builtin_bit_cast(i32x4, B)))

  // One half-step: the 24 MFMAs of k-slice KK of the current step on fragments (fa, fb), with the 10 fragment reads of
  // the next half-step (into (na, nb), ring slot `nslot`, k-slice NKK) and the five LDS-DMA pieces of stage half SH
  // slotted between them in program order.
  auto half_step = [&](f16x8(&fa)[6], f16x8(&fb)[4], f16x8(&na)[6], f16x8(&nb)[4], int nslot_a, int nslot_b, auto nkk_tag, auto sh_tag,
                       const bool do_read) {
    constexpr int NKK = decltype(nkk_tag)::value;
    const uint32_t kx = (uint32_t)(NKK << 5) ^ frag_x;
    const unsigned char* abase = smem + nslot_a * SLOT_A5 + (a_lane + kx);
    const unsigned char* bbase = smem + nslot_b * SLOT_B5 + (b_lane + kx);
    // 24 MFMAs in (mi, ni) order; behind MFMA i: a B read (i = 0..3), an A read (i = 4..9), and one staging piece
    // behind MFMAs 3, 8, 13, 18, 23.
    auto mfma_at = [&](auto i_tag) {
      constexpr int I = decltype(i_tag)::value;
      constexpr int mi = I >> 2, ni = I & 3;
      if constexpr ((ABL & 1) == 0) {
        if constexpr (I < NA_TILES)
          TAVB_MFMA_A(acc_a[I], fa[mi], fb[ni]);
        else
          TAVB_MFMA_V(acc_v[I - NA_TILES], fa[mi], fb[ni]);
      }
      if constexpr ((ABL & 32) == 0) {
        if (do_read) {  // wave-uniform
          if constexpr (I < 4) nb[I] = *reinterpret_cast<const f16x8*>(bbase + I * 2048);
          if constexpr (I >= 4 && I < 10) na[I - 4] = *reinterpret_cast<const f16x8*>(abase + (I - 4) * 2048);
        }
      }
      if constexpr ((ABL & 2) == 0) {
        if constexpr (I == 3) stage_piece(sh_tag, J0{});
        if constexpr (I == 8) stage_piece(sh_tag, J1{});
        if constexpr (I == 13) stage_piece(sh_tag, J2{});
        if constexpr (I == 18) stage_piece(sh_tag, J3{});
        if constexpr (I == 23) stage_piece(sh_tag, J4{});
      }
    };
    [&]<int... I>(std::integer_sequence<int, I...>) { (mfma_at(std::integral_constant<int, I>{}), ...); }
    (std::make_integer_sequence<int, 24>{});
    if constexpr ((ABL & 1) != 0) asm volatile("" ::"v"(fa[0]), "v"(fa[5]), "v"(fb[0]), "v"(fb[3]));
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;

  for (int tile = 0; tile < n_tiles; ++tile) {
    const int64_t row0 = r_begin + (int64_t)tile * BM5;
#pragma unroll
    for (int i = 0; i < NA_TILES; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_a[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 24 - NA_TILES; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_v[i][r] = 0.f;

#pragma unroll 1
    for (int kt = 0; kt < steps_per_tile; ++kt) {
      const int rd_a_next = (rd_a + 1 == RING_A5) ? 0 : rd_a + 1;
      const int rd_b_next = (rd_b + 1 == RING_B5) ? 0 : rd_b + 1;
      // ---- (S,0): multiply k 0-15 of step S; fetch its k 16-31 fragments; stage the second half of step S+2
      half_step(a0, b0, a1, b1, rd_a, rd_b, K1{}, H1{}, true);
      // ---- step S+1 has landed in this wave; slot S is read out; meet
      if constexpr ((ABL & 2) == 0 && (ABL & 1024) == 0) wait_vmcnt<10>();
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0) through the builtin: visible to the compiler's wait-count pass
      if constexpr ((ABL & 2048) == 0) TAVB_BARRIER();
      // ---- (S,1): multiply k 16-31; fetch k 0-15 of step S+1 (across the tile boundary too); stage the first half of S+3.
      //      (Skipping the fetch on a tile's last step -- to free its 40 registers for the epilogue -- needs a run-time
      //      predicate on the reads, whose branches between the MFMAs cost the K loop 30 %.)
      half_step(a1, b1, a0, b0, rd_a_next, rd_b_next, K0{}, H0{}, true);
      rd_a = rd_a_next;
      rd_b = rd_b_next;
    }

    // ---- epilogue: admission test on the raw dot products, append.  The asm MFMAs are invisible to the compiler's
    //      hazard recognizer: a 32x32x16 MFMA needs 18 wait states before its result may be read.
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    // Everything the epilogue derives from the lane id is derived HERE, from a laundered copy: hoisted out of the tile
    // loop these per-lane constants (candidate-buffer pointers, LDS addresses, exchange masks) do not fit next to 384
    // accumulators and would be spilled -- and a spill reload is a VMEM load queued behind the whole in-flight LDS-DMA.
    int zero_e = 0;
    asm volatile("" : "+v"(zero_e));  // (the lane id proper sits in a spill slot by now: recompute it from nothing)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero_e));
    // The VGPR-resident tiles go first: once tested they are dead, and their registers are what the AGPR-resident
    // tiles' copies then live in (interleaved, the allocator runs out and spills).
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int ql = wn * 128 + ni * 32 + (lane_e & 31);
      const float thr = thr_lds[ql];
      const float thr_pre = fmaf(thr, 2.0f, -1.0f) - 4.8e-7f;  // score > thr implies dot > thr_pre (see variant 3)
#pragma unroll
      for (int mi = 0; mi < 6; ++mi) {
        if ((mi * 4 + ni >= NA_TILES) != (pass == 0)) continue;  // pass 0: VGPR tiles, pass 1: AGPR tiles
        const f32x16 dots = (mi * 4 + ni < NA_TILES) ? acc_a[mi * 4 + ni] : acc_v[mi * 4 + ni - NA_TILES];
        float top = dots[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) top = __builtin_fmaxf(top, dots[r]);
        TAVB_SB();  // one block at a time: the scheduler would otherwise pull several blocks' accumulator reads forward (spills)
        const bool any = (ABL == 0 || ABL == 512) && (top > thr_pre);
        if constexpr (ABL == 512) asm volatile("" ::"s"(__builtin_amdgcn_ballot_w64(any)));  // test computed, slow path never taken
        if (ABL != 512 && __builtin_amdgcn_ballot_w64(any) != 0ull) {
          // (scores are recomputed where they are used: sixteen more live registers here would be spilled, and a spill
          //  reload is a VMEM load behind the whole in-flight LDS-DMA queue)
          const int64_t row_base = row0 + wm * 192 + mi * 32 + 4 * (lane_e >> 5);
          unsigned admit = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float sc = fmaf(dots[r], 0.5f, 0.5f);
            float s1 = (sc > 0.0f) ? sc : 0.0f;
            s1 = (s1 > 1.0f) ? 1.0f : s1;
            const bool ok = (sc > thr) && (row_base + (r & 3) + 8 * (r >> 2) < r_end) && (s1 >= p.min_score);
            admit |= ok ? (1u << r) : 0u;
          }
          const int n_adm = __popc(admit);
          int pos = 0;
          if (n_adm > 0) {
            pos = lds_add_rtn(&cnt_lds[ql], n_adm);
            if (pos + n_adm > CAP - BM5) lds_store_i32(need_compact, 1);  // this buffer could overflow on the next tile
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            TAVB_SB();  // one key at a time (sixteen keys and addresses in flight would not fit)
            if ((admit >> r) & 1u) {
              const float sc = fmaf(dots[r], 0.5f, 0.5f);
              float s1 = (sc > 0.0f) ? sc : 0.0f;
              s1 = (s1 > 1.0f) ? 1.0f : s1;
              if (pos < CAP)
                my_cand[(size_t)ql * CAP + pos] = make_key(s1, (uint32_t)(row_base + (r & 3) + 8 * (r >> 2)) + p.index_base);
              ++pos;
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
      for (int q = wave; q < BN; q += NT5 / 64) {
        const int n = cnt_lds[q];
        if (n > CAP - BM5) {
          u64* buf = my_cand + (size_t)q * CAP;
          const WaveTopK<1> best = best_of_buffer(buf, n < CAP ? n : CAP, lane_e);
          if (lane_e < p.k) buf[lane_e] = best.key[0];
          const int kept = __popcll(__ballot(best.key[0] != 0ull && lane_e < p.k));
          const u64 kth = best.at(p.k - 1);
          if (lane_e == 0) {
            cnt_lds[q] = kept;
            const float kth_score = __uint_as_float((uint32_t)(kth >> 32));
            if (kth != 0ull && kth_score > thr_lds[q]) thr_lds[q] = kth_score;
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

  for (int q = wave; q < BN; q += NT5 / 64) {
    const int qg = qtile * BN + q;
    if (qg >= p.nq) continue;
    const int n = cnt_lds[q];
    int lane_f = lane;
    asm volatile("" : "+v"(lane_f));
    const WaveTopK<1> best = best_of_buffer(my_cand + (size_t)q * CAP, n < CAP ? n : CAP, lane_f);
    u64* out = p.lists + ((size_t)qg * p.list_stride + split) * (size_t)p.k;
    if (lane_f < p.k) out[lane_f] = best.key[0];
  }
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
//   * tile = 256 corpus rows x 32 (or 64) queries, 4 waves, each wave owns 64 rows (two, or four, 32 x 32 MFMA tiles);
//     two workgroups per CU (72 KiB of LDS, < 128 VGPRs each) overlap each other's waits.
//   * K advances 64 bytes per row per step for either dtype (16 floats / 32 halves).  A wave stages the four
//     1 KiB pieces of ITS OWN 64 rows by LDS-DMA, so the corpus operand needs no cross-wave synchronisation;
//     waves 0 and 1 (all four for 64 queries) also stage one piece each of the query operand (16 queries x 64 bytes per
//     piece), which all waves read: one barrier per step.  Ring of 3 or 4 slots, counted vmcnt.
//   * LDS image, source-side XOR swizzle and fragment reads are those of variant 3 (64-byte rows).  For fp32
//     a lane's 16-byte fragment is four consecutive k of its row -- lanes 0-31 take k = 8g .. 8g+3, lanes 32-63
//     k = 8g+4 .. 8g+7 -- and feeds four MFMAs: MFMA e multiplies k = 8g+e (lower half-wave) and 8g+4+e
//     (upper), the same pairing on both operands, which is all a dot product needs.
//   * epilogue / candidate buffers / compaction / lists exactly as in the wide kernel (32 or 64 queries per block).
// ---------------------------------------------------------------------------------------------
constexpr int SQ32 = 32;            // queries per 32 x 32 MFMA block; a tile is NI of them (32 or 64 queries)
constexpr int S_THREADS = 256;
constexpr int S_SLOT_A = BM * 64;   // 16 KiB
typedef float f32x4 __attribute__((ext_vector_type(4)));
// fp32: one query operand plane, ring of 4 (72 KiB).  fp16: the fp32 queries are split into an fp16 high and an fp16
// low plane (q = hi + lo to 2^-22; both are multiplied -- the kernel is load-bound, the second MFMA is free), so a
// lookup on an fp16 corpus means the same thing here as in the streaming tiers (fp32 query x fp16 rows); ring of 3 (60 KiB).
// NI = 2 (64 queries per tile): twice the MFMAs per operand byte -- for batches of 33+ queries, which would otherwise
// stream the corpus once per 32 queries (fp32) or pay for a 256-query tile (fp16, 33 .. 64 queries).  Ring of 3.
template <typename T, int NI>
struct SkinnyGeom {
  static constexpr bool F32 = sizeof(T) == 4;
  static constexpr int SQ = NI * SQ32;
  static constexpr int PLANES = F32 ? 1 : 2;
  static constexpr int RING = (F32 && NI == 1) ? 4 : 3;
  static constexpr int PLANE_B = SQ * 64;          // one query operand plane, one step
  static constexpr int SLOT_B = PLANES * PLANE_B;
  static constexpr int B_RING = RING * S_SLOT_A;
  static constexpr int CTRL = RING * (S_SLOT_A + SLOT_B);
  static constexpr int LDS = CTRL + SQ * 8 + 16;
  static constexpr int B_WAVES = SQ / 16;          // waves that stage a query piece (per plane): 2 or 4
};

template <typename T, int NI, int ABL>
__global__ void __launch_bounds__(S_THREADS, 2) skinny_scan_kernel(const MfmaDeviceParams p) {
  using G = SkinnyGeom<T, NI>;
  constexpr int SQ = G::SQ;
  constexpr bool F32 = G::F32;
  constexpr int S_RING = G::RING, S_SLOT_B = G::SLOT_B, S_B_RING = G::B_RING, S_CTRL = G::CTRL;
  extern __shared__ __align__(16) unsigned char smem[];
  float* thr_lds = reinterpret_cast<float*>(smem + S_CTRL);
  int* cnt_lds = reinterpret_cast<int*>(smem + S_CTRL + SQ * 4);
  volatile int* need_compact = reinterpret_cast<volatile int*>(smem + S_CTRL + SQ * 8);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // rows wave * 64 .. of the tile

  const int b = blockIdx.x;
  const int xcd = b & 7;
  const int t = b >> 3;
  const int qtile = t % p.n_qtiles;
  const int split = (t / p.n_qtiles) * 8 + xcd;
  if (split >= p.n_splits) return;
  const int64_t r_begin = (int64_t)split * p.rows_per_split;
  const int64_t r_end = (r_begin + p.rows_per_split < p.rows) ? r_begin + p.rows_per_split : p.rows;
  const int logical_block = split * p.n_qtiles + qtile;
  u64* my_cand = p.cand + (size_t)logical_block * SQ * CAP;

  const float thr0 = (p.min_score > 0.0f) ? __uint_as_float(__float_as_uint(p.min_score) - 1u) : -__builtin_inff();
  if (tid < SQ) {  // SQ <= 64 < S_THREADS
    float t0 = (p.min_score != p.min_score) ? __builtin_inff() : thr0;  // NaN threshold admits nothing
    const int qg0 = qtile * SQ + tid;
    if (qg0 >= p.nq) t0 = __builtin_inff();  // padding queries admit nothing
    else if (p.thr_in && p.thr_in[qg0] > t0) t0 = p.thr_in[qg0];  // k-th best so far: a valid lower bound
    thr_lds[tid] = t0;
    cnt_lds[tid] = 0;
  }
  if (tid == 0) *need_compact = 0;

  const size_t row_bytes = (size_t)p.dim * sizeof(T);
  const int steps_per_tile = (int)(row_bytes / 64);
  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const char* qbase = reinterpret_cast<const char*>(p.queries) + (size_t)qtile * SQ * row_bytes;
  const size_t plane_bytes = (size_t)p.n_qtiles * SQ * row_bytes;  // fp16: the low plane follows the high plane
  const int n_tiles = (r_end > r_begin) ? (int)((r_end - r_begin + BM - 1) / BM) : 0;
  if (n_tiles == 0) {
    for (int q = wave; q < SQ; q += S_THREADS / 64) {
      const int qg = qtile * SQ + q;
      if (qg < p.nq && lane < p.k) p.lists[((size_t)qg * p.list_stride + split) * (size_t)p.k + lane] = 0ull;
    }
    return;
  }

  // ---- stager: piece j of this wave = corpus rows wave * 64 + j * 16 .. + 15 of the tile (its own rows); lane l = row
  //      l >> 2, 16-byte slot l & 3, fetched from the XOR-swizzled source slot.  Waves 0 / 1 add query rows 0-15 / 16-31.
  const int st_row_in_piece = lane >> 2;
  const uint32_t st_slot16 = (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  uint32_t st_off[4];
  auto set_offsets = [&](int64_t row0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int64_t r = wave * 64 + j * 16 + st_row_in_piece;
      if (row0 + r >= p.rows) r = p.rows - 1 - row0;  // stay in bounds; masked in the epilogue
      st_off[j] = (uint32_t)r * (uint32_t)row_bytes + st_slot16;
    }
  };
  const uint32_t st_off_b = (uint32_t)(wave * 16 + st_row_in_piece) * (uint32_t)row_bytes + st_slot16;
  const bool stages_b = wave < G::B_WAVES;
  int st_tile = 0, st_kt = 0, st_slot = 0;
  set_offsets(r_begin);

  auto stage_next = [&]() {
    const int tile = st_tile < n_tiles ? st_tile : n_tiles - 1;  // past the end: harmless reloads of the last tile
    const int64_t src_row0 = (ABL & 4) ? 0 : r_begin + (int64_t)tile * BM;
    const char* ga = sgpr_ptr(corpus + (size_t)src_row0 * row_bytes + (size_t)st_kt * 64);
    unsigned char* la = smem + st_slot * S_SLOT_A + wave * 4096;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((global_void*)(ga + (size_t)st_off[j]), (lds_void*)(la + j * 1024), 16, 0, 0);  // (a non-temporal policy here measured 30 % slower)
    if (stages_b) {
      const char* gb = sgpr_ptr(qbase + (size_t)st_kt * 64);
      unsigned char* lb = smem + S_B_RING + st_slot * S_SLOT_B + wave * 1024;
      __builtin_amdgcn_global_load_lds((global_void*)(gb + (size_t)st_off_b), (lds_void*)lb, 16, 0, 0);
      if constexpr (!F32) {  // the low plane of the split queries
        const char* gl = sgpr_ptr(qbase + plane_bytes + (size_t)st_kt * 64);
        __builtin_amdgcn_global_load_lds((global_void*)(gl + (size_t)st_off_b), (lds_void*)(lb + G::PLANE_B), 16, 0, 0);
      }
    }
    if (++st_slot == S_RING) st_slot = 0;
    if (++st_kt == steps_per_tile) {
      st_kt = 0;
      ++st_tile;
      if (st_tile < n_tiles) set_offsets(r_begin + (int64_t)st_tile * BM);
    }
  };
  auto wait_landed = [&]() {  // all but the newest S_RING - 2 steps of this wave's loads have landed
    if (stages_b)
      wait_vmcnt<(4 + G::PLANES) * (S_RING - 2)>();
    else
      wait_vmcnt<4 * (S_RING - 2)>();
  };

  // ---- fragment addresses: row (lane & 31) of a 32-row block, logical 16-byte slot 2 * g + (lane >> 5)
  const int frag_row = lane & 31;
  const uint32_t frag_x = (uint32_t)(((lane >> 5) ^ ((frag_row >> 2) & 3)) << 4);
  const uint32_t a_lane = (uint32_t)((wave * 64 + frag_row) * 64);  // + mi * 2048
  const uint32_t b_lane = (uint32_t)(S_B_RING + frag_row * 64);

  // ---- prologue: S_RING - 1 steps in flight
#pragma unroll 1
  for (int i = 0; i < S_RING - 1; ++i) stage_next();

  int rd = 0;
  for (int tile = 0; tile < n_tiles; ++tile) {
    const int64_t row0 = r_begin + (int64_t)tile * BM;
    f32x16 acc[2][NI];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

#pragma unroll 1
    for (int kt = 0; kt < steps_per_tile; ++kt) {
      if constexpr ((ABL & 2) == 0) wait_landed();  // this wave's share of step S is in LDS
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TAVB_BARRIER();  // the query piece of step S is visible; everybody is done with the slot of step S - 1
      if constexpr ((ABL & 2) == 0) stage_next();  // step S + S_RING - 1 -> the slot of step S - 1
      const unsigned char* abase = smem + rd * S_SLOT_A;
      const unsigned char* bbase = smem + rd * S_SLOT_B;
      f32x4 af[2][2], bf[2][NI], bl[2][NI];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint32_t kx = (uint32_t)(g << 5) ^ frag_x;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bf[g][ni] = *reinterpret_cast<const f32x4*>(bbase + (b_lane + kx) + ni * 2048);
          if constexpr (!F32) bl[g][ni] = *reinterpret_cast<const f32x4*>(bbase + (b_lane + kx) + ni * 2048 + G::PLANE_B);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) af[g][mi] = *reinterpret_cast<const f32x4*>(abase + (a_lane + kx) + mi * 2048);
      }
      if constexpr ((ABL & 1) == 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if constexpr (F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                  acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g][mi][e], bf[g][ni][e], acc[mi][ni], 0, 0, 0);
          } else {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
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
        asm volatile("" ::"v"(af[0][0]), "v"(af[1][1]), "v"(bf[0][0]), "v"(bf[1][NI - 1]));
        if constexpr (!F32) asm volatile("" ::"v"(bl[0][0]), "v"(bl[1][NI - 1]));
      }
      if (++rd == S_RING) rd = 0;
    }

    // ---- epilogue: admission test on the raw dot products, append (see variant 3)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int ql = ni * 32 + (lane & 31);
      const float thr = thr_lds[ql];
      const float thr_pre = fmaf(thr, 2.0f, -1.0f) - 4.8e-7f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        float top = acc[mi][ni][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) top = __builtin_fmaxf(top, acc[mi][ni][r]);
        const bool any = (ABL == 0) && (top > thr_pre);
        if constexpr (ABL != 0) asm volatile("" ::"v"(acc[mi][ni]));
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
          float sc[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[r] = fmaf(acc[mi][ni][r], 0.5f, 0.5f);
          const int64_t row_base = row0 + wave * 64 + mi * 32 + 4 * (lane >> 5);
          unsigned admit = 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float s1 = sc[r];
            s1 = (s1 > 0.0f) ? s1 : 0.0f;
            s1 = (s1 > 1.0f) ? 1.0f : s1;
            const bool ok = (sc[r] > thr) && (row_base + (r & 3) + 8 * (r >> 2) < r_end) && (s1 >= p.min_score);
            admit |= ok ? (1u << r) : 0u;
          }
          const int n_adm = __popc(admit);
          int pos = 0;
          if (n_adm > 0) {
            pos = lds_add_rtn(&cnt_lds[ql], n_adm);
            if (pos + n_adm > CAP - BM) lds_store_i32(need_compact, 1);  // this buffer could overflow on the next tile
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if ((admit >> r) & 1u) {
              float s1 = sc[r];
              s1 = (s1 > 0.0f) ? s1 : 0.0f;
              s1 = (s1 > 1.0f) ? 1.0f : s1;
              if (pos < CAP)
                my_cand[(size_t)ql * CAP + pos] = make_key(s1, (uint32_t)(row_base + (r & 3) + 8 * (r >> 2)) + p.index_base);
              ++pos;
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
      for (int q = wave; q < SQ; q += S_THREADS / 64) {
        const int n = cnt_lds[q];
        if (n > CAP - BM) {
          u64* buf = my_cand + (size_t)q * CAP;
          const WaveTopK<1> best = best_of_buffer(buf, n < CAP ? n : CAP, lane);
          if (lane < p.k) buf[lane] = best.key[0];
          const int kept = __popcll(__ballot(best.key[0] != 0ull && lane < p.k));
          const u64 kth = best.at(p.k - 1);
          if (lane == 0) {
            cnt_lds[q] = kept;
            const float kth_score = __uint_as_float((uint32_t)(kth >> 32));
            if (kth != 0ull && kth_score > thr_lds[q]) thr_lds[q] = kth_score;
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

  for (int q = wave; q < SQ; q += S_THREADS / 64) {
    const int qg = qtile * SQ + q;
    if (qg >= p.nq) continue;
    const int n = cnt_lds[q];
    const WaveTopK<1> best = best_of_buffer(my_cand + (size_t)q * CAP, n < CAP ? n : CAP, lane);
    u64* out = p.lists + ((size_t)qg * p.list_stride + split) * (size_t)p.k;
    if (lane < p.k) out[lane] = best.key[0];
  }
}

}  // namespace

// thr[q] = the largest float below the k-th best score of the sample pass (so that `score > thr` admits
// every row scoring >= that k-th best), or -inf when the sample did not yield k hits.
__global__ void sample_threshold_kernel(const u64* __restrict__ keys, int nq, int k, float* __restrict__ thr) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const u64 kth = keys[(size_t)q * k + (k - 1)];
  float t = -__builtin_inff();
  if (kth != 0ull) {
    const uint32_t bits = (uint32_t)(kth >> 32);
    t = bits ? __uint_as_float(bits - 1u) : -__builtin_inff();
  }
  thr[q] = t;
}

hipError_t launch_sample_thresholds(const unsigned long long* keys, int nq, int k, float* thr, hipStream_t stream) {
  hipLaunchKernelGGL(sample_threshold_kernel, dim3((nq + 255) / 256), dim3(256), 0, stream, keys, nq, k, thr);
  return hipGetLastError();
}

int mfma_query_tile() { return BN; }

bool mfma_supported(int dim, int k) { return dim % BK == 0 && dim >= BK && dim <= 16384 && k >= 1 && k <= 64; }

int mfma_pick_splits(int64_t rows, int nq_padded, int n_cu) {
  // One workgroup per CU and all of them resident at once: the grid is (groups of 8 row ranges) x query tiles x 8, so the
  // number of row ranges is a multiple of 8 with groups * n_qtiles * 8 <= n_cu.  (85 ranges for 3 query tiles made 264
  // workgroups on 256 CUs: a second scheduling round for the last 8, and a 768-query batch slower than a 1024-query one.)
  const int n_qtiles = nq_padded / BN > 0 ? nq_padded / BN : 1;
  int splits = (n_cu / (8 * n_qtiles)) * 8;
  if (splits < 8) splits = 8;
  const int64_t tiles = (rows + BM - 1) / BM;
  if (splits > tiles) splits = (int)tiles;
  return splits;
}

static size_t mfma_cand_bytes(int n_splits, int nq_padded) {
  return (size_t)n_splits * (size_t)nq_padded * CAP * sizeof(u64);  // nq_padded = tiles x queries per tile (256 or 32)
}

size_t mfma_workspace_bytes(int n_splits, int nq_padded) {
  // candidate buffers, then one rendezvous counter per row range (padded to 256 bytes)
  return mfma_cand_bytes(n_splits, nq_padded) + (((size_t)n_splits * sizeof(int) + 255) & ~(size_t)255);
}

hipError_t launch_mfma_scan(const MfmaParams& p, hipStream_t stream) {
  if (!mfma_supported(p.dim, p.k) || p.nq_padded % BN != 0 || p.n_splits < 1) return hipErrorInvalidValue;
  MfmaDeviceParams d{};
  d.corpus = reinterpret_cast<const _Float16*>(p.corpus);
  d.queries = reinterpret_cast<const _Float16*>(p.queries);
  d.lists = p.lists;
  d.rows = p.rows;
  d.dim = p.dim;
  d.nq = p.nq;
  d.n_qtiles = p.nq_padded / BN;
  d.n_splits = p.n_splits;
  d.list_stride = p.list_stride > p.n_splits ? p.list_stride : p.n_splits;
  d.k = p.k;
  d.index_base = p.index_base;
  d.min_score = p.min_score;
  d.group_sel = p.group_sel;
  d.a_tiled = p.a_tiled;
  d.thr_in = p.thr_in;
  const int64_t per = (p.rows + p.n_splits - 1) / p.n_splits;
  const int bm = (p.variant == 5) ? BM5 : BM;
  d.rows_per_split = ((per + bm - 1) / bm) * bm;
  if (!p.workspace) return hipErrorInvalidValue;
  d.cand = p.workspace;
  d.sync = nullptr;
  if (p.rendezvous && p.variant == 3 && d.n_qtiles > 1) {
    d.sync = reinterpret_cast<int*>(reinterpret_cast<char*>(p.workspace) + mfma_cand_bytes(p.n_splits, p.nq_padded));
    hipError_t e = hipMemsetAsync(d.sync, 0, (size_t)p.n_splits * sizeof(int), stream);
    if (e != hipSuccess) return e;
  }
  const int groups = (p.n_splits + 7) / 8;
  const int grid = groups * d.n_qtiles * 8;
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), LDS_BYTES, stream, d);
    return hipGetLastError();
  };
  // grid: groups of 8 consecutive block ids = 8 different row ranges (one per XCD)
  if (p.variant == 4) {
    constexpr int NA4 = 6, NB4 = 3;
    constexpr int LDS4 = (NA4 + NB4) * 16384 + BN * 8 + 16;
    auto go4 = [&](auto kern) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), LDS4, stream, d);
      return hipGetLastError();
    };
    switch (p.ablate) {
      case 1: return go4(mfma_scan_kernel_v4<NA4, NB4, 1>);
      case 2: return go4(mfma_scan_kernel_v4<NA4, NB4, 2>);
      case 3: return go4(mfma_scan_kernel_v4<NA4, NB4, 3>);
      case 4: return go4(mfma_scan_kernel_v4<NA4, NB4, 4>);
      case 512: return go4(mfma_scan_kernel_v4<NA4, NB4, 512>);  // admission test computed, slow path never taken
      case 256: return go4(mfma_scan_kernel_v4<NA4, NB4, 256>);  // everything except the admission test / appends
      case 34: return go4(mfma_scan_kernel_v4<NA4, NB4, 34>);  // MFMAs + barriers only
      case 32: return go4(mfma_scan_kernel_v4<NA4, NB4, 32>);  // MFMAs + LDS-DMA, no fragment reads
      case 36: return go4(mfma_scan_kernel_v4<NA4, NB4, 36>);  // same, corpus tile 0 only (L2 resident)
      case 33: return go4(mfma_scan_kernel_v4<NA4, NB4, 33>);  // LDS-DMA + barriers only
      case 37: return go4(mfma_scan_kernel_v4<NA4, NB4, 37>);  // same, L2 resident
      case 10: return go4(mfma_scan_kernel_v4<NA4, NB4, 10>);  // no LDS-DMA, barrier every 2nd step
      case 18: return go4(mfma_scan_kernel_v4<NA4, NB4, 18>);  // no LDS-DMA, barrier every 4th step
      default: return go4(mfma_scan_kernel_v4<NA4, NB4, 0>);
    }
  }
  if (p.variant == 5) {
    if (p.a_tiled) return hipErrorInvalidValue;  // row-major operand only
    auto go5 = [&](auto kern) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS5);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(NT5), LDS5, stream, d);
      return hipGetLastError();
    };
    switch (p.ablate) {
      case 1: return go5(mfma_scan_kernel_v5<1>);      // no MFMAs
      case 2: return go5(mfma_scan_kernel_v5<2>);      // no LDS-DMA after the prologue
      case 256: return go5(mfma_scan_kernel_v5<256>);  // everything except admissions
      case 260: return go5(mfma_scan_kernel_v5<260>);  // same, corpus tile 0 re-read by every block (L2 resident)
      case 288: return go5(mfma_scan_kernel_v5<288>);  // same as 256 without the fragment reads
      case 512: return go5(mfma_scan_kernel_v5<512>);  // admission test computed, slow path never taken
      case 1280: return go5(mfma_scan_kernel_v5<1280>);  // 256 without the counted vmcnt wait (garbage: timing only)
      case 3328: return go5(mfma_scan_kernel_v5<3328>);  // ... and without the barrier
      default: return go5(mfma_scan_kernel_v5<0>);
    }
  }
  if (p.variant == 3) {
    constexpr int NA3 = 6, NB3 = 3;
    constexpr int LDS3 = (NA3 + NB3) * 16384 + BN * 8 + 16;
    auto go3 = [&](auto kern) -> hipError_t {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS3);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), LDS3, stream, d);
      return hipGetLastError();
    };
    switch (p.ablate) {
      case 1: return go3(mfma_scan_kernel_v3<NA3, NB3, 1>);
      case 4: return go3(mfma_scan_kernel_v3<NA3, NB3, 4>);
      case 5: return go3(mfma_scan_kernel_v3<NA3, NB3, 5>);
      case 256: return go3(mfma_scan_kernel_v3<NA3, NB3, 256>);  // everything except admissions
      case 2: return go3(mfma_scan_kernel_v3<NA3, NB3, 2>);
      case 3: return go3(mfma_scan_kernel_v3<NA3, NB3, 3>);
      default: return p.a_nt ? go3(mfma_scan_kernel_v3<NA3, NB3, 0, 2>) : go3(mfma_scan_kernel_v3<NA3, NB3, 0>);
    }
  }
  if (p.variant == 1) return go(mfma_scan_kernel<1, 0, 0>);
  const int sel = p.ablate * 4 + p.prio;
  switch (sel) {
    case 0 * 4 + 0: return go(mfma_scan_kernel<2, 0, 0>);
    case 0 * 4 + 1: return go(mfma_scan_kernel<2, 0, 1>);
    case 0 * 4 + 2: return go(mfma_scan_kernel<2, 0, 2>);
    case 1 * 4 + 0: return go(mfma_scan_kernel<2, 1, 0>);
    case 2 * 4 + 0: return go(mfma_scan_kernel<2, 2, 0>);
    case 2 * 4 + 1: return go(mfma_scan_kernel<2, 2, 1>);
    case 2 * 4 + 2: return go(mfma_scan_kernel<2, 2, 2>);
    case 3 * 4 + 0: return go(mfma_scan_kernel<2, 3, 0>);
    default: return go(mfma_scan_kernel<2, 0, 0>);
  }
}

int skinny_query_tile(int nq) { return nq > SQ32 ? 2 * SQ32 : SQ32; }  // 64-query tiles for batches of 33 and more

bool skinny_supported(int dim, int k, bool f32) {
  return (dim * (f32 ? 4 : 2)) % 64 == 0 && dim > 0 && k >= 1 && k <= 64;
}

int skinny_pick_splits(int64_t rows, int nq_padded, int tile, int n_cu) {
  const int n_qtiles = nq_padded / tile;
  int splits = (2 * n_cu) / (n_qtiles > 0 ? n_qtiles : 1);  // two workgroups per CU
  splits = (splits / 8) * 8;                                 // whole groups of 8 (one row range per XCD)
  if (splits < 8) splits = 8;
  const int64_t tiles = (rows + BM - 1) / BM;
  if (splits > tiles) splits = (int)tiles;
  return splits;
}

// Same contract as launch_mfma_scan.  p.queries: fp32 corpus -> [nq_padded, dim] fp32; fp16 corpus -> [2, nq_padded, dim]
// fp16, the high and the low plane of the split fp32 queries (launch_f32_split_f16).  nq_padded is a multiple of the
// tile (p.skinny_tile = 32 or 64 queries).
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
  d.cand = p.workspace;
  d.sync = nullptr;
  const int64_t per = (p.rows + p.n_splits - 1) / p.n_splits;
  d.rows_per_split = ((per + BM - 1) / BM) * BM;
  const int groups = (p.n_splits + 7) / 8;
  const int grid = groups * d.n_qtiles * 8;
  auto go = [&](auto kern, int lds) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S_THREADS), lds, stream, d);
    return hipGetLastError();
  };
  if (f32) {
    if (tile == 64) return go(skinny_scan_kernel<float, 2, 0>, SkinnyGeom<float, 2>::LDS);
    return go(skinny_scan_kernel<float, 1, 0>, SkinnyGeom<float, 1>::LDS);
  }
  if (tile == 64) return go(skinny_scan_kernel<_Float16, 2, 0>, SkinnyGeom<_Float16, 2>::LDS);
  return go(skinny_scan_kernel<_Float16, 1, 0>, SkinnyGeom<_Float16, 1>::LDS);
}

}  // namespace tavb
