// Device-side building blocks shared by the gfx950 kernels: packed result keys,
// the lane-distributed sorted top-k list of one wavefront, and bitonic merging
// of such lists.  CDNA4 only: a wavefront is 64 lanes and that is hard-coded.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tavb {

typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// score + key
// ---------------------------------------------------------------------------
// The reference's public score: clip((cos + 1) / 2, 0, 1) in float32
// (vectorbase.py:44-47).  One f32 add (rounded), an exact halving, a clamp that
// keeps NaN as NaN (np.clip propagates NaN; the row is then dropped by `>=`).
__device__ __forceinline__ float cosine_to_score(float c) {
  float s = (c + 1.0f) * 0.5f;
  if (!(s > 0.0f)) s = (s != s) ? s : 0.0f;  // negative or -0 -> +0, NaN stays
  if (s > 1.0f) s = 1.0f;
  return s;
}

// key = (score bits << 32) | (0xFFFFFFFF - index).  Scores are in [0, 1] so their
// bit patterns order like the floats; a bigger key is a better hit and, among
// equal scores, the smaller index.  0 is the empty slot (no real key is 0 because
// index 0xFFFFFFFF is never used).
__device__ __forceinline__ u64 make_key(float score, uint32_t index) {
  return ((u64)__float_as_uint(score) << 32) | (u64)(0xFFFFFFFFu - index);
}

__device__ __forceinline__ u64 shfl_u64(u64 v, int src_lane) {
  int lo = __shfl((int)(uint32_t)v, src_lane, kWave);
  int hi = __shfl((int)(uint32_t)(v >> 32), src_lane, kWave);
  return ((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo;
}

// Value of lane (l ^ M).  Strides below 16 stay inside a 16-lane DPP row and use DPP moves (VALU,
// a few cycles of latency); 16 and 32 cross rows and go through ds_bpermute (~100 cycles).  The bitonic
// stages that dominate merging and sorting are chains of such exchanges, so latency is what counts.
template <int M>
__device__ __forceinline__ int xor_lane_i32(int x, int lane) {
  if constexpr (M == 1) {
    return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
  } else if constexpr (M == 2) {
    return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false);  // quad_perm [2,3,0,1]
  } else if constexpr (M == 4) {
    // banks (groups of 4 lanes) 0 and 2 read 4 lanes up, banks 1 and 3 read 4 lanes down
    int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);  // row_shl:4 -> lane i <- lane i+4
    t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);      // row_shr:4 -> lane i <- lane i-4
    return t;
  } else if constexpr (M == 8) {
    return __builtin_amdgcn_update_dpp(x, x, 0x128, 0xF, 0xF, false);  // row_ror:8 == lane ^ 8 within the row
  } else {
    return __shfl(x, lane ^ M, kWave);
  }
}

template <int M>
__device__ __forceinline__ u64 xor_lane_u64(u64 v, int lane) {
  const int lo = xor_lane_i32<M>((int)(uint32_t)v, lane);
  const int hi = xor_lane_i32<M>((int)(uint32_t)(v >> 32), lane);
  return ((u64)(uint32_t)hi << 32) | (u64)(uint32_t)lo;
}

__device__ __forceinline__ u64 readlane_u64(u64 v, int lane /* wave-uniform */) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
  return ((u64)hi << 32) | (u64)lo;
}

// Sum over the 64 lanes, returned wave-uniform.  Four DPP adds (quad swaps, half-row mirror, row mirror:
// full-rate VALU, no LDS crossbar) leave each 16-lane row holding its row sum; the four row sums are
// read with v_readlane and added.  Fixed summation order, identical in every lane.
template <int CTRL>
__device__ __forceinline__ float dpp_add_t(float v) {
  const int x = __float_as_int(v);
  return v + __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_add_t<0xB1>(v);   // quad_perm [1,0,3,2]  (lane ^ 1)
  v = dpp_add_t<0x4E>(v);   // quad_perm [2,3,0,1]  (lane ^ 2)
  v = dpp_add_t<0x141>(v);  // row_half_mirror: the two quads of each 8 lanes
  v = dpp_add_t<0x140>(v);  // row_mirror: the two halves of each 16-lane row
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}

__device__ __forceinline__ float wave_uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// ---------------------------------------------------------------------------
// One wavefront's running top-K, K = 64 * KPL, kept sorted best-first and spread
// over the lanes: rank r lives in slot r / 64 of lane r % 64.
// ---------------------------------------------------------------------------
template <int KPL>
struct WaveTopK {
  u64 key[KPL];

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int s = 0; s < KPL; ++s) key[s] = 0;
  }

  // key at rank r (r wave-uniform).  Every slot is read with v_readlane and the
  // choice is made on the scalar results: selecting key[r >> 6] first would make
  // the compiler spill key[] to scratch to index it dynamically.
  __device__ __forceinline__ u64 at(int r) const {
    const int slot = r >> 6, ln = r & 63;
    u64 v = readlane_u64(key[0], ln);
#pragma unroll
    for (int s = 1; s < KPL; ++s) {
      const u64 t = readlane_u64(key[s], ln);
      v = (slot == s) ? t : v;
    }
    return v;
  }

  // Insert the wave-uniform candidate c (distinct from every stored key): every
  // worse key moves one rank down, the worst falls off the end.
  __device__ __forceinline__ void insert(u64 c, int lane) {
    u64 carry = ~0ull;  // what sits "above" rank 0: better than anything
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      const u64 mine = key[s];
      u64 up = shfl_u64(mine, (lane + 63) & 63);  // lane i <- lane i-1
      const u64 last = readlane_u64(mine, 63);
      if (lane == 0) up = carry;
      key[s] = (mine > c) ? mine : ((up > c) ? c : up);
      carry = last;
    }
  }

  // a <- best K of (a U b); both sorted best-first; b is given REVERSED, i.e.
  // brev.key[s] of lane l holds b's rank (K-1) - (s*64 + l).
  __device__ __forceinline__ void merge_reversed(const WaveTopK<KPL>& brev, int lane) {
    // max(a[r], b[K-1-r]) is a bitonic sequence holding the K best of the union
#pragma unroll
    for (int s = 0; s < KPL; ++s) key[s] = (key[s] > brev.key[s]) ? key[s] : brev.key[s];
    // bitonic merge, descending.  Strides >= 64 pair slots of the same lane
    // (written out per KPL: a runtime-indexed key[] would be demoted to scratch).
    auto order = [&](u64& a, u64& b) {
      const u64 hi = (a > b) ? a : b;
      const u64 lo = (a > b) ? b : a;
      a = hi;
      b = lo;
    };
    static_assert(KPL == 1 || KPL == 2 || KPL == 4, "KPL must be 1, 2 or 4");
    if constexpr (KPL == 4) {
      order(key[0], key[2]);
      order(key[1], key[3]);
      order(key[0], key[1]);
      order(key[2], key[3]);
    } else if constexpr (KPL == 2) {
      order(key[0], key[1]);
    }
    exchange_stage<32>(lane);
    exchange_stage<16>(lane);
    exchange_stage<8>(lane);
    exchange_stage<4>(lane);
    exchange_stage<2>(lane);
    exchange_stage<1>(lane);
  }

  // one stage of the descending bitonic merge: lanes l and l ^ M keep the larger / smaller key
  template <int M>
  __device__ __forceinline__ void exchange_stage(int lane) {
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      const u64 mine = key[s];
      const u64 other = xor_lane_u64<M>(mine, lane);
      const bool keep_big = (lane & M) == 0;
      const bool mine_big = mine > other;
      key[s] = (keep_big == mine_big) ? mine : other;
    }
  }

  // Load a stored list of `k` keys (best first) reversed and zero-extended to K,
  // ready for merge_reversed.
  __device__ __forceinline__ void load_reversed(const u64* list, int k, int lane) {
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      const int r = (KPL * 64 - 1) - (s * 64 + lane);
      key[s] = (r < k) ? list[r] : 0ull;
    }
  }

  __device__ __forceinline__ void load(const u64* list, int k, int lane) {
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      const int r = s * 64 + lane;
      key[s] = (r < k) ? list[r] : 0ull;
    }
  }

  __device__ __forceinline__ void store(u64* list, int k, int lane) const {
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      const int r = s * 64 + lane;
      if (r < k) list[r] = key[s];
    }
  }
};

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

// Merge the sorted lists of all waves of a workgroup into wave 0's list through
// LDS (`scratch`: (waves/2) * 64*KPL keys).  Every wave must call this.
template <int KPL>
__device__ __forceinline__ void block_merge(WaveTopK<KPL>& mine, u64* scratch, int wave, int n_waves, int lane) {
  constexpr int K = 64 * KPL;
  for (int stride = 1; stride < n_waves; stride <<= 1) {
    const bool sender = (wave & (2 * stride - 1)) == stride;
    const bool receiver = (wave & (2 * stride - 1)) == 0 && (wave + stride) < n_waves;
    __syncthreads();  // previous level's readers are done with scratch
    if (sender) mine.store(scratch + (size_t)(wave / (2 * stride)) * K, K, lane);
    __syncthreads();
    if (receiver) {
      WaveTopK<KPL> other;
      other.load_reversed(scratch + (size_t)(wave / (2 * stride)) * K, K, lane);
      mine.merge_reversed(other, lane);
    }
  }
}

}  // namespace tavb
