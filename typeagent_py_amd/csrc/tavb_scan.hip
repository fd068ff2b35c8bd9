// Streaming scan: query . corpus-row dot products, score, threshold and top-k
// selection fused into one pass over the row-major corpus (replaces
// np.dot + cosine_to_score + flatnonzero + argpartition + argsort of
// src/typeagent/aitools/vectorbase.py:176-187 and, with a gather list, :217-227).
//
// HBM-bound by construction: every corpus byte is read exactly once with 16-byte
// per-lane loads (one wave instruction = 1 KiB contiguous), nothing proportional
// to N is written (the f32[N] score vector of the reference never exists), and
// the per-row work besides the FMAs is one wave reduction and one compare.
//
// Work split: wave w of W owns row groups w, w+W, ... (interleaved so that the
// set of rows in flight is a compact window of the corpus).  Each wave keeps its
// best-K in registers (WaveTopK); at the end the waves of a workgroup merge
// through LDS and the workgroup writes ONE sorted list per query; a tiny second
// kernel (tavb_merge.hip) merges the per-workgroup lists.
//
// Three kernel families:
//   tier 1  scan_fixed : dim == CH * 64 lanes * 16 B  (1536: CH=6 f32 / CH=3 f16),
//                        single query held in registers, fully unrolled
//   tier 2  scan_vec   : dim a multiple of 16 B, runtime chunk loop, queries in LDS
//   tier 3  scan_scalar: any dim, element loads (tiny / odd dims of the API tests)

#include <cstring>

#include "tavb_device.h"
#include "tavb_internal.h"

namespace tavb {

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int EPL = 4;  // elements in a lane's 16-byte slice
};
template <>
struct Elem<_Float16> {
  static constexpr int EPL = 8;
};

template <bool NT>
__device__ __forceinline__ f32x4 ld16(const void* p) {
  const f32x4* q = reinterpret_cast<const f32x4*>(p);
  if constexpr (NT)
    return __builtin_nontemporal_load(q);
  else
    return *q;
}

// acc += <lane's 16-byte corpus slice, matching query slice>; f16 values are
// widened exactly to f32, products and sums are f32 FMAs.
template <typename T>
__device__ __forceinline__ float dot_slice(f32x4 raw, const float* __restrict__ qf, float acc);

template <>
__device__ __forceinline__ float dot_slice<float>(f32x4 raw, const float* __restrict__ qf, float acc) {
  acc = fmaf(raw.x, qf[0], acc);
  acc = fmaf(raw.y, qf[1], acc);
  acc = fmaf(raw.z, qf[2], acc);
  acc = fmaf(raw.w, qf[3], acc);
  return acc;
}

template <>
__device__ __forceinline__ float dot_slice<_Float16>(f32x4 raw, const float* __restrict__ qf, float acc) {
  const f16x8 h = __builtin_bit_cast(f16x8, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc = fmaf((float)h[i], qf[i], acc);
  return acc;
}

// Per-wave selection state for NQ queries.
template <int NQ, int KPL>
struct Selector {
  WaveTopK<KPL> top[NQ];
  u64 thr[NQ];  // key at rank k-1: a candidate must beat it (wave-uniform)

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      top[q].clear();
      thr[q] = 0;
    }
  }

  // `dot` = the lane-partial sum; the candidate index must be wave-uniform
  __device__ __forceinline__ void offer(int q, float partial, uint32_t index, float min_score, int k, int lane,
                                        u64 bound) {
    const float dot = wave_sum(partial);
    const float s = wave_uniform(cosine_to_score(dot));
    if (s >= min_score) {  // NaN never passes, like numpy's >=
      const u64 c = make_key(s, index);
      if (c > thr[q] && c < bound) {
        top[q].insert(c, lane);
        thr[q] = top[q].at(k - 1);
      }
    }
  }
};

// The queries of this workgroup: all of them (plain form), or group blockIdx.y of the grouped form (ScanParams::group).
struct QueryGroup {
  int q0;  // first query
  int n;   // queries of this workgroup, 1 .. NQ
};
template <int NQ>
__device__ __forceinline__ QueryGroup query_group(const ScanParams& p) {
  QueryGroup g;
  g.q0 = p.group > 0 ? (int)blockIdx.y * p.group : 0;
  g.n = (p.nq - g.q0 < NQ) ? p.nq - g.q0 : NQ;
  return g;
}

template <int NQ, int KPL>
__device__ __forceinline__ void finish_block(Selector<NQ, KPL>& sel, const ScanParams& p, const QueryGroup& qg, u64* scratch, int wave,
                                             int n_waves, int lane) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q < qg.n) {  // uniform
      block_merge<KPL>(sel.top[q], scratch, wave, n_waves, lane);
      if (wave == 0) sel.top[q].store(p.lists + ((size_t)(qg.q0 + q) * gridDim.x + blockIdx.x) * (size_t)p.k, p.k, lane);
    }
  }
}

// ---------------------------------------------------------------------------
// tier 1: dim fixed at compile time
// ---------------------------------------------------------------------------
// `QA` = InlineQuery: the (single, 1536-wide) query travels inside the kernel arguments instead of through a device buffer -- a small-corpus
// lookup then is ONE submission (no hipMemcpyAsync in front of the launch: 17.1 -> ~13.4 us of submit + synchronize,
// profiles/r04_latency_small.md); every wave reads its slice out of the kernarg segment once.
struct InlineQuery {
  float v[1536];
};
struct NoInlineQuery {};

template <typename T, int CH, int NQ, int KPL, int U, bool NT, bool PIPE, int MAXT, typename QA = NoInlineQuery>
__global__ void __launch_bounds__(MAXT) scan_fixed_kernel(const ScanParams p, const QA qa) {
  constexpr int EPL = Elem<T>::EPL;
  constexpr int D = CH * 64 * EPL;
  constexpr bool QREG = (NQ == 1);
  extern __shared__ __align__(16) unsigned char smem[];
  float* qlds = reinterpret_cast<float*>(smem);                                        // [NQ][D] when !QREG
  u64* scratch = reinterpret_cast<u64*>(smem + (QREG ? 0 : (size_t)NQ * D * sizeof(float)));

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  const int64_t n_pos = p.n_pos;
  const int64_t stride = (int64_t)gridDim.x * n_waves * U;
  const int k = p.k;
  const QueryGroup qg = query_group<NQ>(p);
  const float* queries = p.queries + (size_t)qg.q0 * D;

  float qreg[QREG ? CH : 1][EPL];
  if constexpr (QREG) {
#pragma unroll
    for (int c = 0; c < CH; ++c)
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        if constexpr (sizeof(QA) == sizeof(InlineQuery))
          qreg[c][e] = qa.v[(c * 64 + lane) * EPL + e];
        else
          qreg[c][e] = queries[(c * 64 + lane) * EPL + e];
      }
  } else {
    for (int i = threadIdx.x; i < NQ * D; i += blockDim.x) {
      const int q = i / D;
      const int src = (q < qg.n) ? q : (qg.n - 1);
      qlds[i] = queries[(size_t)src * D + (i - q * D)];
    }
    __syncthreads();
  }
  float minsc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) minsc[q] = (q < qg.n) ? p.min_score[qg.q0 + q] : __builtin_inff();  // never passes

  Selector<NQ, KPL> sel;
  sel.clear();

  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const int32_t* row_ids = p.row_ids;

  auto load_rows = [&](f32x4(&x)[U][CH], int64_t base) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pos = base + u;
      if (pos < n_pos) {  // wave-uniform
        const int64_t row = row_ids ? (int64_t)row_ids[pos] : pos;
        const char* rp = corpus + row * (int64_t)(D * sizeof(T)) + lane * 16;
#pragma unroll
        for (int c = 0; c < CH; ++c) x[u][c] = ld16<NT>(rp + c * 1024);
      } else {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[u][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  auto reduce_rows = [&](const f32x4(&x)[U][CH], int64_t base) {
    float acc[U][NQ];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[u][q] = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float qf[EPL];
        if constexpr (QREG) {
#pragma unroll
          for (int e = 0; e < EPL; ++e) qf[e] = qreg[c][e];
        } else {
          const f32x4* src = reinterpret_cast<const f32x4*>(qlds + (size_t)q * D + (c * 64 + lane) * EPL);
#pragma unroll
          for (int v = 0; v < EPL / 4; ++v) {
            const f32x4 t = src[v];
            qf[4 * v + 0] = t.x;
            qf[4 * v + 1] = t.y;
            qf[4 * v + 2] = t.z;
            qf[4 * v + 3] = t.w;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u][q] = dot_slice<T>(x[u][c], qf, acc[u][q]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + u < n_pos) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          sel.offer(q, acc[u][q], (uint32_t)(base + u) + p.index_base, minsc[q], k, lane, p.key_bound);
      }
    }
  };

  int64_t base = ((int64_t)blockIdx.x * n_waves + wave) * U;
  if constexpr (PIPE) {
    f32x4 cur[U][CH], nxt[U][CH];
    if (base < n_pos) load_rows(cur, base);
    while (base < n_pos) {
      const int64_t next = base + stride;
      if (next < n_pos) load_rows(nxt, next);
      reduce_rows(cur, base);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int c = 0; c < CH; ++c) cur[u][c] = nxt[u][c];
      base = next;
    }
  } else {
    for (; base < n_pos; base += stride) {
      f32x4 x[U][CH];
      load_rows(x, base);
      reduce_rows(x, base);
    }
  }

  finish_block<NQ, KPL>(sel, p, qg, scratch, wave, n_waves, lane);
}

// ---------------------------------------------------------------------------
// tier 2: dim % (16 bytes) == 0, runtime chunk loop, queries staged in LDS
// ---------------------------------------------------------------------------
template <typename T, int NQ, int KPL, bool NT>
__global__ void __launch_bounds__(1024) scan_vec_kernel(const ScanParams p) {
  constexpr int EPL = Elem<T>::EPL;
  constexpr int U = 2;  // rows per wave iteration (4 rows measured no faster and spills at 128 VGPRs)
  extern __shared__ __align__(16) unsigned char smem[];
  const int D = p.dim;
  float* qlds = reinterpret_cast<float*>(smem);  // [NQ][D]
  u64* scratch = reinterpret_cast<u64*>(smem + (((size_t)NQ * D * sizeof(float) + 15) & ~(size_t)15));

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  const int64_t n_pos = p.n_pos;
  const int64_t stride = (int64_t)gridDim.x * n_waves * U;
  const int k = p.k;
  const int n_slices = D / EPL;  // 16-byte slices per row
  const QueryGroup qg = query_group<NQ>(p);
  const float* queries = p.queries + (size_t)qg.q0 * D;

  for (int i = threadIdx.x; i < NQ * D; i += blockDim.x) {
    const int q = i / D;
    const int src = (q < qg.n) ? q : (qg.n - 1);
    qlds[i] = queries[(size_t)src * D + (i - q * D)];
  }
  __syncthreads();
  float minsc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) minsc[q] = (q < qg.n) ? p.min_score[qg.q0 + q] : __builtin_inff();

  Selector<NQ, KPL> sel;
  sel.clear();
  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const int32_t* row_ids = p.row_ids;
  const int64_t row_bytes = (int64_t)D * sizeof(T);

  for (int64_t base = ((int64_t)blockIdx.x * n_waves + wave) * U; base < n_pos; base += stride) {
    const char* rp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t pos = (base + u < n_pos) ? base + u : base;  // past the end: re-read the first row, result unused
      const int64_t r = row_ids ? (int64_t)row_ids[pos] : pos;
      rp[u] = corpus + r * row_bytes;
    }
    float acc[U][NQ];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[u][q] = 0.f;
#pragma unroll 4
    for (int sl = lane; sl < n_slices; sl += 64) {
      f32x4 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u) x[u] = ld16<NT>(rp[u] + (size_t)sl * 16);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float qf[EPL];
        const f32x4* src = reinterpret_cast<const f32x4*>(qlds + (size_t)q * D + (size_t)sl * EPL);
#pragma unroll
        for (int v = 0; v < EPL / 4; ++v) {
          const f32x4 t = src[v];
          qf[4 * v + 0] = t.x;
          qf[4 * v + 1] = t.y;
          qf[4 * v + 2] = t.z;
          qf[4 * v + 3] = t.w;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u][q] = dot_slice<T>(x[u], qf, acc[u][q]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + u < n_pos) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          sel.offer(q, acc[u][q], (uint32_t)(base + u) + p.index_base, minsc[q], k, lane, p.key_bound);
      }
    }
  }
  finish_block<NQ, KPL>(sel, p, qg, scratch, wave, n_waves, lane);
}

// ---------------------------------------------------------------------------
// tier 3: any dim, element loads, queries read from global (L1/L2 resident)
// ---------------------------------------------------------------------------
template <typename T, int NQ, int KPL>
__global__ void __launch_bounds__(1024) scan_scalar_kernel(const ScanParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* scratch = reinterpret_cast<u64*>(smem);
  const int D = p.dim;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  const int64_t n_pos = p.n_pos;
  const int64_t stride = (int64_t)gridDim.x * n_waves;
  const int k = p.k;
  const QueryGroup qg = query_group<NQ>(p);
  const float* queries = p.queries + (size_t)qg.q0 * D;
  float minsc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) minsc[q] = (q < qg.n) ? p.min_score[qg.q0 + q] : __builtin_inff();
  Selector<NQ, KPL> sel;
  sel.clear();
  const T* corpus = reinterpret_cast<const T*>(p.corpus);
  const int32_t* row_ids = p.row_ids;
  for (int64_t pos = (int64_t)blockIdx.x * n_waves + wave; pos < n_pos; pos += stride) {
    const int64_t row = row_ids ? (int64_t)row_ids[pos] : pos;
    const T* rp = corpus + row * (int64_t)D;
    float acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
    for (int e = lane; e < D; e += 64) {
      const float x = (float)rp[e];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int src = (q < qg.n) ? q : (qg.n - 1);
        acc[q] = fmaf(x, queries[(size_t)src * D + e], acc[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) sel.offer(q, acc[q], (uint32_t)pos + p.index_base, minsc[q], k, lane, p.key_bound);
  }
  finish_block<NQ, KPL>(sel, p, qg, scratch, wave, n_waves, lane);
}

// ---------------------------------------------------------------------------
// emit: EVERY row with score >= min_score, unsorted, in one pass (the candidate stream of the predicate path,
// vectorbase.py:191-201, of max_hits > TAVB_MAX_FUSED_K and of the max_hits == 0 quirk -- the reference's
// `np.flatnonzero(scores >= min_score)`, :179).  A wave collects the keys of its passing rows across its lanes (slot i in
// lane i) and appends them 64 at a time: one atomic and one coalesced 512-byte store per 64 survivors.
// ---------------------------------------------------------------------------
template <typename T, bool VEC>
__global__ void __launch_bounds__(1024) scan_emit_kernel(const ScanParams p, u64* __restrict__ out, unsigned long long capacity,
                                                         unsigned long long* __restrict__ counter) {
  constexpr int EPL = Elem<T>::EPL;
  extern __shared__ __align__(16) unsigned char smem[];
  float* qlds = reinterpret_cast<float*>(smem);  // [D]
  const int D = p.dim;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  for (int i = threadIdx.x; i < D; i += blockDim.x) qlds[i] = p.queries[i];
  __syncthreads();
  const float min_score = p.min_score[0];
  const char* corpus = reinterpret_cast<const char*>(p.corpus);
  const int32_t* row_ids = p.row_ids;
  const int64_t row_bytes = (int64_t)D * sizeof(T);
  const int n_slices = D / EPL;
  u64 held = 0ull;  // this lane's slot of the wave's pending keys
  int pending = 0;  // wave-uniform
  auto flush = [&]() {
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(counter, (unsigned long long)pending);
    base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    if (lane < pending && base + lane < capacity) out[base + lane] = held;
    pending = 0;
  };
  for (int64_t pos = (int64_t)blockIdx.x * n_waves + wave; pos < p.n_pos; pos += (int64_t)gridDim.x * n_waves) {
    const int64_t row = row_ids ? (int64_t)row_ids[pos] : pos;
    const char* rp = corpus + row * row_bytes;
    float acc = 0.f;
    if constexpr (VEC) {
#pragma unroll 4
      for (int sl = lane; sl < n_slices; sl += 64) {
        const f32x4 x = ld16<true>(rp + (size_t)sl * 16);
        float qf[EPL];
        const f32x4* src = reinterpret_cast<const f32x4*>(qlds + (size_t)sl * EPL);
#pragma unroll
        for (int v = 0; v < EPL / 4; ++v) {
          const f32x4 t = src[v];
          qf[4 * v + 0] = t.x;
          qf[4 * v + 1] = t.y;
          qf[4 * v + 2] = t.z;
          qf[4 * v + 3] = t.w;
        }
        acc = dot_slice<T>(x, qf, acc);
      }
    } else {
      const T* re = reinterpret_cast<const T*>(rp);
      for (int e = lane; e < D; e += 64) acc = fmaf((float)re[e], qlds[e], acc);
    }
    const float s = wave_uniform(cosine_to_score(wave_sum(acc)));
    if (s >= min_score) {  // wave-uniform; NaN never passes
      const u64 key = make_key(s, (uint32_t)pos + p.index_base);
      if (key < p.key_bound) {
        if (lane == pending) held = key;
        if (++pending == 64) flush();
      }
    }
  }
  if (pending) flush();
}

hipError_t launch_scan_emit(const ScanParams& p, int blocks, u64* out, unsigned long long capacity, unsigned long long* counter, hipStream_t stream) {
  if (p.dim < 1 || p.n_pos < 0) return hipErrorInvalidValue;
  const bool f16 = p.dtype == TAVB_F16;
  const int epl = f16 ? 8 : 4;
  const bool vec = ((uintptr_t)p.corpus % 16) == 0 && (p.dim % epl) == 0;
  const size_t lds = (size_t)p.dim * sizeof(float);
  if (lds > 150 * 1024) return hipErrorInvalidValue;
  auto go = [&](auto kern) -> hipError_t {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), lds, stream, p, out, capacity, counter);
    return hipGetLastError();
  };
  if (f16) return vec ? go(scan_emit_kernel<_Float16, true>) : go(scan_emit_kernel<_Float16, false>);
  return vec ? go(scan_emit_kernel<float, true>) : go(scan_emit_kernel<float, false>);
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
namespace {

constexpr size_t scratch_bytes(int kpl, int waves) { return (size_t)((waves + 1) / 2) * 64 * kpl * sizeof(u64); }

// grid of a scan launch: the row workgroups x the query groups of the grouped form (ScanParams::group)
inline dim3 scan_grid(const ScanParams& p, const ScanGeometry& g) { return dim3(g.blocks, p.group > 0 ? (p.nq + p.group - 1) / p.group : 1); }

template <typename T, int CH, int NQ, int KPL, int U, bool NT, bool PIPE>
hipError_t go_fixed(const ScanParams& p, const ScanGeometry& g, hipStream_t s) {
  constexpr int regs_est = U * CH * 4 * (PIPE ? 2 : 1) + (NQ == 1 ? CH * Elem<T>::EPL : 8) + U * NQ + NQ * KPL * 2 + 24;
  constexpr int MAXT = (regs_est > 120 || NQ > 1) ? 512 : 1024;  // multi-query: the compiler keeps many LDS query slices live
  constexpr int D = CH * 64 * Elem<T>::EPL;
  int waves = g.waves;
  if (waves * 64 > MAXT) waves = MAXT / 64;
  const size_t lds = (NQ == 1 ? 0 : (size_t)NQ * D * sizeof(float)) + scratch_bytes(KPL, waves);
  auto kern = scan_fixed_kernel<T, CH, NQ, KPL, U, NT, PIPE, MAXT>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(kern, scan_grid(p, g), dim3(waves * 64), lds, s, p, NoInlineQuery{});
  return hipGetLastError();
}

// the default single-query form (2 rows in flight, non-temporal loads) with the query in the kernel arguments
template <typename T, int CH, int KPL>
hipError_t go_fixed_inline(const ScanParams& p, const ScanGeometry& g, hipStream_t s, const float* host_query) {
  static_assert(CH * 64 * Elem<T>::EPL == 1536, "InlineQuery holds 1536 floats");
  constexpr int MAXT = 1024;
  int waves = g.waves;
  if (waves * 64 > MAXT) waves = MAXT / 64;
  const size_t lds = scratch_bytes(KPL, waves);
  InlineQuery qa;
  std::memcpy(qa.v, host_query, sizeof qa.v);
  hipLaunchKernelGGL((scan_fixed_kernel<T, CH, 1, KPL, 2, true, false, MAXT, InlineQuery>), dim3(g.blocks), dim3(waves * 64), lds, s, p, qa);
  return hipGetLastError();
}

template <typename T, int NQ, int KPL>
hipError_t go_vec(const ScanParams& p, const ScanGeometry& g, hipStream_t s) {
  const size_t qbytes = (((size_t)NQ * p.dim * sizeof(float)) + 15) & ~(size_t)15;
  const size_t lds = qbytes + scratch_bytes(KPL, g.waves);
  hipError_t e = hipSuccess;
  if (g.nt) {
    auto kern = scan_vec_kernel<T, NQ, KPL, true>;
    if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, scan_grid(p, g), dim3(g.waves * 64), lds, s, p);
  } else {
    auto kern = scan_vec_kernel<T, NQ, KPL, false>;
    if (lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, scan_grid(p, g), dim3(g.waves * 64), lds, s, p);
  }
  return hipGetLastError();
}

template <typename T, int NQ, int KPL>
hipError_t go_scalar(const ScanParams& p, const ScanGeometry& g, hipStream_t s) {
  const size_t lds = scratch_bytes(KPL, g.waves);
  hipLaunchKernelGGL((scan_scalar_kernel<T, NQ, KPL>), scan_grid(p, g), dim3(g.waves * 64), lds, s, p);
  return hipGetLastError();
}

// the tuned default + the variants the bench sweep can select (single query, k <= 64)
template <typename T, int CH>
hipError_t go_fixed_q1_variants(const ScanParams& p, const ScanGeometry& g, hipStream_t s) {
  const int u = g.unroll, nt = g.nt, pipe = g.pipe;
#define TAVB_V(UU, NN, PP) \
  if (u == UU && nt == NN && pipe == PP) return go_fixed<T, CH, 1, 1, UU, NN != 0, PP != 0>(p, g, s);
  TAVB_V(1, 1, 0) TAVB_V(2, 1, 0) TAVB_V(4, 1, 0)
  TAVB_V(1, 0, 0) TAVB_V(2, 0, 0) TAVB_V(4, 0, 0)
  TAVB_V(1, 1, 1) TAVB_V(2, 1, 1)
  TAVB_V(1, 0, 1) TAVB_V(2, 0, 1)
#undef TAVB_V
  return go_fixed<T, CH, 1, 1, 2, true, false>(p, g, s);
}

template <typename T, int CH>
hipError_t dispatch_fixed(const ScanParams& p, const ScanGeometry& g, hipStream_t s, int nqt, int kpl) {
  if (nqt == 1) {
    if (kpl == 1) return go_fixed_q1_variants<T, CH>(p, g, s);
    return go_fixed<T, CH, 1, 4, 2, true, false>(p, g, s);
  }
  if (kpl != 1) return hipErrorInvalidValue;  // 256-deep lists with several queries: vector tier
  if (nqt == 2) return go_fixed<T, CH, 2, 1, 2, true, false>(p, g, s);
  if (nqt == 4) return go_fixed<T, CH, 4, 1, 2, true, false>(p, g, s);
  return hipErrorInvalidValue;  // 8 queries do not fit the register file unrolled: vector tier
}

template <typename T>
hipError_t dispatch_vec(const ScanParams& p, const ScanGeometry& g, hipStream_t s, int nqt, int kpl) {
#define TAVB_F(NQ, KPL) \
  if (nqt == NQ && kpl == KPL) return go_vec<T, NQ, KPL>(p, g, s);
  TAVB_F(1, 1) TAVB_F(1, 4) TAVB_F(2, 1) TAVB_F(2, 4) TAVB_F(4, 1) TAVB_F(4, 4) TAVB_F(8, 1)
#undef TAVB_F
  return hipErrorInvalidValue;
}

template <typename T>
hipError_t dispatch_scalar(const ScanParams& p, const ScanGeometry& g, hipStream_t s, int nqt, int kpl) {
#define TAVB_F(NQ, KPL) \
  if (nqt == NQ && kpl == KPL) return go_scalar<T, NQ, KPL>(p, g, s);
  TAVB_F(1, 1) TAVB_F(1, 4) TAVB_F(2, 1) TAVB_F(2, 4) TAVB_F(4, 1) TAVB_F(4, 4) TAVB_F(8, 1)
#undef TAVB_F
  return hipErrorInvalidValue;
}

}  // namespace

bool launch_scan_inline_query(const ScanParams& p, const ScanGeometry& g, hipStream_t stream, const float* host_query, int* tier_used, hipError_t* err) {
  const bool f16 = p.dtype == TAVB_F16;
  if (p.nq != 1 || p.dim != 1536 || p.k < 1 || p.k > TAVB_MAX_FUSED_K || ((uintptr_t)p.corpus % 16) != 0) return false;
  if (!(g.tier == 0 || g.tier == 1) || g.unroll != 2 || !g.nt || g.pipe) return false;  // only the default form of tier 1 has the variant
  if (tier_used) *tier_used = 1;
  if (p.k <= 64)
    *err = f16 ? go_fixed_inline<_Float16, 3, 1>(p, g, stream, host_query) : go_fixed_inline<float, 6, 1>(p, g, stream, host_query);
  else
    *err = f16 ? go_fixed_inline<_Float16, 3, 4>(p, g, stream, host_query) : go_fixed_inline<float, 6, 4>(p, g, stream, host_query);
  return true;
}

hipError_t launch_scan(const ScanParams& p, const ScanGeometry& g, hipStream_t stream, int* tier_used) {
  const bool grouped = p.group > 0;
  if (grouped && !(p.group == 1 || p.group == 2 || p.group == 4 || p.group == 8)) return hipErrorInvalidValue;
  if (p.nq < 1 || p.nq > (grouped ? TAVB_MAX_GROUPED_QUERIES : TAVB_MAX_STREAM_QUERIES) || p.k < 1 || p.k > TAVB_MAX_FUSED_K || p.dim < 1)
    return hipErrorInvalidValue;
  const int per = grouped ? (p.group < p.nq ? p.group : p.nq) : p.nq;  // queries per pass of a wave over a row
  if (p.k > 64 && per > 4) return hipErrorInvalidValue;  // 256-deep lists: at most 4 queries per pass (registers)
  const int nqt = per <= 1 ? 1 : per <= 2 ? 2 : per <= 4 ? 4 : 8;
  const int kpl = p.k <= 64 ? 1 : 4;
  const bool f16 = p.dtype == TAVB_F16;
  const int esize = f16 ? 2 : 4;
  const int epl = 16 / esize;
  const bool aligned = ((uintptr_t)p.corpus % 16) == 0 && (p.dim % epl) == 0;
  // LDS budget of the vector tier: queries + merge scratch must fit 160 KiB
  const size_t vec_lds = (size_t)nqt * p.dim * 4 + 16 + scratch_bytes(kpl, g.waves);
  int tier = g.tier;
  if (tier == 0) {
    // several queries per pass: the unrolled tier-1 form needs ~170 VGPRs (8 waves/CU) and measured slower
    // than the LDS-query vector tier at 16 waves/CU (cfg5 terms pass 8.1 ms vs 6.6 ms), so auto picks tier 2 there
    if (aligned && p.dim == 1536 && nqt == 1)
      tier = 1;
    else if (aligned && vec_lds <= 150 * 1024)
      tier = 2;
    else
      tier = 3;
  }
  if (tier == 1 && !(aligned && p.dim == 1536 && (nqt == 1 || (kpl == 1 && nqt <= 4)))) return hipErrorInvalidValue;
  if (tier == 2 && !(aligned && vec_lds <= 150 * 1024)) return hipErrorInvalidValue;
  if (tier_used) *tier_used = tier;
  if (tier == 1) return f16 ? dispatch_fixed<_Float16, 3>(p, g, stream, nqt, kpl) : dispatch_fixed<float, 6>(p, g, stream, nqt, kpl);
  if (tier == 2) return f16 ? dispatch_vec<_Float16>(p, g, stream, nqt, kpl) : dispatch_vec<float>(p, g, stream, nqt, kpl);
  return f16 ? dispatch_scalar<_Float16>(p, g, stream, nqt, kpl) : dispatch_scalar<float>(p, g, stream, nqt, kpl);
}

}  // namespace tavb
