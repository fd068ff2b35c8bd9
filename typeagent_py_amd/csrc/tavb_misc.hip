// Small kernels around the scan: merging sorted key lists, L2 row normalisation
// (model_adapters.py:181-183 of the reference) and f32 -> f16 conversion.

#include <hip/hip_fp16.h>

#include <map>
#include <mutex>
#include <utility>

#include "tavb_device.h"
#include "tavb_internal.h"

namespace tavb {

// ---------------------------------------------------------------------------
// merge: n_lists sorted lists of k keys per query -> one sorted list of k keys.
// One workgroup per query; each wave folds its share of the lists into a
// register-resident list with bitonic merges, then the waves merge through LDS.
// ---------------------------------------------------------------------------
template <int KPL>
__global__ void __launch_bounds__(1024) merge_kernel(const u64* __restrict__ lists, int n_lists, int nq, int k,
                                                     int query_major, u64* __restrict__ out, const int* __restrict__ active,
                                                     const int* __restrict__ scatter) {
  extern __shared__ __align__(16) unsigned char smem[];
  u64* scratch = reinterpret_cast<u64*>(smem);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  const int q = blockIdx.x;
  if (active != nullptr && q >= *active) return;  // device-side work list: slot not in use
  WaveTopK<KPL> mine;
  mine.clear();
  auto list_ptr = [&](int m) {
    return query_major ? lists + ((size_t)q * n_lists + m) * (size_t)k : lists + ((size_t)m * nq + q) * (size_t)k;
  };
  // four lists per round: all their loads are issued before the first merge, so the (dependent,
  // latency-bound) merges of one round overlap the memory latency of the next
  constexpr int G = (KPL == 1) ? 4 : 1;
  int m = wave;
  for (; m + (G - 1) * n_waves < n_lists; m += G * n_waves) {
    WaveTopK<KPL> other[G];
#pragma unroll
    for (int g = 0; g < G; ++g) other[g].load_reversed(list_ptr(m + g * n_waves), k, lane);
#pragma unroll
    for (int g = 0; g < G; ++g) mine.merge_reversed(other[g], lane);
  }
  for (; m < n_lists; m += n_waves) {
    WaveTopK<KPL> other;
    other.load_reversed(list_ptr(m), k, lane);
    mine.merge_reversed(other, lane);
  }
  block_merge<KPL>(mine, scratch, wave, n_waves, lane);
  if (wave == 0) mine.store(out + (size_t)(scatter != nullptr ? scatter[q] : q) * k, k, lane);
}

hipError_t launch_merge(const unsigned long long* lists, int n_lists, int nq, int k, bool query_major,
                        unsigned long long* out, hipStream_t stream) {
  if (n_lists < 1 || nq < 1 || k < 1 || k > TAVB_MAX_FUSED_K) return hipErrorInvalidValue;
  int waves = 16;
  while (waves > 1 && waves / 2 >= n_lists) waves /= 2;  // no more waves than lists (rounded to a power of two)
  const int kpl = k <= 64 ? 1 : 4;
  const size_t lds = (size_t)((waves + 1) / 2) * 64 * kpl * sizeof(u64);
  if (kpl == 1)
    hipLaunchKernelGGL(merge_kernel<1>, dim3(nq), dim3(waves * 64), lds, stream, lists, n_lists, nq, k,
                       query_major ? 1 : 0, out, nullptr, nullptr);
  else
    hipLaunchKernelGGL(merge_kernel<4>, dim3(nq), dim3(waves * 64), lds, stream, lists, n_lists, nq, k,
                       query_major ? 1 : 0, out, nullptr, nullptr);
  return hipGetLastError();
}

hipError_t launch_merge_scatter(const unsigned long long* lists, int n_lists, int nq, int k, const int* active, const int* scatter,
                                unsigned long long* out, hipStream_t stream) {
  if (n_lists < 1 || nq < 1 || k < 1 || k > 64 || !active) return hipErrorInvalidValue;
  int waves = 16;
  while (waves > 1 && waves / 2 >= n_lists) waves /= 2;
  const size_t lds = (size_t)((waves + 1) / 2) * 64 * sizeof(u64);
  hipLaunchKernelGGL(merge_kernel<1>, dim3(nq), dim3(waves * 64), lds, stream, lists, n_lists, nq, k, 1, out, active, scatter);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// K1: y = x / ||x||_2 per row, float32, zero rows unchanged.  One wave per row;
// HBM-bound (4 B read + 4 B written per element).
// The quotient is an IEEE float32 division so that it rounds like numpy's
// `embeddings / norms` (no reciprocal-multiply).
// ---------------------------------------------------------------------------
// Rows of up to 64 * 4 * NC floats are held in registers between the two phases (sum of squares,
// scale), so HBM sees each element exactly once in and once out; longer rows re-read from L1/L2.
template <int NC>
__global__ void __launch_bounds__(256) normalize_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int64_t rows, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const bool vec = (dim % 4 == 0) && (((uintptr_t)in | (uintptr_t)out) % 16 == 0);
  const int n4 = dim / 4;
  const bool in_regs = vec && NC > 0 && n4 <= 64 * NC;
  for (int64_t r = wave; r < rows; r += n_waves) {
    const float* x = in + r * (int64_t)dim;
    float* y = out + r * (int64_t)dim;
    float ss = 0.f;
    f32x4 keep[NC > 0 ? NC : 1];
    if (in_regs) {
      const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int i = c * 64 + lane;
        keep[c] = (i < n4) ? x4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        ss = fmaf(keep[c].x, keep[c].x, ss);
        ss = fmaf(keep[c].y, keep[c].y, ss);
        ss = fmaf(keep[c].z, keep[c].z, ss);
        ss = fmaf(keep[c].w, keep[c].w, ss);
      }
    } else if (vec) {
      const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
      for (int i = lane; i < n4; i += 64) {
        const f32x4 v = x4[i];
        ss = fmaf(v.x, v.x, ss);
        ss = fmaf(v.y, v.y, ss);
        ss = fmaf(v.z, v.z, ss);
        ss = fmaf(v.w, v.w, ss);
      }
    } else {
      for (int i = lane; i < dim; i += 64) ss = fmaf(x[i], x[i], ss);
    }
    ss = wave_sum(ss);
    float norm = sqrtf(ss);
    if (!(norm > 0.f)) norm = 1.0f;  // np.where(norms > 0, norms, 1): zero (and NaN) norms divide by 1
    if (in_regs) {
      f32x4* y4 = reinterpret_cast<f32x4*>(y);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int i = c * 64 + lane;
        if (i < n4) {
          f32x4 v = keep[c];
          v.x = v.x / norm;
          v.y = v.y / norm;
          v.z = v.z / norm;
          v.w = v.w / norm;
          y4[i] = v;
        }
      }
    } else if (vec) {
      const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
      f32x4* y4 = reinterpret_cast<f32x4*>(y);
      for (int i = lane; i < n4; i += 64) {
        f32x4 v = x4[i];
        v.x = v.x / norm;
        v.y = v.y / norm;
        v.z = v.z / norm;
        v.w = v.w / norm;
        y4[i] = v;
      }
    } else {
      for (int i = lane; i < dim; i += 64) y[i] = x[i] / norm;
    }
  }
}

hipError_t launch_normalize_f32(const float* in, float* out, int64_t rows, int dim, hipStream_t stream) {
  if (rows <= 0) return hipSuccess;
  int64_t blocks = (rows + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (dim % 4 == 0 && dim <= 64 * 4 * 6)
    hipLaunchKernelGGL(normalize_rows_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, rows, dim);
  else if (dim % 4 == 0 && dim <= 64 * 4 * 16)
    hipLaunchKernelGGL(normalize_rows_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, rows, dim);
  else
    hipLaunchKernelGGL(normalize_rows_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, in, out, rows, dim);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// f32 -> f16, round to nearest even (v_cvt_f16_f32), 8 elements per lane per step
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) f32_to_f16_kernel(const float* __restrict__ in, _Float16* __restrict__ out,
                                                         int64_t count) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
  const int64_t n8 = count / 8;
  const bool vec = (((uintptr_t)in) % 16 == 0) && (((uintptr_t)out) % 16 == 0);
  if (vec) {
    const f32x4* in4 = reinterpret_cast<const f32x4*>(in);
    f16x8* out8 = reinterpret_cast<f16x8*>(out);
    for (int64_t i = tid; i < n8; i += nthreads) {
      const f32x4 a = in4[2 * i], b = in4[2 * i + 1];
      f16x8 h;
      h[0] = (_Float16)a.x; h[1] = (_Float16)a.y; h[2] = (_Float16)a.z; h[3] = (_Float16)a.w;
      h[4] = (_Float16)b.x; h[5] = (_Float16)b.y; h[6] = (_Float16)b.z; h[7] = (_Float16)b.w;
      out8[i] = h;
    }
    for (int64_t i = n8 * 8 + tid; i < count; i += nthreads) out[i] = (_Float16)in[i];
  } else {
    for (int64_t i = tid; i < count; i += nthreads) out[i] = (_Float16)in[i];
  }
}

hipError_t launch_f32_to_f16(const float* in, void* out, int64_t count, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  int64_t blocks = (count / 8 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in,
                     reinterpret_cast<_Float16*>(out), count);
  return hipGetLastError();
}

// q = hi + lo with hi = fp16(q) and lo = fp16(q - hi): the pair carries q to ~2^-22 of its magnitude (2^-24 absolute
// where lo is an fp16 subnormal), so hi.x + lo.x on the matrix cores reproduces the fp32-query dot product of the
// streaming kernels on fp16 corpora.  Element-wise; a few KiB per lookup.
__global__ void __launch_bounds__(256) f32_split_f16_kernel(const float* __restrict__ in, _Float16* __restrict__ hi,
                                                            _Float16* __restrict__ lo, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const float q = in[i];
    const _Float16 h = (_Float16)q;
    const float hf = (float)h;
    hi[i] = h;
    lo[i] = (hf - hf == 0.0f) ? (_Float16)(q - hf) : (_Float16)0.0f;  // inf / NaN highs get a zero low part
  }
}

hipError_t launch_f32_split_f16(const float* in, void* hi, void* lo, int64_t count, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  int64_t blocks = (count + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(f32_split_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, in, reinterpret_cast<_Float16*>(hi),
                     reinterpret_cast<_Float16*>(lo), count);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Message re-rank: chunk-row hits -> message hits, on the device (the step right after the lookup in both providers:
// storage/sqlite/messageindex.py:182-257, 296-326; storage/memory/messageindex.py:185-207).
//   hits (sorted by score desc, row asc) -> msg = row_to_msg[row] (rows without a message, -1, are skipped: the SQL
//   `WHERE index_position IN (...)` finds nothing for them) -> optional set-membership filter on msg (the provider's
//   predicate / `ordinals_set`) -> best score per message -> sort by score desc (stable) -> first max_messages.
// Because the hits arrive best first, "best score per message" is the FIRST accepted hit of each message and the stable
// sort leaves that subsequence as it is: the result is the accepted first occurrences, in hit order, cut at
// max_messages.  Output: keys (score bits << 32 | 0xFFFFFFFF - msg), 0-terminated, like every other result list.
// One workgroup per query; k <= 256 hits.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) accept_bitmap_kernel(const int32_t* __restrict__ msgs, int64_t n, uint32_t* __restrict__ bits, int64_t n_bits) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = msgs[i];
    if (m >= 0 && m < n_bits) atomicOr(&bits[m >> 5], 1u << (m & 31));
  }
}

__global__ void __launch_bounds__(256) message_rerank_kernel(const u64* __restrict__ hits /*[nq, k]*/, int k, uint32_t index_base,
                                                             const int32_t* __restrict__ pos_to_row /*optional: subset position -> row*/,
                                                             const int32_t* __restrict__ row_to_msg, int64_t n_rows,
                                                             const uint32_t* __restrict__ accept_bits /*optional*/, int64_t n_bits, int max_messages,
                                                             u64* __restrict__ out /*[nq, k]*/) {
  __shared__ int msg_of[256];
  __shared__ unsigned char keep[256];
  const int i = threadIdx.x;
  const u64* h = hits + (size_t)blockIdx.x * k;
  u64* o = out + (size_t)blockIdx.x * k;
  u64 key = (i < k) ? h[i] : 0ull;
  int msg = -1;
  if (key != 0ull) {
    int64_t row = (int64_t)(0xFFFFFFFFu - (uint32_t)key) - (int64_t)index_base;
    if (pos_to_row != nullptr) row = pos_to_row[row];
    if (row >= 0 && row < n_rows) msg = row_to_msg[row];
    if (msg >= 0 && accept_bits != nullptr && !(msg < n_bits && ((accept_bits[msg >> 5] >> (msg & 31)) & 1u))) msg = -1;
  }
  msg_of[i] = msg;
  __syncthreads();
  bool first = msg >= 0;
  for (int j = 0; first && j < i; ++j) first = (msg_of[j] != msg);  // an earlier (better or equal) hit of the same message wins
  keep[i] = first ? 1 : 0;
  __syncthreads();
  int pos = 0;
  for (int j = 0; j < i; ++j) pos += keep[j];
  if (i < k) o[i] = 0ull;
  __syncthreads();
  if (first && pos < max_messages && pos < k) o[pos] = (key & 0xFFFFFFFF00000000ull) | (u64)(0xFFFFFFFFu - (uint32_t)msg);
}

// keys that carry POSITIONS of a list (0xFFFFFFFF - position in the low word) -> keys that carry map[position]: a shard's subset search
// returns positions into ITS part of the caller's subset; the exchange needs positions into the caller's whole list
__global__ void __launch_bounds__(256) remap_positions_kernel(u64* __restrict__ keys, int64_t n, const int32_t* __restrict__ map, int64_t map_len) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const u64 key = keys[i];
    if (key == 0ull) continue;
    const int64_t pos = (int64_t)(0xFFFFFFFFu - (uint32_t)key);
    const uint32_t to = (pos < map_len) ? (uint32_t)map[pos] : 0xFFFFFFFEu;
    keys[i] = (key & 0xFFFFFFFF00000000ull) | (u64)(0xFFFFFFFFu - to);
  }
}

// fault injection (option "comm_stall_ms"): one wave that keeps the stream busy for `ticks` of the 100 MHz wall clock, as a peer that is late
// for the exchange would; bounded by the option's range (5 s)
__global__ void __launch_bounds__(64) stall_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

hipError_t launch_stall(int ms, hipStream_t stream) {
  hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(64), 0, stream, (unsigned long long)ms * 100000ull);
  return hipGetLastError();
}

hipError_t launch_remap_positions(unsigned long long* keys, int64_t n, const int32_t* map, int64_t map_len, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(remap_positions_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, keys, n, map, map_len);
  return hipGetLastError();
}

hipError_t launch_accept_bitmap(const int32_t* msgs, int64_t n, uint32_t* bits, int64_t n_bits, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(accept_bitmap_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, msgs, n, bits, n_bits);
  return hipGetLastError();
}

hipError_t launch_message_rerank(const unsigned long long* hits, int nq, int k, uint32_t index_base, const int32_t* pos_to_row, const int32_t* row_to_msg,
                                 int64_t n_rows, const uint32_t* accept_bits, int64_t n_bits, int max_messages, unsigned long long* out,
                                 hipStream_t stream) {
  if (nq < 1 || k < 1 || k > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(message_rerank_kernel, dim3(nq), dim3(256), 0, stream, hits, k, index_base, pos_to_row, row_to_msg, n_rows, accept_bits, n_bits,
                     max_messages, out);
  return hipGetLastError();
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size) instead of in front of every launch: a batched lookup on
// the wide path makes ~20 launches of kernels that need it, and the call is host time the GPU waits for when the kernels are short.
hipError_t ensure_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> done;  // largest size granted so far
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  auto it = done.find({dev, kernel});
  if (it != done.end() && it->second >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done[{dev, kernel}] = bytes;
  return e;
}

}  // namespace tavb
