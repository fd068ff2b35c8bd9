// Host-side declarations shared by the translation units of libtavb.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/tavb.h"

namespace tavb {

struct ScanParams {
  const void* corpus;      // [rows, dim] row-major, f32 or f16
  const int32_t* row_ids;  // optional gather list (subset search): position -> row
  const float* queries;    // device, f32 [nq, dim]
  unsigned long long* lists;  // out: [nq, blocks, k] sorted keys, one list per workgroup
  int64_t n_pos;           // number of candidate positions (rows, or subset length)
  int32_t dim;
  int32_t dtype;  // TAVB_F32 / TAVB_F16
  int32_t nq;     // 1..TAVB_MAX_STREAM_QUERIES
  int32_t k;      // 1..TAVB_MAX_FUSED_K
  uint32_t index_base;  // added to the position before it is packed into the key
  unsigned long long key_bound;  // exclusive upper bound on accepted keys (~0 = none): paging cursor
  float min_score[TAVB_MAX_STREAM_QUERIES];
};

struct ScanGeometry {
  int blocks;
  int waves;   // per block
  int unroll;  // rows in flight per wave
  int nt;      // non-temporal loads
  int pipe;    // software prefetch
  int tier;    // 0 auto, 1 fixed, 2 vector, 3 scalar
};

// returns hipSuccess or the launch error; `*tier_used` reports the kernel family chosen
hipError_t launch_scan(const ScanParams& p, const ScanGeometry& g, hipStream_t stream, int* tier_used);

// lists: [n_lists, nq, k] (list-major) or [nq, n_lists, k] (query-major) sorted keys -> out [nq, k]
hipError_t launch_merge(const unsigned long long* lists, int n_lists, int nq, int k, bool query_major,
                        unsigned long long* out, hipStream_t stream);

hipError_t launch_normalize_f32(const float* in, float* out, int64_t rows, int dim, hipStream_t stream);
hipError_t launch_f32_to_f16(const float* in, void* out, int64_t count, hipStream_t stream);
hipError_t launch_f32_split_f16(const float* in, void* hi, void* lo, int64_t count, hipStream_t stream);
size_t tiled_bytes(int64_t rows, int dim);
hipError_t launch_pack_tiled(const void* src, int src_dtype, int64_t rows, int dim, void* dst, hipStream_t stream);

// MFMA batched scan (f16 corpus, f16 queries staged by the launcher)
struct MfmaParams {
  const void* corpus;   // f16 [rows, dim] row-major, or the K-blocked image when a_tiled
  int32_t a_tiled;
  const void* queries;  // f16 [nq_padded, dim] device
  unsigned long long* lists;  // out [nq, n_splits, k]
  unsigned long long* workspace;  // candidate buffers, mfma_workspace_bytes() bytes
  int64_t rows;
  int32_t dim;
  int32_t nq;
  int32_t nq_padded;
  int32_t k;
  uint32_t index_base;
  float min_score;
  int32_t n_splits;  // row ranges the corpus is cut into (one list per (query, split))
  int32_t list_stride;  // lists per query in `lists`; 0 = n_splits (extra slots are the caller's, e.g. a carried-over top-k)
  int32_t variant;   // 1 = lock-step K loop, 2 = ping-pong wave groups
  int32_t group_sel; // ping-pong grouping: 0 = wave>>2, 1 = wave&1, 2 = (wave>>1)&1
  int32_t prio;      // s_setprio placement: 0 none, 1 MFMA phase, 2 LOAD phase
  int32_t ablate;    // measurement only: bit 0 = drop MFMAs, bit 1 = drop LDS-DMA (garbage results)
  const float* thr_in;  // optional per-query admission thresholds (device, [nq_padded]) from a sample pass
  int32_t rendezvous;   // variant 3: the workgroups of a row range meet at every tile start (L2 sharing of corpus slices)
  int32_t a_nt;         // variant 3: non-temporal policy on the corpus LDS-DMA stream
  int32_t f32;          // skinny kernel only: corpus and queries are fp32 (else fp16)
  int32_t skinny_tile;  // skinny kernel only: queries per tile, 32 or 64
};
hipError_t launch_mfma_scan(const MfmaParams& p, hipStream_t stream);
hipError_t launch_sample_thresholds(const unsigned long long* keys, int nq, int k, float* thr, hipStream_t stream);
int mfma_query_tile();                    // queries per workgroup tile
int mfma_pick_splits(int64_t rows, int nq_padded, int n_cu);
bool mfma_supported(int dim, int k);
size_t mfma_workspace_bytes(int n_splits, int nq_padded);
// 32-query tiles at HBM speed, fp32 or fp16 corpora (same parameter block; `queries` in the corpus dtype)
hipError_t launch_skinny_scan(const MfmaParams& p, hipStream_t stream);
int skinny_query_tile(int nq);  // 32, or 64 for batches of 33 and more
int skinny_pick_splits(int64_t rows, int nq_padded, int tile, int n_cu);
bool skinny_supported(int dim, int k, bool f32);

}  // namespace tavb
