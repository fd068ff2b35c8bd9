// Host-side declarations shared by the translation units of libtavb.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/tavb.h"

#define TAVB_MAX_GROUPED_QUERIES 128  // most queries of one grouped streaming launch (ScanParams::group)

namespace tavb {

struct ScanParams {
  const void* corpus;      // [rows, dim] row-major, f32 or f16
  const int32_t* row_ids;  // optional gather list (subset search): position -> row
  const float* queries;    // device, f32 [nq, dim]
  unsigned long long* lists;  // out: [nq, blocks, k] sorted keys, one list per workgroup
  int64_t n_pos;           // number of candidate positions (rows, or subset length)
  int32_t dim;
  int32_t dtype;  // TAVB_F32 / TAVB_F16
  int32_t nq;     // 1..TAVB_MAX_STREAM_QUERIES; with `group` > 0: 1..TAVB_MAX_GROUPED_QUERIES
  int32_t k;      // 1..TAVB_MAX_FUSED_K
  uint32_t index_base;  // added to the position before it is packed into the key
  unsigned long long key_bound;  // exclusive upper bound on accepted keys (~0 = none): paging cursor
  // grouped form (small corpora, 2 .. 128 queries in ONE launch): gridDim.y = ceil(nq / group) query groups, workgroup (x, y) scans the rows
  // of workgroup x for queries y * group .. -- `group` (1, 2, 4 or 8) queries per pass of a wave over a row, as many passes over the (L2-resident)
  // rows as there are groups.  The launch order puts workgroup (x, y) on XCD x % 8 whatever y (gridDim.x a multiple of 8): every group finds the
  // rows of "its" x in that XCD's L2 after the first one read them.  0 = the plain form (gridDim.y = 1, nq <= TAVB_MAX_STREAM_QUERIES).
  int32_t group;
  float min_score[TAVB_MAX_GROUPED_QUERIES];  // one per query (the plain form reads the first TAVB_MAX_STREAM_QUERIES)
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

// single 1536-wide query handed over INSIDE the kernel arguments (`host_query`: 1536 floats on the host; `p.queries` is not read): no copy in front of
// the launch.  Returns false -- nothing launched -- when the shape or geometry has no such variant; otherwise `*err` is the launch result.
bool launch_scan_inline_query(const ScanParams& p, const ScanGeometry& g, hipStream_t stream, const float* host_query, int* tier_used, hipError_t* err);

// every row with score >= min_score[0] (and key < key_bound), unsorted: out[0 .. *counter) (entries past `capacity` are dropped, still counted)
hipError_t launch_scan_emit(const ScanParams& p, int blocks, unsigned long long* out, unsigned long long capacity, unsigned long long* counter,
                            hipStream_t stream);

// lists: [n_lists, nq, k] (list-major) or [nq, n_lists, k] (query-major) sorted keys -> out [nq, k]
hipError_t launch_merge(const unsigned long long* lists, int n_lists, int nq, int k, bool query_major,
                        unsigned long long* out, hipStream_t stream);
// the same over a device-side work list: query slot s is merged only if s < *active, into out[scatter[s]] (scatter == nullptr: out[s])
hipError_t launch_merge_scatter(const unsigned long long* lists, int n_lists, int nq, int k, const int* active, const int* scatter,
                                unsigned long long* out, hipStream_t stream);

// keys carrying list positions -> keys carrying map[position] (tavb_misc.hip)
hipError_t launch_remap_positions(unsigned long long* keys, int64_t n, const int32_t* map, int64_t map_len, hipStream_t stream);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize), once per (device, kernel) (tavb_misc.hip)
hipError_t ensure_dynamic_lds(const void* kernel, int bytes);

// fault injection: holds the stream for `ms` milliseconds (tavb_misc.hip)
hipError_t launch_stall(int ms, hipStream_t stream);

// chunk-row hits -> message hits (tavb_misc.hip)
hipError_t launch_accept_bitmap(const int32_t* msgs, int64_t n, uint32_t* bits, int64_t n_bits, hipStream_t stream);
hipError_t launch_message_rerank(const unsigned long long* hits, int nq, int k, uint32_t index_base, const int32_t* pos_to_row, const int32_t* row_to_msg,
                                 int64_t n_rows, const uint32_t* accept_bits, int64_t n_bits, int max_messages, unsigned long long* out,
                                 hipStream_t stream);

// exact fp32-query semantics for the 256-query tile (tavb_rescore.hip)
hipError_t launch_corpus_max_norm(const void* rows_f16, int64_t n, int dim, float* out_sq, hipStream_t stream);
// The prologue of the wide path in one launch (tavb_rescore.hip::query_prepare_kernel): nq live queries, slots up to nq_pad zeroed.
// q16 may be null (rows_only: the filter uses the exact queries, only the rows' rounding enters the bound)
// band (optional, [nq_pad]): 2 * delta, the width of the band selection
// frag_major: q16 in MFMA-fragment-major order for 256-query tiles (tavb_mfma.hip, BD) instead of row-major
// min_scores: device [nq], the callers' thresholds per query (float32 values, NaN allowed) -- or, ms_fill: ONE threshold for the whole batch,
//   written to ms_out / ms_floor_out [nq_pad] (+inf for the padding) by this launch
// aux (optional): [3][nq_pad] ints zeroed (band counts, lost levels, verdicts); flag64 (optional): 64 ints zeroed (the work list's header)
hipError_t launch_query_prepare(const float* q, int nq, int nq_pad, int dim, const float* min_scores, bool rows_only, const float* max_norm_sq, void* q16,
                                float* delta, float* thr, float* band, hipStream_t stream, bool frag_major, int* aux, int* flag64, bool ms_fill, float ms_value,
                                float ms_floor, float* ms_out, float* ms_floor_out);
// candidates [nq, stride] (+ cand_cnt [nq]: band mode, the set is complete by construction unless incomplete[q]; cand_cnt == nullptr: the
// sorted best `stride` = 64 by approximate score, complete when rank 63 + delta < the exact k-th best) -> exact top k [nq, k]
// (*n_flagged must be zero when the launch starts: the prologue's flag64)
hipError_t launch_rescore(const void* corpus, bool f32_rows, int dim, uint32_t index_base, const float* queries, const unsigned long long* approx,
                          int stride, const int* cand_cnt, const int* incomplete, const float* delta, const float* min_scores, int nq, int k,
                          unsigned long long* out, int* n_flagged, int* flagged, hipStream_t stream, const int* gate = nullptr, int gate_max = 0);
// the same scoring over the device-side work list of flagged queries: slot s (live when s < *slot_active, slot_min < *slot_active <= slot_max) holds
// the candidates an exact tile ranked highest for query slot_query[s] (cand [n_slots, stride]; cand_cnt [n_slots] or nullptr = `stride` each);
// their best k by the streaming kernels' arithmetic go to row slot_query[s] of out [nq, k]
hipError_t launch_rescore_slots(const void* corpus, bool f32_rows, int dim, uint32_t index_base, const float* queries, const unsigned long long* cand,
                                int stride, const int* cand_cnt, const float* min_scores, int n_slots, int k, unsigned long long* out,
                                const int* slot_query, const int* slot_active, int slot_min, int slot_max, hipStream_t stream);
// (gate: device-side counter; when *gate > gate_max there are no candidates -- the last filter phase was skipped -- and EVERY query is flagged)
hipError_t launch_shadow_convert(const float* rows_f32, int64_t n, int dim, void* out_f16, int out_pitch /*halves per shadow row*/, float* stats /*[2]*/, hipStream_t stream);
// seed / delta (optional, [nq]): the filter's cut per query and its error bound -- the exact fallbacks' admission thresholds start from seed - 2 delta;
// band_out (optional, [cap]) <- band_v: the band width of every slot of the wide exact form
hipError_t launch_gather_flagged_f32(const float* queries, int dim, const float* min_scores, const int* n_flagged, const int* flagged, int cap, float* out,
                                     float* thr, const float* seed, const float* delta, hipStream_t stream);
hipError_t launch_gather_flagged(const float* queries, int dim, const float* min_scores, const int* n_flagged, const int* flagged, int cap, void* hi, void* lo,
                                 float* thr, const float* seed, const float* delta, float* band_out, float band_v, hipStream_t stream);

hipError_t launch_normalize_f32(const float* in, float* out, int64_t rows, int dim, hipStream_t stream);
hipError_t launch_f32_to_f16(const float* in, void* out, int64_t count, hipStream_t stream);
hipError_t launch_f32_split_f16(const float* in, void* hi, void* lo, int64_t count, hipStream_t stream);

// MFMA batched scan (f16 corpus, f16 queries staged by the launcher)
struct MfmaParams {
  const void* corpus;   // f16 [rows, dim] row-major (skinny kernel: f32 or f16)
  const void* queries;  // f16 [nq_padded, dim] device
  unsigned long long* lists;  // out [nq, n_splits, k]
  unsigned long long* workspace;  // candidate buffers, mfma_workspace_bytes() bytes
  int* counts;                    // 256-query tile only: [n_splits * nq_padded] keys left per candidate buffer (the tile writes no lists)
  int64_t rows;
  int32_t dim;
  int32_t nq;
  int32_t nq_padded;
  int32_t k;
  uint32_t index_base;
  float min_score;
  int32_t n_splits;  // row ranges the corpus is cut into (one list per (query, split))
  int32_t list_stride;  // lists per query in `lists`; 0 = n_splits (extra slots are the caller's, e.g. a carried-over top-k)
  int32_t sched;     // measurement: staging schedule of the 256-query tile (1, 2); 9 = 64-byte K steps in the 32/64-query tile
  int32_t ablate;    // measurement only (garbage results): see launch_mfma_scan
  const float* thr_in;  // optional per-query admission thresholds (device, [nq_padded]) from the earlier ladder phases
  const int* active;    // optional: device-side count of live queries (query tiles past it return at once) -- a fixed-shape launch over a work list
  int32_t active_min;   // ... and the whole launch returns at once unless active_min < *active <= active_max (0 = no upper bound)
  int32_t active_max;
  const int* gate;      // 256-query kernel only, optional: device-side counter; the launch returns at once when *gate > gate_max
  int32_t gate_max;
  int32_t bdirect;      // 256-query kernel only: `queries` are in MFMA-fragment-major order and go straight from L2 into registers (no LDS staging)
  int64_t split_plane;  // 256-query kernel only: > 0 = the SPLIT form, queries = [2][nq_padded][dim] fp16 planes this many bytes apart (q = hi + lo)
  int32_t f32;          // skinny kernel only: corpus and queries are fp32 (else fp16)
  int32_t skinny_tile;  // skinny kernel only: queries per tile, 32 or 64
  int32_t wide_tile;    // 128/256-query kernel only: queries per tile, 128 or 256 (0 = 256)
  const float* band;    // 128/256-query kernel only, optional [nq_padded]: band selection (keep every key within band[q] of the k-th best)
  unsigned* lost;       // ... [nq_padded]: atomicMax of the score level (bits) below which a query lost band rows to a buffer that could not hold its band
};
hipError_t launch_mfma_scan(const MfmaParams& p, hipStream_t stream);
hipError_t launch_sample_thresholds(const unsigned long long* keys, int nq, int k, const float* floor, float* thr, hipStream_t stream);
int mfma_query_tile(int nq);              // queries per workgroup tile: 128 where that leaves less padding (<= 128, 257..384, 513..640 queries), else 256
int mfma_query_tile_for(int nq, int64_t rows, int n_cu);  // ... and 128 on a corpus so small that 256-query tiles leave half of the CUs idle
int mfma_pick_splits(int64_t rows, int nq_padded, int tile, int n_cu);
bool mfma_supported(int dim, int k);
size_t mfma_workspace_bytes(int n_splits, int nq_padded, bool wide);
// the BAND of every query (all keys within band[q] of its k-th best; at most kc_max, else the strict best k) over the 128/256-query tile's
// candidate buffers (+ an optional carried-over band [nq, kc_max] / carried_cnt [nq]) -> out [nq, kc_max] unsorted, out_cnt [nq];
// thr_out[q] = just below the band's cut (or floor[q]); lost [nq]: highest score level (bits) at which band rows were dropped so far
// (in/out); verdict (optional, last phase) [nq]: 1 = the band handed over is not provably complete; tavb_mfma.hip
constexpr int kBandMax = 2048;  // most candidates per query the rescoring accepts (kc_max <= kBandMax: the context's "band_max" option; also the most the select
                                // kernel's cache keeps when it cuts mid-stream).  1024 until round 6: a band of 1500 near-duplicates cost a 2x exact pass
hipError_t launch_select_band(const unsigned long long* cand, const int* counts, int n_splits, int nq, int nq_padded, int k, int kc_max,
                              const unsigned long long* carried, const int* carried_cnt, const float* floor, const float* band, unsigned long long* out,
                              int* out_cnt, float* thr_out, unsigned* lost, int* verdict, hipStream_t stream, const int* active = nullptr,
                              int active_min = 0, int active_max = 0x7fffffff, const int* gate = nullptr, int gate_max = 0, int* doomed = nullptr,
                              int doom_limit = 0);
// 32-query tiles at HBM speed, fp32 or fp16 corpora (same parameter block; `queries` in the corpus dtype)
hipError_t launch_skinny_scan(const MfmaParams& p, hipStream_t stream);
int skinny_query_tile(int nq);  // 32, or 64 for batches of 33 and more
int skinny_pick_splits(int64_t rows, int nq_padded, int tile, int n_cu, int dim, bool f32, int sched);
bool skinny_supported(int dim, int k, bool f32);

}  // namespace tavb
