// C ABI of libtavb.so (declared in include/tavb.h): context, workspaces, the
// host-synchronous lookups and the asynchronous device-resident ones.  Host code
// only -- the kernels live in tavb_scan.hip / tavb_misc.hip / tavb_mfma.hip.

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the functions are resolved with dlsym (tavb_comm_init)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "tavb_internal.h"

typedef unsigned long long u64_t;

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define TAVB_HIP(expr)                                                                                     \
  do {                                                                                                     \
    hipError_t e__ = (expr);                                                                               \
    if (e__ != hipSuccess)                                                                                 \
      return fail(TAVB_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

// bumped by every (re)allocation or release of a workspace: captured HIP graphs hold raw pointers into these buffers
std::atomic<unsigned long long> g_alloc_epoch{1};  // (contexts on several threads share it)

struct Buffer {
  void* ptr = nullptr;
  size_t cap = 0;
  bool pinned_host = false;
  int reserve(size_t bytes) {
    if (bytes <= cap) return TAVB_OK;
    size_t want = std::max(bytes, cap * 2);
    want = (want + 255) & ~(size_t)255;
    ++g_alloc_epoch;
    if (ptr) {
      hipError_t e = pinned_host ? hipHostFree(ptr) : hipFree(ptr);
      ptr = nullptr;
      cap = 0;
      if (e != hipSuccess) return fail(TAVB_E_HIP, "free of workspace failed: %s", hipGetErrorString(e));
    }
    hipError_t e = pinned_host ? hipHostMalloc(&ptr, want, hipHostMallocDefault) : hipMalloc(&ptr, want);
    if (e != hipSuccess) {
      ptr = nullptr;
      return fail(TAVB_E_NOMEM, "workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
    }
    cap = want;
    return TAVB_OK;
  }
  void release() {
    if (ptr) ++g_alloc_epoch;
    if (ptr) (void)(pinned_host ? hipHostFree(ptr) : hipFree(ptr));
    ptr = nullptr;
    cap = 0;
  }
};

struct PendingTiming {
  int kernel;
  hipEvent_t start, stop;
};

}  // namespace

struct tavb_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int n_cu = 256;

  const void* corpus = nullptr;
  int64_t rows = 0;
  int32_t dim = 0;
  int32_t dtype = TAVB_F32;
  int64_t ordinal_base = 0;

  tavb::ScanGeometry geom{0, 16, 2, 1, 0, 0};
  int64_t mfma_min_batch = 65;  // batches from this size up use the 128/256-query tile + rescoring (smaller ones the 32/64-query tile) ...
  // ... and on corpora of `mfma_big_bytes` (256 MiB) or more already from `mfma_min_batch_big` = 33 queries (round 5): padded to 128 queries the wide
  // tile serves 33 / 48 / 64 queries over 10M fp16 rows in 5.42 / 5.48 / 5.46 ms against 5.54 / 5.81 / 5.91 ms on the 64-query split-plane tile,
  // and 64 queries over 1M fp32 rows (through the fp16 shadow) in 0.85 ms against 2.17 ms (profiles/r05_raw/b64.txt).  On small corpora its ~40
  // launches per batch cost more than the 64-query tile's pass.
  int64_t mfma_min_batch_big = 33;
  int64_t mfma_big_bytes = (int64_t)256 << 20;
  // ... and on FP32 corpora of `mfma_big_bytes_f32` (2 GiB) or more from `mfma_min_batch_big_f32` = 5 queries (round 6): the wide tile streams the fp16
  // shadow -- half the bytes of the fp32 rows the 32-query fp32 tile reads -- and its candidates are rescored with the fp32 rows: 5 / 8 / 16 / 32
  // queries over 1M x 1536 fp32 rows in 0.76 / 0.77 / 0.78 / 0.79 ms against 1.14 / 1.16 / 1.21 / 1.24 ms (profiles/r06_raw/f32_mid.txt); 32 queries
  // over 700k / 400k / 200k / 100k rows: 0.64 / 0.48 / 0.36 / 0.53 ms against 0.95 / 0.61 / 0.40 / 0.22 (f32_few.txt: the wide path's ~0.35 ms of
  // selection and rescoring launches against half a pass).  Batches of 2 .. 4 queries (one pass of the fp32 streaming scan) from TWICE that size:
  // 1M rows 0.77 against 0.92 .. 0.97 ms, 700k rows 0.64 against 0.68, 400k rows 0.48 against 0.39.  Single queries keep the fp32 scan (option
  // f32_shadow = 2 moves them too).  Needs the shadow (f32_shadow >= 1: +50 % device memory, built on first use); without the memory for it the fp32
  // kernels serve the batch.
  // End of round 6 (tools/regime_sweep.py, profiles/r06_raw/regime_sweep_before.md, after the wide path's launch diet): 5+ queries from 1e9 bytes
  // (165k x 1536 rows: 0.26 ms through the shadow against 0.27 .. 0.33 on the fp32 tile, whose workgroups start compacting with their second
  // tile; at 120k rows the fp32 tile still wins, 0.17 .. 0.21 against 0.24), 2 .. 4 queries from 4 GiB as before (500k rows: 0.45 either way),
  // and 33+ queries at ANY size (`mfma_min_batch_f32`): one tile of the 64-query fp32 kernel is 82 us of fp32 matrix work however small the
  // corpus -- 64 queries over 1000 / 5000 / 20000 fp32 rows 0.189 / 0.202 / 0.204 ms against 0.087 / 0.112 / 0.140 for 65 queries on the wide tile.
  int64_t mfma_min_batch_big_f32 = 5;
  int64_t mfma_big_bytes_f32 = 1000000000;
  int64_t mfma_few_bytes_f32 = (int64_t)4 << 30;
  int64_t mfma_min_batch_f32 = 33;
  int64_t mfma_splits = 0;  // 0 = auto
  int64_t mfma_ablate = 0;
  int64_t mfma_sched = 0;
  int64_t mfma_tile = 0;  // 0 = auto (128 queries per tile up to 128 queries, else 256)
  int64_t mfma_sample_rows = 0;  // rows of the first (threshold-seeding) phase: 0 = auto (two tiles per workgroup), -1 = one phase, no seeding
  int64_t skinny_min_batch_f32 = 5;   // fp32 corpus: batches from this size up use the 32-query MFMA tile
  int64_t skinny_min_batch_f16 = 3;   // fp16 corpus: batches from this size up to mfma_min_batch - 1 use it
  int64_t mfma_ladder = 4;            // each further phase scans this many times the rows scanned so far (0 = seed once)

  Buffer d_queries, d_queries_f16, d_lists, d_out, d_rows, d_cand, d_thr, d_sample_keys;
  Buffer d_counts;  // 256-query tile: keys left per candidate buffer
  Buffer d_delta, d_approx, d_flag, d_fb_queries, d_norm;  // exact rescoring of the 256-query tile (tavb_rescore.hip)
  Buffer d_minscores;      // per-query thresholds of a batch on the device: [nq_pad] min_scores, then [nq_pad] exclusive admission floors (the tile paths)
  Buffer d_fb_cand;        // what the 64-query exact tile ranked highest for the flagged queries: [slots][64] keys, rescored into the callers' rows
  Buffer d_shadow;         // fp32 corpora (and fp16 ones whose width is not a multiple of 64): fp16 copy of rows [0, norm_rows), each padded with zeros to a
                           // multiple of 64 halves -- the filter operand of the 128/256-query tile
  Buffer d_queries_pad;    // the queries of a batch zero-padded to that width (odd widths only)
  int64_t f32_shadow = 1;  // option: 1 = batches of mfma_min_batch+ queries on fp32 corpora go through that shadow (+50 % HBM); 2 = every lookup on
                           // fp32 corpora of f32_shadow_min_bytes and more (half the bytes per pass); 0 = never
  int64_t f32_shadow_min_bytes = (int64_t)2 << 30;  // level 2 only: fp32 corpora from this size up (below it the extra launches cost more than half a pass saves)
  int last_shadow = 0;     // the last lookup's filter pass read the shadow
  Buffer d_accept, d_bits;  // message re-rank: accepted message ordinals, their bitmap
  Buffer d_emit;            // survivors of tavb_search_all: a counter, then the keys
  // load path (tavb_upload_rows): two pinned staging slots + two device scratch slots, recycled through events
  Buffer h_ring[2] = {{nullptr, 0, true}, {nullptr, 0, true}};
  Buffer d_ring[2];
  hipEvent_t ring_done[2] = {nullptr, nullptr};
  const int32_t* row_to_msg = nullptr;  // borrowed device map chunk row -> message ordinal
  int64_t row_to_msg_rows = 0, n_messages = 0;
  int64_t norm_rows = 0;  // rows of the corpus covered by the cached row-norm maxima (d_norm) -- and, for fp32 corpora, by the fp16 shadow
  Buffer h_stage{nullptr, 0, true};
  Buffer h_out{nullptr, 0, true};  // pinned + device-visible: the last kernel of a synchronous lookup writes its keys straight here
  Buffer h_lists{nullptr, 0, true};  // pinned + device-visible: per-workgroup lists of a small single-query lookup (merged on the host)
  Buffer h_flag{nullptr, 0, true};   // pinned: the work list of flagged queries read back by the one route that needs a host round trip (fp32 corpus, k > 64)
  int64_t mfma_bdirect = 0;  // option (measurement for now): the 256-query tile takes its query operand straight from L2 (fragment-major layout), not through LDS
  int64_t band_max = tavb::kBandMax;  // option: keys of a query's band the wide tile's selection hands to the rescoring (256 .. kBandMax); a band that does not fit flags the query
  int64_t early_exact = 1;    // option: ... and a batch found to be mostly such queries BEFORE the last filter phase skips that phase (needs wide_fallback)
  int64_t wide_fallback = 1;  // option: batches of 256+ queries re-run MANY (> 64) flagged queries on the 256-query tile's exact (split-plane) form
  int64_t small_direct_bytes = (int64_t)128 << 20;  // option: single-query lookups on corpora up to this size take the one-launch path (0 = never)
  int64_t small_direct_keys = 8192;                 // option: most keys the per-workgroup lists of such a lookup may hold (the grid is cut to fit; x 2 for a batch of 2 .. 8 queries)
  int64_t last_direct = 0;                          // option "last_direct" (get): 1 when the last lookup took it, 2 = with the query inside the kernel arguments
  int inline_query = 1;                             // option: 1536-wide single queries of that path ride in the kernel arguments (no H2D copy before the launch)
  // the GROUPED form of that path (ScanParams::group): batches of 2 .. direct_group_max_nq queries in one launch of gridDim.y query groups
  int64_t direct_group_max_nq = TAVB_MAX_GROUPED_QUERIES;  // option: biggest batch that may take it (0 / 1 = never: batches of up to 8 keep the plain form, bigger ones the tiles)
  int64_t direct_group = 0;                         // option: queries per group, 1 / 2 / 4 / 8, taken whatever the cost model says (0 = plan_direct_group)
  int64_t direct_group_wgs = 0;                     // option: most workgroups of such a launch (row workgroups x groups); 0 = plan_direct_group (256 or 512)
  bool dispatch_no_group = false;                   // set by tavb_search_batch around its fall-through: the host-synchronous cost model already said no
  int64_t direct_group_keys = 32768;                // option: most keys the lists of such a launch may hold (nq x workgroups-per-group x k; 256 KiB over PCIe)

  bool profiling = false;
  double total_ms[TAVB_KERNEL_COUNT] = {0};
  int64_t launches[TAVB_KERNEL_COUNT] = {0};
  std::vector<PendingTiming> pending;
  std::vector<hipEvent_t> free_events;

  int last_tier = 0;
  int pending_nq = 0, pending_k = 0;  // shape of the lookup enqueued by tavb_search_begin

  // small corpora (the reference's own scale: 10k x 1536, 41 us per call as three submissions): the H2D copy of the query, the scan and the merge
  // of a single-query lookup replayed as ONE captured HIP graph.  A few (corpus, k, min_score) shapes are kept.  OFF by default: measured on
  // MI355X / ROCm 7.2 (profiles/r03_latency_cfg1.md) the replay takes 48.3 us against 41.3 us for the three plain submissions -- hipGraphLaunch
  // costs more than it saves for a 3-node graph; the GPU-side floor of the lookup is the two kernels (scan 14.8 us + merge 11.8 us).
  struct SmallGraph {
    const void* corpus = nullptr;
    int64_t rows = 0;
    int32_t dim = 0, dtype = 0, k = 0;
    uint32_t thr_bits = 0;
    unsigned long long epoch = 0, geom_tag = 0;
    hipGraphExec_t exec = nullptr;
    int seen = 0;  // calls with this shape so far (the first one runs un-captured: it sizes the workspaces)
    unsigned long long last_used = 0;
  };
  SmallGraph graphs[4];
  unsigned long long graph_clock = 0;
  int64_t graph_max_bytes = 0;  // option "graph_max_bytes": single-query lookups on corpora up to this size replay a graph (0 = never, the default)
  int64_t last_graph = 0;                          // option "last_graph" (get): 1 when the last lookup was a graph replay

  // row-sharded corpora: this context's RCCL communicator (tavb_comm_init) and the buffers of the exchange
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  int64_t comm_force = 0;  // option: run the all-gather + merge even in a world of one (tests, dry runs of the N > 1 path)
  int64_t comm_fail_rank = -1;  // option (fault injection): the local search of tavb_search_allgather "fails" on this rank of the communicator
  int64_t comm_fail_alloc = 0;  // option (fault injection): the per-call allocations of tavb_search_allgather "fail" (lists beyond comm_reserve_keys)
  int64_t comm_stall_ms = 0;    // option (fault injection): the next exchange is held up on the stream for this long, as by a peer that is late
  int64_t comm_timeout_ms = 0;  // option: tavb_synchronize gives an exchange in flight this long before it aborts the communicator (0 = wait for ever)
  // keys of the exchange buffers reserved by tavb_comm_init (d_xlocal: that many, d_gather: x world): an exchange of up to that many keys per
  // rank allocates NOTHING between entering the call and ncclAllGather; a bigger one goes through the same buffers in chunks of whole queries
  int64_t comm_reserve_keys = (int64_t)1 << 20;
  bool comm_inflight = false;   // an exchange was enqueued since the last successful tavb_synchronize
  Buffer d_local;   // this shard's [nq, k] lists when they do not fit d_xlocal
  Buffer d_xlocal;  // this shard's lists of an exchange up to comm_reserve_keys keys; the TAVB_KEY_PEER_FAILED lists of a rank that failed
  Buffer d_gather;  // the all-gathered [world][chunk queries][k]
};

namespace {

struct Timed {
  tavb_ctx* c;
  int kernel;
  hipEvent_t a = nullptr, b = nullptr;
  Timed(tavb_ctx* ctx, int k) : c(ctx), kernel(k) {
    if (!c->profiling) return;
    auto get = [&](hipEvent_t* ev) {
      if (!c->free_events.empty()) {
        *ev = c->free_events.back();
        c->free_events.pop_back();
        return true;
      }
      return hipEventCreate(ev) == hipSuccess;
    };
    if (get(&a) && get(&b)) {
      (void)hipEventRecord(a, c->stream);
    } else {
      a = b = nullptr;
    }
  }
  ~Timed() {
    if (!a) return;
    (void)hipEventRecord(b, c->stream);
    c->pending.push_back({kernel, a, b});
  }
};

int drain_timings(tavb_ctx* c) {
  if (c->pending.empty()) return TAVB_OK;
  TAVB_HIP(hipStreamSynchronize(c->stream));
  for (auto& p : c->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
      c->total_ms[p.kernel] += ms;
      c->launches[p.kernel] += 1;
    }
    c->free_events.push_back(p.start);
    c->free_events.push_back(p.stop);
  }
  c->pending.clear();
  return TAVB_OK;
}

// Shape of a grouped one-launch lookup (tavb_search_batch on a small corpus, 2 .. 128 queries; ScanParams::group) and whether it is expected to
// beat the tiles.  Fitted to tools/group_sweep.py on MI355X (profiles/r06_group_sweep.md: rows 1000 .. 40000, D = 384 / 1536 / 3072, k = 10 at
// min_score 0 and k = 50 at 0.85), all in us per host-synchronous call:
//  * queries per group: ONE on fp16 corpora (1536-wide rows: the query stays in registers) and for up to ~10k (row, query) pairs, two on fp32
//    corpora beyond -- the smaller the group, the less a workgroup does besides reading rows (query staging, one 16-wave list merge per query),
//    and the rows are L2 / Infinity-Cache resident from the second group on;
//  * one workgroup per CU in all (256); two (512) for groups of two when one would walk a wave over more than ~6 row pairs;
//  * both routes pay 20 + 0.4 nq around their kernels on the host-synchronous call (staging and H2D copy of the queries, host merges / decode);
//  * grouped: 12 (launch + synchronise) + c x (rows x nq / 1000) for the scan, c = 0.08 / 0.19 / 0.41 (fp32) and 0.08 / 0.14 / 0.34 (fp16) at
//    D = 384 / 1536 / 3072 (with temporal row loads; a fifth more with the evict-first hint the single-query scan uses), + 0.6 per 1000 list
//    keys beyond 5000 (their way over PCIe and the host merge);
//  * the tiles (32/64-query tile, wide tile over the shadow) depend on how many rows survive `min_score` (fp16, 64 queries over 1000 rows: 154 at
//    min_score 0, 77 at 0.85) -- the estimate sits between the two: 35 + 0.035 D - 0.2 nq on fp32 corpora, 18 + 0.008 D + 0.4 nq on fp16 ones.
//    Up to 4 queries (2 on fp16) the alternative is the plain one-launch form or the streaming passes: the grouped form is never slower there;
//  * k <= 64 only (the 64-deep lists are what was measured).
struct DirectGroupPlan {
  int group;   // queries per group
  int blocks;  // row workgroups per group
  bool worth;  // predicted faster than the other routes (or forced by the `direct_group` option)
};
DirectGroupPlan plan_direct_group(const tavb_ctx* c, int nq, int k, int full_blocks, bool host) {
  const bool f16 = c->dtype == TAVB_F16;
  DirectGroupPlan p{};
  p.group = (f16 || (double)c->rows * nq <= 10000.0) ? 1 : 2;
  if (c->direct_group > 0) p.group = (int)c->direct_group;
  const int n_groups = (nq + p.group - 1) / p.group;
  int wgs = (p.group >= 2 && (double)c->rows * n_groups / (256.0 * 32.0) > 6.0) ? 512 : 256;
  if (c->direct_group_wgs > 0) wgs = (int)c->direct_group_wgs;
  // lists: nq x blocks x k keys over PCIe into pinned memory (`direct_group_keys`, 32768 = 256 KiB); blocks in whole rounds of the eight XCDs
  // (the device-resident form keeps its lists in device memory and merges them with a second launch: no such budget)
  int blocks = host ? (int)std::min<int64_t>(full_blocks, c->direct_group_keys / ((int64_t)k * nq)) : full_blocks;
  // whole rounds of the XCDs, within `wgs` in all (3 groups of 88 = 264 workgroups leave 8 CUs with two: 39 us against 27 for 4 groups of 64) --
  // except that 15 per group are 16, not 8 (33 groups of two at 512)
  const int per_group = wgs / n_groups;
  blocks = std::min(blocks, std::max(8, per_group >= 12 && per_group < 16 ? 16 : per_group / 8 * 8));
  p.blocks = blocks >= 8 ? blocks / 8 * 8 : blocks;
  if (p.blocks < 1 || (p.blocks < 8 && p.blocks != full_blocks)) return p;  // (worth = false)
  const double d = c->dim, wide = std::max(0.0, d - 1536.0);
  const double per_kpair = f16 ? 0.065 + 0.00005 * d + 0.00008 * wide : 0.045 + 0.000095 * d + 0.00005 * wide;
  const double keys = (double)nq * p.blocks * k;
  // what both routes pay around their kernels on the host-synchronous call (staging + H2D copy of the queries, Python-free part of the call);
  // the device-resident form pays a second launch (the merge) instead of the lists' way over PCIe
  const double around = host ? 20.0 + 0.4 * nq : 0.0;
  const double grouped_us = around + 12.0 + (host ? 0.0006 * std::max(0.0, keys - 5000.0) : 0.0) + per_kpair * ((double)c->rows * nq / 1000.0);
  // (device-resident form, measured as back-to-back submissions: 32 queries over 1000 fp32 rows 109 -> 18 us, 64 over 1000 fp16 rows 112 -> 20)
  const double tiles_us = around + (host ? (f16 ? 18.0 + 0.008 * d + 0.4 * nq : 35.0 + 0.035 * d - 0.2 * nq)
                                         : (f16 ? 0.85 * (22.0 + 0.008 * d + 0.8 * nq) : 38.0 + 0.03 * d));
  p.worth = c->direct_group > 0 || (host && nq <= (f16 ? 2 : 4)) || grouped_us <= tiles_us;
  return p;
}

int scan_blocks_for(const tavb_ctx* c, int64_t n_pos, int waves, int unroll) {
  int blocks = c->geom.blocks > 0 ? c->geom.blocks : c->n_cu;
  const int64_t per_block = (int64_t)waves * unroll;
  const int64_t needed = (n_pos + per_block - 1) / per_block;
  if (needed < blocks) blocks = (int)std::max<int64_t>(needed, 1);
  return blocks;
}

// Small corpus, 2 .. 128 device-resident queries: ONE grouped scan launch (ScanParams::group; plan_direct_group) + ONE merge launch -> d_out [nq, k]
// (async on the stream).  Bit for bit the answers of nq single-query scans.
int search_device_grouped(tavb_ctx* c, const float* d_q, int nq, int k, const float* min_scores /*host, nq*/, uint32_t index_base, u64_t* d_out,
                          const DirectGroupPlan& plan) {
  tavb::ScanGeometry g = c->geom;
  if (g.waves < 1) g.waves = 1;
  if (g.waves > 16) g.waves = 16;
  g.blocks = plan.blocks;
  g.nt = 0;  // (the rows are read again by every further group: no evict-first hint)
  if (int rc = c->d_lists.reserve((size_t)nq * g.blocks * k * sizeof(u64_t))) return rc;
  tavb::ScanParams p{};
  p.corpus = c->corpus;
  p.row_ids = nullptr;
  p.queries = d_q;
  p.lists = reinterpret_cast<u64_t*>(c->d_lists.ptr);  // [nq][blocks][k]
  p.n_pos = c->rows;
  p.dim = c->dim;
  p.dtype = c->dtype;
  p.nq = nq;
  p.k = k;
  p.index_base = index_base;
  p.key_bound = ~0ull;
  p.group = plan.group;
  for (int i = 0; i < TAVB_MAX_GROUPED_QUERIES; ++i) p.min_score[i] = (i < nq) ? min_scores[i] : INFINITY;
  {
    Timed t(c, TAVB_KERNEL_SCAN);
    hipError_t e = tavb::launch_scan(p, g, c->stream, &c->last_tier);
    if (e != hipSuccess) return fail(TAVB_E_HIP, "scan kernel launch failed: %s", hipGetErrorString(e));
  }
  {
    Timed t(c, TAVB_KERNEL_MERGE);
    hipError_t e = tavb::launch_merge(p.lists, g.blocks, nq, k, /*query_major=*/true, d_out, c->stream);
    if (e != hipSuccess) return fail(TAVB_E_HIP, "merge kernel launch failed: %s", hipGetErrorString(e));
  }
  return TAVB_OK;
}

// Core: queries on device (f32 [nq, dim]) -> sorted key lists d_out [nq, k] (async on the stream).
int search_device_impl(tavb_ctx* c, const float* d_q, int nq, int k, const float* min_scores /*host, nq*/,
                       const int32_t* d_row_ids, int64_t n_pos, uint32_t index_base, u64_t* d_out,
                       u64_t key_bound = ~0ull) {
  if (!c->corpus && c->rows != 0) return fail(TAVB_E_NO_CORPUS, "no corpus set (call tavb_set_corpus first)");
  if (n_pos <= 0) {
    TAVB_HIP(hipMemsetAsync(d_out, 0, (size_t)nq * k * sizeof(u64_t), c->stream));
    return TAVB_OK;
  }
  const int per_pass = (k > 64) ? 4 : TAVB_MAX_STREAM_QUERIES;
  tavb::ScanGeometry g = c->geom;
  if (g.waves < 1) g.waves = 1;
  if (g.waves > 16) g.waves = 16;
  g.blocks = scan_blocks_for(c, n_pos, g.waves, g.unroll);
  const size_t list_bytes = (size_t)per_pass * g.blocks * k * sizeof(u64_t);
  int rc = c->d_lists.reserve(list_bytes);
  if (rc) return rc;
  for (int q0 = 0; q0 < nq; q0 += per_pass) {
    const int n = std::min(per_pass, nq - q0);
    tavb::ScanParams p{};
    p.corpus = c->corpus;
    p.row_ids = d_row_ids;
    p.queries = d_q + (size_t)q0 * c->dim;
    p.lists = reinterpret_cast<u64_t*>(c->d_lists.ptr);
    p.n_pos = n_pos;
    p.dim = c->dim;
    p.dtype = c->dtype;
    p.nq = n;
    p.k = k;
    p.index_base = index_base;
    p.key_bound = key_bound;
    for (int i = 0; i < TAVB_MAX_STREAM_QUERIES; ++i) p.min_score[i] = (i < n) ? min_scores[q0 + i] : INFINITY;
    {
      Timed t(c, TAVB_KERNEL_SCAN);
      hipError_t e = tavb::launch_scan(p, g, c->stream, &c->last_tier);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "scan kernel launch failed: %s", hipGetErrorString(e));
    }
    {
      Timed t(c, TAVB_KERNEL_MERGE);
      hipError_t e = tavb::launch_merge(p.lists, g.blocks, n, k, /*query_major=*/true, d_out + (size_t)q0 * k, c->stream);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "merge kernel launch failed: %s", hipGetErrorString(e));
    }
  }
  return TAVB_OK;
}

int check_ctx(tavb_ctx* c) {
  if (!c) return fail(TAVB_E_INVALID, "null context");
  return TAVB_OK;
}

int check_search_args(tavb_ctx* c, int k) {
  if (int rc = check_ctx(c)) return rc;
  if (!c->corpus && c->rows != 0) return fail(TAVB_E_NO_CORPUS, "no corpus set (call tavb_set_corpus first)");
  if (c->dim <= 0) return fail(TAVB_E_NO_CORPUS, "no corpus set (call tavb_set_corpus first)");
  if (k < 1) return fail(TAVB_E_INVALID, "k must be >= 1 (got %d)", k);
  if (k > TAVB_MAX_FUSED_K)
    return fail(TAVB_E_UNSUPPORTED, "k=%d exceeds the fused-select limit %d; page with tavb_search_after / tavb_search_subset_after", k,
                TAVB_MAX_FUSED_K);
  return TAVB_OK;
}

u64_t host_key(float score, uint32_t index) {
  uint32_t bits;
  memcpy(&bits, &score, sizeof bits);
  return ((u64_t)bits << 32) | (u64_t)(0xFFFFFFFFu - index);
}

int cursor_key(float after_score, int64_t after_index, int64_t limit, u64_t* out) {
  if (!(after_score >= 0.0f && after_score <= 1.0f)) return fail(TAVB_E_INVALID, "cursor score must be in [0, 1]");
  if (after_index < 0 || after_index >= limit) return fail(TAVB_E_INVALID, "cursor index out of range");
  *out = host_key(after_score, (uint32_t)after_index);
  return TAVB_OK;
}

void decode(const u64_t* keys, int nq, int k, int64_t base, int64_t* ordinals, float* scores, int32_t* counts) {
  for (int q = 0; q < nq; ++q) {
    int m = 0;
    for (int i = 0; i < k; ++i) {
      const u64_t key = keys[(size_t)q * k + i];
      if (key == 0) break;  // lists are sorted: the first empty slot ends the list
      const uint32_t hi = (uint32_t)(key >> 32), lo = (uint32_t)key;
      float s;
      memcpy(&s, &hi, sizeof s);
      ordinals[(size_t)q * k + i] = (int64_t)(0xFFFFFFFFu - lo) + base;
      scores[(size_t)q * k + i] = s;
      ++m;
    }
    counts[q] = m;
  }
}

}  // namespace

int tavb_search_device_dispatch(tavb_ctx* c, const float* d_q, int nq, int k, const float* min_scores,
                                uint32_t index_base, u64_t* d_out);
extern "C" {

namespace {
int comm_wait_or_abort(tavb_ctx* c);  // (defined next to the RCCL bindings)
}

int tavb_version(void) { return TAVB_ABI_VERSION; }

const char* tavb_last_error(void) { return g_last_error.c_str(); }

int tavb_device_count(int* out_count) {
  if (!out_count) return fail(TAVB_E_INVALID, "null out_count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *out_count = 0;
    return fail(TAVB_E_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  *out_count = n;
  return TAVB_OK;
}

int tavb_create(int device, void* stream, tavb_ctx** out) {
  if (!out) return fail(TAVB_E_INVALID, "null out");
  *out = nullptr;
  int n = 0;
  TAVB_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(TAVB_E_INVALID, "device %d out of range (have %d)", device, n);
  DeviceGuard guard(device);
  if (!guard.ok) return fail(TAVB_E_HIP, "hipSetDevice(%d) failed", device);
  hipDeviceProp_t prop;
  TAVB_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(TAVB_E_UNSUPPORTED, "device %d is %s; libtavb is built for gfx950 (MI355X) only", device, prop.gcnArchName);
  tavb_ctx* c = new (std::nothrow) tavb_ctx();
  if (!c) return fail(TAVB_E_NOMEM, "out of host memory");
  c->device = device;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (stream) {
    c->stream = reinterpret_cast<hipStream_t>(stream);
    c->own_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      return fail(TAVB_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    c->own_stream = true;
  }
  *out = c;
  return TAVB_OK;
}

int tavb_destroy(tavb_ctx* c) {
  if (!c) return TAVB_OK;
  DeviceGuard guard(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& p : c->pending) {
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
  }
  for (auto& e : c->free_events) (void)hipEventDestroy(e);
  c->d_queries.release();
  c->d_queries_f16.release();
  c->d_lists.release();
  c->d_out.release();
  c->d_rows.release();
  c->d_cand.release();
  c->d_thr.release();
  c->d_sample_keys.release();
  c->d_counts.release();
  c->d_delta.release();
  c->d_approx.release();
  c->d_flag.release();
  c->d_fb_queries.release();
  c->d_norm.release();
  c->d_minscores.release();
  c->d_fb_cand.release();
  c->d_shadow.release();       // (was missing until round 5: a context that had served a 65+-query batch on an fp32 corpus left its shadow -- half the corpus' size -- behind)
  c->d_queries_pad.release();
  c->d_accept.release();
  c->d_bits.release();
  c->d_emit.release();
  for (int i = 0; i < 2; ++i) {
    c->h_ring[i].release();
    c->d_ring[i].release();
    if (c->ring_done[i]) (void)hipEventDestroy(c->ring_done[i]);
  }
  c->h_stage.release();
  c->h_out.release();
  c->h_lists.release();
  c->h_flag.release();
  for (auto& g : c->graphs)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  (void)tavb_comm_destroy(c);
  c->d_local.release();
  c->d_xlocal.release();
  c->d_gather.release();
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return TAVB_OK;
}

int tavb_synchronize(tavb_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  DeviceGuard guard(c->device);
  if (c->comm && c->comm_inflight && c->comm_timeout_ms > 0) return comm_wait_or_abort(c);
  TAVB_HIP(hipStreamSynchronize(c->stream));
  c->comm_inflight = false;
  return TAVB_OK;
}

int tavb_set_option(tavb_ctx* c, const char* name, int64_t v) {
  if (int rc = check_ctx(c)) return rc;
  if (!name) return fail(TAVB_E_INVALID, "null option name");
  std::string n(name);
  if (n == "scan_blocks") {
    if (v < 0 || v > 65535) return fail(TAVB_E_INVALID, "scan_blocks out of range");
    c->geom.blocks = (int)v;
  } else if (n == "scan_waves") {
    if (v < 1 || v > 16) return fail(TAVB_E_INVALID, "scan_waves must be 1..16");
    c->geom.waves = (int)v;
  } else if (n == "scan_unroll") {
    if (v != 1 && v != 2 && v != 4) return fail(TAVB_E_INVALID, "scan_unroll must be 1, 2 or 4");
    c->geom.unroll = (int)v;
  } else if (n == "scan_nt") {
    c->geom.nt = v ? 1 : 0;
  } else if (n == "scan_pipe") {
    c->geom.pipe = v ? 1 : 0;
  } else if (n == "force_tier") {
    if (v < 0 || v > 3) return fail(TAVB_E_INVALID, "force_tier must be 0..3");
    c->geom.tier = (int)v;
  } else if (n == "mfma_min_batch") {
    if (v < 1) return fail(TAVB_E_INVALID, "mfma_min_batch must be >= 1");
    c->mfma_min_batch = v;
  } else if (n == "mfma_min_batch_big") {
    if (v < 1) return fail(TAVB_E_INVALID, "mfma_min_batch_big must be >= 1");
    c->mfma_min_batch_big = v;
  } else if (n == "mfma_big_bytes") {
    if (v < 0) return fail(TAVB_E_INVALID, "mfma_big_bytes must be >= 0");
    c->mfma_big_bytes = v;
  } else if (n == "mfma_min_batch_big_f32") {
    if (v < 1) return fail(TAVB_E_INVALID, "mfma_min_batch_big_f32 must be >= 1");
    c->mfma_min_batch_big_f32 = v;
  } else if (n == "mfma_big_bytes_f32") {
    if (v < 0) return fail(TAVB_E_INVALID, "mfma_big_bytes_f32 must be >= 0");
    c->mfma_big_bytes_f32 = v;
  } else if (n == "mfma_few_bytes_f32") {
    if (v < 0) return fail(TAVB_E_INVALID, "mfma_few_bytes_f32 must be >= 0");
    c->mfma_few_bytes_f32 = v;
  } else if (n == "mfma_min_batch_f32") {
    if (v < 1) return fail(TAVB_E_INVALID, "mfma_min_batch_f32 must be >= 1");
    c->mfma_min_batch_f32 = v;
  } else if (n == "mfma_sample_rows") {
    if (v < -1) return fail(TAVB_E_INVALID, "mfma_sample_rows must be >= -1");
    c->mfma_sample_rows = v;
  } else if (n == "f32_shadow") {
    if (v < 0 || v > 2) return fail(TAVB_E_INVALID, "f32_shadow must be 0, 1 or 2");
    c->f32_shadow = v;
    if (!v) {
      c->d_shadow.release();
      if (c->dtype == TAVB_F32) c->norm_rows = 0;
    }
  } else if (n == "f32_shadow_min_bytes") {
    if (v < 0) return fail(TAVB_E_INVALID, "f32_shadow_min_bytes must be >= 0");
    c->f32_shadow_min_bytes = v;
  } else if (n == "mfma_tile") {
    if (v != 0 && v != 128 && v != 256) return fail(TAVB_E_INVALID, "mfma_tile must be 0 (auto), 128 or 256");
    c->mfma_tile = v;
  } else if (n == "mfma_sched") {
    if (v < 0 || v > 9) return fail(TAVB_E_INVALID, "mfma_sched must be 0..9");
    c->mfma_sched = v;
  } else if (n == "skinny_min_batch_f32") {
    if (v < 1) return fail(TAVB_E_INVALID, "skinny_min_batch_f32 must be >= 1");
    c->skinny_min_batch_f32 = v;
  } else if (n == "skinny_min_batch_f16") {
    if (v < 1) return fail(TAVB_E_INVALID, "skinny_min_batch_f16 must be >= 1");
    c->skinny_min_batch_f16 = v;
  } else if (n == "mfma_ladder") {
    if (v < 0 || v > 64) return fail(TAVB_E_INVALID, "mfma_ladder must be 0..64");
    c->mfma_ladder = v;
  } else if (n == "mfma_ablate") {
    if (v < 0 || v > 4095) return fail(TAVB_E_INVALID, "mfma_ablate must be 0..4095");
    c->mfma_ablate = v;
  } else if (n == "mfma_splits") {
    if (v < 0 || v > 4096) return fail(TAVB_E_INVALID, "mfma_splits out of range");
    c->mfma_splits = v;
  } else if (n == "band_max") {
    if (v < TAVB_MAX_FUSED_K || v > tavb::kBandMax) return fail(TAVB_E_INVALID, "band_max must be %d .. %d", TAVB_MAX_FUSED_K, tavb::kBandMax);
    c->band_max = v;
  } else if (n == "early_exact") {
    c->early_exact = v ? 1 : 0;
  } else if (n == "wide_fallback") {
    c->wide_fallback = v ? 1 : 0;
  } else if (n == "mfma_bdirect") {
    c->mfma_bdirect = v ? 1 : 0;
  } else if (n == "small_direct_keys") {
    if (v < 64 || v > (1 << 20)) return fail(TAVB_E_INVALID, "small_direct_keys must be 64 .. 1048576");
    c->small_direct_keys = v;
  } else if (n == "inline_query") {
    c->inline_query = v ? 1 : 0;
  } else if (n == "direct_group_max_nq") {
    if (v < 0 || v > TAVB_MAX_GROUPED_QUERIES) return fail(TAVB_E_INVALID, "direct_group_max_nq must be 0 .. %d", TAVB_MAX_GROUPED_QUERIES);
    c->direct_group_max_nq = v;
  } else if (n == "direct_group") {
    if (!(v == 0 || v == 1 || v == 2 || v == 4 || v == 8)) return fail(TAVB_E_INVALID, "direct_group must be 0, 1, 2, 4 or 8");
    c->direct_group = v;
  } else if (n == "direct_group_wgs") {
    if (v != 0 && (v < 8 || v > 65536)) return fail(TAVB_E_INVALID, "direct_group_wgs must be 0 or 8 .. 65536");
    c->direct_group_wgs = v;
  } else if (n == "direct_group_keys") {
    if (v < 64 || v > (1 << 22)) return fail(TAVB_E_INVALID, "direct_group_keys must be 64 .. 4194304");
    c->direct_group_keys = v;
  } else if (n == "small_direct_bytes") {
    if (v < 0) return fail(TAVB_E_INVALID, "small_direct_bytes must be >= 0");
    c->small_direct_bytes = v;
  } else if (n == "comm_force") {
    c->comm_force = v ? 1 : 0;
  } else if (n == "comm_fail_rank") {
    if (v < -1) return fail(TAVB_E_INVALID, "comm_fail_rank must be >= -1");
    c->comm_fail_rank = v;
  } else if (n == "comm_fail_alloc") {
    c->comm_fail_alloc = v ? 1 : 0;
  } else if (n == "comm_stall_ms") {
    if (v < 0 || v > 5000) return fail(TAVB_E_INVALID, "comm_stall_ms must be 0..5000");
    c->comm_stall_ms = v;
  } else if (n == "comm_timeout_ms") {
    if (v < 0) return fail(TAVB_E_INVALID, "comm_timeout_ms must be >= 0");
    c->comm_timeout_ms = v;
  } else if (n == "comm_reserve_keys") {
    if (c->comm) return fail(TAVB_E_INVALID, "comm_reserve_keys is read by tavb_comm_init: set it before");
    if (v < TAVB_MAX_FUSED_K || v > ((int64_t)1 << 28)) return fail(TAVB_E_INVALID, "comm_reserve_keys must be %d .. 2^28", TAVB_MAX_FUSED_K);
    c->comm_reserve_keys = v;
  } else if (n == "graph_max_bytes") {
    if (v < 0) return fail(TAVB_E_INVALID, "graph_max_bytes must be >= 0");
    c->graph_max_bytes = v;
  } else {
    return fail(TAVB_E_INVALID, "unknown option '%s'", name);
  }
  return TAVB_OK;
}

int tavb_get_option(tavb_ctx* c, const char* name, int64_t* out) {
  if (int rc = check_ctx(c)) return rc;
  if (!name || !out) return fail(TAVB_E_INVALID, "null argument");
  std::string n(name);
  if (n == "scan_blocks") *out = c->geom.blocks;
  else if (n == "scan_waves") *out = c->geom.waves;
  else if (n == "scan_unroll") *out = c->geom.unroll;
  else if (n == "scan_nt") *out = c->geom.nt;
  else if (n == "scan_pipe") *out = c->geom.pipe;
  else if (n == "force_tier") *out = c->geom.tier;
  else if (n == "mfma_min_batch") *out = c->mfma_min_batch;
  else if (n == "mfma_min_batch_big") *out = c->mfma_min_batch_big;
  else if (n == "mfma_big_bytes") *out = c->mfma_big_bytes;
  else if (n == "mfma_min_batch_big_f32") *out = c->mfma_min_batch_big_f32;
  else if (n == "mfma_big_bytes_f32") *out = c->mfma_big_bytes_f32;
  else if (n == "mfma_few_bytes_f32") *out = c->mfma_few_bytes_f32;
  else if (n == "mfma_min_batch_f32") *out = c->mfma_min_batch_f32;
  else if (n == "mfma_splits") *out = c->mfma_splits;
  else if (n == "mfma_tile") *out = c->mfma_tile;
  else if (n == "f32_shadow") *out = c->f32_shadow;
  else if (n == "last_shadow") *out = c->last_shadow;
  else if (n == "f32_shadow_min_bytes") *out = c->f32_shadow_min_bytes;
  else if (n == "mfma_ladder") *out = c->mfma_ladder;
  else if (n == "skinny_min_batch_f32") *out = c->skinny_min_batch_f32;
  else if (n == "skinny_min_batch_f16") *out = c->skinny_min_batch_f16;
  else if (n == "mfma_sample_rows") *out = c->mfma_sample_rows;
  else if (n == "compute_units") *out = c->n_cu;
  else if (n == "comm_force") *out = c->comm_force;
  else if (n == "comm_fail_rank") *out = c->comm_fail_rank;
  else if (n == "comm_fail_alloc") *out = c->comm_fail_alloc;
  else if (n == "comm_stall_ms") *out = c->comm_stall_ms;
  else if (n == "comm_timeout_ms") *out = c->comm_timeout_ms;
  else if (n == "comm_reserve_keys") *out = c->comm_reserve_keys;
  else if (n == "small_direct_bytes") *out = c->small_direct_bytes;
  else if (n == "wide_fallback") *out = c->wide_fallback;
  else if (n == "early_exact") *out = c->early_exact;
  else if (n == "band_max") *out = c->band_max;
  else if (n == "last_doomed") {  // queries of the last 256-query-tile lookup counted by the early verdict (synchronises); above nq / 2 the last filter phase was skipped
    *out = 0;
    if (c->d_flag.ptr) {
      DeviceGuard guard(c->device);
      int v = 0;
      TAVB_HIP(hipStreamSynchronize(c->stream));
      TAVB_HIP(hipMemcpy(&v, reinterpret_cast<const int*>(c->d_flag.ptr) + 1, sizeof v, hipMemcpyDeviceToHost));
      *out = v;
    }
  }
  else if (n == "mfma_bdirect") *out = c->mfma_bdirect;
  else if (n == "last_direct") *out = c->last_direct;
  else if (n == "inline_query") *out = c->inline_query;
  else if (n == "small_direct_keys") *out = c->small_direct_keys;
  else if (n == "direct_group_max_nq") *out = c->direct_group_max_nq;
  else if (n == "direct_group") *out = c->direct_group;
  else if (n == "direct_group_wgs") *out = c->direct_group_wgs;
  else if (n == "direct_group_keys") *out = c->direct_group_keys;
  else if (n == "graph_max_bytes") *out = c->graph_max_bytes;
  else if (n == "last_graph") *out = c->last_graph;
  else if (n == "comm_world") *out = c->comm ? c->comm_world : 0;
  else if (n == "comm_rank") *out = c->comm ? c->comm_rank : -1;
  else if (n == "last_tier") *out = c->last_tier;
  else if (n == "last_flagged") {  // queries of the last 256-query-tile lookup that were re-run on the exact tile (synchronises)
    *out = 0;
    if (c->d_flag.ptr) {
      DeviceGuard guard(c->device);
      int v = 0;
      TAVB_HIP(hipStreamSynchronize(c->stream));
      TAVB_HIP(hipMemcpy(&v, c->d_flag.ptr, sizeof v, hipMemcpyDeviceToHost));
      *out = v;
    }
  }
  else return fail(TAVB_E_INVALID, "unknown option '%s'", name);
  return TAVB_OK;
}

int tavb_set_corpus(tavb_ctx* c, const void* dev_rows, int64_t rows, int32_t dim, int32_t dtype, int64_t ordinal_base) {
  if (int rc = check_ctx(c)) return rc;
  if (rows < 0) return fail(TAVB_E_INVALID, "rows must be >= 0");
  if (dim < 1) return fail(TAVB_E_INVALID, "dim must be >= 1");
  if (dtype != TAVB_F32 && dtype != TAVB_F16) return fail(TAVB_E_INVALID, "dtype must be TAVB_F32 or TAVB_F16");
  if (rows > 0 && !dev_rows) return fail(TAVB_E_INVALID, "null corpus pointer with rows > 0");
  if (rows >= 0x7FFFFFFFll) return fail(TAVB_E_UNSUPPORTED, "at most 2^31-2 rows per device shard (got %lld)", (long long)rows);
  if (ordinal_base < 0) return fail(TAVB_E_INVALID, "ordinal_base must be >= 0");
  if (dev_rows != c->corpus || dim != c->dim || dtype != c->dtype || rows < c->norm_rows) c->norm_rows = 0;  // cached row-norm maximum: keep it across appends only
  if ((dtype != TAVB_F32 && dim % 64 == 0) || rows == 0) c->d_shadow.release();  // the fp16 shadow belongs to an fp32 corpus, or to an fp16 one of an odd width
  c->corpus = dev_rows;
  c->rows = rows;
  c->dim = dim;
  c->dtype = dtype;
  c->ordinal_base = ordinal_base;
  return TAVB_OK;
}

namespace {
// host -> pinned copy on a few threads: one core moves ~10 GB/s, PCIe Gen5 x16 takes ~50
void parallel_copy(void* dst, const void* src, size_t bytes) {
  constexpr size_t kMinPerThread = 2u << 20;
  unsigned hw = std::thread::hardware_concurrency();
  size_t n = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), bytes / kMinPerThread);
  if (n <= 1) {
    memcpy(dst, src, bytes);
    return;
  }
  const size_t per = ((bytes / n) + 63) & ~(size_t)63;
  std::vector<std::thread> pool;
  for (size_t i = 1; i < n; ++i) {
    const size_t off = i * per;
    if (off >= bytes) break;
    const size_t len = std::min(per, bytes - off);
    pool.emplace_back([=] { memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len); });
  }
  memcpy(dst, src, std::min(per, bytes));
  for (auto& t : pool) t.join();
}
}  // namespace

int tavb_upload_rows(tavb_ctx* c, const float* rows_host, int64_t n_rows, int32_t dim, void* dev_dst, int32_t dst_dtype) {
  if (int rc = check_ctx(c)) return rc;
  if (n_rows < 0 || dim < 1) return fail(TAVB_E_INVALID, "bad shape");
  if (dst_dtype != TAVB_F32 && dst_dtype != TAVB_F16) return fail(TAVB_E_INVALID, "dtype must be TAVB_F32 or TAVB_F16");
  if (n_rows == 0) return TAVB_OK;
  if (!rows_host || !dev_dst) return fail(TAVB_E_INVALID, "null pointer");
  DeviceGuard guard(c->device);
  const size_t row_bytes = (size_t)dim * sizeof(float);
  constexpr size_t kSlot = 16u << 20;  // 16 MiB per staging slot
  const int64_t rows_per_chunk = std::max<int64_t>(1, (int64_t)(kSlot / row_bytes));
  const size_t slot_bytes = (size_t)rows_per_chunk * row_bytes;
  for (int i = 0; i < 2; ++i) {
    if (int rc = c->h_ring[i].reserve(slot_bytes)) return rc;
    if (dst_dtype == TAVB_F16)
      if (int rc = c->d_ring[i].reserve(slot_bytes)) return rc;
    if (!c->ring_done[i]) TAVB_HIP(hipEventCreateWithFlags(&c->ring_done[i], hipEventDisableTiming));
  }
  const size_t dst_elem = dst_dtype == TAVB_F16 ? 2 : 4;
  int64_t done = 0;
  for (int chunk = 0; done < n_rows; ++chunk) {
    const int slot = chunk & 1;
    const int64_t n = std::min(rows_per_chunk, n_rows - done);
    const size_t bytes = (size_t)n * row_bytes;
    if (chunk >= 2) TAVB_HIP(hipEventSynchronize(c->ring_done[slot]));  // the copy (and convert) that used this slot two chunks ago are done
    parallel_copy(c->h_ring[slot].ptr, reinterpret_cast<const char*>(rows_host) + (size_t)done * row_bytes, bytes);  // overlaps the previous chunk's DMA
    char* dst = reinterpret_cast<char*>(dev_dst) + (size_t)done * dim * dst_elem;
    if (dst_dtype == TAVB_F32) {
      TAVB_HIP(hipMemcpyAsync(dst, c->h_ring[slot].ptr, bytes, hipMemcpyHostToDevice, c->stream));
    } else {
      TAVB_HIP(hipMemcpyAsync(c->d_ring[slot].ptr, c->h_ring[slot].ptr, bytes, hipMemcpyHostToDevice, c->stream));
      Timed t(c, TAVB_KERNEL_CONVERT);
      hipError_t e = tavb::launch_f32_to_f16(reinterpret_cast<const float*>(c->d_ring[slot].ptr), dst, n * dim, c->stream);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "convert launch failed: %s", hipGetErrorString(e));
    }
    TAVB_HIP(hipEventRecord(c->ring_done[slot], c->stream));
    done += n;
  }
  TAVB_HIP(hipStreamSynchronize(c->stream));  // the caller may free / reuse rows_host and read dev_dst from other streams
  return TAVB_OK;
}

int tavb_corpus_modified(tavb_ctx* c, int64_t first_row) {
  if (int rc = check_ctx(c)) return rc;
  if (first_row < 0) return fail(TAVB_E_INVALID, "first_row must be >= 0");
  if (first_row < c->norm_rows) c->norm_rows = 0;  // the cached row-norm maximum may be stale: recompute on the next batched lookup
  return TAVB_OK;
}

int tavb_normalize_rows_f32(tavb_ctx* c, const float* dev_in, float* dev_out, int64_t rows, int32_t dim) {
  if (int rc = check_ctx(c)) return rc;
  if (rows < 0 || dim < 1) return fail(TAVB_E_INVALID, "bad shape");
  if (rows == 0) return TAVB_OK;
  if (!dev_in || !dev_out) return fail(TAVB_E_INVALID, "null pointer");
  DeviceGuard guard(c->device);
  Timed t(c, TAVB_KERNEL_NORMALIZE);
  hipError_t e = tavb::launch_normalize_f32(dev_in, dev_out, rows, dim, c->stream);
  if (e != hipSuccess) return fail(TAVB_E_HIP, "normalize launch failed: %s", hipGetErrorString(e));
  return TAVB_OK;
}

int tavb_convert_f32_to_f16(tavb_ctx* c, const float* dev_in, void* dev_out, int64_t count) {
  if (int rc = check_ctx(c)) return rc;
  if (count < 0) return fail(TAVB_E_INVALID, "bad count");
  if (count == 0) return TAVB_OK;
  if (!dev_in || !dev_out) return fail(TAVB_E_INVALID, "null pointer");
  DeviceGuard guard(c->device);
  Timed t(c, TAVB_KERNEL_CONVERT);
  hipError_t e = tavb::launch_f32_to_f16(dev_in, dev_out, count, c->stream);
  if (e != hipSuccess) return fail(TAVB_E_HIP, "convert launch failed: %s", hipGetErrorString(e));
  return TAVB_OK;
}

int tavb_search_batch(tavb_ctx* c, const float* queries_host, int32_t nq, int32_t k, const float* min_scores,
                      int64_t* out_ordinals, float* out_scores, int32_t* out_counts) {
  if (int rc = check_search_args(c, k)) return rc;
  if (nq < 0) return fail(TAVB_E_INVALID, "nq must be >= 0");
  if (nq == 0) return TAVB_OK;
  if (!queries_host || !min_scores || !out_ordinals || !out_scores || !out_counts)
    return fail(TAVB_E_INVALID, "null argument");
  if (c->rows == 0) {
    for (int q = 0; q < nq; ++q) out_counts[q] = 0;
    return TAVB_OK;
  }
  DeviceGuard guard(c->device);
  const size_t qbytes = (size_t)nq * c->dim * sizeof(float);
  const size_t obytes = (size_t)nq * k * sizeof(u64_t);
  if (int rc = c->h_stage.reserve(qbytes)) return rc;
  if (int rc = c->h_out.reserve(obytes)) return rc;
  if (int rc = c->d_queries.reserve(qbytes)) return rc;
  parallel_copy(c->h_stage.ptr, queries_host, qbytes);  // (a 1024 x 1536 batch is 6 MiB: 0.6 ms on one core, a few threads from 4 MiB up)
  c->last_graph = 0;
  c->last_direct = 0;
  // ---- small corpus, one query: replay the captured (H2D, scan, merge) graph -- one submission instead of three
  const int64_t corpus_bytes = c->rows * c->dim * (c->dtype == TAVB_F16 ? 2 : 4);
  const bool streaming = nq == 1 && !(c->dtype == TAVB_F32 && c->f32_shadow >= 2 && corpus_bytes >= c->f32_shadow_min_bytes);
  tavb_ctx::SmallGraph* slot = nullptr;
  if (streaming && !c->profiling && c->graph_max_bytes > 0 && corpus_bytes <= c->graph_max_bytes) {
    uint32_t thr_bits;
    memcpy(&thr_bits, &min_scores[0], sizeof thr_bits);
    const unsigned long long geom_tag = ((unsigned long long)c->geom.blocks << 40) ^ ((unsigned long long)c->geom.waves << 32) ^ ((unsigned long long)c->geom.unroll << 24) ^
                                        ((unsigned long long)c->geom.nt << 16) ^ ((unsigned long long)c->geom.pipe << 8) ^ (unsigned long long)c->geom.tier;
    tavb_ctx::SmallGraph* oldest = &c->graphs[0];
    for (auto& g : c->graphs) {
      if (g.corpus == c->corpus && g.rows == c->rows && g.dim == c->dim && g.dtype == c->dtype && g.k == k && g.thr_bits == thr_bits && g.geom_tag == geom_tag) slot = &g;
      if (g.last_used < oldest->last_used) oldest = &g;
    }
    if (!slot) {  // a new shape takes the least recently used slot
      slot = oldest;
      if (slot->exec) (void)hipGraphExecDestroy(slot->exec);
      *slot = tavb_ctx::SmallGraph{};
      slot->corpus = c->corpus;
      slot->rows = c->rows;
      slot->dim = c->dim;
      slot->dtype = c->dtype;
      slot->k = k;
      slot->thr_bits = thr_bits;
      slot->geom_tag = geom_tag;
    }
    slot->last_used = ++c->graph_clock;
    if (slot->exec && slot->epoch != g_alloc_epoch) {  // a workspace moved since the capture: the graph holds stale pointers
      (void)hipGraphExecDestroy(slot->exec);
      slot->exec = nullptr;
      slot->seen = 1;
    }
    if (slot->exec) {
      TAVB_HIP(hipGraphLaunch(slot->exec, c->stream));
      TAVB_HIP(hipStreamSynchronize(c->stream));
      c->last_graph = 1;
      decode(reinterpret_cast<const u64_t*>(c->h_out.ptr), nq, k, c->ordinal_base, out_ordinals, out_scores, out_counts);
      return TAVB_OK;
    }
  }
  // ---- small corpus (the scale typeagent itself runs at: ~1.3k .. 10k rows), one query or a FEW (batched related-term lookups,
  //      adapters.install_batched_lookup_terms): ONE launch.  The scan's per-workgroup lists go straight into pinned host memory and are merged
  //      here -- the second launch that merged them on the device cost 11.8 us of a 41 us lookup (profiles/r03_latency_cfg1.md), and a batch of
  //      2 .. 8 queries took 36 .. 110 us through the device merge or the 32-query MFMA tile (profiles/r04_latency_small.md).  The grid is cut to
  //      what keeps the lists within `small_direct_keys` keys (8192 = 64 KiB over PCIe; twice that for a batch): 163 workgroups at k = 50,
  //      all of them at k <= 32.  A batch takes this path when its share of that budget still covers the rows in two rounds of the grid, and on
  //      fp16 corpora up to 4 queries: beyond that the multi-query scan (6 us more per query) loses to the 32-query tile (measured).
  //      Batches of up to `direct_group_max_nq` (128) queries take it in its GROUPED form (end of round 6): gridDim.y query groups of `group` queries each
  //      (ScanParams::group), every group a pass of its own over the rows -- which sit in L2 after the first one (workgroup (x, y) runs on XCD
  //      x % 8 for every y) -- wherever plan_direct_group expects it to beat the tiles.  Until then 9 .. 64 queries (5+ on fp16) went to
  //      the 32/64-query tile or the wide tile, both several launches and, on a corpus of a few thousand rows, one or two busy CUs:
  //      32 queries over 1000 fp32 rows 141 -> 52 us, 64 over 1000 fp16 rows 153 -> 62 us, the answers now the sequential lookups' bit for bit.
  const int direct_nq_max = (k > 64) ? 4 : TAVB_MAX_STREAM_QUERIES;  // queries one pass of the streaming kernels serves
  const bool shadow2 = c->dtype == TAVB_F32 && c->f32_shadow >= 2 && corpus_bytes >= c->f32_shadow_min_bytes;
  const bool few = nq >= 2 && nq <= direct_nq_max && !shadow2;
  const bool many = nq >= 2 && nq <= std::min<int64_t>(c->direct_group_max_nq, TAVB_MAX_GROUPED_QUERIES) && !shadow2 && k <= 64;  // (fitted for the 64-deep lists only)
  if ((streaming || few || many) && slot == nullptr && c->small_direct_bytes > 0 && corpus_bytes <= c->small_direct_bytes) {
    tavb::ScanGeometry g = c->geom;
    if (g.waves < 1) g.waves = 1;
    if (g.waves > 16) g.waves = 16;
    const int full_blocks = scan_blocks_for(c, c->rows, g.waves, g.unroll);
    // the grouped form (plan_direct_group): a launch of (row workgroups) x (query groups), every group a pass of its own over the rows
    DirectGroupPlan plan{};
    if (many) plan = plan_direct_group(c, nq, k, full_blocks, /*host=*/true);
    const bool grouped = many && plan.worth;
    bool take = grouped;
    if (grouped) {
      g.blocks = plan.blocks;
      g.nt = 0;  // the rows are read again by every further group: no evict-first hint (16 queries over 10k fp32 rows: scan 47 -> 41 us)
    }
    if (!take && (streaming || few)) {
      const int64_t budget = c->small_direct_keys * (nq > 1 ? 2 : 1);
      g.blocks = std::min(full_blocks, (int)std::max<int64_t>(8, budget / ((int64_t)k * nq)));
      const int64_t rounds = (c->rows + (int64_t)g.blocks * g.waves * g.unroll - 1) / ((int64_t)g.blocks * g.waves * g.unroll);
      // one query: only while the cut grid keeps at least half of the full one (k = 256 would leave 32 workgroups to stream up to 128 MiB: slower
      // than the full grid + the device merge; measured at k <= 50, where 163+ of 204 workgroups stay)
      take = (nq == 1 && 2 * g.blocks >= full_blocks) || (nq > 1 && rounds <= 2 && (c->dtype == TAVB_F32 || nq <= 4));
    }
    if (take) {
      const size_t list_keys = (size_t)nq * g.blocks * k;
      if (int rc = c->h_lists.reserve((list_keys + (size_t)nq * k) * sizeof(u64_t))) return rc;  // + the merged keys
      tavb::ScanParams p{};
      p.corpus = c->corpus;
      p.row_ids = nullptr;
      p.queries = reinterpret_cast<const float*>(c->d_queries.ptr);
      p.lists = reinterpret_cast<u64_t*>(c->h_lists.ptr);  // [nq][blocks][k]
      p.n_pos = c->rows;
      p.dim = c->dim;
      p.dtype = c->dtype;
      p.nq = nq;
      p.k = k;
      p.index_base = 0u;
      p.key_bound = ~0ull;
      p.group = grouped ? plan.group : 0;
      for (int i = 0; i < TAVB_MAX_GROUPED_QUERIES; ++i) p.min_score[i] = (i < nq) ? min_scores[i] : INFINITY;
      {
        // one 1536-wide query (the embedding size typeagent runs at): it rides in the kernel arguments -- one submission, no copy in front of the launch
        hipError_t e = hipSuccess;
        bool launched = false;
        if (c->inline_query && nq == 1) {
          Timed t(c, TAVB_KERNEL_SCAN);
          launched = tavb::launch_scan_inline_query(p, g, c->stream, queries_host, &c->last_tier, &e);
        }
        if (!launched) {
          TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, c->h_stage.ptr, qbytes, hipMemcpyHostToDevice, c->stream));
          Timed t(c, TAVB_KERNEL_SCAN);
          e = tavb::launch_scan(p, g, c->stream, &c->last_tier);
        }
        if (e != hipSuccess) return fail(TAVB_E_HIP, "scan kernel launch failed: %s", hipGetErrorString(e));
        c->last_direct = launched ? 2 : (grouped ? 3 : 1);
      }
      TAVB_HIP(hipStreamSynchronize(c->stream));
      tavb_key* merged = reinterpret_cast<tavb_key*>(c->h_lists.ptr) + list_keys;
      // The lists were just written by the device: every cache line of them is a miss to DRAM for this core (~100 ns), and a merge hops between
      // its lists -- 16 lines per query one after the other were 1.6 us per query, 51 us of a 32-term batch.  The heads of the NEXT query's lists
      // are prefetched while this one is merged (two lines per list: the merge rarely reads further), so the misses overlap.
      const tavb_key* all = reinterpret_cast<const tavb_key*>(c->h_lists.ptr);
      auto prefetch_query = [&](int q) {
        const tavb_key* base = all + (size_t)q * g.blocks * k;
        const int n = std::min(g.blocks, 256);
        for (int l = 0; l < n; ++l) {
          __builtin_prefetch(base + (size_t)l * k);
          if (k > 8 && g.blocks <= 64) __builtin_prefetch(base + (size_t)l * k + 8);
        }
      };
      prefetch_query(0);
      for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) prefetch_query(q + 1);
        if (int rc = tavb_merge_keys_host(all + (size_t)q * g.blocks * k, g.blocks, 1, k, merged + (size_t)q * k)) return rc;
      }
      decode(reinterpret_cast<const u64_t*>(merged), nq, k, c->ordinal_base, out_ordinals, out_scores, out_counts);
      return TAVB_OK;
    }
  }
  // (a batch the host-synchronous cost model kept off the grouped form stays off it: the device-resident model below prices submissions that
  //  are not waited for one by one)
  struct NoGroup {
    tavb_ctx* c;
    explicit NoGroup(tavb_ctx* ctx) : c(ctx) { c->dispatch_no_group = true; }
    ~NoGroup() { c->dispatch_no_group = false; }
  } no_group(c);
  const bool capture = slot != nullptr && slot->seen >= 1;  // (the first call of a shape sizes the workspaces: no allocation may happen inside a capture)
  if (slot) ++slot->seen;
  if (capture) TAVB_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  hipError_t copy_err = hipMemcpyAsync(c->d_queries.ptr, c->h_stage.ptr, qbytes, hipMemcpyHostToDevice, c->stream);
  int rc = copy_err == hipSuccess ? tavb_search_device_dispatch(c, reinterpret_cast<const float*>(c->d_queries.ptr), nq, k, min_scores, 0u,
                                                                reinterpret_cast<u64_t*>(c->h_out.ptr))
                                  : fail(TAVB_E_HIP, "hipMemcpyAsync of the query failed: %s", hipGetErrorString(copy_err));
  if (capture) {
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(c->stream, &graph);
    if (rc == TAVB_OK && e == hipSuccess && graph) {
      hipGraphExec_t exec = nullptr;
      e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      if (e == hipSuccess) {
        slot->exec = exec;
        slot->epoch = g_alloc_epoch;
      }
    }
    if (graph) (void)hipGraphDestroy(graph);
    if (rc) return rc;
    if (!slot->exec) {  // capture or instantiation failed: this shape stays on the plain path
      (void)hipGetLastError();
      slot->seen = -1000000;
      TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, c->h_stage.ptr, qbytes, hipMemcpyHostToDevice, c->stream));
      if (int rc2 = tavb_search_device_dispatch(c, reinterpret_cast<const float*>(c->d_queries.ptr), nq, k, min_scores, 0u, reinterpret_cast<u64_t*>(c->h_out.ptr)))
        return rc2;
    } else {
      TAVB_HIP(hipGraphLaunch(slot->exec, c->stream));  // nothing ran during the capture: this is the lookup
      c->last_graph = 1;
    }
  }
  if (rc) return rc;
  // no D2H copy: the merge kernel wrote the keys into pinned host memory
  TAVB_HIP(hipStreamSynchronize(c->stream));
  decode(reinterpret_cast<const u64_t*>(c->h_out.ptr), nq, k, c->ordinal_base, out_ordinals, out_scores, out_counts);
  return TAVB_OK;
}

int tavb_search_begin(tavb_ctx* c, const float* queries_host, int32_t nq, int32_t k, const float* min_scores, const tavb_key* cursor) {
  if (int rc = check_search_args(c, k)) return rc;
  if (nq < 1) return fail(TAVB_E_INVALID, "nq must be >= 1");
  if (!queries_host || !min_scores) return fail(TAVB_E_INVALID, "null argument");
  if (cursor && nq != 1) return fail(TAVB_E_INVALID, "a cursor goes with exactly one query");
  if (c->ordinal_base + c->rows >= 0xFFFFFFFFll)
    return fail(TAVB_E_UNSUPPORTED, "keys hold 32-bit ordinals: ordinal_base + rows must be < 2^32 - 1");
  DeviceGuard guard(c->device);
  const size_t qbytes = (size_t)nq * c->dim * sizeof(float);
  const size_t obytes = (size_t)nq * k * sizeof(u64_t);
  if (int rc = c->h_stage.reserve(qbytes)) return rc;
  if (int rc = c->h_out.reserve(obytes)) return rc;
  if (int rc = c->d_queries.reserve(qbytes)) return rc;
  c->pending_nq = c->pending_k = 0;
  if (c->rows == 0) {
    memset(c->h_out.ptr, 0, obytes);
  } else {
    memcpy(c->h_stage.ptr, queries_host, qbytes);
    TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, c->h_stage.ptr, qbytes, hipMemcpyHostToDevice, c->stream));
    int rc;
    if (cursor)
      rc = search_device_impl(c, reinterpret_cast<const float*>(c->d_queries.ptr), 1, k, min_scores, nullptr, c->rows, (uint32_t)c->ordinal_base,
                              reinterpret_cast<u64_t*>(c->h_out.ptr), (u64_t)*cursor);
    else
      rc = tavb_search_device_dispatch(c, reinterpret_cast<const float*>(c->d_queries.ptr), nq, k, min_scores, (uint32_t)c->ordinal_base,
                                       reinterpret_cast<u64_t*>(c->h_out.ptr));
    if (rc) return rc;
  }
  c->pending_nq = nq;
  c->pending_k = k;
  return TAVB_OK;
}

int tavb_search_end(tavb_ctx* c, int32_t nq, int32_t k, tavb_key* out_keys_host) {
  if (int rc = check_ctx(c)) return rc;
  if (!out_keys_host) return fail(TAVB_E_INVALID, "null argument");
  if (nq != c->pending_nq || k != c->pending_k || nq < 1) return fail(TAVB_E_INVALID, "tavb_search_end does not match the pending tavb_search_begin");
  DeviceGuard guard(c->device);
  TAVB_HIP(hipStreamSynchronize(c->stream));
  memcpy(out_keys_host, c->h_out.ptr, (size_t)nq * k * sizeof(u64_t));
  c->pending_nq = c->pending_k = 0;
  return TAVB_OK;
}

int tavb_merge_keys_host(const tavb_key* lists, int32_t n_lists, int32_t nq, int32_t k, tavb_key* out) {
  if (n_lists < 1 || nq < 0 || k < 1) return fail(TAVB_E_INVALID, "bad merge shape");
  if (nq == 0) return TAVB_OK;
  if (!lists || !out) return fail(TAVB_E_INVALID, "null argument");
  // Every list is sorted best first, so its j-th key bounds j of its keys from below.  With j = ceil(k / n_lists) and t = the m-th largest of the
  // lists' j-th keys, m = ceil(k / j), at least m * j >= k keys are >= t: the k best overall all are, and they sit in the prefixes (down to t) of
  // the lists whose head is >= t.  One pass over n_lists keys, a selection among them, a sort of a few dozen keys: 0.6 us for the 204 lists of a
  // 10k-row lookup and 0.9 us for 40 lists of 50, where picking the maximum head k times took k * n_lists steps (3.1 us of a 30 us call;
  // profiles/r04_latency_small.md).
  // A FEW lists (the grouped one-launch form leaves 8 .. 32 per query, and there are up to 64 queries to merge): a plain k-way merge, the
  // largest head k times -- with 8 lists of 50 the selection above keeps most of their 400 keys for the sort (~2 us per query, 64 us for
  // a 32-term batch of a 91 us call); k * n_lists steps are ~0.3 us.
  if (n_lists <= 16 && nq == 1) {
    const tavb_key* head[16];
    int left[16];
    for (int l = 0; l < n_lists; ++l) {
      head[l] = lists + (size_t)l * k;
      left[l] = k;
    }
    for (int i = 0; i < k; ++i) {
      int best = -1;
      u64_t best_key = 0;
      for (int l = 0; l < n_lists; ++l)
        if (left[l] > 0 && *head[l] > best_key) {
          best_key = *head[l];
          best = l;
        }
      out[i] = best_key;  // 0 once every list is exhausted (an empty slot of a list is 0 too, and the lists are sorted: nothing behind it)
      if (best >= 0) {
        ++head[best];
        --left[best];
      }
    }
    return TAVB_OK;
  }
  static thread_local std::vector<u64_t> pool;
  const int j = (k + n_lists - 1) / n_lists;
  const int m = (k + j - 1) / j;  // <= n_lists
  for (int q = 0; q < nq; ++q) {
    pool.resize((size_t)n_lists);
    for (int l = 0; l < n_lists; ++l) pool[l] = lists[((size_t)l * nq + q) * k + (j - 1)];
    std::nth_element(pool.begin(), pool.begin() + (m - 1), pool.end(), std::greater<u64_t>());
    const u64_t t = std::max<u64_t>(pool[m - 1], 1);  // (0 = an empty slot, never a result: fewer than k keys in all, take whatever there is)
    pool.clear();
    for (int l = 0; l < n_lists; ++l) {
      const tavb_key* list = lists + ((size_t)l * nq + q) * k;
      for (int i = 0; i < k && list[i] >= t; ++i) pool.push_back(list[i]);
    }
    const size_t take = std::min<size_t>((size_t)k, pool.size());
    std::partial_sort(pool.begin(), pool.begin() + take, pool.end(), std::greater<u64_t>());
    for (size_t i = 0; i < (size_t)k; ++i) out[(size_t)q * k + i] = i < take ? pool[i] : 0;  // 0 once every list is exhausted
  }
  return TAVB_OK;
}

int tavb_search(tavb_ctx* c, const float* query_host, int32_t k, float min_score, int64_t* out_ordinals,
                float* out_scores, int32_t* out_count) {
  return tavb_search_batch(c, query_host, 1, k, &min_score, out_ordinals, out_scores, out_count);
}

static int search_subset_impl(tavb_ctx* c, const float* query_host, const int64_t* rows_host, int64_t n_subset, int32_t k,
                              float min_score, bool has_cursor, float after_score, int64_t after_position,
                              int64_t* out_positions, float* out_scores, int32_t* out_count) {
  if (int rc = check_search_args(c, k)) return rc;
  if (n_subset < 0) return fail(TAVB_E_INVALID, "n_subset must be >= 0");
  if (!query_host || !out_positions || !out_scores || !out_count) return fail(TAVB_E_INVALID, "null argument");
  if (n_subset == 0 || c->rows == 0) {
    *out_count = 0;
    return TAVB_OK;
  }
  if (!rows_host) return fail(TAVB_E_INVALID, "null rows_host");
  if (n_subset >= 0x7FFFFFFFll) return fail(TAVB_E_UNSUPPORTED, "subset too long");
  u64_t bound = ~0ull;
  if (has_cursor) {
    if (int rc = cursor_key(after_score, after_position, n_subset, &bound)) return rc;
  }
  DeviceGuard guard(c->device);
  const size_t qbytes = (size_t)c->dim * sizeof(float);
  const size_t rbytes = (size_t)n_subset * sizeof(int32_t);
  const size_t obytes = (size_t)k * sizeof(u64_t);
  const size_t qoff = (rbytes + 255) & ~(size_t)255;
  if (int rc = c->h_stage.reserve(qoff + qbytes)) return rc;
  if (int rc = c->h_out.reserve(obytes)) return rc;
  if (int rc = c->d_queries.reserve(qbytes)) return rc;
  if (int rc = c->d_rows.reserve(rbytes)) return rc;
  int32_t* r32 = reinterpret_cast<int32_t*>(c->h_stage.ptr);
  for (int64_t i = 0; i < n_subset; ++i) {
    const int64_t r = rows_host[i];
    if (r < 0 || r >= c->rows)
      return fail(TAVB_E_INVALID, "subset row %lld out of range [0, %lld)", (long long)r, (long long)c->rows);
    r32[i] = (int32_t)r;
  }
  memcpy(reinterpret_cast<char*>(c->h_stage.ptr) + qoff, query_host, qbytes);
  TAVB_HIP(hipMemcpyAsync(c->d_rows.ptr, c->h_stage.ptr, rbytes, hipMemcpyHostToDevice, c->stream));
  TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, reinterpret_cast<char*>(c->h_stage.ptr) + qoff, qbytes,
                          hipMemcpyHostToDevice, c->stream));
  int rc = search_device_impl(c, reinterpret_cast<const float*>(c->d_queries.ptr), 1, k, &min_score,
                              reinterpret_cast<const int32_t*>(c->d_rows.ptr), n_subset, 0u,
                              reinterpret_cast<u64_t*>(c->h_out.ptr), bound);
  if (rc) return rc;
  TAVB_HIP(hipStreamSynchronize(c->stream));
  decode(reinterpret_cast<const u64_t*>(c->h_out.ptr), 1, k, 0, out_positions, out_scores, out_count);
  return TAVB_OK;
}

// one pass, every survivor: keys on the device -> host, sorted best first, the first max_out decoded
static int search_all_impl(tavb_ctx* c, const float* query_host, const int64_t* rows_host, int64_t n_subset, bool subset, float min_score,
                           int64_t max_out, int64_t* out_items, float* out_scores, int64_t* out_count, int64_t* out_total) {
  if (int rc = check_ctx(c)) return rc;
  if (c->dim <= 0 || (!c->corpus && c->rows != 0)) return fail(TAVB_E_NO_CORPUS, "no corpus set (call tavb_set_corpus first)");
  if (!query_host || !out_count || !out_total || max_out < 0 || (max_out > 0 && (!out_items || !out_scores)))
    return fail(TAVB_E_INVALID, "bad argument");
  *out_count = 0;
  *out_total = 0;
  const int64_t n_pos = subset ? n_subset : c->rows;
  if (n_pos < 0 || n_pos >= 0x7FFFFFFFll) return fail(TAVB_E_INVALID, "bad subset length");
  if (n_pos == 0 || c->rows == 0) return TAVB_OK;
  if (subset && !rows_host) return fail(TAVB_E_INVALID, "null rows_host");
  DeviceGuard guard(c->device);
  const size_t qbytes = (size_t)c->dim * sizeof(float);
  const size_t rbytes = subset ? (size_t)n_subset * sizeof(int32_t) : 0;
  const size_t qoff = (rbytes + 255) & ~(size_t)255;
  if (int rc = c->h_stage.reserve(qoff + qbytes)) return rc;
  if (int rc = c->d_queries.reserve(qbytes)) return rc;
  if (int rc = c->d_emit.reserve(256 + (size_t)n_pos * sizeof(u64_t))) return rc;
  if (subset) {
    if (int rc = c->d_rows.reserve(rbytes)) return rc;
    int32_t* r32 = reinterpret_cast<int32_t*>(c->h_stage.ptr);
    for (int64_t i = 0; i < n_subset; ++i) {
      const int64_t r = rows_host[i];
      if (r < 0 || r >= c->rows) return fail(TAVB_E_INVALID, "subset row %lld out of range [0, %lld)", (long long)r, (long long)c->rows);
      r32[i] = (int32_t)r;
    }
    TAVB_HIP(hipMemcpyAsync(c->d_rows.ptr, c->h_stage.ptr, rbytes, hipMemcpyHostToDevice, c->stream));
  }
  memcpy(reinterpret_cast<char*>(c->h_stage.ptr) + qoff, query_host, qbytes);
  TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, reinterpret_cast<char*>(c->h_stage.ptr) + qoff, qbytes, hipMemcpyHostToDevice, c->stream));
  unsigned long long* d_counter = reinterpret_cast<unsigned long long*>(c->d_emit.ptr);
  u64_t* d_keys = reinterpret_cast<u64_t*>(reinterpret_cast<char*>(c->d_emit.ptr) + 256);
  TAVB_HIP(hipMemsetAsync(d_counter, 0, sizeof(unsigned long long), c->stream));
  tavb::ScanParams p{};
  p.corpus = c->corpus;
  p.row_ids = subset ? reinterpret_cast<const int32_t*>(c->d_rows.ptr) : nullptr;
  p.queries = reinterpret_cast<const float*>(c->d_queries.ptr);
  p.n_pos = n_pos;
  p.dim = c->dim;
  p.dtype = c->dtype;
  p.nq = 1;
  p.k = 1;
  p.index_base = 0u;
  p.key_bound = ~0ull;
  p.min_score[0] = min_score;
  {
    Timed t(c, TAVB_KERNEL_SCAN);
    const int blocks = (int)std::min<int64_t>(c->n_cu, (n_pos + 15) / 16);
    hipError_t e = tavb::launch_scan_emit(p, std::max(blocks, 1), d_keys, (unsigned long long)n_pos, d_counter, c->stream);
    if (e != hipSuccess) return fail(TAVB_E_HIP, "emit scan launch failed: %s", hipGetErrorString(e));
  }
  unsigned long long total = 0;
  TAVB_HIP(hipMemcpyAsync(&total, d_counter, sizeof total, hipMemcpyDeviceToHost, c->stream));
  TAVB_HIP(hipStreamSynchronize(c->stream));
  *out_total = (int64_t)total;
  if (total == 0 || max_out == 0) return TAVB_OK;
  std::vector<u64_t> keys((size_t)total);
  TAVB_HIP(hipMemcpy(keys.data(), d_keys, (size_t)total * sizeof(u64_t), hipMemcpyDeviceToHost));
  const size_t want = (size_t)std::min<int64_t>((int64_t)total, max_out);
  if (want < keys.size())
    std::partial_sort(keys.begin(), keys.begin() + want, keys.end(), std::greater<u64_t>());
  else
    std::sort(keys.begin(), keys.end(), std::greater<u64_t>());
  const int64_t base = subset ? 0 : c->ordinal_base;
  for (size_t i = 0; i < want; ++i) {
    const uint32_t hi = (uint32_t)(keys[i] >> 32), lo = (uint32_t)keys[i];
    float sc;
    memcpy(&sc, &hi, sizeof sc);
    out_items[i] = (int64_t)(0xFFFFFFFFu - lo) + base;
    out_scores[i] = sc;
  }
  *out_count = (int64_t)want;
  return TAVB_OK;
}

int tavb_search_all(tavb_ctx* c, const float* query_host, float min_score, int64_t max_out, int64_t* out_ordinals, float* out_scores,
                    int64_t* out_count, int64_t* out_total) {
  return search_all_impl(c, query_host, nullptr, 0, false, min_score, max_out, out_ordinals, out_scores, out_count, out_total);
}

int tavb_search_subset_all(tavb_ctx* c, const float* query_host, const int64_t* rows_host, int64_t n_subset, float min_score, int64_t max_out,
                           int64_t* out_positions, float* out_scores, int64_t* out_count, int64_t* out_total) {
  return search_all_impl(c, query_host, rows_host, n_subset, true, min_score, max_out, out_positions, out_scores, out_count, out_total);
}

int tavb_set_row_messages(tavb_ctx* c, const int32_t* dev_row_to_msg, int64_t rows, int64_t n_messages) {
  if (int rc = check_ctx(c)) return rc;
  if (rows < 0 || n_messages < 0) return fail(TAVB_E_INVALID, "bad shape");
  if (rows > 0 && !dev_row_to_msg) return fail(TAVB_E_INVALID, "null map with rows > 0");
  if (n_messages >= 0xFFFFFFFFll) return fail(TAVB_E_UNSUPPORTED, "message ordinals must stay below 2^32 - 1");
  c->row_to_msg = dev_row_to_msg;
  c->row_to_msg_rows = rows;
  c->n_messages = n_messages;
  return TAVB_OK;
}

// hits (device keys [k]) -> message keys in pinned host memory -> caller's arrays
static int rerank_and_return(tavb_ctx* c, const u64_t* d_hits, int k, const int32_t* d_pos_to_row, const int32_t* accept_msgs_host, int64_t n_accept,
                             bool filtered, int32_t max_messages, int64_t* out_messages, float* out_scores, int32_t* out_count) {
  const uint32_t* d_bits = nullptr;
  if (filtered) {
    const size_t words = (size_t)((c->n_messages + 31) / 32) + 1;
    if (int rc = c->d_bits.reserve(words * 4)) return rc;
    TAVB_HIP(hipMemsetAsync(c->d_bits.ptr, 0, words * 4, c->stream));
    if (n_accept > 0) {
      if (int rc = c->d_accept.reserve((size_t)n_accept * 4)) return rc;
      // (pageable source: the copy is staged by the runtime before the call returns)
      TAVB_HIP(hipMemcpyAsync(c->d_accept.ptr, accept_msgs_host, (size_t)n_accept * 4, hipMemcpyHostToDevice, c->stream));
      hipError_t e = tavb::launch_accept_bitmap(reinterpret_cast<const int32_t*>(c->d_accept.ptr), n_accept, reinterpret_cast<uint32_t*>(c->d_bits.ptr),
                                                c->n_messages, c->stream);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "bitmap launch failed: %s", hipGetErrorString(e));
    }
    d_bits = reinterpret_cast<const uint32_t*>(c->d_bits.ptr);
  }
  hipError_t e = tavb::launch_message_rerank(d_hits, 1, k, 0u, d_pos_to_row, c->row_to_msg, c->row_to_msg_rows, d_bits, c->n_messages, max_messages,
                                             reinterpret_cast<u64_t*>(c->h_out.ptr), c->stream);
  if (e != hipSuccess) return fail(TAVB_E_HIP, "re-rank launch failed: %s", hipGetErrorString(e));
  TAVB_HIP(hipStreamSynchronize(c->stream));
  decode(reinterpret_cast<const u64_t*>(c->h_out.ptr), 1, k, 0, out_messages, out_scores, out_count);
  return TAVB_OK;
}

static int check_message_args(tavb_ctx* c, int k, int32_t max_messages) {
  if (int rc = check_search_args(c, k)) return rc;
  if (!c->row_to_msg && c->rows > 0) return fail(TAVB_E_NO_CORPUS, "no row -> message map set (call tavb_set_row_messages first)");
  if (c->row_to_msg_rows < c->rows) return fail(TAVB_E_INVALID, "the row -> message map covers %lld rows, the corpus has %lld", (long long)c->row_to_msg_rows, (long long)c->rows);
  if (max_messages < 0) return fail(TAVB_E_INVALID, "max_messages must be >= 0");
  return TAVB_OK;
}

int tavb_search_messages(tavb_ctx* c, const float* query_host, int32_t k, float min_score, const int32_t* accept_msgs_host, int64_t n_accept,
                         int32_t max_messages, int64_t* out_messages, float* out_scores, int32_t* out_count) {
  if (int rc = check_message_args(c, k, max_messages)) return rc;
  if (!query_host || !out_messages || !out_scores || !out_count) return fail(TAVB_E_INVALID, "null argument");
  if (n_accept < -1 || (n_accept > 0 && !accept_msgs_host)) return fail(TAVB_E_INVALID, "bad accept list");
  *out_count = 0;
  if (c->rows == 0) return TAVB_OK;
  DeviceGuard guard(c->device);
  const size_t qbytes = (size_t)c->dim * sizeof(float);
  const size_t obytes = (size_t)k * sizeof(u64_t);
  if (int rc = c->h_stage.reserve(qbytes)) return rc;
  if (int rc = c->h_out.reserve(obytes)) return rc;
  if (int rc = c->d_queries.reserve(qbytes)) return rc;
  if (int rc = c->d_out.reserve(obytes)) return rc;
  memcpy(c->h_stage.ptr, query_host, qbytes);
  TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, c->h_stage.ptr, qbytes, hipMemcpyHostToDevice, c->stream));
  // keys carry LOCAL rows here (index_base 0): they only index the map
  if (int rc = search_device_impl(c, reinterpret_cast<const float*>(c->d_queries.ptr), 1, k, &min_score, nullptr, c->rows, 0u,
                                  reinterpret_cast<u64_t*>(c->d_out.ptr)))
    return rc;
  return rerank_and_return(c, reinterpret_cast<const u64_t*>(c->d_out.ptr), k, nullptr, accept_msgs_host, n_accept, n_accept >= 0, max_messages, out_messages,
                           out_scores, out_count);
}

int tavb_search_messages_subset(tavb_ctx* c, const float* query_host, const int64_t* rows_host, int64_t n_subset, int32_t k, float min_score,
                                int32_t max_messages, int64_t* out_messages, float* out_scores, int32_t* out_count) {
  if (int rc = check_message_args(c, k, max_messages)) return rc;
  if (!query_host || !out_messages || !out_scores || !out_count) return fail(TAVB_E_INVALID, "null argument");
  if (n_subset < 0 || n_subset >= 0x7FFFFFFFll) return fail(TAVB_E_INVALID, "bad subset length");
  *out_count = 0;
  if (n_subset == 0 || c->rows == 0) return TAVB_OK;
  if (!rows_host) return fail(TAVB_E_INVALID, "null rows_host");
  DeviceGuard guard(c->device);
  const size_t qbytes = (size_t)c->dim * sizeof(float);
  const size_t rbytes = (size_t)n_subset * sizeof(int32_t);
  const size_t obytes = (size_t)k * sizeof(u64_t);
  const size_t qoff = (rbytes + 255) & ~(size_t)255;
  if (int rc = c->h_stage.reserve(qoff + qbytes)) return rc;
  if (int rc = c->h_out.reserve(obytes)) return rc;
  if (int rc = c->d_queries.reserve(qbytes)) return rc;
  if (int rc = c->d_rows.reserve(rbytes)) return rc;
  if (int rc = c->d_out.reserve(obytes)) return rc;
  int32_t* r32 = reinterpret_cast<int32_t*>(c->h_stage.ptr);
  for (int64_t i = 0; i < n_subset; ++i) {
    const int64_t r = rows_host[i];
    if (r < 0 || r >= c->rows) return fail(TAVB_E_INVALID, "subset row %lld out of range [0, %lld)", (long long)r, (long long)c->rows);
    r32[i] = (int32_t)r;
  }
  memcpy(reinterpret_cast<char*>(c->h_stage.ptr) + qoff, query_host, qbytes);
  TAVB_HIP(hipMemcpyAsync(c->d_rows.ptr, c->h_stage.ptr, rbytes, hipMemcpyHostToDevice, c->stream));
  TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, reinterpret_cast<char*>(c->h_stage.ptr) + qoff, qbytes, hipMemcpyHostToDevice, c->stream));
  if (int rc = search_device_impl(c, reinterpret_cast<const float*>(c->d_queries.ptr), 1, k, &min_score, reinterpret_cast<const int32_t*>(c->d_rows.ptr),
                                  n_subset, 0u, reinterpret_cast<u64_t*>(c->d_out.ptr)))
    return rc;
  return rerank_and_return(c, reinterpret_cast<const u64_t*>(c->d_out.ptr), k, reinterpret_cast<const int32_t*>(c->d_rows.ptr), nullptr, 0, false,
                           max_messages, out_messages, out_scores, out_count);
}

int tavb_search_subset(tavb_ctx* c, const float* query_host, const int64_t* rows_host, int64_t n_subset, int32_t k,
                       float min_score, int64_t* out_positions, float* out_scores, int32_t* out_count) {
  return search_subset_impl(c, query_host, rows_host, n_subset, k, min_score, false, 0.f, 0, out_positions, out_scores,
                            out_count);
}

int tavb_search_subset_after(tavb_ctx* c, const float* query_host, const int64_t* rows_host, int64_t n_subset,
                             int32_t k, float min_score, float after_score, int64_t after_position,
                             int64_t* out_positions, float* out_scores, int32_t* out_count) {
  return search_subset_impl(c, query_host, rows_host, n_subset, k, min_score, true, after_score, after_position,
                            out_positions, out_scores, out_count);
}

int tavb_search_device(tavb_ctx* c, const float* dev_queries, int32_t nq, int32_t k, float min_score,
                       tavb_key* dev_out_keys) {
  if (int rc = check_search_args(c, k)) return rc;
  if (nq < 1) return fail(TAVB_E_INVALID, "nq must be >= 1");
  if (!dev_queries || !dev_out_keys) return fail(TAVB_E_INVALID, "null argument");
  if (c->ordinal_base + c->rows >= 0xFFFFFFFFll)
    return fail(TAVB_E_UNSUPPORTED, "device-resident keys hold 32-bit ordinals: ordinal_base + rows must be < 2^32 - 1");
  DeviceGuard guard(c->device);
  std::vector<float> ms((size_t)nq, min_score);
  return tavb_search_device_dispatch(c, dev_queries, nq, k, ms.data(), (uint32_t)c->ordinal_base,
                                     reinterpret_cast<u64_t*>(dev_out_keys));
}

int tavb_search_subset_device(tavb_ctx* c, const float* dev_query, const int32_t* dev_rows, int64_t n_subset, int32_t k,
                              float min_score, tavb_key* dev_out_keys) {
  if (int rc = check_search_args(c, k)) return rc;
  if (!dev_query || !dev_out_keys) return fail(TAVB_E_INVALID, "null argument");
  if (n_subset < 0 || n_subset >= 0x7FFFFFFFll) return fail(TAVB_E_INVALID, "bad subset length");
  if (n_subset > 0 && !dev_rows) return fail(TAVB_E_INVALID, "null dev_rows");
  DeviceGuard guard(c->device);
  return search_device_impl(c, dev_query, 1, k, &min_score, dev_rows, c->rows == 0 ? 0 : n_subset, 0u,
                            reinterpret_cast<u64_t*>(dev_out_keys));
}

int tavb_search_subset_resident(tavb_ctx* c, const float* query_host, const int32_t* dev_rows, int64_t n_subset, int32_t k, float min_score,
                                int64_t* out_positions, float* out_scores, int32_t* out_count) {
  if (int rc = check_search_args(c, k)) return rc;
  if (n_subset < 0 || n_subset >= 0x7FFFFFFFll) return fail(TAVB_E_INVALID, "bad subset length");
  if (!query_host || !out_positions || !out_scores || !out_count) return fail(TAVB_E_INVALID, "null argument");
  if (n_subset == 0 || c->rows == 0) {
    *out_count = 0;
    return TAVB_OK;
  }
  if (!dev_rows) return fail(TAVB_E_INVALID, "null dev_rows");
  DeviceGuard guard(c->device);
  const size_t qbytes = (size_t)c->dim * sizeof(float);
  if (int rc = c->h_stage.reserve(qbytes)) return rc;
  if (int rc = c->h_out.reserve((size_t)k * sizeof(u64_t))) return rc;
  if (int rc = c->d_queries.reserve(qbytes)) return rc;
  memcpy(c->h_stage.ptr, query_host, qbytes);
  c->last_direct = 0;
  // a small subset (the reference script's 1000 of 10k; the memory provider's scope lists): ONE launch, as tavb_search does it for small corpora --
  // the scan's per-workgroup lists land in pinned host memory and are merged here; a 1536-wide query rides in the kernel arguments
  const int64_t subset_bytes = n_subset * c->dim * (c->dtype == TAVB_F16 ? 2 : 4);
  if (c->small_direct_bytes > 0 && subset_bytes <= c->small_direct_bytes) {
    tavb::ScanGeometry g = c->geom;
    if (g.waves < 1) g.waves = 1;
    if (g.waves > 16) g.waves = 16;
    const int full_blocks = scan_blocks_for(c, n_subset, g.waves, g.unroll);
    g.blocks = std::min(full_blocks, (int)std::max<int64_t>(8, c->small_direct_keys / (int64_t)k));
    if (2 * g.blocks >= full_blocks) {
      const size_t list_keys = (size_t)g.blocks * k;
      if (int rc = c->h_lists.reserve((list_keys + (size_t)k) * sizeof(u64_t))) return rc;
      tavb::ScanParams p{};
      p.corpus = c->corpus;
      p.row_ids = dev_rows;
      p.queries = reinterpret_cast<const float*>(c->d_queries.ptr);
      p.lists = reinterpret_cast<u64_t*>(c->h_lists.ptr);
      p.n_pos = n_subset;
      p.dim = c->dim;
      p.dtype = c->dtype;
      p.nq = 1;
      p.k = k;
      p.index_base = 0u;
      p.key_bound = ~0ull;
      for (int i = 0; i < TAVB_MAX_STREAM_QUERIES; ++i) p.min_score[i] = (i == 0) ? min_score : INFINITY;
      hipError_t e = hipSuccess;
      bool launched = false;
      if (c->inline_query) {
        Timed t(c, TAVB_KERNEL_SCAN);
        launched = tavb::launch_scan_inline_query(p, g, c->stream, query_host, &c->last_tier, &e);
      }
      if (!launched) {
        TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, c->h_stage.ptr, qbytes, hipMemcpyHostToDevice, c->stream));
        Timed t(c, TAVB_KERNEL_SCAN);
        e = tavb::launch_scan(p, g, c->stream, &c->last_tier);
      }
      if (e != hipSuccess) return fail(TAVB_E_HIP, "scan kernel launch failed: %s", hipGetErrorString(e));
      c->last_direct = launched ? 2 : 1;
      TAVB_HIP(hipStreamSynchronize(c->stream));
      tavb_key* merged = reinterpret_cast<tavb_key*>(c->h_lists.ptr) + list_keys;
      if (int rc = tavb_merge_keys_host(reinterpret_cast<const tavb_key*>(c->h_lists.ptr), g.blocks, 1, k, merged)) return rc;
      decode(reinterpret_cast<const u64_t*>(merged), 1, k, 0, out_positions, out_scores, out_count);
      return TAVB_OK;
    }
  }
  TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, c->h_stage.ptr, qbytes, hipMemcpyHostToDevice, c->stream));
  if (int rc = search_device_impl(c, reinterpret_cast<const float*>(c->d_queries.ptr), 1, k, &min_score, dev_rows, n_subset, 0u,
                                  reinterpret_cast<u64_t*>(c->h_out.ptr)))
    return rc;
  TAVB_HIP(hipStreamSynchronize(c->stream));
  decode(reinterpret_cast<const u64_t*>(c->h_out.ptr), 1, k, 0, out_positions, out_scores, out_count);
  return TAVB_OK;
}

int tavb_merge_device(tavb_ctx* c, const tavb_key* dev_lists, int32_t n_lists, int32_t nq, int32_t k,
                      tavb_key* dev_out_keys) {
  if (int rc = check_ctx(c)) return rc;
  if (n_lists < 1 || nq < 1 || k < 1 || k > TAVB_MAX_FUSED_K) return fail(TAVB_E_INVALID, "bad merge shape");
  if (!dev_lists || !dev_out_keys) return fail(TAVB_E_INVALID, "null argument");
  DeviceGuard guard(c->device);
  Timed t(c, TAVB_KERNEL_MERGE);
  hipError_t e = tavb::launch_merge(reinterpret_cast<const u64_t*>(dev_lists), n_lists, nq, k, /*query_major=*/false,
                                    reinterpret_cast<u64_t*>(dev_out_keys), c->stream);
  if (e != hipSuccess) return fail(TAVB_E_HIP, "merge launch failed: %s", hipGetErrorString(e));
  return TAVB_OK;
}

int tavb_decode_keys(const tavb_key* keys_host, int32_t nq, int32_t k, int64_t* out_ordinals, float* out_scores,
                     int32_t* out_counts) {
  if (nq < 0 || k < 1) return fail(TAVB_E_INVALID, "bad shape");
  if (nq == 0) return TAVB_OK;
  if (!keys_host || !out_ordinals || !out_scores || !out_counts) return fail(TAVB_E_INVALID, "null argument");
  // a rank whose local search failed joins the collective with TAVB_KEY_PEER_FAILED in every slot of its lists; the key sorts above every
  // real one, so it leads every merged list on every rank: the answer is missing a shard and must not be used
  for (int q = 0; q < nq; ++q)
    if (keys_host[(size_t)q * k] == TAVB_KEY_PEER_FAILED) {
      for (int i = 0; i < nq; ++i) out_counts[i] = 0;
      return fail(TAVB_E_PEER, "a rank of the collective lookup failed in its local search: the merged lists are missing its shard");
    }
  decode(reinterpret_cast<const u64_t*>(keys_host), nq, k, 0, out_ordinals, out_scores, out_counts);
  return TAVB_OK;
}

int tavb_search_after(tavb_ctx* c, const float* query_host, int32_t k, float min_score, float after_score,
                      int64_t after_ordinal, int64_t* out_ordinals, float* out_scores, int32_t* out_count) {
  if (int rc = check_search_args(c, k)) return rc;
  if (!query_host || !out_ordinals || !out_scores || !out_count) return fail(TAVB_E_INVALID, "null argument");
  if (c->rows == 0) {
    *out_count = 0;
    return TAVB_OK;
  }
  u64_t bound;
  if (int rc = cursor_key(after_score, after_ordinal - c->ordinal_base, c->rows, &bound)) return rc;
  DeviceGuard guard(c->device);
  const size_t qbytes = (size_t)c->dim * sizeof(float);
  const size_t obytes = (size_t)k * sizeof(u64_t);
  if (int rc = c->h_stage.reserve(qbytes)) return rc;
  if (int rc = c->h_out.reserve(obytes)) return rc;
  if (int rc = c->d_queries.reserve(qbytes)) return rc;
  memcpy(c->h_stage.ptr, query_host, qbytes);
  TAVB_HIP(hipMemcpyAsync(c->d_queries.ptr, c->h_stage.ptr, qbytes, hipMemcpyHostToDevice, c->stream));
  int rc = search_device_impl(c, reinterpret_cast<const float*>(c->d_queries.ptr), 1, k, &min_score, nullptr, c->rows,
                              0u, reinterpret_cast<u64_t*>(c->h_out.ptr), bound);
  if (rc) return rc;
  TAVB_HIP(hipStreamSynchronize(c->stream));
  decode(reinterpret_cast<const u64_t*>(c->h_out.ptr), 1, k, c->ordinal_base, out_ordinals, out_scores, out_count);
  return TAVB_OK;
}

// ---- RCCL (resolved at run time: libtavb.so has no link-time dependency on librccl) -------------------------------------------
namespace {
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;  // (optional: only the timeout path needs it)
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_error;

int load_rccl() {
  std::call_once(g_rccl_once, [] {
    // the copy the process already has (torch ships one with the same SONAME) before a fresh one from the ROCm tree
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
      if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    for (const char* n : names)
      if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.handle) {
      const char* e = dlerror();
      g_rccl_error = std::string("cannot load librccl.so.1: ") + (e ? e : "not found");
      return;
    }
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(g_rccl.handle, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(g_rccl.handle, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(g_rccl.handle, "ncclCommDestroy"));
    g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(g_rccl.handle, "ncclCommAbort"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(dlsym(g_rccl.handle, "ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(g_rccl.handle, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString)
      g_rccl_error = "librccl.so.1 lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString";
  });
  if (!g_rccl_error.empty()) return fail(TAVB_E_UNSUPPORTED, "%s", g_rccl_error.c_str());
  return TAVB_OK;
}
static_assert(TAVB_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the rendezvous id is RCCL's");

#define TAVB_RCCL(expr)                                                                                             \
  do {                                                                                                              \
    ncclResult_t r__ = (expr);                                                                                      \
    if (r__ != ncclSuccess) return fail(TAVB_E_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r__), __FILE__, __LINE__); \
  } while (0)

// tavb_synchronize with an exchange in flight and "comm_timeout_ms" set: polls the stream; when the deadline passes (a peer never joined the
// all-gather, or died in it) the communicator is ABORTED -- ncclCommAbort makes the collective's kernel return, so the stream drains -- and the
// context is left without one (tavb_comm_init again to rejoin): TAVB_E_TIMEOUT, never a process stuck in a collective for ever.
int comm_wait_or_abort(tavb_ctx* c) {
  const auto t0 = std::chrono::steady_clock::now();
  const auto deadline = t0 + std::chrono::milliseconds(c->comm_timeout_ms);
  int polls = 0;
  for (;;) {
    const hipError_t e = hipStreamQuery(c->stream);
    if (e == hipSuccess) {
      c->comm_inflight = false;
      return TAVB_OK;
    }
    if (e != hipErrorNotReady) return fail(TAVB_E_HIP, "hipStreamQuery failed: %s", hipGetErrorString(e));
    if (std::chrono::steady_clock::now() >= deadline) break;
    if (++polls > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));  // (the first polls spin: a lookup of a small shard is that short)
  }
  ncclComm_t comm = c->comm;
  c->comm = nullptr;
  c->comm_rank = 0;
  c->comm_world = 1;
  c->comm_inflight = false;
  const long long waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  if (g_rccl.CommAbort) (void)g_rccl.CommAbort(comm);
  (void)hipStreamSynchronize(c->stream);  // (drains once the aborted collective has let go of the stream)
  return fail(TAVB_E_TIMEOUT, "the exchange did not complete within %lld ms (option comm_timeout_ms): a peer never joined the all-gather; "
              "the communicator was aborted -- tavb_comm_init to rejoin", waited);
}
}  // namespace

int tavb_comm_unique_id(void* out_id) {
  if (!out_id) return fail(TAVB_E_INVALID, "null out_id");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  TAVB_RCCL(g_rccl.GetUniqueId(&id));
  memcpy(out_id, id.internal, TAVB_COMM_ID_BYTES);
  return TAVB_OK;
}

int tavb_comm_init(tavb_ctx* c, const void* id_bytes, int32_t rank, int32_t world) {
  if (int rc = check_ctx(c)) return rc;
  if (!id_bytes) return fail(TAVB_E_INVALID, "null id");
  if (world < 1 || rank < 0 || rank >= world) return fail(TAVB_E_INVALID, "rank %d out of range for world %d", rank, world);
  if (c->comm) return fail(TAVB_E_INVALID, "this context already has a communicator (tavb_comm_destroy first)");
  if (int rc = load_rccl()) return rc;
  DeviceGuard guard(c->device);
  ncclUniqueId id;
  memcpy(id.internal, id_bytes, TAVB_COMM_ID_BYTES);
  // the exchange buffers come first: a rank that cannot have them fails HERE, in a collective every rank is still free to fail in, and no
  // exchange of up to comm_reserve_keys keys allocates anything afterwards (8 MiB + world x 8 MiB at the default)
  if (int rc = c->d_xlocal.reserve((size_t)c->comm_reserve_keys * sizeof(u64_t))) return rc;
  if (int rc = c->d_gather.reserve((size_t)c->comm_reserve_keys * sizeof(u64_t) * world)) return rc;
  ncclComm_t comm = nullptr;
  TAVB_RCCL(g_rccl.CommInitRank(&comm, world, id, rank));
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  c->comm_inflight = false;
  return TAVB_OK;
}

int tavb_comm_destroy(tavb_ctx* c) {
  if (!c || !c->comm) return TAVB_OK;
  DeviceGuard guard(c->device);
  (void)hipStreamSynchronize(c->stream);
  ncclComm_t comm = c->comm;
  c->comm = nullptr;
  c->comm_rank = 0;
  c->comm_world = 1;
  if (g_rccl.CommDestroy) TAVB_RCCL(g_rccl.CommDestroy(comm));
  return TAVB_OK;
}

// local [nq, k] lists (device; nullptr = this rank FAILED: it sends TAVB_KEY_PEER_FAILED in every slot) -> ncclAllGather on the context's
// stream -> merge kernel -> out_keys [nq, k].  Nothing here allocates: the lists travel through the buffers tavb_comm_init reserved, in chunks
// of whole queries when they hold more than comm_reserve_keys keys (every rank makes the same call, so every rank cuts the same chunks).
static int exchange_and_merge(tavb_ctx* c, const u64_t* local, int32_t nq, int32_t k, tavb_key* out_keys) {
  const int64_t reserve_keys = (int64_t)(c->d_xlocal.cap / sizeof(u64_t));
  const int qc = (int)std::min<int64_t>(nq, std::max<int64_t>(1, reserve_keys / k));  // queries per chunk
  if ((size_t)qc * k * sizeof(u64_t) * c->comm_world > c->d_gather.cap || (size_t)qc * k * sizeof(u64_t) > c->d_xlocal.cap)
    return fail(TAVB_E_INVALID, "the exchange buffers of this communicator are gone (tavb_comm_init reserves them)");
  u64_t* gathered = reinterpret_cast<u64_t*>(c->d_gather.ptr);
  if (!local) (void)hipMemsetAsync(c->d_xlocal.ptr, 0xFF, (size_t)qc * k * sizeof(u64_t), c->stream);
  if (c->comm_stall_ms > 0) {  // fault injection: one shot
    (void)tavb::launch_stall((int)c->comm_stall_ms, c->stream);
    c->comm_stall_ms = 0;
  }
  c->comm_inflight = true;
  for (int q0 = 0; q0 < nq; q0 += qc) {
    const int qn = std::min(qc, nq - q0);
    const u64_t* src = local ? local + (size_t)q0 * k : reinterpret_cast<const u64_t*>(c->d_xlocal.ptr);
    {
      Timed t(c, TAVB_KERNEL_EXCHANGE);
      TAVB_RCCL(g_rccl.AllGather(src, gathered, (size_t)qn * k, ncclUint64, c->comm, c->stream));
    }
    Timed t(c, TAVB_KERNEL_MERGE);
    hipError_t e = tavb::launch_merge(gathered, c->comm_world, qn, k, /*query_major=*/false, reinterpret_cast<u64_t*>(out_keys) + (size_t)q0 * k, c->stream);
    if (e != hipSuccess) return fail(TAVB_E_HIP, "merge launch failed: %s", hipGetErrorString(e));
  }
  return TAVB_OK;
}

int tavb_search_allgather(tavb_ctx* c, const float* dev_queries, int32_t nq, int32_t k, float min_score, tavb_key* out_keys) {
  // argument errors every rank makes alike (the ranks make the same call) return at once ...
  if (int rc = check_ctx(c)) return rc;
  if (k < 1) return fail(TAVB_E_INVALID, "k must be >= 1 (got %d)", k);
  if (k > TAVB_MAX_FUSED_K)
    return fail(TAVB_E_UNSUPPORTED, "k=%d exceeds the fused-select limit %d; page with tavb_search_after / tavb_search_subset_after", k, TAVB_MAX_FUSED_K);
  if (nq < 1) return fail(TAVB_E_INVALID, "nq must be >= 1");
  if (!dev_queries || !out_keys) return fail(TAVB_E_INVALID, "null argument");
  if (!c->comm || (c->comm_world == 1 && !c->comm_force)) return tavb_search_device(c, dev_queries, nq, k, min_score, out_keys);
  DeviceGuard guard(c->device);
  // ... everything that can fail on ONE rank -- the state of its shard, an allocation, a launch -- still joins the collectives, with
  // TAVB_KEY_PEER_FAILED lists, so that the peers are never left waiting in ncclAllGather for a rank that has returned an error to its caller.
  // Lists of up to comm_reserve_keys keys live in the buffer tavb_comm_init reserved: no allocation between here and the all-gather.
  const size_t list_keys = (size_t)nq * k;
  int rc_local = TAVB_OK;
  u64_t* local = nullptr;
  if (!c->corpus && c->rows != 0) rc_local = fail(TAVB_E_NO_CORPUS, "no corpus set (call tavb_set_corpus first)");
  else if (c->dim <= 0) rc_local = fail(TAVB_E_NO_CORPUS, "no corpus set (call tavb_set_corpus first)");
  else if (c->ordinal_base + c->rows >= 0xFFFFFFFFll)
    rc_local = fail(TAVB_E_UNSUPPORTED, "device-resident keys hold 32-bit ordinals: ordinal_base + rows must be < 2^32 - 1");
  else if (list_keys * sizeof(u64_t) <= c->d_xlocal.cap) local = reinterpret_cast<u64_t*>(c->d_xlocal.ptr);
  else if (c->comm_fail_alloc) rc_local = fail(TAVB_E_NOMEM, "injected failure of the list allocation (option comm_fail_alloc)");
  else if ((rc_local = c->d_local.reserve(list_keys * sizeof(u64_t))) == TAVB_OK) local = reinterpret_cast<u64_t*>(c->d_local.ptr);
  std::vector<float> ms((size_t)nq, min_score);
  if (rc_local != TAVB_OK) {
  } else if (c->comm_fail_rank >= 0 && c->comm_fail_rank == c->comm_rank) {  // fault injection (option "comm_fail_rank"): what a failed launch / allocation inside the local search looks like
    rc_local = fail(TAVB_E_HIP, "injected failure of the local search on rank %d (option comm_fail_rank)", c->comm_rank);
  } else if (c->rows == 0) {  // an empty shard still takes part in the collectives
    const hipError_t e = hipMemsetAsync(local, 0, list_keys * sizeof(u64_t), c->stream);
    if (e != hipSuccess) rc_local = fail(TAVB_E_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e));
  } else {
    rc_local = tavb_search_device_dispatch(c, dev_queries, nq, k, ms.data(), (uint32_t)c->ordinal_base, local);
  }
  // a failed rank's lists = TAVB_KEY_PEER_FAILED (all bits set) in every slot: it sorts above every real key, so it leads every merged list on
  // EVERY rank -- the peers' answers would silently miss this shard otherwise; tavb_decode_keys turns it into TAVB_E_PEER
  const std::string local_error = rc_local != TAVB_OK ? g_last_error : std::string();
  const int rc_x = exchange_and_merge(c, rc_local == TAVB_OK ? local : nullptr, nq, k, out_keys);
  if (rc_local != TAVB_OK) {
    g_last_error = local_error;
    return rc_local;
  }
  return rc_x;
}

int tavb_allgather_merge(tavb_ctx* c, const tavb_key* dev_local_keys, int32_t nq, int32_t k, tavb_key* out_keys) {
  if (int rc = check_ctx(c)) return rc;
  if (nq < 1 || k < 1 || k > TAVB_MAX_FUSED_K) return fail(TAVB_E_INVALID, "bad list shape");
  if (!dev_local_keys || !out_keys) return fail(TAVB_E_INVALID, "null argument");
  DeviceGuard guard(c->device);
  if (!c->comm || (c->comm_world == 1 && !c->comm_force)) {
    if (reinterpret_cast<const void*>(dev_local_keys) != reinterpret_cast<const void*>(out_keys))
      TAVB_HIP(hipMemcpyAsync(out_keys, dev_local_keys, (size_t)nq * k * sizeof(u64_t), hipMemcpyDefault, c->stream));
    return TAVB_OK;
  }
  return exchange_and_merge(c, reinterpret_cast<const u64_t*>(dev_local_keys), nq, k, out_keys);
}

int tavb_remap_key_positions(tavb_ctx* c, tavb_key* dev_keys, int64_t count, const int32_t* dev_map, int64_t map_len) {
  if (int rc = check_ctx(c)) return rc;
  if (count < 0 || map_len < 0) return fail(TAVB_E_INVALID, "bad shape");
  if (count == 0) return TAVB_OK;
  if (!dev_keys || (map_len > 0 && !dev_map)) return fail(TAVB_E_INVALID, "null argument");
  DeviceGuard guard(c->device);
  hipError_t e = tavb::launch_remap_positions(reinterpret_cast<u64_t*>(dev_keys), count, dev_map, map_len, c->stream);
  if (e != hipSuccess) return fail(TAVB_E_HIP, "remap launch failed: %s", hipGetErrorString(e));
  return TAVB_OK;
}

int tavb_profile_enable(tavb_ctx* c, int32_t on) {
  if (int rc = check_ctx(c)) return rc;
  DeviceGuard guard(c->device);
  if (!on) {
    if (int rc = drain_timings(c)) return rc;
  }
  c->profiling = on != 0;
  return TAVB_OK;
}

int tavb_profile_reset(tavb_ctx* c) {
  if (int rc = check_ctx(c)) return rc;
  DeviceGuard guard(c->device);
  if (int rc = drain_timings(c)) return rc;
  for (int i = 0; i < TAVB_KERNEL_COUNT; ++i) {
    c->total_ms[i] = 0;
    c->launches[i] = 0;
  }
  return TAVB_OK;
}

int tavb_profile_read(tavb_ctx* c, int32_t kernel_id, double* out_total_ms, int64_t* out_launches) {
  if (int rc = check_ctx(c)) return rc;
  if (kernel_id < 0 || kernel_id >= TAVB_KERNEL_COUNT) return fail(TAVB_E_INVALID, "bad kernel id");
  DeviceGuard guard(c->device);
  if (int rc = drain_timings(c)) return rc;
  if (out_total_ms) *out_total_ms = c->total_ms[kernel_id];
  if (out_launches) *out_launches = c->launches[kernel_id];
  return TAVB_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Tile kernels (tavb_mfma.hip) behind the threshold ladder.
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct TileRun {
  bool skinny;            // 32/64-query tile (fp32 or split-fp16 queries) instead of the 256-query fp16 tile
  bool q32;               // skinny tile on an fp32 corpus
  int qt;                 // queries per tile
  int nq, nq_pad, k;
  uint32_t index_base;
  float kernel_min_score; // uniform threshold applied inside the kernel
  const float* floor;     // optional device [nq_pad]: per-query exclusive admission thresholds valid from the first row on
  const void* queries;    // operand in the kernel's layout
  const void* corpus;     // corpus operand (nullptr: the context's corpus; the fp16 shadow of an fp32 corpus for the filter pass)
  int dim;                // halves / floats per row of that operand and of `queries` (0: the context's dim; the zero-padded width of a shadow whose corpus is not a multiple of 64 wide)
  const int* active;      // optional device-side live-query count (fixed-shape launch over a work list)
  int active_min, active_max;  // ... served only when active_min < *active <= active_max (0 = no upper bound): two fallbacks share one list
  bool bdirect;           // 256-query tile: `queries` are in fragment-major order (straight from L2 into registers)
  int64_t split_plane;    // 128/256-query tile: > 0 = exact form, `queries` = [2][nq_pad][dim] fp16 planes this many bytes apart (final scores, no band)
  bool ladder;            // scan in phases of growing size (else one phase)
  // 128/256-query tile only: band selection (tavb_mfma.hip::select_band_kernel).  d_out then receives [nq, kBandMax] unsorted keys,
  const float* band;      // device [nq_pad]: width of the band below the k-th best
  int* band_cnt;          // device [nq]: out, keys per query in d_out
  unsigned* lost;         // device [nq_pad]: scratch (zeroed by the caller), score level below which a query lost band rows
  int* verdict;           // device [nq]: out, 1 where the band handed over is not provably complete
  // ... early verdict on the whole batch: after the phase before the last, queries whose band over the rows seen so far extrapolates to more than
  // the band buffer are counted in *doomed (zeroed by the caller); with more than doomed_max of them the last phase's launches return at once
  int* doomed;
  int doomed_max;
  // a work-list run of the 128/256-query tile (r.active: the SPLIT fallback) ends with its candidates rescored by the streaming kernels' arithmetic
  // (tavb_rescore.hip, slot mode): the callers' fp32 queries [*, dim], their thresholds [*] (device), indexed by scatter[slot]
  const float* rs_queries;
  const float* rs_min_scores;
};

// Width (in score) of the band the exact fallbacks keep below their k-th best before the candidates are scored again with the streaming
// kernels' arithmetic: the two arithmetics (fp32 accumulation inside the matrix pipe vs the streaming kernels' per-lane fma chains) differ by
// a few 1e-7 on unit vectors of 1536 dimensions (measured: <= 4e-7 against float64), the hi + lo split of a query carries it to 2^-22.
// Ten times that: a row the streaming arithmetic ranks in the top k is inside the band unless the two disagree by more than 4e-6.
constexpr float kExactBand = 4e-6f;

// Phase boundaries of the threshold ladder (see run_tile_ladder): phase i scans rows [b[i], b[i+1]).  `sample_opt` / `growth` = the options
// mfma_sample_rows (0 = auto, -1 = one phase) / mfma_ladder.  A pure function of its arguments: tavb_plan_ladder() hands it to callers that
// want to know how many tile launches a lookup makes (tests/test_bench_contract.py checks the committed PMC pass against it).
std::vector<int64_t> ladder_bounds(int64_t rows, int splits, int nq_pad, bool skinny, bool ladder, int64_t sample_opt, int64_t growth) {
  std::vector<int64_t> bounds;
  bounds.push_back(0);
  // first phase: `mfma_sample_rows`, or (0 = auto) part of ONE tile per workgroup of the 128/256-query kernel -- nothing compacts while
  // everything is still being admitted, and every unfiltered row of this phase is a key the select kernel has to stream (one workgroup per
  // QUERY).  Round 2 used two tiles per workgroup (40960 rows), round 3 one (20480: 4 % faster on a 1.25M-row shard, the same on 10M rows;
  // profiles/r03_shard_ladder.md).  Round 4: with one LDS atomic per admitted row (tavb_mfma.hip) the all-admitted first phase is best kept
  // to 32 ranges' worth, 10240 rows -- 1 % faster on the shard, the same on 10M rows, half the keys for the select kernel
  // (profiles/r04_cfg3_kernel.md).  (The 32/64-query tile keeps round 2's 40960 rows.)
  // One or two query tiles (up to 256 queries: 128 .. 256 row ranges) keep 64 ranges' worth: 1 - 2 % faster there (profiles/r04_raw/mid_batch.txt).
  const int64_t auto_sample = skinny ? (int64_t)std::min(splits, 64) * 320 * 2 : (int64_t)std::min(splits, nq_pad >= 512 ? 32 : 64) * 320;
  const int64_t sample = sample_opt > 0 ? (sample_opt + 255) / 256 * 256 : (sample_opt == 0 ? auto_sample : 0);
  // 32/64-query tile on corpora of a few hundred thousand to ~2M rows: the default ladder's first phases are smaller than one tile per
  // workgroup (40960 rows = 160 tiles for 512 resident workgroups) and each costs a launch + ~one tile time whatever its size; ONE seeding
  // phase of exactly one tile per workgroup, then the rest, is faster (1M x 1536 fp32, 32 queries: 1.18 -> 1.07 ms of kernels per batch,
  // profiles/r03_mid_batch.md); a single un-seeded phase is slower still (1.23 ms: every workgroup pays the cold start)
  const int64_t one_tile_each = (int64_t)splits * 256;
  if (ladder && skinny && sample_opt == 0 && rows >= 4 * one_tile_each && rows < 2048000) {
    bounds.push_back(one_tile_each);
  } else if (ladder && skinny && sample_opt == 0 && rows >= 2048000 && rows >= 32 * one_tile_each) {
    // ... and on bigger corpora THREE phases: one tile per workgroup, twelve times that, the rest (10M rows: 65536 / 851968 / 9.08M).  The
    // 32-query tile is HBM-bound and admits little (k ln(n / seen) rows per query): what its early phases cost is their launches and tails,
    // 0.62 ms for three phases over 1.02M rows against 0.51 ms for two over 0.92M (profiles/r05_mid_batch.md; one phase fewer than the
    // generic ladder below, +1.5 % on cfg3_b32)
    bounds.push_back(one_tile_each);
    if (growth > 0) bounds.push_back(13 * one_tile_each);
  } else if (ladder && sample > 0 && rows >= 8 * sample) {
    int64_t done = sample;
    bounds.push_back(done);
    while (growth > 0 && done * (growth + 1) * 2 <= rows && bounds.size() < 8) {
      done += done * growth;
      bounds.push_back(done);
    }
  } else if (ladder && !skinny && sample_opt == 0 && growth > 0 && rows >= 8 * 320 && (rows >= 12 * 2560 || rows > (int64_t)splits * 640)) {
    // the wide tile on a SMALL corpus (below eight first phases' worth: 82k rows at 1024 queries, 164k at up to 128): until the end of round 6
    // ONE un-seeded phase -- every row admitted; at 1024 queries the candidate buffers compact every other tile from the third tile of a row
    // range on (50k rows: 0.99 ms of tile kernel, twice what 100k rows took), at up to 128 queries the select kernel streams every row of the
    // corpus per query (150k rows: 0.29 ms of selection next to 0.16 ms of tile kernel).  Two phases instead: an eighth of the rows in whole
    // tiles, then the rest behind its thresholds: 1024 queries over 50k rows 1.13 -> 0.47 ms, 128 over 150k rows 0.52 -> 0.28 ms; from 30720
    // rows up (20k rows: one phase is as fast; profiles/r06_raw/small_wide.txt) -- and below that whenever a workgroup would walk more than
    // two tiles un-seeded (many query tiles leave few row ranges: 2048 queries over 20000 rows are 16 ranges of four tiles, 0.68 ms in one
    // phase against 0.44).
    bounds.push_back((rows / 8 / 320) * 320);
  }
  bounds.push_back(rows);
  return bounds;
}

// Threshold ladder.  The corpus is scanned in phases of growing size -- the first `mfma_sample_rows` rows, then
// `mfma_ladder` times everything scanned so far, ..., then the rest -- every row exactly once.  After each phase the
// exact top-k so far is merged; its k-th best score is a valid admission threshold for every later row (the k-th best
// of a subset never exceeds the k-th best of the whole corpus), so each phase starts selective instead of admitting
// whatever comes first and compacting, and the running top-k rides along as one more list of the next phase's merge.
// Expected admissions per query drop from k * rows / sample (one seeding phase) to ~k * ladder per phase.  Results do
// not depend on the phase boundaries.  Output: sorted key lists [nq, k] at `d_out` (or, with `scatter`, rows
// scatter[slot] of it for the slots below *active; a work-list run of the wide tile rescoring its band first: TileRun::rs_queries).
int run_tile_ladder(tavb_ctx* c, const TileRun& r, u64_t* d_out, const int* scatter, bool scatter_identity = false) {
  const int dim = r.dim > 0 ? r.dim : c->dim;  // of the tile's operands (the candidates' ordinals are the corpus' own either way)
  auto pick_splits = [&](int64_t rows) {
    return r.skinny ? tavb::skinny_pick_splits(rows, r.nq_pad, r.qt, c->n_cu, dim, r.q32, (int)c->mfma_sched) : tavb::mfma_pick_splits(rows, r.nq_pad, r.qt, c->n_cu);
  };
  auto launch = [&](const tavb::MfmaParams& q) { return r.skinny ? tavb::launch_skinny_scan(q, c->stream) : tavb::launch_mfma_scan(q, c->stream); };
  const int nq = r.nq, k = r.k;
  const int splits = c->mfma_splits > 0 ? (int)c->mfma_splits : pick_splits(c->rows);
  const bool wide = !r.skinny;  // the 256-query tile leaves unsorted buffers + counts, one select kernel picks the best k over them
  if (!wide)
    if (int rc = c->d_lists.reserve((size_t)nq * (splits + 1) * k * sizeof(u64_t))) return rc;  // + the carried-over top-k
  if (int rc = c->d_cand.reserve(tavb::mfma_workspace_bytes(splits, r.nq_pad, wide))) return rc;
  if (wide)
    if (int rc = c->d_counts.reserve((size_t)splits * r.nq_pad * sizeof(int))) return rc;
  tavb::MfmaParams p{};
  p.corpus = r.corpus ? r.corpus : c->corpus;
  p.queries = r.queries;
  p.lists = reinterpret_cast<u64_t*>(c->d_lists.ptr);
  p.workspace = reinterpret_cast<u64_t*>(c->d_cand.ptr);
  p.counts = reinterpret_cast<int*>(c->d_counts.ptr);
  p.rows = c->rows;
  p.dim = dim;
  p.nq = nq;
  p.nq_padded = r.nq_pad;
  p.k = k;
  p.index_base = r.index_base;
  p.min_score = r.kernel_min_score;
  p.n_splits = splits;
  p.ablate = (int)c->mfma_ablate;
  p.sched = (int)c->mfma_sched;
  p.f32 = r.q32 ? 1 : 0;
  p.skinny_tile = r.skinny ? r.qt : 0;
  p.wide_tile = r.skinny ? 0 : r.qt;
  p.active = r.active;
  p.active_min = r.active_min;
  p.active_max = r.active_max;
  p.split_plane = r.split_plane;
  p.bdirect = r.bdirect ? 1 : 0;
  const std::vector<int64_t> bounds = ladder_bounds(c->rows, splits, r.nq_pad, r.skinny, r.ladder, c->mfma_sample_rows, c->mfma_ladder);  // phase i scans rows [bounds[i], bounds[i+1])
  const int n_phases = (int)bounds.size() - 1;
  const int kc = wide ? (int)c->band_max : k;  // keys per query of the running selection between phases
  if (wide)  // (every phase's selection leaves its cut here -- the last one's seeds the exact fallbacks' admission thresholds, search_wide_exact)
    if (int rc = c->d_thr.reserve((size_t)r.nq_pad * sizeof(float))) return rc;
  if (n_phases > 1 || (wide && r.active)) {
    if (int rc = c->d_thr.reserve((size_t)r.nq_pad * sizeof(float))) return rc;
    if (int rc = c->d_sample_keys.reserve((size_t)2 * nq * kc * sizeof(u64_t) + (size_t)2 * nq * sizeof(int))) return rc;  // running selection: two copies (ping-pong) + counts
  }
  p.band = r.band;
  p.lost = r.lost;
  const float* floor = r.floor;  // per-query thresholds valid for every row
  const size_t row_bytes = (size_t)dim * (r.q32 ? 4 : 2);  // of the corpus operand
  for (int ph = 0; ph < n_phases; ++ph) {
    const bool last = (ph == n_phases - 1);
    tavb::MfmaParams pp = p;
    pp.corpus = reinterpret_cast<const char*>(p.corpus) + (size_t)bounds[ph] * row_bytes;
    pp.rows = bounds[ph + 1] - bounds[ph];
    pp.index_base = r.index_base + (uint32_t)bounds[ph];
    pp.n_splits = pick_splits(pp.rows);
    if (c->mfma_splits > 0 || pp.n_splits > splits) pp.n_splits = splits;  // lists / candidate buffers are sized for `splits`
    const int carried = ph > 0 ? 1 : 0;  // the running top-k of the earlier phases occupies one more list slot
    pp.list_stride = pp.n_splits + carried;
    pp.thr_in = ph > 0 ? reinterpret_cast<const float*>(c->d_thr.ptr) : floor;
    // the early verdict (r.doomed): the select launch of the phase before the last counts, the last phase's launches gate themselves on the count
    const bool doom_count = wide && r.doomed && n_phases >= 2 && ph == n_phases - 2;
    const bool doom_gate = wide && r.doomed && n_phases >= 2 && last;
    // a band of c keys over `seen` of `rows` rows grows to about c * rows / seen when its rows are spread evenly (a cluster of near-duplicates around
    // the k-th best; on ordinary data the band is k plus a key or two whatever the row count): counted when that is 1.25 x the band buffer,
    // and only with at least 16 keys beyond k in hand
    const int doom_limit = std::max(k + 15, (int)std::min<int64_t>(1 << 30, (int64_t)(1.25 * kc * (double)bounds[ph + 1] / (double)c->rows)));
    if (doom_gate) {
      pp.gate = r.doomed;
      pp.gate_max = r.doomed_max;
    }
    u64_t* const running = reinterpret_cast<u64_t*>(c->d_sample_keys.ptr);  // [2][nq][kc] (+ [2][nq] counts); not allocated for a single phase
    const u64_t* const run_in = running ? running + (size_t)((ph + 1) & 1) * nq * kc : nullptr;  // what phase ph - 1 left
    u64_t* const run_out = running ? running + (size_t)(ph & 1) * nq * kc : nullptr;
    int* const run_cnt = running ? reinterpret_cast<int*>(running + (size_t)2 * nq * kc) : nullptr;
    const int* const cnt_in = run_cnt ? run_cnt + (size_t)((ph + 1) & 1) * nq : nullptr;
    int* const cnt_out = run_cnt ? run_cnt + (size_t)(ph & 1) * nq : nullptr;
    if (carried && !wide) {
      TAVB_HIP(hipMemcpy2DAsync(pp.lists + (size_t)pp.n_splits * k, (size_t)pp.list_stride * k * sizeof(u64_t), run_in,
                                (size_t)k * sizeof(u64_t), (size_t)k * sizeof(u64_t), (size_t)nq, hipMemcpyDeviceToDevice, c->stream));
    }
    {
      Timed t(c, r.active ? TAVB_KERNEL_RESCORE : !last ? TAVB_KERNEL_MFMA_SAMPLE : (r.skinny ? TAVB_KERNEL_SKINNY : TAVB_KERNEL_MFMA));
      hipError_t e = launch(pp);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "mfma scan launch failed (phase %d): %s", ph, hipGetErrorString(e));
    }
    if (wide) {
      Timed t(c, r.active ? TAVB_KERNEL_RESCORE : TAVB_KERNEL_MERGE);
      // (thresholds of the padding queries are never read: the tiles give every query past the live ones +inf themselves.  Until round 6 a
      //  memset per phase filled them with NaNs -- one launch per phase for nothing.)
      float* d_thr = reinterpret_cast<float*>(c->d_thr.ptr);
      // a work-list run (r.active: the SPLIT fallback) ends in its own band buffer; the strict best k of it is scattered to the callers' rows below
      u64_t* const last_out = r.active ? run_out : d_out;
      int* const last_cnt = r.active ? cnt_out : r.band_cnt;
      hipError_t e = tavb::launch_select_band(pp.workspace, pp.counts, pp.n_splits, nq, r.nq_pad, k, kc, carried ? run_in : nullptr, carried ? cnt_in : nullptr,
                                              floor, r.band, last ? last_out : run_out, last ? last_cnt : cnt_out, (last && r.active) ? nullptr : d_thr, r.lost,
                                              last ? r.verdict : nullptr, c->stream, r.active, r.active_min, r.active_max > 0 ? r.active_max : 0x7fffffff,
                                              doom_gate ? r.doomed : nullptr, r.doomed_max, doom_count ? r.doomed : nullptr, doom_limit);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "select launch failed: %s", hipGetErrorString(e));
      if (last && r.active) {  // the band of every live slot, scored again the streaming kernels' way: its best k go to the caller's row scatter[slot]
        e = tavb::launch_rescore_slots(c->corpus, /*f32_rows=*/false, c->dim, r.index_base, r.rs_queries, last_out, kc, last_cnt, r.rs_min_scores, nq, k, d_out,
                                       scatter, r.active, r.active_min, r.active_max > 0 ? r.active_max : 0x7fffffff, c->stream);
        if (e != hipSuccess) return fail(TAVB_E_HIP, "fallback rescore launch failed: %s", hipGetErrorString(e));
      }
    } else if (last) {
      Timed t(c, r.active ? TAVB_KERNEL_RESCORE : TAVB_KERNEL_MERGE);
      // (scatter_identity: a work-list run whose lists stay slot-indexed -- merged only for the live slots)
      hipError_t e = (scatter || scatter_identity) ? tavb::launch_merge_scatter(pp.lists, pp.list_stride, nq, k, r.active, scatter, d_out, c->stream)
                                                    : tavb::launch_merge(pp.lists, pp.list_stride, nq, k, /*query_major=*/true, d_out, c->stream);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "merge launch failed: %s", hipGetErrorString(e));
    } else {
      hipError_t e = tavb::launch_merge(pp.lists, pp.list_stride, nq, k, /*query_major=*/true, run_out, c->stream);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "phase merge launch failed: %s", hipGetErrorString(e));
      e = tavb::launch_sample_thresholds(run_out, nq, k, r.floor, reinterpret_cast<float*>(c->d_thr.ptr), c->stream);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "threshold launch failed: %s", hipGetErrorString(e));
    }
  }
  return TAVB_OK;
}

// The lowest threshold of a batch (NaN thresholds aside; NaN when every one is NaN): the ONE threshold a tile launch takes -- the per-query
// thresholds ride in the `floor` array.
float lowest_min_score(const float* min_scores, int nq) {
  float lo = NAN;
  for (int i = 0; i < nq; ++i)
    if (min_scores[i] == min_scores[i]) lo = (lo != lo || min_scores[i] < lo) ? min_scores[i] : lo;
  return lo;
}

// The exclusive admission floor that goes with a threshold: `score > floor` <=> `score >= min_score` (+inf for NaN / > 1: nothing passes).
float floor_of_min_score(float ms) {
  if (ms != ms || ms > 1.0f) return INFINITY;
  if (!(ms > 0.0f)) return -INFINITY;
  uint32_t bits;
  memcpy(&bits, &ms, sizeof bits);
  --bits;
  float f;
  memcpy(&f, &bits, sizeof f);
  return f;
}

// min_scores (host, [nq]) -> c->d_minscores: [nq_pad] the thresholds themselves (padding: +inf), then [nq_pad] the exclusive admission floors
// that go with them (+inf for NaN / > 1 / padding).  A mixed batch is copied from pageable memory (staged by the runtime before the call
// returns).  A uniform one -- every caller of the reference -- needs no host buffer in flight (the device-resident forms stay asynchronous):
// *uniform_out = true, NOTHING is written here, and the caller's prologue kernel fills both arrays from the one value (query_prepare_kernel).
int upload_min_scores(tavb_ctx* c, const float* min_scores, int nq, int nq_pad, float** d_ms_out, float** d_floor_out, bool* uniform_out) {
  if (int rc = c->d_minscores.reserve((size_t)2 * nq_pad * sizeof(float))) return rc;
  float* d_ms = reinterpret_cast<float*>(c->d_minscores.ptr);
  float* d_floor = d_ms + nq_pad;
  bool uniform = true;
  for (int i = 1; i < nq; ++i) uniform = uniform && (memcmp(&min_scores[i], &min_scores[0], sizeof(float)) == 0);
  if (!uniform) {
    std::vector<float> h((size_t)2 * nq_pad, INFINITY);
    for (int i = 0; i < nq; ++i) {
      h[i] = min_scores[i];
      h[(size_t)nq_pad + i] = floor_of_min_score(min_scores[i]);
    }
    TAVB_HIP(hipMemcpyAsync(d_ms, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    TAVB_HIP(hipStreamSynchronize(c->stream));  // (pageable source: be sure the runtime is done with `h` before it goes out of scope)
  }
  *d_ms_out = d_ms;
  *d_floor_out = d_floor;
  *uniform_out = uniform;
  return TAVB_OK;
}

// The 128/256-query fp16 tile as an exact filter + fp32-query rescoring of its candidates (tavb_rescore.hip).  fp32 corpora:
// the filter reads the fp16 shadow (d_shadow, reserved by the caller), the rescoring and the fallback tile the fp32 rows.
// `small` (fp32 corpora only): the filter is the 32/64-query tile over the shadow with the EXACT queries (split fp16 planes), for batches
// below the wide tile's range -- half the bytes of an fp32 pass.
// min_scores: host [nq], one threshold per query (the reference takes `min_score` per call, vectorbase.py:163-173: a batch of Q calls has Q of them).
int search_wide_exact(tavb_ctx* c, const float* d_q, int nq, int k, const float* min_scores, uint32_t index_base, u64_t* d_out, bool small = false) {
  // candidates per query handed to the rescoring: the wide tile selects a BAND (every row within 2 delta of the approximate k-th best: as many
  // as the data makes it, up to kBandMax), the 32/64-query tile (`small`) the best 64 by approximate score
  const int KC = small ? 64 : (int)c->band_max;
  const bool f32c = (c->dtype == TAVB_F32);
  // a corpus whose width is not a multiple of 64 (the tile's K step is a whole 128-byte line): the filter -- and, on fp16 corpora, the exact
  // fallbacks -- read a zero-padded fp16 copy of the rows (d_shadow, `fdim` halves per row; for fp32 corpora the shadow they have anyway) and a
  // zero-padded copy of the queries: zeros add nothing to a dot product or a norm, the delta bound is unchanged.  The candidates are rescored
  // with the corpus' own rows and the callers' own queries.
  const bool padded = !small && (c->dim % 64 != 0);
  const int fdim = padded ? ((c->dim + 63) / 64) * 64 : c->dim;
  const bool shadow_ops = f32c || padded;  // the filter's corpus operand is d_shadow
  const bool big_k = k > 64;  // beyond what the 64-query exact tile ranks: every flagged query goes to the wide split-plane form (fp16 corpora only: the caller checked)
  const int qt = small ? tavb::skinny_query_tile(nq) : (c->mfma_tile > 0 ? (int)c->mfma_tile : tavb::mfma_query_tile_for(nq, c->rows, c->n_cu));
  const int nq_pad = ((nq + qt - 1) / qt) * qt;
  const bool bdirect = !small && qt == 256 && c->mfma_bdirect && c->mfma_ablate == 0;
  // Work list of queries that need an exact pass (a band that did not fit).  Few of them (<= 64): ONE pass of the 64-query exact tile.  Many: the
  // 256-query tile in its SPLIT form (fp32 queries as two fp16 planes, the K loop run once per plane: twice the MFMAs of a filter pass, exact) --
  // 16 passes of the 64-query tile per 1024 flagged queries otherwise (round 2-3: "stated, not solved").  Both are fixed-shape
  // launches over the same device-side list and return at once when it is empty or is the other one's share.  Either one hands its best rows
  // (and a small band below them) to the rescoring kernel in slot mode: a query served by a fallback gets the streaming kernels' float32 scores.
  const bool wide_fallback = !small && !f32c && c->wide_fallback && (nq >= 256 || big_k);
  // k > 64 on an FP32 corpus (end of round 6): the filter, the band and the rescoring serve any k up to TAVB_MAX_FUSED_K, but no exact tile ranks
  // more than 64 fp32 rows per query.  A flagged query -- more than band_max near-duplicates around its k-th best: rare -- is therefore re-run on
  // the streaming kernels, which takes the one host round trip of this file (the work list is read back; nothing flagged: nothing more to do).
  // Until then such batches took the streaming kernels four queries per corpus pass: 128 queries over 2M x 1536 fp32 rows, k = 65: 66 ms against 1.3.
  const bool f32_big_k = !small && f32c && big_k;
  if (big_k && !wide_fallback && !f32_big_k) return fail(TAVB_E_UNSUPPORTED, "k > 64 on the batched tile of an fp16 corpus needs the wide_fallback option");
  const int cap = wide_fallback ? ((nq + 255) / 256) * 256 : ((nq + 63) / 64) * 64;  // slots of the work list
  const size_t q16_bytes = (size_t)nq_pad * fdim * 2 * (small ? 2 : 1);  // small: high and low plane
  if (int rc = c->d_queries_f16.reserve(q16_bytes)) return rc;
  if (int rc = c->d_delta.reserve((size_t)nq_pad * 6 * sizeof(float))) return rc;  // delta, the relaxed thresholds, the band widths; band counts, lost levels, verdicts
  if (int rc = c->d_approx.reserve((size_t)nq * KC * sizeof(u64_t))) return rc;
  if (int rc = c->d_flag.reserve((size_t)(cap + 64) * sizeof(int))) return rc;
  if (int rc = c->d_fb_queries.reserve((size_t)2 * cap * fdim * 2 + (size_t)2 * cap * sizeof(float))) return rc;  // + per-slot thresholds, per-slot band widths
  const float* fq = d_q;  // the queries as the filter and the padded fallbacks read them
  if (padded) {
    if (int rc = c->d_queries_pad.reserve((size_t)nq * fdim * sizeof(float))) return rc;
    TAVB_HIP(hipMemsetAsync(c->d_queries_pad.ptr, 0, (size_t)nq * fdim * sizeof(float), c->stream));
    TAVB_HIP(hipMemcpy2DAsync(c->d_queries_pad.ptr, (size_t)fdim * sizeof(float), d_q, (size_t)c->dim * sizeof(float), (size_t)c->dim * sizeof(float), (size_t)nq,
                              hipMemcpyDeviceToDevice, c->stream));
    fq = reinterpret_cast<const float*>(c->d_queries_pad.ptr);
  }
  if (!big_k || f32_big_k)
    if (int rc = c->d_fb_cand.reserve((size_t)cap * (f32_big_k ? k : 64) * sizeof(u64_t))) return rc;
  if (f32_big_k)
    if (int rc = c->h_flag.reserve((size_t)(64 + cap) * sizeof(int))) return rc;
  if (int rc = c->d_norm.reserve(256)) return rc;
  float *d_ms = nullptr, *d_ms_floor = nullptr;
  bool ms_uniform = false;
  if (int rc = upload_min_scores(c, min_scores, nq, nq_pad, &d_ms, &d_ms_floor, &ms_uniform)) return rc;
  const float ms_lo = lowest_min_score(min_scores, nq);
  float* d_norm = reinterpret_cast<float*>(c->d_norm.ptr);
  float* d_delta = reinterpret_cast<float*>(c->d_delta.ptr);
  float* d_floor = d_delta + nq_pad;
  float* d_band = d_floor + nq_pad;
  int* d_band_cnt = reinterpret_cast<int*>(d_band + nq_pad);
  unsigned* d_lost = reinterpret_cast<unsigned*>(d_band_cnt + nq_pad);
  int* d_verdict = d_band_cnt + 2 * nq_pad;
  int* d_nflag = reinterpret_cast<int*>(c->d_flag.ptr);
  int* d_flagged = d_nflag + 64;
  {
    Timed t(c, TAVB_KERNEL_RESCORE);
    if (c->norm_rows > c->rows || c->norm_rows == 0) {  // first use on this corpus (or it shrank: the old maxima are still upper bounds, but start over)
      TAVB_HIP(hipMemsetAsync(d_norm, 0, 2 * sizeof(float), c->stream));
      c->norm_rows = 0;
    }
    if (c->norm_rows < c->rows) {  // rows appended since: extend the maxima (and the shadow)
      hipError_t e;
      const int sdim = ((c->dim + 63) / 64) * 64;  // halves per shadow row
      char* shadow_new = c->d_shadow.ptr ? reinterpret_cast<char*>(c->d_shadow.ptr) + (size_t)c->norm_rows * sdim * 2 : nullptr;
      if (f32c) {
        e = tavb::launch_shadow_convert(reinterpret_cast<const float*>(c->corpus) + (size_t)c->norm_rows * c->dim, c->rows - c->norm_rows, c->dim, shadow_new, sdim,
                                        d_norm, c->stream);
      } else {
        if (padded) {  // fp16 rows of an odd width: the same values, rows zero-padded to whole K steps
          const size_t n_new = (size_t)(c->rows - c->norm_rows);
          TAVB_HIP(hipMemsetAsync(shadow_new, 0, n_new * sdim * 2, c->stream));
          TAVB_HIP(hipMemcpy2DAsync(shadow_new, (size_t)sdim * 2, reinterpret_cast<const char*>(c->corpus) + (size_t)c->norm_rows * c->dim * 2, (size_t)c->dim * 2,
                                    (size_t)c->dim * 2, n_new, hipMemcpyDeviceToDevice, c->stream));
        }
        // (the norm kernel loads 16 bytes at a time: widths that are no multiple of 8 are read from the padded copy -- zeros add nothing to a norm)
        const bool norm_from_pad = padded && (c->dim % 8 != 0);
        e = norm_from_pad ? tavb::launch_corpus_max_norm(shadow_new, c->rows - c->norm_rows, sdim, d_norm, c->stream)
                          : tavb::launch_corpus_max_norm(reinterpret_cast<const char*>(c->corpus) + (size_t)c->norm_rows * c->dim * 2, c->rows - c->norm_rows,
                                                         c->dim, d_norm, c->stream);
      }
      if (e != hipSuccess) return fail(TAVB_E_HIP, "corpus norm / shadow launch failed: %s", hipGetErrorString(e));
      c->norm_rows = c->rows;
    }
    // ONE launch: the filter's query operand (padding slots zero), delta / relaxed thresholds / band widths, the selection's counters zeroed,
    // the work list's header zeroed, a uniform batch's thresholds filled in (round 6: a fill kernel, three memsets and this kernel until then)
    if (small) TAVB_HIP(hipMemsetAsync(c->d_queries_f16.ptr, 0, q16_bytes, c->stream));  // (the split planes' padding queries: launch_f32_split_f16 writes the live ones)
    hipError_t e = tavb::launch_query_prepare(fq, nq, nq_pad, fdim, d_ms, small, d_norm, small ? nullptr : c->d_queries_f16.ptr, d_delta, d_floor,
                                              small ? nullptr : d_band, c->stream, bdirect, d_band_cnt, d_nflag, ms_uniform, min_scores[0],
                                              floor_of_min_score(min_scores[0]), d_ms, d_ms_floor);
    if (e != hipSuccess) return fail(TAVB_E_HIP, "query prepare launch failed: %s", hipGetErrorString(e));
    if (small) {
      e = tavb::launch_f32_split_f16(d_q, c->d_queries_f16.ptr, reinterpret_cast<char*>(c->d_queries_f16.ptr) + q16_bytes / 2, (int64_t)nq * c->dim, c->stream);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "query split launch failed: %s", hipGetErrorString(e));
    }
  }
  TileRun filt{};
  filt.skinny = small;
  filt.q32 = false;
  filt.qt = qt;
  filt.nq = nq;
  filt.nq_pad = nq_pad;
  filt.k = small ? KC : k;  // the wide tile ranks by the caller's k and keeps the band below it
  filt.band = small ? nullptr : d_band;
  filt.band_cnt = small ? nullptr : d_band_cnt;
  filt.lost = small ? nullptr : d_lost;
  filt.verdict = small ? nullptr : d_verdict;
  filt.index_base = index_base;
  filt.kernel_min_score = (ms_lo > 0.0f) ? 0.0f : ms_lo;  // the per-query relaxed thresholds (floor) do the filtering; NaN stays NaN
  filt.floor = d_floor;
  filt.bdirect = bdirect;
  filt.queries = c->d_queries_f16.ptr;
  filt.corpus = shadow_ops ? c->d_shadow.ptr : nullptr;
  filt.dim = fdim;
  filt.ladder = true;
  // a batch MOST of whose bands are not going to fit (every query next to more near-duplicates than a band holds) is found out before the last --
  // the big -- filter phase and goes straight to the exact split-plane form: the filter's last phase, its selection and the rescoring return at once
  const bool early = wide_fallback && c->early_exact;
  filt.doomed = early ? d_nflag + 1 : nullptr;
  filt.doomed_max = nq / 2;
  c->last_shadow = shadow_ops ? 1 : 0;
  if (int rc = run_tile_ladder(c, filt, reinterpret_cast<u64_t*>(c->d_approx.ptr), nullptr)) return rc;
  char* fb = reinterpret_cast<char*>(c->d_fb_queries.ptr);
  // behind the gathered operand: fp32 [cap][dim] on fp32 corpora, two fp16 planes of [cap][fdim] otherwise (fdim, not dim: an odd width's planes are padded)
  float* fb_thr = reinterpret_cast<float*>(fb + (size_t)2 * cap * (f32c ? c->dim : fdim) * 2);
  float* fb_band = fb_thr + cap;
  {
    Timed t(c, TAVB_KERNEL_RESCORE);
    hipError_t e = tavb::launch_rescore(c->corpus, f32c, c->dim, index_base, d_q, reinterpret_cast<const u64_t*>(c->d_approx.ptr), KC,
                                        small ? nullptr : d_band_cnt, small ? nullptr : d_verdict, d_delta, d_ms, nq, k, d_out, d_nflag, d_flagged,
                                        c->stream, filt.doomed, filt.doomed_max);
    if (e != hipSuccess) return fail(TAVB_E_HIP, "rescore launch failed: %s", hipGetErrorString(e));
    // (the exact tiles of an fp32 corpus read its own rows -- the dispatch admits only widths they take; those of an fp16 corpus of an odd
    //  width read the padded copy, which holds the same values)
    // the exact fallbacks start from what the filter has proven: the cut its last selection left in d_thr (the one before it when the early
    // verdict skipped the last phase) less the filter's error bound is a valid admission threshold on exact scores, so ONE phase each
    const float* seed = small ? nullptr : reinterpret_cast<const float*>(c->d_thr.ptr);
    e = f32c ? tavb::launch_gather_flagged_f32(d_q, c->dim, d_ms, d_nflag, d_flagged, cap, reinterpret_cast<float*>(fb), fb_thr, seed, d_delta, c->stream)
             : tavb::launch_gather_flagged(fq, fdim, d_ms, d_nflag, d_flagged, cap, fb, fb + (size_t)cap * fdim * 2, fb_thr, seed, d_delta,
                                           wide_fallback ? fb_band : nullptr, kExactBand, c->stream);
    if (e != hipSuccess) return fail(TAVB_E_HIP, "gather launch failed: %s", hipGetErrorString(e));
  }
  if (f32_big_k) {
    int* h = reinterpret_cast<int*>(c->h_flag.ptr);
    TAVB_HIP(hipMemcpyAsync(h, d_nflag, (size_t)(64 + cap) * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    TAVB_HIP(hipStreamSynchronize(c->stream));
    const int n_flagged = h[0] < cap ? h[0] : cap;
    if (n_flagged > 0) {  // (gather_flagged_f32_kernel has put their fp32 queries into fb[0 .. n_flagged))
      std::vector<float> ms_f((size_t)n_flagged);
      for (int i = 0; i < n_flagged; ++i) ms_f[i] = min_scores[h[64 + i]];
      u64_t* d_redo = reinterpret_cast<u64_t*>(c->d_fb_cand.ptr);
      const int tier = c->last_tier;  // (the batch's route stays what "last_tier" reports: the re-run is a detail of it)
      const int rc_redo = search_device_impl(c, reinterpret_cast<const float*>(fb), n_flagged, k, ms_f.data(), nullptr, c->rows, index_base, d_redo);
      c->last_tier = tier;
      if (rc_redo) return rc_redo;
      for (int i = 0; i < n_flagged; ++i)
        TAVB_HIP(hipMemcpyAsync(d_out + (size_t)h[64 + i] * k, d_redo + (size_t)i * k, (size_t)k * sizeof(u64_t), hipMemcpyDefault, c->stream));
    }
    return TAVB_OK;
  }
  {  // (run_tile_ladder times its own launches, in the same bucket)
    if (!big_k) {
      // the exact tile over the work list: returns at once when the list is empty (the normal case).  It ranks 64 rows per slot whatever k: the
      // rows beyond the k-th are the band the rescoring (slot mode) re-orders with the streaming kernels' arithmetic
      TileRun ex{};
      ex.skinny = true;
      ex.q32 = f32c;
      ex.qt = 64;
      ex.nq = cap;
      ex.nq_pad = cap;
      ex.k = 64;
      ex.index_base = index_base;
      ex.kernel_min_score = ms_lo;
      ex.floor = fb_thr;
      ex.queries = fb;
      if (!f32c && padded) {
        ex.corpus = c->d_shadow.ptr;
        ex.dim = fdim;
      }
      ex.active = d_nflag;
      ex.active_min = 0;
      ex.active_max = wide_fallback ? 64 : 0;
      ex.ladder = false;
      u64_t* fb_cand = reinterpret_cast<u64_t*>(c->d_fb_cand.ptr);
      if (int rc = run_tile_ladder(c, ex, fb_cand, nullptr, /*scatter_identity=*/true)) return rc;
      Timed t(c, TAVB_KERNEL_RESCORE);
      hipError_t e = tavb::launch_rescore_slots(c->corpus, f32c, c->dim, index_base, d_q, fb_cand, 64, nullptr, d_ms, wide_fallback ? 64 : cap, k, d_out,
                                                d_flagged, d_nflag, 0, wide_fallback ? 64 : 0x7fffffff, c->stream);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "fallback rescore launch failed: %s", hipGetErrorString(e));
    }
    if (wide_fallback) {
      TileRun wx{};
      wx.skinny = false;
      wx.q32 = false;
      wx.qt = 256;
      wx.nq = cap;
      wx.nq_pad = cap;
      wx.k = k;
      wx.index_base = index_base;
      wx.kernel_min_score = ms_lo;
      wx.floor = fb_thr;  // (+inf for the unused slots: they admit nothing)
      wx.band = fb_band;  // kExactBand below the k-th best: what the rescoring re-orders
      wx.queries = fb;    // [2][cap][dim]: the high plane, then the low plane
      wx.split_plane = (int64_t)cap * fdim * 2;
      if (padded) {
        wx.corpus = c->d_shadow.ptr;
        wx.dim = fdim;
      }
      wx.active = d_nflag;
      wx.active_min = big_k ? 0 : 64;
      wx.active_max = 0;
      wx.ladder = false;  // one phase, seeded by the filter's cut (fb_thr): three launches that return at once when the list is not this form's share
      wx.rs_queries = d_q;
      wx.rs_min_scores = d_ms;
      if (int rc = run_tile_ladder(c, wx, d_out, d_flagged)) return rc;
    }
  }
  return TAVB_OK;
}

}  // namespace

// Routes a device-resident query batch: streaming scan (few queries), 32/64-query tile (small batches; every batch on
// fp32 corpora), or the 256-query fp16 tile with exact rescoring (large batches on fp16 corpora).  Not part of the public ABI.
// min_scores: one threshold per query -- the tiles take them per query (a batch of Q `fuzzy_lookup_embedding` calls has Q of them,
// vectorbase.py:163-173), so a mixed batch takes the same route as a uniform one.
int tavb_search_device_dispatch(tavb_ctx* c, const float* d_q, int nq, int k, const float* min_scores,
                                uint32_t index_base, u64_t* d_out) {
  bool uniform_thr = true;
  for (int i = 1; i < nq; ++i) uniform_thr = uniform_thr && (memcmp(&min_scores[i], &min_scores[0], sizeof(float)) == 0);
  const bool f16c = (c->dtype == TAVB_F16);
  c->last_direct = 0;
  {  // small corpora, 2 .. 128 queries: the grouped streaming scan + one merge where it beats the tiles (plan_direct_group)
    const int64_t bytes = (int64_t)c->rows * c->dim * (f16c ? 2 : 4);
    const bool shadow2 = !f16c && c->f32_shadow >= 2 && bytes >= c->f32_shadow_min_bytes;
    if (c->corpus && c->rows > 0 && !c->dispatch_no_group && nq >= 2 && nq <= std::min<int64_t>(c->direct_group_max_nq, TAVB_MAX_GROUPED_QUERIES) && !shadow2 &&
        c->small_direct_bytes > 0 && bytes <= c->small_direct_bytes && k <= 64) {
      int waves = c->geom.waves < 1 ? 1 : (c->geom.waves > 16 ? 16 : c->geom.waves);
      const DirectGroupPlan plan = plan_direct_group(c, nq, k, scan_blocks_for(c, c->rows, waves, c->geom.unroll), /*host=*/false);
      if (plan.worth) {
        c->last_direct = 4;
        return search_device_grouped(c, d_q, nq, k, min_scores, index_base, d_out, plan);
      }
    }
  }
  // the wide tile keeps a band below the k-th best (any k the fused selections serve: the reference's max_matches = 50, convsettings.py:61-63,
  // included).  Its flagged queries need an exact tile: the 64-query one up to k = 64, beyond that the wide split-plane form (fp16 corpora).
  // A width that is not a multiple of 64 (the tile's K step) rides the wide tile on a zero-padded copy of the rows (search_wide_exact): any
  // width on fp16 corpora (the exact fallbacks read the padded copy too: the same values), multiples of 16 on fp32 ones (their exact
  // tile reads the corpus' own fp32 rows).
  const bool odd_width = c->dim % 64 != 0;
  const int wide_dim = ((c->dim + 63) / 64) * 64;
  // (round 6: ANY width on fp16 corpora -- the rescoring reads rows that are not 16-byte aligned element by element, in the scalar streaming
  //  kernel's order)
  const bool width_ok = !odd_width || f16c || c->dim % 16 == 0;
  const bool exact_tile = (k <= 64) ? ((f16c && odd_width) ? tavb::skinny_supported(wide_dim, k, false) : tavb::skinny_supported(c->dim, k, !f16c))
                                    : (f16c ? c->wide_fallback != 0 : true);  // (fp32, k > 64: flagged queries are re-run on the streaming kernels)
  const int64_t corpus_bytes = (int64_t)c->rows * c->dim * (f16c ? 2 : 4);
  // k > 64: the 32/64-query tile does not serve it and the streaming kernels take FOUR such queries per corpus pass -- from 9 queries (more than two
  // passes), or 3 on corpora of mfma_big_bytes and more, the wide tile (32 queries over 2M x 1536 fp16 rows, k = 65: 12.5 ms against 1.3)
  const bool big_k_batch = k > 64 && (nq >= 9 || (nq >= 3 && corpus_bytes >= c->mfma_big_bytes));
  // (narrow rows: the streaming scan's cost per row does not shrink with the row -- 3 queries over 1M x 384 fp32 rows 0.39 ms against 0.28 on the
  //  wide tile, profiles/r06_raw/regime_sweep_d384.md -- so the byte thresholds measured at D = 1536 scale down with the width)
  const int64_t few_bytes_f32 = c->dim < 1536 ? c->mfma_few_bytes_f32 / 1536 * c->dim : c->mfma_few_bytes_f32;
  const bool wide_batch = nq >= c->mfma_min_batch || big_k_batch || (nq >= c->mfma_min_batch_big && corpus_bytes >= c->mfma_big_bytes) ||
                          (!f16c && nq >= c->mfma_min_batch_f32) ||
                          (!f16c && nq >= c->mfma_min_batch_big_f32 && corpus_bytes >= c->mfma_big_bytes_f32) ||
                          (!f16c && nq >= 2 && c->mfma_min_batch_big_f32 <= 64 && corpus_bytes >= few_bytes_f32);
  bool wide = (f16c || c->f32_shadow) && c->corpus && wide_batch && width_ok && tavb::mfma_supported(wide_dim, k) && c->rows > 0 && exact_tile;
  // f32_shadow = 2: smaller batches (and single queries) on big fp32 corpora filter on the shadow too, with the 32/64-query tile
  // (it keeps the best 64 candidates per query: k up to 48 leaves the slack the completeness test needs)
  bool shadow_small = !wide && !f16c && c->f32_shadow >= 2 && c->corpus && nq <= 64 && tavb::mfma_supported(c->dim, 64) && k <= 48 &&
                      tavb::skinny_supported(c->dim, k, false) && (int64_t)c->rows * c->dim * 4 >= c->f32_shadow_min_bytes;
  if ((wide || shadow_small) && (!f16c || odd_width)) {  // the filter needs the fp16 shadow / padded copy; without the memory for it the other kernels serve the batch
    const size_t need = (size_t)c->rows * wide_dim * 2;
    if (c->d_shadow.cap < need) {
      c->norm_rows = 0;  // reserve() does not keep the old contents
      if (c->d_shadow.reserve(need) != TAVB_OK) wide = shadow_small = false;
    }
  }
  c->last_shadow = 0;
  if (shadow_small) {
    c->last_tier = 5;
    return search_wide_exact(c, d_q, nq, k, min_scores, index_base, d_out, /*small=*/true);
  }
  // 32/64-query tiles at HBM speed: small batches on fp16 corpora, every batch from `skinny_min_batch_f32` up on fp32 ones
  const bool skinny = !wide && c->corpus && c->rows > 0 && tavb::skinny_supported(c->dim, k, !f16c) &&
                      nq >= (f16c ? ((c->dim <= 768 && c->rows >= 200000) ? std::min<int64_t>(2, c->skinny_min_batch_f16) : c->skinny_min_batch_f16)
                                  : c->skinny_min_batch_f32);  // (two queries over 1M x 384 fp16 rows: 0.23 ms on the streaming scan, 0.17 on the tile)
  if (wide) {
    c->last_tier = 4;  // 1-3 = streaming tiers, 4 = 256-query MFMA tile, 5 = 32/64-query MFMA tile
    return search_wide_exact(c, d_q, nq, k, min_scores, index_base, d_out);
  }
  if (skinny) {
    const int qt = tavb::skinny_query_tile(nq);
    const int nq_pad = ((nq + qt - 1) / qt) * qt;
    const bool q32 = !f16c;  // on an fp32 corpus the tile multiplies fp32 queries, on an fp16 one fp32 queries split into fp16 high + low planes
    const size_t plane = (size_t)nq_pad * c->dim * (q32 ? 4 : 2);
    const size_t qbytes = plane * (q32 ? 1 : 2);
    if (int rc = c->d_queries_f16.reserve(qbytes)) return rc;
    float *d_ms = nullptr, *d_ms_floor = nullptr;
    if (!uniform_thr) {  // per-query thresholds: exclusive admission floors valid from the first row on
      bool uni = false;
      if (int rc = upload_min_scores(c, min_scores, nq, nq_pad, &d_ms, &d_ms_floor, &uni)) return rc;
    }
    TAVB_HIP(hipMemsetAsync(c->d_queries_f16.ptr, 0, qbytes, c->stream));
    if (q32) {
      TAVB_HIP(hipMemcpyAsync(c->d_queries_f16.ptr, d_q, (size_t)nq * c->dim * 4, hipMemcpyDeviceToDevice, c->stream));
    } else {
      hipError_t e = tavb::launch_f32_split_f16(d_q, c->d_queries_f16.ptr, reinterpret_cast<char*>(c->d_queries_f16.ptr) + plane,
                                                (int64_t)nq * c->dim, c->stream);
      if (e != hipSuccess) return fail(TAVB_E_HIP, "query split launch failed: %s", hipGetErrorString(e));
    }
    c->last_tier = 5;
    TileRun r{};
    r.skinny = true;
    r.q32 = q32;
    r.qt = qt;
    r.nq = nq;
    r.nq_pad = nq_pad;
    r.k = k;
    r.index_base = index_base;
    r.kernel_min_score = uniform_thr ? min_scores[0] : lowest_min_score(min_scores, nq);
    r.floor = d_ms_floor;
    r.queries = c->d_queries_f16.ptr;
    r.ladder = true;
    return run_tile_ladder(c, r, d_out, nullptr);
  }
  return search_device_impl(c, d_q, nq, k, min_scores, nullptr, c->rows, index_base, d_out);
}

extern "C" int tavb_plan_ladder(int64_t rows, int32_t nq, int32_t n_cu, int64_t* out_bounds, int32_t cap) {
  if (rows < 0 || nq < 1 || n_cu < 8) return fail(TAVB_E_INVALID, "bad shape");
  const int qt = tavb::mfma_query_tile_for(nq, rows, n_cu);
  const int nq_pad = ((nq + qt - 1) / qt) * qt;
  const int splits = tavb::mfma_pick_splits(rows, nq_pad, qt, n_cu);
  const std::vector<int64_t> b = ladder_bounds(rows, splits, nq_pad, /*skinny=*/false, /*ladder=*/true, /*sample_opt=*/0, /*growth=*/4);
  for (size_t i = 0; out_bounds && i < b.size() && (int)i < cap; ++i) out_bounds[i] = b[i];
  return (int)b.size() - 1;
}

