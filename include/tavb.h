/*
 * tavb.h -- C ABI of libtavb.so, the MI355X (gfx950) engine behind typeagent's
 * VectorBase nearest-neighbour lookup.
 *
 * The reference (microsoft/typeagent-py) has no FFI: the boundary it offers is the
 * Python class `typeagent.aitools.vectorbase.VectorBase`
 * (src/typeagent/aitools/vectorbase.py:82-287).  The entry points below are what a
 * ctypes binding for that class's numeric methods binds; each one cites the
 * reference lines whose arithmetic it replaces.  `typeagent_py_amd/_native.py` is
 * that binding; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *  - plain C types only; no torch / HIP types in signatures (a `void* stream` is an
 *    opaque hipStream_t, NULL = the context's own stream).
 *  - every function returns TAVB_OK (0) or a negative TAVB_E_* code; the message for
 *    the calling thread's last failure is `tavb_last_error()`.  The library never
 *    calls abort()/exit().
 *  - "dev" pointers are device memory owned by the caller (normally a torch tensor);
 *    the library never frees them and keeps `tavb_set_corpus`'s pointer only until
 *    the next tavb_set_corpus / tavb_destroy.  Host pointers are not retained after
 *    a call returns; all outputs are caller-allocated.
 *  - one in-flight call per context (the Python wrapper enforces this).
 *  - results are ordered by (score descending, ordinal ascending); the reference's
 *    order among exactly equal float32 scores is numpy-implementation-defined
 *    (vectorbase.py:183-187), ours is deterministic.
 *  - scores are the reference's public 0..1 scale: clip((cos + 1) / 2, 0, 1) evaluated
 *    in float32 (vectorbase.py:44-47); `min_score` is compared as float32
 *    (vectorbase.py:179 under NEP 50); NaN scores never pass (numpy `>=`).
 */
#ifndef TAVB_H
#define TAVB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever an entry point, a signature, an option or a kernel id changes incompatibly; the binding refuses a library of another version */
#define TAVB_ABI_VERSION 6

#define TAVB_OK 0
#define TAVB_E_INVALID (-1)     /* bad argument */
#define TAVB_E_HIP (-2)         /* a HIP runtime call failed; message has hipGetErrorString */
#define TAVB_E_NO_CORPUS (-3)   /* search before tavb_set_corpus */
#define TAVB_E_UNSUPPORTED (-4) /* shape outside what the kernels cover (see message) */
#define TAVB_E_NOMEM (-5)
#define TAVB_E_TIMEOUT (-7)     /* tavb_synchronize: an exchange did not complete within "comm_timeout_ms"; the communicator was aborted */
#define TAVB_E_PEER (-6)        /* tavb_decode_keys: the lists come from a collective lookup in which a rank's local search failed (they lead with TAVB_KEY_PEER_FAILED) */

#define TAVB_F32 0
#define TAVB_F16 1

/* Largest k served by the fused select-while-streaming kernels. */
#define TAVB_MAX_FUSED_K 256
/* Largest number of queries one streaming pass serves (bigger batches are split). */
#define TAVB_MAX_STREAM_QUERIES 8

typedef struct tavb_ctx tavb_ctx;

/* ---- library ---------------------------------------------------------------- */
int tavb_version(void);
const char* tavb_last_error(void);
int tavb_device_count(int* out_count);

/* ---- context ---------------------------------------------------------------- */
/* One context per (process, GPU).  `stream` = an existing hipStream_t to launch on
 * (e.g. torch.cuda.current_stream().cuda_stream) or NULL to create a private one. */
int tavb_create(int device, void* stream, tavb_ctx** out);
int tavb_destroy(tavb_ctx* ctx);
int tavb_synchronize(tavb_ctx* ctx);

/* Tunables (launch geometry etc.); unknown names -> TAVB_E_INVALID.
 *   "scan_blocks"   workgroups of the streaming scan (0 = auto: one per CU)
 *   "scan_waves"    waves per workgroup, 1..16 (default 16)
 *   "scan_unroll"   rows in flight per wave: 1, 2 or 4 (default 2)
 *   "scan_nt"       1 = non-temporal corpus loads (default 1)
 *   "scan_pipe"     1 = software-prefetch next rows before reducing current ones
 *   "force_tier"    0 = auto, 1 = fixed-D kernel, 2 = vector kernel, 3 = scalar kernel
 *   "mfma_min_batch" smallest batch routed to the 128/256-query MFMA tile (default 65; on corpora of "mfma_big_bytes" (default 256 MiB) or more: "mfma_min_batch_big", default 33; dim % 64 == 0 directly, other widths -- multiples of 8 on
 *                   fp16 corpora, of 16 on fp32 ones -- on a zero-padded fp16 copy of the rows (+ device memory: rows x pad64(dim) x 2 bytes); k up to TAVB_MAX_FUSED_K on either
 *                   (fp32 corpora through their fp16 shadow; beyond k = 64 a flagged query of an fp32 corpus is re-run on the streaming kernels after ONE
 *                   host round trip, so such a call may block before it returns); thresholds may differ per query; k > 64 rides the tile from 9 queries,
 *                   3 on corpora of mfma_big_bytes and more -- the streaming kernels take four such queries per corpus pass)
 *   "mfma_min_batch_f32", "mfma_min_batch_big_f32" / "mfma_big_bytes_f32", "mfma_few_bytes_f32"  FP32 corpora: batches that take the wide tile over the fp16 shadow
 *                   (needs "f32_shadow" >= 1) + exact fp32 rescoring instead of the fp32 kernels -- half the bytes per pass and fp16 matrix rates:
 *                   from mfma_min_batch_f32 (default 33) queries at any corpus size (64 queries over 5000 x 1536 rows: 0.11 against 0.20 ms on the 64-query
 *                   fp32 tile), from mfma_min_batch_big_f32 (default 5) queries on corpora of mfma_big_bytes_f32 (default 1e9 bytes) or more (32 queries over
 *                   1M x 1536 rows: 0.73 against 1.24 ms), and batches of 2 .. 4 queries on corpora of mfma_few_bytes_f32 (default 4 GiB) or more (0.78 against
 *                   0.97 ms at 1M rows x 1.5); mfma_min_batch_big_f32 > 64 switches the last two rules off
 *   "skinny_min_batch_f32" / "skinny_min_batch_f16"  smallest batch routed to the 32/64-query MFMA tile on fp32 / fp16
 *                   corpora (defaults 5 / 3, the measured break-even; used up to
 *                   mfma_min_batch - 1); smaller batches use the streaming tiers
 *   "mfma_sample_rows", "mfma_ladder"  phases of the MFMA paths' threshold ladder: rows of the first phase (0 = auto: one
 *                   tile per workgroup; -1 = a single phase, no seeding) and the growth factor of the following ones
 *   "f32_shadow"            1 (default): batches of mfma_min_batch+ queries on FP32 corpora run the fp16 tile over an fp16 shadow copy of the corpus
 *                           (built on first use, +50 % device memory, extended on append) and rescore the candidates with the fp32 rows: same
 *                           answers, ~6x the throughput of the fp32 matrix path; 0: fp32 kernels only (the shadow is freed); 2: EVERY lookup
 *                           (single queries included) on fp32 corpora of "f32_shadow_min_bytes" (default 2 GiB) and more filters on the
 *                           shadow with the 32/64-query tile: half the bytes per pass (1M x 1536: 0.93 -> 0.70 ms per query, 32 queries 1.46 -> 0.94 ms)
 *   "last_shadow"           (get) 1 when the last lookup's corpus pass read the shadow
 *   "mfma_tile"             queries per workgroup tile of the wide fp16 kernel: 0 = auto (128 where that pads less: up to 128, 257..384, 513..640 queries; else 256), 128, 256
 *   "mfma_splits", "mfma_sched", "mfma_ablate"  measurement knobs: row ranges per launch; staging variant (256-query tile: 1 / 2 = other piece
 *                           schedules, 3 / 4 / 5 = other issue orders of a slice's MFMAs -- profiles/r06_mfma_power.md; 32-query tile: 7 = LDS-DMA ring (the default), 8 = deep corpus ring, 5 = register staging four K steps
 *                           deep, 9 = 64-byte K steps -- profiles/r05_mid_batch.md); parts of a tile kernel compiled out (answers are garbage)
 *   "graph_max_bytes"       single-query lookups (tavb_search / tavb_search_batch with nq = 1) on corpora of at most this many bytes replay ONE
 *                           captured HIP graph (query H2D, scan, merge into pinned host memory) instead of three submissions; the first call
 *                           of a (corpus, k, min_score) shape runs plain, the second captures, later ones replay.  Default 0 = never: on
 *                           ROCm 7.2 the replay measured SLOWER than the plain submissions (48 vs 41 us at 10k x 1536)
 *   "last_graph"            (get) 1 when the last lookup was such a replay
 *   "comm_force"            1: tavb_search_allgather runs its all-gather + merge even in a world of one rank (tests, dry runs)
 *   "comm_fail_rank"        fault injection (default -1 = off): on the rank of the communicator with this number the local search of
 *                           tavb_search_allgather fails as a launch or an allocation inside it would -- that rank still joins the all-gather
 *                           (with TAVB_KEY_PEER_FAILED lists) and returns its error, every other rank's lists decode to TAVB_E_PEER
 *   "comm_reserve_keys"     keys per rank of the exchange buffers tavb_comm_init reserves (default 2^20 = 8 MiB + world x 8 MiB; set BEFORE tavb_comm_init):
 *                           an exchange of up to that many keys (nq x k) allocates nothing between entering tavb_search_allgather and ncclAllGather;
 *                           bigger ones travel through the same buffers in chunks of whole queries (every rank cuts the same chunks)
 *   "comm_timeout_ms"       0 (default) = tavb_synchronize waits for an exchange for ever; > 0: after that many milliseconds the communicator is
 *                           aborted (ncclCommAbort: the stream drains), the context is left without one and tavb_synchronize returns TAVB_E_TIMEOUT
 *   "comm_fail_alloc"       fault injection: 1 = the per-call list allocation of tavb_search_allgather (lists beyond comm_reserve_keys) fails; the rank
 *                           still joins every chunk of the exchange with TAVB_KEY_PEER_FAILED lists
 *   "comm_stall_ms"         fault injection (one shot, 0..5000): the next exchange is held up on the stream for that long, as by a late peer
 *   "comm_world", "comm_rank" (read only) shape of the context's communicator (0 / -1 without one)
 *   "last_tier" (read only) the kernel family of the last lookup: 1-3 streaming tiers, 4 = 256-query MFMA tile (exact
 *                   filter + fp32-query rescoring of its candidates), 5 = 32/64-query MFMA tile
 *   "last_flagged" (read only; synchronises) queries of the last 256-query-tile lookup whose candidate set could not be
 *                   proven complete and were re-run exactly (0 on ordinary data)
 *   "band_max"      256 .. 2048 (default 2048; 1024 until ABI 6): the most candidates per query the 128/256-query tile's selection hands to the exact
 *                   rescoring -- every row within 2 delta of the approximate k-th best.  A query with MORE rows than that inside its band (next to a
 *                   bigger cluster of near-duplicates) is flagged and re-run exactly; a band of 1500 near-duplicates per query (bench.py cfg3_dup) used
 *                   to cost the whole batch an exact pass of twice the MFMAs and now costs 1500 gathered rows per query in the rescoring
 *   "wide_fallback" 1 (default): when MORE than 64 queries of a batch of 256+ are flagged, they are re-run on the 256-query tile in its
 *                   exact split-plane form (fp32 queries as two fp16 planes) instead of 64 at a time on the 64-query exact tile; 0 = never
 *   "early_exact"   1 (default): with wide_fallback, a batch MOST of whose queries (> nq / 2) already hold, before the last and biggest filter phase, a band that
 *                   extrapolates past the band buffer (every query next to more near-duplicates than a band holds) skips that phase, its selection and the
 *                   rescoring, and all its queries take the exact split-plane form at once; "last_doomed" (read only; synchronises) = that count
 *   "mfma_bdirect"  0 (default) / 1: the 256-query tile takes its query operand in MFMA-fragment-major order straight from L2 into registers
 *                   (no LDS staging; three corpus slots instead of two): measured +0.8 %, kept as an option (profiles/r04_cfg3_kernel.md section 10)
 *   "small_direct_bytes" host-synchronous lookups of one query (tavb_search) or a few (tavb_search_batch with nq <= 8; <= 4 for k > 64) on corpora up
 *                   to this many bytes (default 128 MiB; 0 = never) are ONE launch ("small_direct_keys", default 8192: the most keys the lists may hold
 *                   -- the grid is cut to fit; twice that for a batch, which goes the usual way when its share would starve the grid): the scan's per-workgroup lists go to pinned host memory and are merged
 *                   on the host; "last_direct" (read only) = 1 when the last lookup took that path, 2 when in addition the query rode
 *                   inside the kernel arguments ("inline_query", default 1: 1536-wide queries on the default scan form; no H2D copy before the launch)
 *   "direct_group_max_nq" (default 128; 0 = never) batches of 2 .. this many queries (k <= 64) on corpora up to "small_direct_bytes" take that ONE launch
 *                   in its GROUPED form wherever a fitted cost model expects it to beat the 32/64-query and wide tiles (a corpus of a few
 *                   thousand rows is one or two busy CUs for a tile): a 2-D grid of row workgroups x query groups of one or two queries
 *                   ("direct_group": 1 / 2 / 4 / 8 forces the group size and the form, 0 = pick), "direct_group_wgs" workgroups in all (0 = pick:
 *                   256 or 512), lists of up to "direct_group_keys" keys (default 32768) merged on the host; "last_direct" = 3.  The
 *                   device-resident calls (tavb_search_device, tavb_search_allgather) take the same scan + ONE merge launch ("last_direct" = 4).
 *                   Either way the answers are the single-query lookups' bit for bit.  profiles/r06_group_sweep.md
 */
int tavb_set_option(tavb_ctx* ctx, const char* name, int64_t value);
int tavb_get_option(tavb_ctx* ctx, const char* name, int64_t* out_value);

/* ---- corpus ----------------------------------------------------------------- */
/* Borrow a row-major [rows, dim] matrix in device memory as the corpus
 * (replaces the host ndarray `VectorBase._vectors`, vectorbase.py:84, 176).
 * dtype TAVB_F32 (the reference's layout) or TAVB_F16 (our storage extension: the
 * values the kernels see are the fp16 values widened to f32).  `ordinal_base` is
 * added to every returned ordinal (row-sharded corpora: the shard's first row). */
int tavb_set_corpus(tavb_ctx* ctx, const void* dev_rows, int64_t rows, int32_t dim, int32_t dtype,
                    int64_t ordinal_base);

/* Load path (what knowpro/serialization.py:207-221 and the BLOB reload loops of
 * storage/sqlite/messageindex.py:33-45 / reltermsindex.py:144-156 feed): float32 host rows [n_rows, dim] -> device
 * memory at dev_dst (inside the caller's capacity-doubling corpus buffer) as dst_dtype.  The rows go through two pinned
 * 16 MiB staging slots (filled by a few host threads) and asynchronous copies, so the host-side copy of chunk i+1 overlaps
 * the DMA of chunk i; for TAVB_F16 the conversion (round to nearest even) runs on the device from a scratch slot, the
 * host never builds an fp16 (or a second fp32) copy.  Returns when the rows are in place. */
int tavb_upload_rows(tavb_ctx* ctx, const float* rows_host, int64_t n_rows, int32_t dim, void* dev_dst, int32_t dst_dtype);

/* Tell the library that rows [first_row, rows) of the borrowed corpus buffer were rewritten in place (e.g. an append into
 * spare capacity followed by tavb_set_corpus with the new row count, or a re-upload after the host matrix was edited):
 * per-corpus quantities it caches (the row-norm maxima of the batched path's exactness proof; the fp16 shadow copy of an
 * fp32 corpus, see "f32_shadow") are refreshed on the next lookup.  tavb_set_corpus with a different pointer / shape /
 * dtype implies it; the caches are keyed on the ADDRESS, so a new buffer that happens to sit where a freed one lived needs
 * tavb_corpus_modified(ctx, 0) (the Python binding does that whenever it is handed a different tensor object). */
int tavb_corpus_modified(tavb_ctx* ctx, int64_t first_row);

/* K1: rows / ||row||_2 in float32, zero rows unchanged
 * (model_adapters.py:181-183; tools/benchmark_vectorbase.py:85-86).  in == out allowed. */
int tavb_normalize_rows_f32(tavb_ctx* ctx, const float* dev_in, float* dev_out, int64_t rows, int32_t dim);

/* float32 -> float16 (round to nearest even), used to build f16 corpora on device. */
int tavb_convert_f32_to_f16(tavb_ctx* ctx, const float* dev_in, void* dev_out, int64_t count);

/* A packed result key: (float32 score bits << 32) | (0xFFFFFFFF - ordinal); 0 = empty
 * slot.  Bigger key = better hit, so per-shard lists merge with integer compares. */
typedef uint64_t tavb_key;
/* Not a result: what a rank whose local search FAILED contributes to the all-gather of tavb_search_allgather, in every slot of its lists.  All
 * bits set sorts above every real key (scores are in [0, 1]), so it leads every merged list on every rank: no rank can mistake a result that
 * is missing a shard for an answer.  tavb_decode_keys returns TAVB_E_PEER for lists that carry it. */
#define TAVB_KEY_PEER_FAILED (~(tavb_key)0)

/* ---- synchronous lookups (host in, host out) ---------------------------------- */
/* fuzzy_lookup_embedding without predicate (vectorbase.py:163-190):
 * up to k best rows with score >= min_score.  out_* hold k entries; *out_count = M. */
int tavb_search(tavb_ctx* ctx, const float* query_host, int32_t k, float min_score, int64_t* out_ordinals,
                float* out_scores, int32_t* out_count);

/* fuzzy_lookup_embedding_in_subset (vectorbase.py:203-230).  `rows_host[i]` is the
 * (already wrapped, range-checked) corpus row of subset position i; the call returns
 * POSITIONS into the subset list (the wrapper maps them back through the caller's
 * list, reproducing `subset[indices[i]]`, :229).  Ties: ascending position. */
int tavb_search_subset(tavb_ctx* ctx, const float* query_host, const int64_t* rows_host, int64_t n_subset,
                       int32_t k, float min_score, int64_t* out_positions, float* out_scores,
                       int32_t* out_count);

/* Q independent lookups in one submission (the batching the reference leaves as a TODO,
 * storage/sqlite/reltermsindex.py:259-271).  Semantics == Q calls of tavb_search.
 * queries_host: float32 [nq, dim].  min_scores: nq thresholds -- one per query, as Q calls of the reference have Q `min_score`
 * arguments (vectorbase.py:163-173); a mixed batch runs on the same kernels as a uniform one.  Outputs [nq, k] / [nq]. */
int tavb_search_batch(tavb_ctx* ctx, const float* queries_host, int32_t nq, int32_t k, const float* min_scores,
                      int64_t* out_ordinals, float* out_scores, int32_t* out_counts);

/* ---- message re-rank on the device (the step right after the lookup in both providers) -------- */
/* Chunk row -> message ordinal map of the corpus: device int32 [rows], borrowed like the corpus (-1 = the row belongs to
 * no message).  n_messages bounds the ordinals (size of the accept bitmap). */
int tavb_set_row_messages(tavb_ctx* ctx, const int32_t* dev_row_to_msg, int64_t rows, int64_t n_messages);

/* SqliteMessageTextIndex.lookup_by_embedding / lookup_in_subset_by_embedding (storage/sqlite/messageindex.py:296-326,
 * 182-257) in one submission: the best k chunk rows with score >= min_score over the WHOLE corpus, THEN the
 * set-membership filter on their message ordinals (accept_msgs_host: n_accept ordinals, the provider's
 * `ordinals_set`; n_accept = -1: no filter), THEN the best score per message, sorted by score (stable: ties keep hit
 * order), cut at max_messages -- the provider's order of operations, so it returns exactly what the provider returns
 * (possibly fewer than max_messages).  out_* hold k entries. */
int tavb_search_messages(tavb_ctx* ctx, const float* query_host, int32_t k, float min_score, const int32_t* accept_msgs_host, int64_t n_accept,
                         int32_t max_messages, int64_t* out_messages, float* out_scores, int32_t* out_count);

/* The in-memory provider's form (storage/memory/messageindex.py:173-207 via knowpro/textlocindex.py:164-177): a true
 * subset gather (rows_host: corpus row per subset position, as tavb_search_subset), then the same aggregation. */
int tavb_search_messages_subset(tavb_ctx* ctx, const float* query_host, const int64_t* rows_host, int64_t n_subset, int32_t k, float min_score,
                                int32_t max_messages, int64_t* out_messages, float* out_scores, int32_t* out_count);

/* Split form of tavb_search_batch for a caller that drives SEVERAL contexts (one per GPU, row shards of one corpus) from
 * one thread: tavb_search_begin copies the queries and enqueues the kernels on the context's stream, tavb_search_end
 * waits for them and returns nq sorted lists of k keys.  Keys carry ordinal_base + row (must stay below 2^32 - 1), so the
 * lists of the shards merge with tavb_merge_keys_host (lists [n_lists, nq, k] -> out [nq, k]; a pure host helper) and
 * decode with tavb_decode_keys.  `cursor` (optional, nq == 1): only hits strictly AFTER that key in the result order,
 * i.e. the paging cursor of tavb_search_after as a key -- the same key on every shard. */
int tavb_search_begin(tavb_ctx* ctx, const float* queries_host, int32_t nq, int32_t k, const float* min_scores, const tavb_key* cursor);
int tavb_search_end(tavb_ctx* ctx, int32_t nq, int32_t k, tavb_key* out_keys_host);
int tavb_merge_keys_host(const tavb_key* lists, int32_t n_lists, int32_t nq, int32_t k, tavb_key* out);

/* Every row with score >= min_score in ONE pass -- the reference's `np.flatnonzero(scores >= min_score)`
 * (vectorbase.py:179, 219) -- sorted best first on the host; the first min(total, max_out) are returned, *out_total is the
 * number of survivors.  This is the candidate set of the predicate path (vectorbase.py:191-201), of max_hits >
 * TAVB_MAX_FUSED_K and of the max_hits == 0 quirk (`[-0:]`: all survivors, sorted).  Call with max_out = 0 to learn the
 * count first.  The subset form returns subset POSITIONS (like tavb_search_subset). */
int tavb_search_all(tavb_ctx* ctx, const float* query_host, float min_score, int64_t max_out, int64_t* out_ordinals, float* out_scores,
                    int64_t* out_count, int64_t* out_total);
int tavb_search_subset_all(tavb_ctx* ctx, const float* query_host, const int64_t* rows_host, int64_t n_subset, float min_score, int64_t max_out,
                           int64_t* out_positions, float* out_scores, int64_t* out_count, int64_t* out_total);

/* Continuation ("cursor") forms: the next k hits strictly AFTER the hit
 * (after_score, after_ordinal) in the (score descending, ordinal ascending) order.
 * Feeding the last hit of one page as the cursor of the next enumerates every row
 * with score >= min_score, best first, k (<= TAVB_MAX_FUSED_K) at a time.  This is the
 * candidate stream of the predicate path (vectorbase.py:191-201), of max_hits >
 * TAVB_MAX_FUSED_K and of the max_hits == 0 quirk (all survivors, sorted).  For the
 * subset form the cursor is a subset POSITION.  Each page is one corpus pass. */
int tavb_search_after(tavb_ctx* ctx, const float* query_host, int32_t k, float min_score, float after_score,
                      int64_t after_ordinal, int64_t* out_ordinals, float* out_scores, int32_t* out_count);
int tavb_search_subset_after(tavb_ctx* ctx, const float* query_host, const int64_t* rows_host, int64_t n_subset,
                             int32_t k, float min_score, float after_score, int64_t after_position,
                             int64_t* out_positions, float* out_scores, int32_t* out_count);


/* ---- asynchronous device-resident lookups (sharding, benchmarking) -------------- */
/* Queries already on the device (float32 [nq, dim]); writes nq sorted lists of k keys to
 * dev_out_keys [nq, k] on the context's stream and returns without synchronising.
 * Keys carry ordinal_base + row (must stay below 2^32 - 1). */
int tavb_search_device(tavb_ctx* ctx, const float* dev_queries, int32_t nq, int32_t k, float min_score,
                       tavb_key* dev_out_keys);

/* Subset form of the above: one query (device, float32 [dim]) against the rows listed in
 * dev_rows (device int32 [n_subset], already wrapped / range-checked by the caller); keys carry
 * subset POSITIONS.  Asynchronous.  Used by the fused multi-index submission. */
int tavb_search_subset_device(tavb_ctx* ctx, const float* dev_query, const int32_t* dev_rows, int64_t n_subset,
                              int32_t k, float min_score, tavb_key* dev_out_keys);
/* Host-synchronous tavb_search_subset (vectorbase.py:203-230) over a subset whose row list is ALREADY on the device: callers that search the
 * same subset again and again (the memory provider hands `lookup_in_subset_by_embedding` the same scope list per query term,
 * storage/memory/messageindex.py:173-183; tools/benchmark_vectorbase.py:133-163 passes one list for every round) upload and range-check it
 * once.  query_host float32 [dim]; dev_rows device int32 [n_subset], wrapped and range-checked by the caller; positions / scores / count as
 * tavb_search_subset.  A subset of up to "small_direct_bytes" of rows is ONE launch (per-workgroup lists into pinned memory, merged on the host;
 * a 1536-wide query rides in the kernel arguments), like tavb_search on small corpora: 21 us against 28 for the 1000-of-10k case. */
int tavb_search_subset_resident(tavb_ctx* ctx, const float* query_host, const int32_t* dev_rows, int64_t n_subset, int32_t k, float min_score,
                                int64_t* out_positions, float* out_scores, int32_t* out_count);

/* Merge `n_lists` sorted key lists per query (dev_lists [n_lists, nq, k], e.g. the
 * all-gathered per-shard results) into one list per query: dev_out_keys [nq, k]. */
int tavb_merge_device(tavb_ctx* ctx, const tavb_key* dev_lists, int32_t n_lists, int32_t nq, int32_t k,
                      tavb_key* dev_out_keys);

/* Decode host copies of keys into ordinals/scores/counts (pure host helper). */
int tavb_decode_keys(const tavb_key* keys_host, int32_t nq, int32_t k, int64_t* out_ordinals, float* out_scores,
                     int32_t* out_counts);

/* ---- row-sharded corpora: one process per GPU, RCCL over xGMI ------------------------------------------------------------
 * Every rank holds a contiguous range of the corpus rows in its own context (tavb_set_corpus with ordinal_base = the shard's first
 * row); a lookup is every rank scanning its shard for the same queries, ONE all-gather of the per-shard [nq, k] key lists
 * (nq * k * 8 bytes per rank: 256 KiB at 1024 x 32) and a merge kernel on every rank, so that every rank returns the whole-corpus
 * answer of vectorbase.py:163-190 (keys carry global ordinals and order by (score desc, ordinal asc): the merged answer is the
 * single-device one, ties included).  A rank whose local search fails still joins the all-gather -- the peers are never left waiting -- with
 * TAVB_KEY_PEER_FAILED lists, returns its own error, and every other rank's merged lists decode to TAVB_E_PEER (tavb_decode_keys): a lookup
 * either returns the whole-corpus answer on a rank or an error, never an answer that silently misses a shard.  That covers every failure
 * local to a rank between entering the call and the all-gather: the state of its shard, its launches, its allocations (the exchange buffers
 * themselves are reserved by tavb_comm_init, option "comm_reserve_keys", so the exchange path allocates nothing).  What it cannot cover -- a
 * peer that never makes the call, or dies inside it -- is bounded by "comm_timeout_ms": tavb_synchronize aborts the communicator and returns
 * TAVB_E_TIMEOUT instead of waiting for ever.
 * The collective is issued by the library itself, on the context's stream, behind the scan and
 * in front of the merge -- RCCL (librccl.so.1, resolved with dlopen at tavb_comm_init: no link-time dependency) is the only
 * communication layer; torch.distributed is not needed on the lookup path.
 *
 * tavb_comm_unique_id: rank 0 creates the 128-byte rendezvous id (ncclGetUniqueId) and hands it to the other ranks by any means
 * (a file, a socket, an existing torch.distributed / MPI broadcast).  tavb_comm_init: collective over all `world` ranks
 * (ncclCommInitRank on the context's device).  One communicator per context; tavb_destroy / tavb_comm_destroy release it. */
#define TAVB_COMM_ID_BYTES 128
int tavb_comm_unique_id(void* out_id /* TAVB_COMM_ID_BYTES */);
int tavb_comm_init(tavb_ctx* ctx, const void* id /* TAVB_COMM_ID_BYTES */, int32_t rank, int32_t world);
int tavb_comm_destroy(tavb_ctx* ctx);
/* Collective tavb_search_device: queries (device float32 [nq, dim], the same on every rank) -> out_keys [nq, k] = the merged
 * whole-corpus lists, on every rank.  out_keys may be device memory or pinned host memory the device can write (the merge kernel
 * writes it directly).  Asynchronous on the context's stream; tavb_synchronize before reading.  Every rank must call with the same
 * nq and k.  Without a communicator (or world == 1) this is tavb_search_device. */
int tavb_search_allgather(tavb_ctx* ctx, const float* dev_queries, int32_t nq, int32_t k, float min_score, tavb_key* out_keys);
/* The exchange on its own, for lists a rank produced by other means (its part of a subset search, vectorbase.py:203-230; its survivors
 * of a predicate, :191-201): dev_local_keys [nq, k] sorted lists with GLOBAL ordinals / positions -> out_keys [nq, k] merged over all
 * ranks (device or device-writable pinned memory).  Asynchronous; collective. */
int tavb_allgather_merge(tavb_ctx* ctx, const tavb_key* dev_local_keys, int32_t nq, int32_t k, tavb_key* out_keys);
/* keys [count] that carry POSITIONS into a list -> the same keys carrying dev_map[position] (int32 [map_len]): a rank's subset search
 * (tavb_search_subset_device) numbers the part of the caller's subset that lies in its shard; the map leads back to the positions in the
 * caller's whole list.  In place, asynchronous.  The map must be monotonic for the lists to stay sorted among equal scores. */
int tavb_remap_key_positions(tavb_ctx* ctx, tavb_key* dev_keys, int64_t count, const int32_t* dev_map, int64_t map_len);

/* ---- measurement ------------------------------------------------------------ */
/* HIP-event timing of the kernels this context launches, on the stream they run on.
 * kernel ids: 0 = streaming scan (dot + score + select), 1 = list merge,
 *             2 = MFMA batched scan (256-query tile: the last phase of the threshold ladder), 3 = normalise,
 *             4 = f32->f16 convert, 5 = the earlier phases of the ladder (either MFMA tile), 6 = 32-query MFMA tile
 *             (last phase), 7 = candidate rescoring of the 256-query tile, 8 = the all-gather of tavb_search_allgather. */
#define TAVB_KERNEL_SCAN 0
#define TAVB_KERNEL_MERGE 1
#define TAVB_KERNEL_MFMA 2
#define TAVB_KERNEL_NORMALIZE 3
#define TAVB_KERNEL_CONVERT 4
#define TAVB_KERNEL_MFMA_SAMPLE 5 /* threshold-seeding phases of the MFMA paths (all ladder phases but the last) */
#define TAVB_KERNEL_SKINNY 6 /* 32-query MFMA tile (small batches; every batch on fp32 corpora) */
#define TAVB_KERNEL_RESCORE 7 /* exact fp32-query rescoring of the 256-query tile's candidates (+ query preparation) */
#define TAVB_KERNEL_EXCHANGE 8 /* the RCCL all-gather of tavb_search_allgather (stream time between its two events: includes waiting for the slowest rank) */
#define TAVB_KERNEL_COUNT 9
int tavb_profile_enable(tavb_ctx* ctx, int32_t on);
int tavb_profile_reset(tavb_ctx* ctx);
int tavb_profile_read(tavb_ctx* ctx, int32_t kernel_id, double* out_total_ms, int64_t* out_launches);

/* How the library scans a corpus of `rows` rows for a batch of `nq` (>= 65) queries on a part with `n_cu` compute units, with default
 * options: the phase boundaries of the threshold ladder of the 128/256-query tile (phase i scans rows [out_bounds[i], out_bounds[i+1]);
 * at most `cap` entries are written, out_bounds may be NULL).  Returns the number of phases = tile-kernel launches per lookup (> 0), or
 * a negative error code.  No reference counterpart (the reference scans once, vectorbase.py:176); a pure function, needs no context and no
 * GPU -- what lets a committed profile be checked against the code that ships (tests/test_bench_contract.py). */
int tavb_plan_ladder(int64_t rows, int32_t nq, int32_t n_cu, int64_t* out_bounds, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* TAVB_H */
