// Exact fp32-query semantics for the 256-query MFMA tile.
//
// `fuzzy_lookup_embeddings(E)` must equal `[fuzzy_lookup_embedding(e) for e in E]` (vectorbase.py:163-190 per query):
// fp32 query x corpus row, fp32 accumulation.  The 256-query tile multiplies fp16 x fp16, so it sees the queries ROUNDED to
// fp16 -- up to ~2e-5 off in score, above the 1e-5 bar.  Instead of paying a second (low-plane) MFMA pass over the
// corpus, the tile is used as an exact FILTER:
//
//   1. query_prepare_kernel: q16 = fp16(q); delta_q = a rigorous bound on |score_fp16query(x) - score_fp32query(x)| over
//      every corpus row x:   |x.(q - q16)| <= ||x|| ||q - q16||   (Cauchy-Schwarz), ||x|| <= R = the largest row norm of
//      the corpus (corpus_max_norm_kernel, cached per corpus), plus the fp32 summation slack of the two dot products.
//   2. the tile + select_band_kernel (tavb_mfma.hip) keep, per query, the BAND: every row whose approximate score is within
//      2 delta_q of the approximate k-th best a_k (and above min_score - 2 delta_q) -- as many rows as the data puts there
//      (k + 2..3 on isotropic corpora, a whole cluster of near-duplicates on clustered ones), up to kBandMax.
//   3. rescore_kernel: the band is scored exactly (fp32 query, fp16 row widened, fp32 accumulate -- the arithmetic of the
//      streaming kernels), filtered by the exact min_score, and its best k by exact key are the answer.  Complete by
//      construction: a row outside the band has approx < a_k - 2 delta, so exact < a_k - delta <= the exact score of each of
//      the k rows that lead the approximate ranking -- it cannot be in the exact top-k, however tightly the scores around
//      rank k are packed.  (Round 2 kept a fixed 64 candidates and had to re-run every query whose 64 best approximate
//      scores sat inside the bound: a cliff on clustered corpora.)
//   4. only a band that did not fit (more than kBandMax rows, or more than a candidate buffer holds inside one row range, at
//      a level the final band reaches: select_band_kernel's verdict) makes a query incomplete; it is appended to a
//      device-side list; one fixed-shape launch of the 64-query tile with split hi/lo query planes (exact by construction,
//      tavb_mfma.hip) serves the list and returns at once when it is empty.  No host round trip anywhere: the asynchronous
//      device-resident form stays asynchronous.
//   The 32/64-query tile over an fp16 shadow (f32_shadow = 2) still hands over its best 64 by approximate score; there the
//   set is complete when it was not cut, or when rank 63 + delta is below the exact k-th best (rescore_kernel, cut mode).

//
// fp32 corpora (the reference's own layout) ride the same filter through an fp16 SHADOW copy of the corpus
// (shadow_convert_kernel: built once, extended on append, +50 % HBM; `f32_shadow` option): the bound gains the corpus
// rounding term,  |x.q - x16.q16| <= ||x - x16|| ||q|| + ||x16|| ||q - q16||  with E = max ||x - x16|| and R = max ||x16||
// over the rows, the candidates are rescored with the fp32 rows and fp32 queries, and flagged queries fall back to the
// exact fp32 64-query tile.  1024 queries over 1M x 1536 fp32 rows: one fp16 MFMA pass instead of 16 passes at the fp32
// matrix rate.

#include <hip/hip_runtime.h>

#include "tavb_device.h"
#include "tavb_internal.h"

namespace tavb {

namespace {

__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
  // non-negative floats (and +inf) order like their bit patterns
  atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// max over rows of sum(x^2), rows fp16; a row whose sum is not finite makes the result +inf (the bound is then
// useless and every query takes the exact path).  One wave per row, 16-byte loads.
__global__ void __launch_bounds__(256) corpus_max_norm_kernel(const _Float16* __restrict__ rows, int64_t n, int dim, float* __restrict__ out_sq) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int n8 = dim / 8;
  float best = 0.f;
  for (int64_t r = wave; r < n; r += n_waves) {
    const f16x8* x = reinterpret_cast<const f16x8*>(rows + r * (int64_t)dim);
    float ss = 0.f;
    for (int i = lane; i < n8; i += 64) {
      const f16x8 v = __builtin_nontemporal_load(x + i);
#pragma unroll
      for (int e = 0; e < 8; ++e) ss = fmaf((float)v[e], (float)v[e], ss);
    }
    ss = wave_sum(ss);
    if (!(ss < __builtin_inff())) ss = __builtin_inff();  // NaN / inf rows
    best = fmaxf(best, ss);
  }
  if (lane == 0) atomic_max_nonneg(out_sq, best);
}

// fp32 rows -> fp16 shadow rows, and the two maxima the bound needs: stats[0] = max sum(x16^2), stats[1] = max sum((x - x16)^2).
// A non-finite row makes the maxima +inf (every query then takes the exact path).  One wave per row.
__global__ void __launch_bounds__(256) shadow_convert_kernel(const float* __restrict__ rows, int64_t n, int dim, _Float16* __restrict__ out, int out_pitch,
                                                             float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int n4 = dim / 4;  // dim % 4 == 0 on this path; out_pitch (>= dim, a multiple of 64) = halves per shadow row, the tail zero
  float best16 = 0.f, best_err = 0.f;
  for (int64_t r = wave; r < n; r += n_waves) {
    const f32x4* x = reinterpret_cast<const f32x4*>(rows + r * (int64_t)dim);
    _Float16* y = out + r * (int64_t)out_pitch;
    for (int i = dim + lane; i < out_pitch; i += 64) y[i] = (_Float16)0.0f;
    float ss = 0.f, se = 0.f;
    for (int i = lane; i < n4; i += 64) {
      const f32x4 v = __builtin_nontemporal_load(x + i);
      const float e[4] = {v.x, v.y, v.z, v.w};
      _Float16 h[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = (_Float16)e[j];
        const float hf = (float)h[j];
        const float d = e[j] - hf;
        ss = fmaf(hf, hf, ss);
        se = fmaf(d, d, se);
      }
      *reinterpret_cast<uint2*>(y + i * 4) = *reinterpret_cast<const uint2*>(h);
    }
    ss = wave_sum(ss);
    se = wave_sum(se);
    if (!(ss < __builtin_inff())) ss = __builtin_inff();
    if (!(se < __builtin_inff())) se = __builtin_inff();
    best16 = fmaxf(best16, ss);
    best_err = fmaxf(best_err, se);
  }
  if (lane == 0) {
    atomic_max_nonneg(stats, best16);
    atomic_max_nonneg(stats + 1, best_err);
  }
}

// The prologue of a batched lookup on the wide path, ONE launch (until round 6: fill_thresholds + three memsets + this kernel; every launch of
// a few microseconds of work costs about five, and a lookup of a 1.25M-row shard is 3.5 ms).  One wave per query slot, nq_pad of them:
//   * a live query (slot < nq): q16 <- fp16(q); delta <- the score-units bound described above; thr <- the exclusive admission threshold of
//     the approximate pass (just below min_score - 2 delta; -inf when every row qualifies; +inf for NaN); band <- 2 delta;
//   * a padding slot: a zero query (it admits nothing: the tiles give it a threshold of +inf), band 0;
//   * every slot: band count, lost level and verdict zeroed (aux, [3][nq_pad] ints behind `band`); with `ms_fill` the callers' uniform
//     threshold goes into ms_out / ms_floor_out (+inf for the padding) -- a mixed batch arrives in `min_scores` instead;
//   * the first wave zeroes the 64 ints of the work list's header (flag64: flagged count, doomed count).
// rows_only: the filter multiplies the EXACT queries (split fp16 high + low planes) with the shadow rows: only the rows' rounding counts.
__global__ void __launch_bounds__(256) query_prepare_kernel(const float* __restrict__ q, int nq, int nq_pad, int dim, const float* __restrict__ min_scores /*[nq] unless ms_fill*/,
                                                            int rows_only, const float* __restrict__ max_norm_sq /*[2]: max |x16|^2, max |x - x16|^2*/,
                                                            _Float16* __restrict__ q16, float* __restrict__ delta, float* __restrict__ thr, float* __restrict__ band,
                                                            int frag_major, int* __restrict__ aux, int* __restrict__ flag64, int ms_fill, float ms_value, float ms_floor,
                                                            float* __restrict__ ms_out, float* __restrict__ ms_floor_out) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (qi == 0 && flag64 != nullptr) flag64[lane] = 0;
  if (qi >= nq_pad) return;
  const bool live = qi < nq;
  if (lane == 0) {
    if (aux != nullptr) {
      aux[qi] = 0;
      aux[nq_pad + qi] = 0;
      aux[2 * nq_pad + qi] = 0;
    }
    if (ms_fill) {
      ms_out[qi] = live ? ms_value : __builtin_inff();
      ms_floor_out[qi] = live ? ms_floor : __builtin_inff();
    }
  }
  const float* src = q + (size_t)qi * dim;
  _Float16* dst = q16 ? q16 + (size_t)qi * dim : nullptr;
  // fragment-major (the 256-query tile's direct query operand): 1 KiB per (query tile t, K step s of 64 halves, k16 slice c, 32-query block j);
  // in it lane l = (half-slice h) * 32 + (query row r & 31) holds halves 8 h .. 8 h + 7 of the slice
  const int steps = dim / 64;
  const int qt = qi >> 8, qr = qi & 255;
  float err = 0.f, qq = 0.f;
  auto one = [&](int i, float v, _Float16* h_out) {
    const _Float16 h = (_Float16)v;
    *h_out = h;
    const float d = v - (float)h;
    err = fmaf(d, d, err);
    qq = fmaf(v, v, qq);
  };
  if (dim % 4 == 0) {  // (always, on the routes that come here: four elements per lane per round, one 16-byte load, one 8-byte store)
    for (int i = lane * 4; i < dim; i += 256) {
      const f32x4 v = live ? *reinterpret_cast<const f32x4*>(src + i) : f32x4{0.f, 0.f, 0.f, 0.f};
      _Float16 h[4];
      one(i, v.x, &h[0]);
      one(i + 1, v.y, &h[1]);
      one(i + 2, v.z, &h[2]);
      one(i + 3, v.w, &h[3]);
      if (dst) {
        // (fragment-major: elements i .. i + 3 share K step, slice and half-slice -- four consecutive halves there too)
        const int ks = i >> 6, c = (i >> 4) & 3, hs = (i >> 3) & 1, e = i & 7;
        _Float16* out = frag_major ? q16 + ((((size_t)(qt * steps + ks) * 4 + c) * 8 + (qr >> 5)) * 512) + (size_t)((hs * 32 + (qr & 31)) * 8 + e) : dst + i;
        *reinterpret_cast<uint2*>(out) = *reinterpret_cast<const uint2*>(h);
      }
    }
  } else {
    for (int i = lane; i < dim; i += 64) {
      _Float16 h;
      one(i, live ? src[i] : 0.0f, &h);
      if (dst) {
        if (frag_major) {
          const int ks = i >> 6, c = (i >> 4) & 3, hs = (i >> 3) & 1, e = i & 7;
          q16[((((size_t)(qt * steps + ks) * 4 + c) * 8 + (qr >> 5)) * 512) + (size_t)((hs * 32 + (qr & 31)) * 8 + e)] = h;
        } else {
          dst[i] = h;
        }
      }
    }
  }
  if (!live) {
    if (lane == 0 && band) band[qi] = 0.0f;
    return;
  }
  err = wave_sum(err);
  qq = wave_sum(qq);
  if (lane == 0) {
    const float R = sqrtf(max_norm_sq[0]);
    const float E = sqrtf(max_norm_sq[1]);  // 0 for fp16 corpora (the rows ARE the fp16 values); the shadow's rounding for fp32 ones
    // rounding of the query (+ of the rows) + summation slack of two fp32 dot products (blocked accumulation: dim / 8 effective terms)
    float d = 0.5f * ((rows_only ? 0.0f : sqrtf(err) * R * 1.0001f) + E * sqrtf(qq) * 1.0001f + 2.0f * (float)(dim / 8 + 8) * 5.9604645e-8f * (R + E) * sqrtf(qq)) +
              1.2e-7f;
    if (!(d < __builtin_inff())) d = __builtin_inff();  // inf / NaN query or corpus: nothing can be proven
    delta[qi] = d;
    if (band) band[qi] = 2.0f * d;
    float t;
    const float min_score = ms_fill ? ms_value : min_scores[qi];
    if (min_score != min_score) {
      t = __builtin_inff();
    } else {
      const float lo = min_score - 2.0f * d;
      t = (lo > 0.0f) ? __uint_as_float(__float_as_uint(lo) - 1u) : -__builtin_inff();
    }
    thr[qi] = t;
  }
}

// One workgroup (4 waves) per query: exact scores of its candidates, exact threshold, best k sorted, completeness.
//   * band mode (cand_cnt != nullptr; the 128/256-query tile): cand_cnt[q] <= kBandMax unsorted candidates = every row whose
//     approximate score is within 2 delta of the approximate k-th best.  A row outside has approx < a_k - 2 delta, so its exact
//     score is < a_k - delta <= the exact score of each of the k rows that lead the approximate ranking: it cannot be in the
//     exact top k.  Complete by construction -- unless incomplete[q] (select_band_kernel's verdict) says that a band did not fit somewhere on the way at a level that matters.
//   * cut mode (cand_cnt == nullptr; the 32/64-query tile over an fp16 shadow): the best `stride` = 64 rows by approximate
//     score, sorted.  Fewer than 64: nothing was cut.  Else a row outside has approx <= a_63, exact <= a_63 + delta, and is
//     out when that is below the exact k-th best of the candidates (one delta: the exact k-th is known by now).
//   * SLOT mode (slot_query != nullptr): the workgroup serves slot blockIdx.x of the device-side work list of flagged queries (live when
//     slot < *slot_active and slot_min < *slot_active <= slot_max): the candidates at approx[slot] / cand_cnt[slot] are what an EXACT
//     tile (fp32 query as hi + lo fp16 planes, or fp32 MFMAs) ranked highest -- its best k and every row within a small band below the
//     k-th (tavb_abi.hip: kExactBand) -- and belong to query slot_query[slot].  Their scores come out of the matrix pipe's accumulation
//     order; scoring them again HERE gives them the streaming kernels' arithmetic, so that a query served by a fallback returns the
//     same float32 scores (and the same order among near-ties) as `fuzzy_lookup_embedding` on its own.  Nothing is flagged.
// KPL: result keys per lane (1: k <= 64, 4: k <= 256).
template <typename T, int KPL>
__global__ void __launch_bounds__(256) rescore_kernel(const T* __restrict__ corpus, int dim, uint32_t index_base,
                                                      const float* __restrict__ queries, const u64* __restrict__ approx /*[nq, stride]*/, int stride,
                                                      const int* __restrict__ cand_cnt, const int* __restrict__ incomplete,
                                                      const float* __restrict__ delta, const float* __restrict__ min_scores /*[nq]*/, int k,
                                                      u64* __restrict__ out /*[nq, k]*/, int* __restrict__ n_flagged,
                                                      int* __restrict__ flagged, const int* __restrict__ gate, int gate_max,
                                                      const int* __restrict__ slot_query, const int* __restrict__ slot_active, int slot_min, int slot_max) {
  __shared__ u64 exact[kBandMax];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int slot = blockIdx.x;
  int qi = slot;
  if (slot_query != nullptr) {
    const int live = *slot_active;
    if (live <= slot_min || live > slot_max || slot >= live) return;
    qi = slot_query[slot];
  } else if (gate != nullptr && *gate > gate_max) {  // the last filter phase did not run (most bands were not going to fit): no candidates, every query takes the exact form
    if (threadIdx.x == 0) flagged[atomicAdd(n_flagged, 1)] = qi;
    return;
  }
  const u64* cand = approx + (size_t)slot * stride;
  const float* q = queries + (size_t)qi * dim;
  const float min_score = min_scores[qi];
  const int n8 = dim / 8;
  int n_cand = cand_cnt ? cand_cnt[slot] : stride;
  if (n_cand > kBandMax) n_cand = kBandMax;
  // Two candidates per wave per round (c and c + 4): their row loads are in flight together -- a wave works through ~9 rows of a band one
  // after the other, each a dependent HBM round trip, and at a few dozen queries nothing else hides it.  Every row keeps its own fma chain in
  // its own order: the scores are those of the one-row loop bit for bit.
  const bool aligned = dim % (16 / (int)sizeof(T)) == 0;
  for (int c = wave; c < n_cand; c += 8) {
    const u64 key0 = cand[c];  // wave-uniform
    const u64 key1 = (c + 4 < n_cand) ? cand[c + 4] : 0ull;
    const uint32_t ord0 = 0xFFFFFFFFu - (uint32_t)key0, ord1 = 0xFFFFFFFFu - (uint32_t)key1;
    // (an empty slot reads the row of the candidate next to it: harmless, its result is dropped)
    const T* x0 = corpus + (size_t)((key0 != 0ull ? ord0 : ord1) - index_base) * dim;
    const T* x1 = corpus + (size_t)((key1 != 0ull ? ord1 : ord0) - index_base) * dim;
    float dot0 = 0.f, dot1 = 0.f;
    if (key0 == 0ull && key1 == 0ull) {
    } else if (!aligned) {
      // rows that are not 16-byte aligned (fp16 widths that are no multiple of 8): element loads in the order of the streaming kernel that
      // serves such widths (scan_scalar_kernel: lane l takes elements l, l + 64, ...), so a batch still is its sequential lookups bit for bit
      for (int e = lane; e < dim; e += 64) {
        const float qe = q[e];
        dot0 = fmaf((float)x0[e], qe, dot0);
        dot1 = fmaf((float)x1[e], qe, dot1);
      }
    } else if constexpr (sizeof(T) == 2) {
      const f16x8* r0 = reinterpret_cast<const f16x8*>(x0);
      const f16x8* r1 = reinterpret_cast<const f16x8*>(x1);
      for (int i = lane; i < n8; i += 64) {
        const f16x8 v = r0[i];
        const f16x8 w = r1[i];
        const f32x4 qa = *reinterpret_cast<const f32x4*>(q + i * 8);
        const f32x4 qb = *reinterpret_cast<const f32x4*>(q + i * 8 + 4);
        dot0 = fmaf((float)v[0], qa.x, dot0);
        dot0 = fmaf((float)v[1], qa.y, dot0);
        dot0 = fmaf((float)v[2], qa.z, dot0);
        dot0 = fmaf((float)v[3], qa.w, dot0);
        dot0 = fmaf((float)v[4], qb.x, dot0);
        dot0 = fmaf((float)v[5], qb.y, dot0);
        dot0 = fmaf((float)v[6], qb.z, dot0);
        dot0 = fmaf((float)v[7], qb.w, dot0);
        dot1 = fmaf((float)w[0], qa.x, dot1);
        dot1 = fmaf((float)w[1], qa.y, dot1);
        dot1 = fmaf((float)w[2], qa.z, dot1);
        dot1 = fmaf((float)w[3], qa.w, dot1);
        dot1 = fmaf((float)w[4], qb.x, dot1);
        dot1 = fmaf((float)w[5], qb.y, dot1);
        dot1 = fmaf((float)w[6], qb.z, dot1);
        dot1 = fmaf((float)w[7], qb.w, dot1);
      }
    } else {
      const f32x4* r0 = reinterpret_cast<const f32x4*>(x0);
      const f32x4* r1 = reinterpret_cast<const f32x4*>(x1);
      for (int i = lane; i < 2 * n8; i += 64) {
        const f32x4 v = r0[i];
        const f32x4 w = r1[i];
        const f32x4 qa = *reinterpret_cast<const f32x4*>(q + i * 4);
        dot0 = fmaf(v.x, qa.x, dot0);
        dot0 = fmaf(v.y, qa.y, dot0);
        dot0 = fmaf(v.z, qa.z, dot0);
        dot0 = fmaf(v.w, qa.w, dot0);
        dot1 = fmaf(w.x, qa.x, dot1);
        dot1 = fmaf(w.y, qa.y, dot1);
        dot1 = fmaf(w.z, qa.z, dot1);
        dot1 = fmaf(w.w, qa.w, dot1);
      }
    }
    dot0 = wave_sum(dot0);
    dot1 = wave_sum(dot1);
    const float s0 = cosine_to_score(dot0), s1 = cosine_to_score(dot1);
    // a NaN score ranks nowhere (numpy's >= drops it)
    if (lane == 0) {
      exact[c] = (key0 != 0ull && s0 == s0) ? make_key(s0, ord0) : 0ull;
      if (c + 4 < n_cand) exact[c + 4] = (key1 != 0ull && s1 == s1) ? make_key(s1, ord1) : 0ull;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  // best 64 * KPL of the exact keys, sorted best first (rank r in slot r / 64 of lane r % 64)
  WaveTopK<KPL> best;
  best.clear();
  for (int off = 0; off < n_cand; off += 64) {
    WaveTopK<KPL> chunk;
    chunk.clear();
    chunk.key[KPL - 1] = sort64_ascending((off + lane < n_cand) ? exact[off + lane] : 0ull, lane);  // ascending == "reversed best-first", zero-extended to 64 * KPL
    best.merge_reversed(chunk, lane);
  }
#pragma unroll
  for (int s_ = 0; s_ < KPL; ++s_) {
    const u64 mine = best.key[s_];
    const float my_score = __uint_as_float((uint32_t)(mine >> 32));
    const int r = s_ * 64 + lane;
    if (r < k) out[(size_t)qi * k + r] = (mine != 0ull && my_score >= min_score) ? mine : 0ull;  // sorted by score: the rows that fail are a tail
  }
  if (slot_query != nullptr) return;
  // completeness of the candidate set
  bool ok = true;
  if (cand_cnt != nullptr) {
    ok = (incomplete[qi] == 0);
  } else {
    const u64 a_last = cand[stride - 1];
    if (a_last != 0ull) {  // the set was cut
      const float a_cut = __uint_as_float((uint32_t)(a_last >> 32));
      const u64 kth = best.at(k - 1);
      const float e_k = __uint_as_float((uint32_t)(kth >> 32));
      ok = (kth != 0ull) && (a_cut + delta[qi] < e_k);  // false for delta = inf
    }
  }
  if (!ok && lane == 0) {
    const int slot_f = atomicAdd(n_flagged, 1);
    flagged[slot_f] = qi;
  }
}

// Compact the flagged queries into the operand of the exact 64-query tile: split hi / lo fp16 planes [2, cap, dim]
// (unused slots zero) and per-slot exclusive thresholds (+inf for unused slots: they admit nothing).
// seed (optional, [nq]) / delta: what the filter's selection has proven about a flagged query -- its cut, a level at least k rows reach by
// APPROXIMATE score -- is a valid admission threshold for the exact fallbacks once the filter's error bound is taken off (k rows score
// >= seed - delta exactly; a second delta and 1e-6 cover the exact tile's own arithmetic): the fallbacks start selective in ONE phase instead of
// climbing their own ladder (round 6: fifteen launches per batch that did nothing whenever the work list was empty -- the normal case).
// band_out (optional, [cap]): the band width of every slot of the wide exact form (fill_f32 until round 6: one more launch).
__device__ __forceinline__ float slot_threshold(bool used, float min_score, const float* seed, const float* delta, int qi) {
  if (!used || min_score != min_score) return __builtin_inff();
  float t = (min_score > 0.0f) ? __uint_as_float(__float_as_uint(min_score) - 1u) : -__builtin_inff();
  if (seed != nullptr) {
    const float s = seed[qi] - 2.0f * delta[qi] - 1e-6f;  // (-inf, or NaN from inf - inf: the comparison is false, nothing is seeded)
    if (s > t) t = s;
  }
  return t;
}

__global__ void __launch_bounds__(256) gather_flagged_kernel(const float* __restrict__ queries, int dim, const float* __restrict__ min_scores,
                                                             const int* __restrict__ n_flagged, const int* __restrict__ flagged, int cap,
                                                             _Float16* __restrict__ hi, _Float16* __restrict__ lo, float* __restrict__ thr,
                                                             const float* __restrict__ seed, const float* __restrict__ delta, float* __restrict__ band_out,
                                                             float band_v) {
  const int n = *n_flagged;
  if (n == 0) return;  // the common case: nothing to do, nothing written
  const int slot = blockIdx.x;
  if (slot >= cap) return;
  const bool used = slot < n;
  const int qi = used ? flagged[slot] : 0;
  if (threadIdx.x == 0) {
    thr[slot] = slot_threshold(used, used ? min_scores[qi] : 0.0f, seed, delta, qi);
    if (band_out != nullptr) band_out[slot] = band_v;
  }
  const float* src = used ? queries + (size_t)qi * dim : nullptr;
  for (int i = threadIdx.x; i < dim; i += blockDim.x) {
    _Float16 h = (_Float16)0.0f, l = (_Float16)0.0f;
    if (used) {
      const float v = src[i];
      h = (_Float16)v;
      const float hf = (float)h;
      l = (hf - hf == 0.0f) ? (_Float16)(v - hf) : (_Float16)0.0f;
    }
    hi[(size_t)slot * dim + i] = h;
    lo[(size_t)slot * dim + i] = l;
  }
}

// the same for the exact fp32 tile: plain fp32 queries [cap, dim]
__global__ void __launch_bounds__(256) gather_flagged_f32_kernel(const float* __restrict__ queries, int dim, const float* __restrict__ min_scores,
                                                                 const int* __restrict__ n_flagged, const int* __restrict__ flagged, int cap,
                                                                 float* __restrict__ out, float* __restrict__ thr, const float* __restrict__ seed,
                                                                 const float* __restrict__ delta) {
  const int n = *n_flagged;
  if (n == 0) return;
  const int slot = blockIdx.x;
  if (slot >= cap) return;
  const bool used = slot < n;
  const int qi = used ? flagged[slot] : 0;
  if (threadIdx.x == 0) thr[slot] = slot_threshold(used, used ? min_scores[qi] : 0.0f, seed, delta, qi);
  const float* src = used ? queries + (size_t)qi * dim : nullptr;
  for (int i = threadIdx.x; i < dim; i += blockDim.x) out[(size_t)slot * dim + i] = used ? src[i] : 0.0f;
}

}  // namespace

hipError_t launch_corpus_max_norm(const void* rows_f16, int64_t n, int dim, float* out_sq, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(corpus_max_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<const _Float16*>(rows_f16), n, dim, out_sq);
  return hipGetLastError();
}

hipError_t launch_shadow_convert(const float* rows_f32, int64_t n, int dim, void* out_f16, int out_pitch, float* stats, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(shadow_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, rows_f32, n, dim, reinterpret_cast<_Float16*>(out_f16), out_pitch, stats);
  return hipGetLastError();
}

hipError_t launch_query_prepare(const float* q, int nq, int nq_pad, int dim, const float* min_scores, bool rows_only, const float* max_norm_sq, void* q16,
                                float* delta, float* thr, float* band, hipStream_t stream, bool frag_major, int* aux, int* flag64, bool ms_fill, float ms_value,
                                float ms_floor, float* ms_out, float* ms_floor_out) {
  if (nq_pad < nq || (ms_fill && (!ms_out || !ms_floor_out)) || (!ms_fill && !min_scores)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(query_prepare_kernel, dim3((nq_pad + 3) / 4), dim3(256), 0, stream, q, nq, nq_pad, dim, min_scores, rows_only ? 1 : 0, max_norm_sq,
                     reinterpret_cast<_Float16*>(q16), delta, thr, band, frag_major ? 1 : 0, aux, flag64, ms_fill ? 1 : 0, ms_value, ms_floor, ms_out,
                     ms_floor_out);
  return hipGetLastError();
}

namespace {
template <typename T>
void launch_rescore_t(const void* corpus, int dim, uint32_t index_base, const float* queries, const unsigned long long* approx, int stride,
                      const int* cand_cnt, const int* incomplete, const float* delta, const float* min_scores, int n_blocks, int k, unsigned long long* out,
                      int* n_flagged, int* flagged, hipStream_t stream, const int* gate, int gate_max, const int* slot_query, const int* slot_active,
                      int slot_min, int slot_max) {
  if (k <= 64)
    hipLaunchKernelGGL((rescore_kernel<T, 1>), dim3(n_blocks), dim3(256), 0, stream, reinterpret_cast<const T*>(corpus), dim, index_base, queries, approx,
                       stride, cand_cnt, incomplete, delta, min_scores, k, out, n_flagged, flagged, gate, gate_max, slot_query, slot_active, slot_min, slot_max);
  else
    hipLaunchKernelGGL((rescore_kernel<T, 4>), dim3(n_blocks), dim3(256), 0, stream, reinterpret_cast<const T*>(corpus), dim, index_base, queries, approx,
                       stride, cand_cnt, incomplete, delta, min_scores, k, out, n_flagged, flagged, gate, gate_max, slot_query, slot_active, slot_min, slot_max);
}
}  // namespace

hipError_t launch_rescore(const void* corpus, bool f32_rows, int dim, uint32_t index_base, const float* queries, const unsigned long long* approx,
                          int stride, const int* cand_cnt, const int* incomplete, const float* delta, const float* min_scores, int nq, int k,
                          unsigned long long* out, int* n_flagged, int* flagged, hipStream_t stream, const int* gate, int gate_max) {
  if (stride < 1 || stride > kBandMax || k < 1 || k > TAVB_MAX_FUSED_K || (cand_cnt != nullptr && incomplete == nullptr) || !min_scores) return hipErrorInvalidValue;
  if (cand_cnt == nullptr && k > 64) return hipErrorInvalidValue;  // cut mode hands over 64 candidates
  // (*n_flagged is zeroed by the caller: query_prepare_kernel's flag64)
  if (f32_rows)
    launch_rescore_t<float>(corpus, dim, index_base, queries, approx, stride, cand_cnt, incomplete, delta, min_scores, nq, k, out, n_flagged, flagged, stream,
                            gate, gate_max, nullptr, nullptr, 0, 0);
  else
    launch_rescore_t<_Float16>(corpus, dim, index_base, queries, approx, stride, cand_cnt, incomplete, delta, min_scores, nq, k, out, n_flagged, flagged,
                               stream, gate, gate_max, nullptr, nullptr, 0, 0);
  return hipGetLastError();
}

hipError_t launch_rescore_slots(const void* corpus, bool f32_rows, int dim, uint32_t index_base, const float* queries, const unsigned long long* cand,
                                int stride, const int* cand_cnt, const float* min_scores, int n_slots, int k, unsigned long long* out,
                                const int* slot_query, const int* slot_active, int slot_min, int slot_max, hipStream_t stream) {
  if (stride < 1 || stride > kBandMax || k < 1 || k > TAVB_MAX_FUSED_K || n_slots < 1 || !slot_query || !slot_active || !min_scores) return hipErrorInvalidValue;
  if (f32_rows)
    launch_rescore_t<float>(corpus, dim, index_base, queries, cand, stride, cand_cnt, nullptr, nullptr, min_scores, n_slots, k, out, nullptr, nullptr, stream,
                            nullptr, 0, slot_query, slot_active, slot_min, slot_max);
  else
    launch_rescore_t<_Float16>(corpus, dim, index_base, queries, cand, stride, cand_cnt, nullptr, nullptr, min_scores, n_slots, k, out, nullptr, nullptr, stream,
                               nullptr, 0, slot_query, slot_active, slot_min, slot_max);
  return hipGetLastError();
}

hipError_t launch_gather_flagged_f32(const float* queries, int dim, const float* min_scores, const int* n_flagged, const int* flagged, int cap, float* out,
                                     float* thr, const float* seed, const float* delta, hipStream_t stream) {
  hipLaunchKernelGGL(gather_flagged_f32_kernel, dim3(cap), dim3(256), 0, stream, queries, dim, min_scores, n_flagged, flagged, cap, out, thr, seed, delta);
  return hipGetLastError();
}

hipError_t launch_gather_flagged(const float* queries, int dim, const float* min_scores, const int* n_flagged, const int* flagged, int cap, void* hi, void* lo,
                                 float* thr, const float* seed, const float* delta, float* band_out, float band_v, hipStream_t stream) {
  hipLaunchKernelGGL(gather_flagged_kernel, dim3(cap), dim3(256), 0, stream, queries, dim, min_scores, n_flagged, flagged, cap,
                     reinterpret_cast<_Float16*>(hi), reinterpret_cast<_Float16*>(lo), thr, seed, delta, band_out, band_v);
  return hipGetLastError();
}

}  // namespace tavb
