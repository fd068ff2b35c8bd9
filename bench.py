#!/usr/bin/env python3
"""Benchmark of the VectorBase kNN hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak] [--workload NAME]

One JSON line on stdout (rank 0).  Without --workload this is the north-star suite:

  headline   cfg3 = BASELINE.json configs[2]: 10M x 1536 fp16 corpus, ONE 1024-query batch per step, top-32,
             through the C-ABI call `tavb_search_device` + `tavb_synchronize`: the query batches (four in rotation) are RESIDENT
             in HBM when the timed region starts, the step is the MFMA scan over the whole corpus, candidate selection and
             rescoring, and the last kernel writing the result keys into pinned host memory.  `value` = queries/s over exactly
             K timed steps.  The same batch handed over as a HOST buffer (`tavb_search_batch`: query H2D included -- what
             `VectorBase.fuzzy_lookup_embeddings` costs) is reported beside it as `host_buffer_form`, never as `value`.
             `roofline` = 2*Q*N*D flops per batch / the summed duration of the MFMA scan launches of one batch (HIP events on
             the library's stream) against the 2.5 PFLOP/s dense fp16 MFMA peak; `roofline.sclk_mhz` / `power_w` = the
             board's clock and power sampled in-process (amdgpu hwmon) while those launches ran, `frac_at_clock` = the same
             fraction against the peak at THAT clock (2500 x sclk / 2400).
  sub.cfg3_q1  the same corpus, ONE query per step (north star's single-query target: HBM-bound, 30.72 GB/query)
  sub.cfg2     configs[1]: 1M x 1536 fp32 corpus, one query per step (HBM-bound, 6.144 GB/query)

  Every record carries `roofline` (kernel, mean launch duration, algorithmic work, counter traffic from the committed
  rocprofv3 --pmc pass), a `parity` object (the GPU answers for >= 16 sampled queries checked against the CPU oracle
  over the WHOLE corpus, delivered to the oracle in 1M-row chunks) and, at N = 1, `cpu_baseline` (the reference's own
  VectorBase -- verbatim when /root/reference exists, else its numpy port -- timed on this host's cores).

--gpus N > 1: one process per GPU over RCCL.  When WORLD_SIZE is not set the script re-launches itself under
  `python -m torch.distributed.run --nproc-per-node N`; when it is set it must equal N.  The corpus is row-sharded
  (contiguous ranges), every rank scans its shard for every query, the per-shard top-k key lists are all-gathered and
  merged on every rank.  `value` is always the end-to-end rate of global lookups (queries/s).
    --scaling strong (default): the SAME 10M-row corpus split over the N GPUs (total work fixed).
    --scaling weak: cfg4 = configs[3], 12.5M rows PER GPU (100M rows at N = 8); `row_queries_per_sec` is reported too.

Synthetic data: gaussian rows generated on the device per 262144-row chunk (seeded per chunk, so any rank can
reproduce any row range), L2-normalised by our K1 kernel, rounded to fp16 by our convert kernel.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

_PROCESS_T0 = time.perf_counter()  # (the suite's time budget counts from here: --budget-seconds)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
BASELINE_METRIC = "queries/sec + p50 lookup latency, 1536-d top-32 kNN at 1/2/4/8 MI355X"  # BASELINE.json `metric`, verbatim
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32 matrix rate (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md peak table
CHUNK_ROWS = 262_144           # generation granule of the synthetic corpus
ORACLE_CHUNK = 1_000_000       # rows handed to the CPU oracle at a time (a reference-sized VectorBase)
BATCH_ROTATION = 4             # different query batches that take turns in the timed region of a batched workload
PARITY_QUERIES = 16

WORKLOADS = {
    #            rows (total)  dim   dtype   queries/step  k   bound   seed
    "cfg2": dict(rows=1_000_000, dim=1536, dtype="fp32", nq=1, k=32, bound="hbm", seed=1043),
    "cfg2_f16": dict(rows=1_000_000, dim=1536, dtype="fp16", nq=1, k=32, bound="hbm", seed=1043),
    "cfg3": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=1024, k=32, bound="mfma", seed=10043),
    # the same shape on a CLUSTERED corpus (100-row clusters of near-duplicates incl. exact duplicates; every query sits next to a cluster
    # centre, so its top-64 spans < 2e-4 in score -- inside the fp16 filter's error bound): what real embedding corpora do to the wide tile
    "cfg3_clustered": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=1024, k=32, bound="mfma", seed=10143, kind="clustered"),
    # the duplication cliff: clusters of 1500 near-duplicates -- more than a band (1024 candidates) holds, so EVERY query is flagged and re-run
    # exactly (the 256-query tile's split-plane form: twice the MFMAs of a filter pass on top of the filter pass itself)
    "cfg3_dup": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=1024, k=32, bound="mfma", seed=10243, kind="clustered", cluster_rows=1500),
    # the north star's single-query target on the cfg3 corpus: HBM-bound, 30.72 GB per query
    "cfg3_q1": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=1, k=32, bound="hbm", seed=10043),
    "cfg4": dict(rows=12_500_000, dim=1536, dtype="fp16", nq=1024, k=32, bound="mfma", seed=100043),  # PER GPU (weak scaling)
    "cfg1": dict(rows=10_000, dim=1536, dtype="fp32", nq=1, k=10, bound="latency", seed=43),
    # batches on the reference's own dtype (fp32): 32 queries ride one HBM pass; 1024 are bound by the fp32 matrix rate
    "cfg2_b32": dict(rows=1_000_000, dim=1536, dtype="fp32", nq=32, k=32, bound="hbm", seed=1043),
    "cfg2_b1024": dict(rows=1_000_000, dim=1536, dtype="fp32", nq=1024, k=32, bound="mfma", seed=1043),  # fp16 shadow + exact fp32 rescoring (--opt f32_shadow=0: fp32 MFMAs)
    # middle batch sizes on the fp16 corpus (the 64/128-query tiles)
    "cfg3_b32": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=32, k=32, bound="hbm", seed=10043),
    "cfg3_b128": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=128, k=32, bound="hbm", seed=10043),  # 128-query tile: one HBM pass (3.9 PFLOP would take 3.2 ms at 0.49 of the matrix peak)
    # fused multi-index user query
    "cfg5": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=6, k=50, bound="hbm", seed=50043),
    # the other widths of the reference's model table (vectorbase.py:31-35: text-embedding-3-large = 3072) and of its benchmark script
    # (tools/benchmark_vectorbase.py:55-76: --dim 384), same bytes per corpus as cfg2 / cfg3 x 2 / x 1
    "cfg2_d3072": dict(rows=1_000_000, dim=3072, dtype="fp32", nq=1, k=32, bound="hbm", seed=2043),
    "cfg3_d3072_q1": dict(rows=5_000_000, dim=3072, dtype="fp16", nq=1, k=32, bound="hbm", seed=30043),
    "cfg3_d3072": dict(rows=5_000_000, dim=3072, dtype="fp16", nq=1024, k=32, bound="mfma", seed=30043),
    "cfg1_d384": dict(rows=10_000, dim=384, dtype="fp32", nq=1, k=10, bound="latency", seed=43),
    "cfg1_1k_d384": dict(rows=1_000, dim=384, dtype="fp32", nq=1, k=10, bound="latency", seed=42),  # that script's FIRST row (1k vectors): where a GPU lookup cannot win
    # fuzzy_lookup_embedding_in_subset (vectorbase.py:203-230): the reference script's third row (1000 of 10k, subset seed 99,
    # tools/benchmark_vectorbase.py:133-163) and a subset at bench scale (1M random ordinals of the cfg3 corpus: S * D * 2 + S * 4 bytes)
    "cfg1_subset": dict(rows=10_000, dim=1536, dtype="fp32", nq=1, k=10, bound="latency", seed=43, subset=1000),
    # the reference's batched call site at its own scale: `lookup_terms` (storage/memory/reltermsindex.py:320-332: a loop of single lookups) with 32
    # terms over a term index of 1294 rows (the shape of its one real fixture), k = 50 @ 0.85 (knowpro/convsettings.py:61-63), on the
    # real-embedding-like corpus (most rows survive 0.85): ONE grouped streaming launch since the end of round 6 (profiles/r06_group_sweep.md)
    "cfg1_terms32": dict(rows=1294, dim=1536, dtype="fp32", nq=32, k=50, bound="latency", seed=44, kind="aniso", min_score=0.85),
    "cfg3_subset": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=1, k=32, bound="hbm", seed=10043, subset=1_000_000),
    # a real-embedding-like corpus: rows = normalise(g + c * mu), mean pairwise cosine 0.75 (scores 0.875 +- 0.013), at the reference's
    # related-terms threshold 0.85 (knowpro/convsettings.py:61-63; vectorbase.py:16-35) -- MOST rows survive min_score, the regime the
    # reference's defaults live in (gaussian rows at 0.85: none does)
    "cfg3_aniso": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=1024, k=32, bound="mfma", seed=10343, kind="aniso", min_score=0.85),
    "cfg3_aniso_q1": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=1, k=32, bound="hbm", seed=10343, kind="aniso", min_score=0.85),
}
ANISO_C = float(np.sqrt(3.0))  # |c * mu| over |g|: cosine between two rows = c^2 / (1 + c^2) = 0.75


# ----------------------------------------------------------------------------------------------------------------
# synthetic corpus: reproducible per chunk, generated where it is used
# ----------------------------------------------------------------------------------------------------------------
CLUSTER_ROWS = 100      # rows per cluster of the clustered corpus
CLUSTER_SPREAD = 0.002  # |row - centre| before normalisation: scores inside a cluster spread over ~2e-4
CLUSTER_MULT = 7_368_787  # row -> cluster hash (a prime): the rows of a cluster are scattered over the whole corpus


def cluster_centres(eng, rows_total: int, dim: int, seed: int, cluster_rows: int = CLUSTER_ROWS):
    """[rows_total // cluster_rows, dim] fp32 unit vectors on the device (same on every rank: one seeded torch generator)."""
    import torch

    dev = torch.device("cuda", eng.device)
    n_c = max(1, rows_total // cluster_rows)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed * 1_000_003 + 999_983)
    c = torch.empty((n_c, dim), dtype=torch.float32, device=dev)
    c.normal_(generator=gen)
    eng.normalize_rows_(c)
    return c


def aniso_direction(eng, dim: int, seed: int):
    """the common direction mu of the anisotropic corpus `seed`: a unit vector [dim] on the device (same on every rank)"""
    import torch

    dev = torch.device("cuda", eng.device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed * 1_000_003 + 999_979)
    mu = torch.empty((1, dim), dtype=torch.float32, device=dev)
    mu.normal_(generator=gen)
    eng.normalize_rows_(mu)
    return mu


def aniso_queries(eng, count: int, dim: int, seed: int) -> np.ndarray:
    """`count` unit queries from the distribution of the anisotropic corpus' rows (what a model that embeds queries and rows alike gives)"""
    mu = aniso_direction(eng, dim, seed).cpu().numpy()
    rng = np.random.default_rng(seed + 23)
    q = rng.standard_normal((count, dim)).astype(np.float32) + np.float32(ANISO_C * np.sqrt(dim)) * mu
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def gen_rows(eng, lo: int, hi: int, dim: int, seed: int, dtype: str, kind: str = "gaussian", rows_total: int | None = None, cluster_rows: int = CLUSTER_ROWS):
    """Rows [lo, hi) of the synthetic corpus `seed` as a device tensor (fp32 or fp16).  Chunk c (CHUNK_ROWS rows) is
    torch.randn with generator seed `seed * 1000003 + c`, L2-normalised by our K1 kernel and (fp16) rounded by our
    convert kernel, so every rank / the parity checker reproduce the same bytes for any row range.
    kind = "clustered": row i belongs to cluster (i * CLUSTER_MULT) % n_clusters and is centre + CLUSTER_SPREAD * noise / sqrt(dim),
    normalised; every row with i % 8 == 5 takes the NEXT cluster's centre as its noise instead -- all such rows of a cluster are
    exact duplicates of one another (about a dozen per cluster).
    kind = "aniso": row = normalise(g + ANISO_C * |g| * mu), one direction mu for the whole corpus (mean pairwise cosine 0.75)."""
    import torch

    dev = torch.device("cuda", eng.device)
    out = torch.empty((hi - lo, dim), dtype=torch.float16 if dtype == "fp16" else torch.float32, device=dev)
    gen = torch.Generator(device=dev)
    centres = cluster_centres(eng, rows_total if rows_total is not None else hi, dim, seed, cluster_rows) if kind == "clustered" else None
    mu = aniso_direction(eng, dim, seed) * float(ANISO_C * np.sqrt(dim)) if kind == "aniso" else None
    c = lo // CHUNK_ROWS
    while c * CHUNK_ROWS < hi:
        c_lo = c * CHUNK_ROWS
        gen.manual_seed(seed * 1_000_003 + c)
        tmp = torch.empty((CHUNK_ROWS, dim), dtype=torch.float32, device=dev)
        tmp.normal_(generator=gen)
        if centres is not None:
            n_c = centres.shape[0]
            ids = torch.arange(c_lo, c_lo + CHUNK_ROWS, dtype=torch.int64, device=dev)
            cl = (ids * CLUSTER_MULT) % n_c
            tmp.mul_(CLUSTER_SPREAD / float(np.sqrt(dim)))
            dup = (ids % 8) == 5
            tmp[dup] = centres[(cl[dup] + 1) % n_c] * CLUSTER_SPREAD
            tmp.add_(centres[cl])
            del ids, cl, dup
        if mu is not None:
            tmp.add_(mu)
        a, b = max(lo, c_lo), min(hi, c_lo + CHUNK_ROWS)
        part = tmp[a - c_lo : b - c_lo]
        eng.normalize_rows_(part)
        out[a - lo : b - lo].copy_(eng.to_f16(part) if dtype == "fp16" else part)
        del tmp
        c += 1
    torch.cuda.synchronize(dev)
    return out


def clustered_queries(eng, count: int, rows_total: int, dim: int, seed: int, cluster_rows: int = CLUSTER_ROWS) -> np.ndarray:
    """`count` unit queries next to cluster centres of the clustered corpus `seed` (centre + 0.05 * noise: cosine ~0.9988 to the centre)."""
    centres = cluster_centres(eng, rows_total, dim, seed, cluster_rows)
    rng = np.random.default_rng(seed + 17)
    pick = rng.choice(centres.shape[0], size=count, replace=centres.shape[0] < count)
    q = centres[np.sort(pick)].cpu().numpy() + 0.05 * rng.standard_normal((count, dim)).astype(np.float32) / np.sqrt(dim)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def make_device_corpus(eng, rows: int, dim: int, seed: int, dtype: str):
    """Whole corpus on one device (tests import this)."""
    return gen_rows(eng, 0, rows, dim, seed, dtype)


def host_queries(count: int, dim: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((count, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


# ----------------------------------------------------------------------------------------------------------------
# CPU legs: parity oracle and the reported baseline
# ----------------------------------------------------------------------------------------------------------------
def oracle_chunks(eng, resident, resident_lo: int, total_rows: int, dim: int, seed: int, dtype: str, kind: str = "gaussian", cluster_rows: int = CLUSTER_ROWS):
    """The whole corpus as float32 host chunks of ORACLE_CHUNK rows (fp16 values widened, as the kernels see them):
    rows this rank holds come from the resident tensor, the rest are regenerated on the device."""
    for lo in range(0, total_rows, ORACLE_CHUNK):
        hi = min(total_rows, lo + ORACLE_CHUNK)
        if resident is not None and lo >= resident_lo and hi <= resident_lo + resident.shape[0]:
            t = resident[lo - resident_lo : hi - resident_lo]
        else:
            t = gen_rows(eng, lo, hi, dim, seed, dtype, kind, total_rows, cluster_rows)
        yield t.float().cpu().numpy()


class ParityTally:
    """Sums of the per-query reports of `check_topk_parity_large` (with its float64 referee) for one `parity` object."""

    def __init__(self):
        self.exact = self.permuted = self.near = self.inv_gpu = self.inv_ref = 0
        self.worst = self.gap = self.noise_ref = self.noise_gpu = self.width = self.inv_gap_gpu = self.inv_gap_ref = 0.0

    def add(self, rep, n_near: int) -> None:
        self.exact += rep.exact_positions
        self.permuted += rep.tie_permuted_positions
        self.near += n_near
        self.inv_gpu += rep.gpu_inversions_vs_f64
        self.inv_ref += rep.reference_inversions_vs_f64
        self.gap = max(self.gap, rep.max_permuted_gap)
        self.noise_ref = max(self.noise_ref, rep.noise_ref or 0.0)
        self.noise_gpu = max(self.noise_gpu, rep.noise_gpu or 0.0)
        self.width = max(self.width, rep.tie_width or 0.0)
        self.inv_gap_gpu = max(self.inv_gap_gpu, rep.max_inverted_gap_gpu)
        self.inv_gap_ref = max(self.inv_gap_ref, rep.max_inverted_gap_ref)

    def fields(self) -> dict:
        # near-tie rule: oracle/vectorbase_oracle.py::check_topk_parity (float64 referee; nothing hand-set) -- profiles/README.md spells the fields out
        return {
            "positions_exact": self.exact,
            "positions_permuted": self.permuted,
            "max_permuted_gap": self.gap,
            "gpu_inversions_vs_f64": self.inv_gpu,
            "reference_inversions_vs_f64": self.inv_ref,
            "max_inverted_gap_gpu": self.inv_gap_gpu,
            "max_inverted_gap_ref": self.inv_gap_ref,
            "noise_gpu": self.noise_gpu,
            "noise_ref": self.noise_ref,
            "tie_width": self.width,
            "near_tie_pairs": self.near,
            "max_abs_score_error": self.worst,
        }


def parity_check(eng, resident, resident_lo, wl, queries: np.ndarray, sample: list[int], got: dict, min_score: float, subset=None) -> dict:
    """got[qi] = (ordinals, scores) from the GPU path.  Oracle = numpy restatement of vectorbase.py:163-190 over the
    whole corpus (oracle/vectorbase_oracle.py) in ORACLE_CHUNK-row chunks; near-ties are decided by a float64 referee computed in the
    same pass (the reference's best k + 256 rows of every chunk and the returned rows)."""
    from oracle import vectorbase_oracle as vo

    t0 = time.perf_counter()
    ref, referee = vo.scores_full_chunked_refereed(
        oracle_chunks(eng, resident, resident_lo, wl["rows_total"], wl["dim"], wl["seed"], wl["dtype"], wl.get("kind", "gaussian"), wl.get("cluster_rows", CLUSTER_ROWS)),
        queries[sample], [got[qi][0] for qi in sample], keep=wl["k"] + 256, restrict=subset)
    tally = ParityTally()
    pos_of = {int(o): i for i, o in enumerate(subset)} if subset is not None else None
    try:
        for j, qi in enumerate(sample):
            o, s = got[qi]
            if subset is not None:  # vectorbase.py:217-227: the scores of the subset's rows, ranked among themselves
                truth = referee.for_query(j)

                def sub_truth(p_, t=truth):
                    return t(subset[np.asarray(p_)])
                sub_truth.dim = wl["dim"]
                rep, n_near = vo.check_topk_parity_large(ref[j][subset], [pos_of[int(x)] for x in o], s.tolist(), wl["k"], min_score, referee=sub_truth)
            else:
                rep, n_near = vo.check_topk_parity_large(ref[j], o.tolist(), s.tolist(), wl["k"], min_score, referee=referee.for_query(j))
            tally.add(rep, n_near)
            tally.worst = max(tally.worst, float(np.max(np.abs(ref[j][o] - s))) if len(o) else 0.0)
    except AssertionError as exc:
        return {"ok": False, "error": str(exc)[:300], "queries_checked": len(sample), "rows": wl["rows_total"]}
    return {"ok": True, "queries_checked": len(sample), "rows": wl["rows_total"], **tally.fields(), "seconds": round(time.perf_counter() - t0, 1)}


def cpu_baseline(host: np.ndarray, queries: np.ndarray, k: int, rows_total: int, nq: int, budget_s: float, subset=None) -> dict:
    """The reference's VectorBase.fuzzy_lookup_embedding (`subset`: fuzzy_lookup_embedding_in_subset with that list) on this host's cores,
    same corpus bytes (first rows of it), sequential single-query calls (the reference has no batch entry point:
    storage/memory/reltermsindex.py:320-332).  Verbatim class when /root/reference is present (build container), else its numpy port
    (oracle/vectorbase_oracle.py -- pinned to the verbatim class by tests/test_oracle_vs_reference.py and the committed goldens)."""
    from oracle import ref_loader
    from oracle import vectorbase_oracle as vo

    kind = "port"
    sub_list = None if subset is None else [int(x) for x in subset]  # a Python list, as tools/benchmark_vectorbase.py:136 passes it
    if ref_loader.reference_available():
        vb = ref_loader.make_reference_vectorbase(host)
        kind = "reference"

        def one(q):
            if sub_list is not None:
                return vb.fuzzy_lookup_embedding_in_subset(q, sub_list, max_hits=k, min_score=0.0)
            return vb.fuzzy_lookup_embedding(q, max_hits=k, min_score=0.0)
    else:
        def one(q):
            if sub_list is not None:
                return vo.lookup_in_subset(host, q, sub_list, k, 0.0)
            return vo.lookup(host, q, k, 0.0)

    cores = len(os.sched_getaffinity(0))
    try:
        from threadpoolctl import threadpool_limits
    except Exception:  # pragma: no cover - threadpoolctl is part of the image
        threadpool_limits = None

    def timed(n_calls: int, first: int = 0) -> list[float]:
        out = []
        for j in range(n_calls):
            t0 = time.perf_counter_ns()
            one(queries[(first + j) % len(queries)])
            out.append((time.perf_counter_ns() - t0) / 1e9)
        return out

    # 20 warm-up calls (BLAS thread pool, page-in of the matrix), then the timed rounds; default threads = all host cores AND
    # the best thread count of a sweep (OpenBLAS sgemv on a few-GB matrix is memory-bound: 256 threads oversubscribe it)
    t_end = time.perf_counter() + budget_s
    warm = timed(20)
    per_call = float(np.median(warm))
    sweep = {}
    best_threads, best_med = cores, None
    if threadpool_limits is not None:
        for th in sorted({1, 4, 8, 16, 32, 64, 128, cores}):
            if th > cores:
                continue
            if th == 1 and per_call * 20 > budget_s:  # (one thread on 1M rows: ~0.6 s per call; measured below with 3 calls)
                continue
            with threadpool_limits(limits=th, user_api="blas"):
                one(queries[0])
                med = float(np.median(timed(3, 1)))
            sweep[th] = med
            if best_med is None or med < best_med:
                best_threads, best_med = th, med
    times_default = timed(max(3, min(200, int(max(0.0, t_end - time.perf_counter()) / 2 / max(per_call, 1e-6)))), 4)
    if threadpool_limits is not None and best_threads != cores:
        with threadpool_limits(limits=best_threads, user_api="blas"):
            one(queries[0])
            times = timed(max(10, min(200, int(max(0.0, t_end - time.perf_counter()) / max(best_med, 1e-6)))), 4)
    else:
        times = times_default
    med = float(np.median(times))
    med_default = float(np.median(times_default))
    scale = rows_total / host.shape[0]
    out = {
        "value": 1.0 / (med * scale),
        "unit": "queries/s",
        "cores": best_threads,
        "host_cores": cores,
        "kind": kind,
        # (what the fields mean, the warm-up and the sweep: profiles/README.md "bench line")
        "sample": f"{len(times)} calls on {host.shape[0]}x{host.shape[1]} fp32 rows of the corpus, median {med * 1e3:.2f} ms"
                  + (f", x{scale:g} to {rows_total} rows" if scale != 1 else "")
                  + (f"; subset of {len(sub_list)} ordinals" if sub_list is not None else "")
                  + ("; numpy port of the reference class (no /root/reference on this box), pinned to the verbatim class by the committed goldens" if kind == "port" else ""),
        "p50_ms_per_query_on_sample": med * 1e3,
        # `value` = the CPU's best: the thread count (`cores`) that won a BLAS thread sweep.  What the reference gets out of the box --
        # OpenBLAS's default of one thread per host core, SURVEY 8d's stated baseline -- is `default_threads` (oversubscribed on a 256-core host)
        "value_is": "best thread count of a BLAS thread sweep; default_threads = OpenBLAS default (all host cores)",
        "default_threads": {"value": 1.0 / (med_default * scale), "cores": cores},
    }
    if 1 in sweep:
        out["one_thread"] = {"value": 1.0 / (sweep[1] * scale)}
    elif threadpool_limits is not None:
        with threadpool_limits(limits=1, user_api="blas"):
            t1 = timed(3)
        out["one_thread"] = {"value": 1.0 / (float(np.median(t1)) * scale)}
    return out


# ----------------------------------------------------------------------------------------------------------------
# one workload on the resident corpus
# ----------------------------------------------------------------------------------------------------------------
class Ctx:
    """Process-wide state: rank layout, engine, optional sharded searcher."""

    def __init__(self, args):
        import torch

        self.args = args
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.distributed = self.world > 1 or os.environ.get("TAVB_BENCH_FORCE_DIST") == "1"  # the latter: 1-rank dry run of the N>1 code
        # TAVB_BENCH_DIST_BACKEND=gloo: the DRY RUN of this script's N > 1 branches on a box with ONE GPU -- every rank on cuda:0, torch.distributed
        # over gloo, the exchange of the lookups through the searcher's `gather_fn` hook (RCCL refuses two ranks on one device).  Exercises
        # what the first real N > 1 run executes for the first time otherwise: `rank != 0`, the per-rank timing gather, rank 0 regenerating the
        # other ranks' rows for the whole-corpus parity check, `sub.cfg4_weak`.  Its rates mean nothing (two ranks share one GPU).
        self.dist_backend = os.environ.get("TAVB_BENCH_DIST_BACKEND", "nccl")
        self.dry_run = self.dist_backend != "nccl"
        self.dist = None
        if self.distributed:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                # only the launcher-less one-rank dry run gets here (TAVB_BENCH_FORCE_DIST=1): a free port of this host, so that two benches
                # on one node cannot collide (a launcher -- the driver's torch.distributed.run, or respawn_under_torchrun -- hands the port down)
                if self.world > 1:
                    raise SystemExit("bench.py: WORLD_SIZE > 1 without MASTER_PORT (launch through torch.distributed.run, or `python bench.py --gpus N`)")
                os.environ["MASTER_PORT"] = str(free_port())
            for var, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):  # (TAVB_BENCH_FORCE_DIST=1 without a launcher)
                os.environ.setdefault(var, val)
            if self.dry_run:
                self.local_rank = self.local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(self.local_rank)
            import datetime

            # (rank 0 checks parity against the CPU oracle over the WHOLE corpus -- 100M rows at N = 8 take minutes -- while the others wait in a barrier)
            if self.dry_run:
                dist.init_process_group(self.dist_backend, timeout=datetime.timedelta(minutes=45))
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank), timeout=datetime.timedelta(minutes=45))
            self.dist = dist
            if dist.get_world_size() != args.gpus:
                raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
        self.dev = self.local_rank if self.distributed else 0
        torch.cuda.set_device(self.dev)
        from typeagent_py_amd import _native

        self.native = _native
        if self.distributed:
            from typeagent_py_amd.sharded import DeviceShardBackend

            self.backend = DeviceShardBackend(self.dev)
            self.eng = self.backend.engine
            # the exchange of the lookup path is libtavb's own RCCL communicator (tavb_search_allgather); torch.distributed only carries
            # the rendezvous id, the barriers and the max-over-ranks of the timing
            if not self.dry_run:
                self.backend.init_comm(dist.get_rank(), dist.get_world_size())
                if dist.get_world_size() == 1:
                    self.eng.set_option("comm_force", 1)  # 1-rank dry run of the N > 1 code
        else:
            self.backend = None
            self.eng = _native.Engine(self.dev)
        for item in args.opt:
            name, val = item.split("=")
            self.eng.set_option(name, int(val))

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize(self.dev)

    def coll_device(self):
        """where the tensors of this script's own collectives (timings, not lookups) live: the GPU under RCCL, the host under gloo"""
        return self.torch.device("cpu") if self.dry_run else self.torch.device("cuda", self.dev)

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.coll_device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_fn(self):
        """dry run only: the exchange of a lookup by other means than RCCL -- this rank's [nq, k] keys -> host -> gloo all-gather -> device
        [world, nq, k] (ShardedSearcher's test hook; the merge kernel and everything around it are the product's)."""
        if not self.dry_run:
            return None
        torch, dist, backend = self.torch, self.dist, self.backend

        def gather(local):
            backend.stream.synchronize()
            mine = local.cpu().contiguous()
            parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(parts, mine)
            with torch.cuda.stream(backend.stream):
                return torch.stack(parts).to(local.device)
        return gather


class HwmonSampler:
    """Board clock and power while a leg runs, read in-process from the amdgpu hwmon files of THIS device (freq1_input = sclk in Hz,
    power1_input = package power in uW; matched to the HIP device by PCI address) every 10 ms on a thread.  Used around the event pass of a
    record -- the same K steps as the timed region, outside it -- so the driver's own line carries the clock the roofline's kernel ran at."""

    def __init__(self, torch, dev: int):
        import glob
        import threading

        self.freq = self.power = None
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            pass
        found = []
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            hw = glob.glob(os.path.join(card, "hwmon", "hwmon*", "freq1_input"))
            if hw:
                found.append((os.path.basename(os.path.realpath(card)), os.path.dirname(hw[0])))
        pick = [h for addr, h in found if want and addr.lower() == want.lower()] or ([found[0][1]] if len(found) == 1 else [])
        if pick:
            self.freq, self.power = os.path.join(pick[0], "freq1_input"), os.path.join(pick[0], "power1_input")
        self.sclk, self.watts = [], []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                with open(self.freq) as f:
                    self.sclk.append(int(f.read()) / 1e6)
                with open(self.power) as f:
                    self.watts.append(int(f.read()) / 1e6)
            except Exception:
                return
            self._stop.wait(0.01)

    def __enter__(self):
        if self.freq:
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.freq:
            self._thread.join()

    def summary(self) -> dict | None:
        if len(self.sclk) < 3:
            return None
        tail = lambda a: a[len(a) // 4:]  # (the first quarter: the clock is still settling from the barrier's idle moment)
        return {"sclk_mhz": float(np.median(tail(self.sclk))), "power_w": float(np.median(tail(self.watts))), "samples": len(self.sclk)}


def kernel_times(ctx: Ctx) -> dict:
    n = ctx.native
    ids = {"scan": n.KERNEL_SCAN, "merge": n.KERNEL_MERGE, "mfma_last_phase": n.KERNEL_MFMA, "mfma_earlier_phases": n.KERNEL_MFMA_SAMPLE,
           "skinny_last_phase": n.KERNEL_SKINNY, "convert": n.KERNEL_CONVERT, "rescore": n.KERNEL_RESCORE, "exchange": n.KERNEL_EXCHANGE}
    return {name: ctx.eng.profile_read(kid) for name, kid in ids.items()}


def run_record(ctx: Ctx, name: str, wl: dict, corpus, shard_lo: int, steps: int, warmup: int, with_cpu: bool) -> dict:
    """Time `steps` lookups of workload `wl` on the corpus already resident on this rank; returns the record (rank 0)."""
    args, eng, torch = ctx.args, ctx.eng, ctx.torch
    dim, k, nq = wl["dim"], wl["k"], wl["nq"]
    rows_local, rows_total = int(corpus.shape[0]), wl["rows_total"]
    min_score = wl.get("min_score", args.min_score)
    thr = float(ctx.native.f32_threshold(min_score))
    # identical on every rank; arbitrary fp32 values (not fp16-representable).  BATCH_ROTATION different batches take turns in the timed region.
    n_rot = BATCH_ROTATION if nq > 1 else 1
    if wl.get("kind") == "clustered":
        queries = clustered_queries(eng, max(64, nq * n_rot), rows_total, dim, wl["seed"], wl.get("cluster_rows", CLUSTER_ROWS))
        queries = queries[np.random.default_rng(7).permutation(len(queries))]  # (neighbouring clusters do not share a query tile)
    elif wl.get("kind") == "aniso":
        queries = aniso_queries(eng, max(64, nq * n_rot), dim, wl["seed"])
    else:
        queries = host_queries(max(64, nq * n_rot), dim, 4242)

    searcher = None
    if ctx.distributed:
        from typeagent_py_amd.sharded import ShardedSearcher

        ctx.backend.set_shard(corpus, row_offset=shard_lo)
        searcher = ShardedSearcher(ctx.backend, gather_fn=ctx.gather_fn())
        dq_all = torch.from_numpy(queries).to(torch.device("cuda", ctx.dev))
    else:
        eng.set_corpus_tensor(corpus)
        if nq > 1:  # batches: the queries are resident too when the timed region starts (the host-buffer rate is reported beside it)
            dq_all = torch.from_numpy(queries[: nq * n_rot]).to(torch.device("cuda", ctx.dev))
            keys_buf = torch.empty((nq, k), dtype=torch.int64, pin_memory=True)  # the last kernel writes the result keys straight into pinned host memory
            keys_np = keys_buf.numpy()
            torch.cuda.synchronize()

    # the subset form (vectorbase.py:203-230): `subset` random ordinals (seed 99, without replacement: tools/benchmark_vectorbase.py:133-136),
    # their row list resident on the device -- uploaded once, as the class does for a subset it is handed again (tavb_search_subset_resident)
    subset = dev_subset = None
    if wl.get("subset") and searcher is None:
        subset = np.random.default_rng(99).choice(rows_total, size=min(int(wl["subset"]), rows_total), replace=False).astype(np.int64)
        dev_subset = eng.rows_to_device(subset)
        torch.cuda.synchronize()

    n_lookups = [0]

    def one_step(i: int, host_queries_form: bool = False):
        n_lookups[0] += 1
        if nq == 1:
            qi = i % len(queries)
            if subset is not None:
                pos, scs = eng.search_subset_resident(queries[qi], dev_subset, k, np.float32(thr))
                return subset[pos], scs
            if searcher is None:
                # C ABI as the drop-in class calls it: 6 KiB host query in, host results out (the kernel writes them into pinned memory)
                return eng.search(queries[qi], k, np.float32(thr))
            r = searcher.search(dq_all[qi : qi + 1], k, min_score)
            return r.ordinals[0, : r.counts[0]], r.scores[0, : r.counts[0]]
        b0 = (i % n_rot) * nq  # the batch of this step
        if searcher is None:
            if host_queries_form:
                return eng.search_batch(queries[b0 : b0 + nq], k, np.float32(thr))
            eng.search_device(dq_all[b0 : b0 + nq], k, thr, out_keys=keys_buf)
            eng.synchronize()
            return ctx.native.decode_keys(keys_np)
        r = searcher.search(dq_all[b0 : b0 + nq], k, min_score)
        return r.ordinals, r.scores, r.counts

    w0, n_w = time.perf_counter(), 0
    while n_w < warmup or (args.warmup is None and time.perf_counter() - w0 < 0.3):  # default: at least 0.3 s, the clocks ramp for that long
        one_step(n_w)
        n_w += 1
    warmup = n_w
    # ---- the timed region: exactly `steps` steps, nothing instrumented.  The interpreter's cyclic collector is held off for its duration (as
    #      `timeit` does): a generation-2 collection over torch's module graph takes 30-40 ms -- one such pause inside a 60-step region of
    #      3.6 ms steps reads as +17 % (found in round 4: the "sporadic host stalls" of the earlier rounds; whether one lands inside the region
    #      depends on the allocation count up to it, i.e. on the command line).
    import gc

    gc.collect()
    gc.disable()
    ctx.barrier()
    lat = []
    t0 = time.perf_counter()
    for i in range(steps):
        s0 = time.perf_counter_ns()
        one_step(warmup + i)
        lat.append((time.perf_counter_ns() - s0) / 1e3)
    t_loop = time.perf_counter() - t0
    ctx.barrier()
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
    gc.enable()
    if os.environ.get("TAVB_BENCH_DEBUG"):
        sys.stderr.write(f"[bench debug] {name}: loop {t_loop * 1e3:.2f} ms, sum of steps {sum(lat) / 1e3:.2f} ms, max step {max(lat) / 1e3:.3f} ms, "
                         f"closing barrier {(elapsed - t_loop) * 1e3:.2f} ms\n")
    # ---- the same `steps` steps again with a HIP event pair around every kernel launch (on the library's stream): the kernel times of the
    #      roofline.  Kept out of the timed region: an event pair costs the stream ~0.03-0.07 ms of bubbles per launch (10 % of a 1.4 ms step).
    eng.profile_enable(True)
    eng.profile_reset()
    ctx.barrier()
    e0 = time.perf_counter()
    with HwmonSampler(torch, ctx.dev) as hw:
        for i in range(steps):
            one_step(warmup + i)
        ctx.barrier()
    elapsed_with_events = ctx.max_over_ranks(time.perf_counter() - e0)
    kt = kernel_times(ctx)
    eng.profile_enable(False)
    dist_extra = None
    if ctx.dist is not None:
        # every rank's own scan time and what it spent in the exchange (ncclAllGather inside tavb_search_allgather, HIP events on the library's
        # stream: a rank that finishes its scan early waits there for the slowest one, so the MINIMUM over ranks is the cost of the exchange
        # itself and max - min of the scan times is the skew)
        scan_ms = sum(kt[p][0] for p in ("scan", "mfma_last_phase", "mfma_earlier_phases", "skinny_last_phase")) / steps
        mine = torch.tensor([scan_ms, kt["exchange"][0] / steps, kt["merge"][0] / steps, float(rows_local)], dtype=torch.float64, device=ctx.coll_device())
        every = [torch.zeros_like(mine) for _ in range(ctx.dist.get_world_size())]
        ctx.dist.all_gather(every, mine)
        per = torch.stack(every).cpu().numpy()
        dist_extra = {
            "scan_ms_per_rank": [float(x) for x in per[:, 0]],
            "rank_skew_ms": float(per[:, 0].max() - per[:, 0].min()),
            "exchange_ms": float(per[:, 1].min()),           # all-gather of the per-shard [nq, k] key lists (256 KiB per rank at 1024 x 32)
            "exchange_ms_incl_wait": float(per[:, 1].max()), # ... on the rank that waited longest for its peers
            "merge_ms": float(per[:, 2].max()),
            "rows_per_rank": [int(x) for x in per[:, 3]],
        }
        if ctx.dry_run:  # (no RCCL all-gather ran: the keys went through the host; exchange_ms is 0 and the rates are two ranks sharing one GPU)
            dist_extra["backend"] = f"{ctx.dist_backend} dry run on one GPU: keys exchanged through the host, rates not meaningful"
    host_form = None
    if searcher is None and nq > 1:  # the same batch handed over as a host buffer (what VectorBase.fuzzy_lookup_embeddings does): PCIe-inclusive
        one_step(0, True)
        n_host = max(3, min(steps, 20))
        h0 = time.perf_counter()
        for i in range(n_host):
            one_step(i, True)
        h_ms = (time.perf_counter() - h0) / n_host * 1e3
        host_form = {"ms_per_step": h_ms, "queries_per_sec": nq / (h_ms * 1e-3), "steps": n_host}  # tavb_search_batch: host queries in, host results out

    # answers for the parity sample (outside the timed region)
    if nq == 1:
        sample = list(range(PARITY_QUERIES))
        got = {qi: one_step(qi)[:2] for qi in sample}
    else:
        sample = sorted(set(np.linspace(0, nq - 1, PARITY_QUERIES).astype(int).tolist()))
        o, s, c = one_step(0)  # (batch 0 of the rotation)
        got = {qi: (o[qi, : c[qi]], s[qi, : c[qi]]) for qi in sample}
        flagged = []
        if searcher is None:
            try:  # queries whose candidate band did not fit and were re-run on the exact tile (0 on ordinary data), per batch of the rotation
                for b in range(n_rot):
                    if b:
                        one_step(b)
                    flagged.append(int(eng.get_option("last_flagged")))
            except Exception:
                flagged = []
    if ctx.rank != 0:
        # rank 0 regenerates what it needs for the oracle; the others only keep the collective calls aligned
        return {}
    sys.stderr.write(f"bench.py: {name}: {n_lookups[0]} lookups in this process (warm-up, timed steps, event pass, host-buffer form, answers)\n")

    qps = steps * nq / elapsed
    esize = 2 if wl["dtype"] == "fp16" else 4
    shadow = False
    try:
        shadow = wl["dtype"] == "fp32" and bool(eng.get_option("last_shadow"))
    except Exception:
        pass
    if shadow:
        esize = 2  # the pass that streams the corpus read its fp16 shadow (exact fp32 answers: candidates rescored with the fp32 rows)
    if wl["bound"] == "hbm" and kt["scan"][1]:
        kern_name, parts = "scan_*_kernel", ["scan"]
    elif kt["mfma_last_phase"][1]:
        kern_name, parts = "mfma_scan_kernel", ["mfma_last_phase", "mfma_earlier_phases"]
    elif kt["skinny_last_phase"][1]:
        kern_name, parts = "skinny_scan_kernel", ["skinny_last_phase", "mfma_earlier_phases"]
    else:
        kern_name, parts = "scan_*_kernel", ["scan"]
    kern_ms_per_step = sum(kt[p][0] for p in parts) / steps
    launches_per_step = sum(kt[p][1] for p in parts) / steps
    passes = kt[parts[0]][1] / steps  # corpus passes per step (a batch bigger than one pass serves is split)
    if wl["bound"] in ("hbm", "latency"):
        alg = rows_local * dim * esize * passes  # bytes the scan must read per step: this rank's shard once per pass
        if subset is not None:
            alg = len(subset) * (dim * esize + 4) * passes  # the subset's rows + its int32 row list
        achieved = alg / (kern_ms_per_step * 1e-3) / 1e9 if kern_ms_per_step > 0 else 0.0
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None}
        if wl["bound"] == "latency":
            # a corpus of a few MB (the reference's own scale) sits in the 256 MiB Infinity Cache and a lookup is two or three submissions: the
            # record is bound by launch + synchronise latency, not by HBM -- no fraction of the HBM peak is claimed for it
            roof = {"bound": "latency", "kernel_us_per_step": kern_ms_per_step * 1e3, "bytes_per_step": alg, "traffic": None,
                    "note": "launch-bound: the corpus is cache-resident; p50_latency_us is the figure of merit"}
    else:
        alg = 2.0 * nq * rows_local * dim
        achieved = alg / (kern_ms_per_step * 1e-3) / 1e12 if kern_ms_per_step > 0 else 0.0
        # the pipe that did the work: fp16 MFMAs for the 128/256-query tile (fp16 corpora, and fp32 corpora through their fp16 shadow + exact
        # rescoring), fp32 MFMAs for the 64-query tile on fp32 corpora
        fp16_pipe = wl["dtype"] == "fp16" or parts[0] == "mfma_last_phase"
        peak = MFMA_F16_PEAK_TFLOPS if fp16_pipe else MFMA_F32_PEAK_TFLOPS
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                "pipe": "v_mfma_f32_32x32x16_f16" if fp16_pipe else "v_mfma_f32_32x32x2_f32"}
    try:  # HBM traffic comes from a separate rocprofv3 --pmc pass (bench.py cannot count it itself): profiles/pmc_traffic.json, B per step
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f).get(name)
        if pmc and pmc.get("traffic_bytes_per_step") and rows_local == WORKLOADS[name]["rows"] and ctx.world == 1:
            roof["traffic"] = pmc["traffic_bytes_per_step"]
    except Exception:
        pass
    if shadow:
        roof["scanned"] = "fp16 shadow"  # of the fp32 corpus, as an exact filter; the band of candidates is rescored with the fp32 rows
    hwm = hw.summary()
    if hwm:
        # board clock / power sampled in-process while the event pass ran (amdgpu hwmon, 10 ms): dense MFMA work holds this board well below
        # the 2.4 GHz the peak is quoted at -- `frac_at_clock` prices the kernel against the peak at the clock it actually ran at
        roof.update(hwm)
        if roof["bound"] == "mfma" and hwm["sclk_mhz"] > 0:
            roof["frac_at_clock"] = roof["achieved"] / (roof["peak"] * hwm["sclk_mhz"] / 2400.0)
    roof.update({
        "kernel": kern_name,
        "kernel_ms_per_step": kern_ms_per_step,          # HIP event pairs around every launch, over a second pass of the same steps
        "kernel_launches_per_step": launches_per_step,
        "algorithmic_per_step": alg,                     # bytes (hbm) or flops (mfma)
        "ms_per_step_with_events": elapsed_with_events / steps * 1e3,
        "kernel_parts_ms_per_step": {p: kt[p][0] / steps for p in parts},
        "other_kernels_ms_per_step": {p: kt[p][0] / steps for p in kt if p not in parts and kt[p][1]},
    })
    rec = {
        "workload": f"{name}: {rows_total}x{dim} {wl['dtype']}"
                    + (f" over {ctx.world} GPUs ({rows_local} rows on rank 0)" if ctx.world > 1 else "")
                    + f", {nq} q/step, top-{k}, min_score {min_score:g}",
        "queries_per_sec": qps,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "p50_latency_us": float(np.percentile(lat, 50)),
        "p99_latency_us": float(np.percentile(lat, 99)),
        "min_latency_us": float(np.min(lat)),
        "dtype": "f32" if wl["dtype"] == "fp32" else "f16 storage, f32 accumulate",
        "roofline": roof,
    }
    if nq > 1:
        rec["query_batches_in_rotation"] = n_rot
        if searcher is None and flagged:
            rec["flagged_fraction"] = float(sum(flagged)) / (len(flagged) * nq)  # queries re-run on the exact tile, over the batches of the rotation
            exact_ms = kt["rescore"][0] / steps
            if rec["flagged_fraction"] > 0.99 and roof["bound"] == "mfma" and exact_ms > kern_ms_per_step:
                # the whole batch took the 256-query tile's exact split-plane form (the K loop once per fp16 plane of the fp32 queries: twice the
                # MFMAs of a filter pass) and, found out early, skipped the filter's last phase: THAT launch is the dominant kernel of this record
                alg2 = 2.0 * alg
                ach2 = alg2 / (exact_ms * 1e-3) / 1e12
                roof.update({"kernel": "mfma_scan_kernel, exact split-plane form (+ its selection)", "achieved": ach2, "frac": ach2 / roof["peak"],
                             "kernel_ms_per_step": exact_ms, "algorithmic_per_step": alg2, "kernel_launches_per_step": None,
                             "filter_phases_ms_per_step": kern_ms_per_step})
    if host_form:
        rec["host_buffer_form"] = host_form
    if dist_extra:
        rec["exchange"] = dist_extra
    if subset is not None:
        rec["workload"] += f", subset of {len(subset)} ordinals (row list resident)"
    if not args.no_parity:
        rec["parity"] = parity_check(eng, corpus, shard_lo, wl, queries, sample, got, min_score, subset)
    if with_cpu and not args.no_cpu_baseline:
        n_host = min(rows_local, ORACLE_CHUNK)
        host = corpus[:n_host].float().cpu().numpy()
        rec["cpu_baseline"] = cpu_baseline(host, queries, k, rows_total, nq, wl.get("cpu_seconds", args.cpu_seconds), subset)
    return rec


def sustained_calibration(ctx: Ctx, wl: dict, corpus, kern_ms_per_step: float) -> dict:
    """Two calibrations of what this box sustains on the cfg3 contraction RIGHT NOW (same process, same corpus bytes, same power state),
    outside every timed region and never `value`:
      mfma_only_tflops    the shipping tile kernel with its operand staging and admissions compiled out (`mfma_ablate=258`: the MFMA stream,
                          fragment reads and barriers only; answers are garbage) over the whole corpus, all ladder phases;
      vendor_gemm_tflops  torch.matmul (hipBLASLt) on [327680, 1536] rows of the corpus x [1536, 1024] queries in fp16, product written, no
                          selection.
    `frac_of_mfma_only` = the shipping kernel's rate over the first: how much of what the matrix pipe sustains at this power state the
    fused kernel keeps."""
    eng, torch = ctx.eng, ctx.torch
    nq, k, dim = wl["nq"], wl["k"], wl["dim"]
    rows = int(corpus.shape[0])
    flops = 2.0 * nq * rows * dim
    dq = torch.from_numpy(host_queries(nq, dim, 4242)).to(torch.device("cuda", ctx.dev))
    keys = torch.empty((nq, k), dtype=torch.int64, pin_memory=True)
    out = {}
    try:
        eng.set_option("mfma_ablate", 258)
        for _ in range(3):
            eng.search_device(dq, k, 0.0, out_keys=keys)
        eng.synchronize()
        eng.profile_enable(True)
        eng.profile_reset()
        n = 8
        with HwmonSampler(torch, ctx.dev) as hw:
            for _ in range(n):
                eng.search_device(dq, k, 0.0, out_keys=keys)
            eng.synchronize()
        kt = kernel_times(ctx)
        eng.profile_enable(False)
        ms = (kt["mfma_last_phase"][0] + kt["mfma_earlier_phases"][0]) / n
        out["mfma_only_ms_per_step"] = ms
        out["mfma_only_tflops"] = flops / (ms * 1e-3) / 1e12
        out["frac_of_mfma_only"] = ms / kern_ms_per_step if kern_ms_per_step > 0 else None
        if hw.summary():  # (the clock and power the ablation ran at: how much of the gap to it is clock, how much pipe occupancy)
            out["mfma_only_sclk_mhz"], out["mfma_only_power_w"] = hw.summary()["sclk_mhz"], hw.summary()["power_w"]
    finally:
        eng.set_option("mfma_ablate", 0)
    g_rows = min(rows, 327_680)
    a = corpus[:g_rows]
    b = dq.to(torch.float16)  # [nq, dim]: the product is rows x queries^T, hipBLASLt's NT form
    prod = torch.empty((g_rows, nq), dtype=torch.float16, device=a.device)
    for _ in range(3):
        torch.matmul(a, b.t(), out=prod)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 200  # (~0.2 s: long enough for the clock / power sampler)
    with HwmonSampler(torch, ctx.dev) as hw:
        e0.record()
        for _ in range(iters):
            torch.matmul(a, b.t(), out=prod)
        e1.record()
        torch.cuda.synchronize()
    out["vendor_gemm_tflops"] = 2.0 * g_rows * nq * dim / (e0.elapsed_time(e1) / iters * 1e-3) / 1e12
    if hw.summary():
        out["vendor_gemm_sclk_mhz"], out["vendor_gemm_power_w"] = hw.summary()["sclk_mhz"], hw.summary()["power_w"]
    del prod
    return out


def terms_variants(ctx: Ctx, wl: dict, corpus) -> dict:
    """cfg1_terms32 as the batched `lookup_terms` patch calls it (`tavb_search_batch`: host queries in, host results out), median us per call:
    as shipped (the grouped one-launch form), with that form off (the tiles: the routing until the end of round 6), and as nq sequential single
    lookups (what the reference's loop costs on this engine).  Outside every timed region."""
    eng = ctx.eng
    nq, k = wl["nq"], wl["k"]
    thr = np.float32(ctx.native.f32_threshold(wl["min_score"]))
    q = aniso_queries(eng, nq, wl["dim"], wl["seed"])
    eng.set_corpus_tensor(corpus)

    def med(fn, n=200):
        for _ in range(20):
            fn()
        t = []
        for _ in range(n):
            t0 = time.perf_counter_ns()
            fn()
            t.append((time.perf_counter_ns() - t0) / 1e3)
        return float(np.median(t))

    out = {"grouped_us": med(lambda: eng.search_batch(q, k, thr)), "last_direct": int(eng.get_option("last_direct"))}
    max_nq = eng.get_option("direct_group_max_nq")
    eng.set_option("direct_group_max_nq", 0)
    try:
        out["tiles_us"] = med(lambda: eng.search_batch(q, k, thr))
    finally:
        eng.set_option("direct_group_max_nq", max_nq)
    out["sequential_us"] = med(lambda: [eng.search(q[i], k, thr) for i in range(nq)], n=50)
    return out


def class_api_rates(ctx: Ctx, wl: dict, corpus, min_score: float, steps: int) -> dict:
    """The same batch through the drop-in class: `VectorBase.fuzzy_lookup_embeddings` to `list[list[ScoredInt]]` (what the reference's
    callers get, vectorbase.py:188-190 per query) and with `as_arrays=True`.  Host queries in, host objects out."""
    from typeagent_py_amd import TextEmbeddingIndexSettings, VectorBase

    class _Null:
        model_name = "bench"

    nq, k, dim = wl["nq"], wl["k"], wl["dim"]
    vb = VectorBase(TextEmbeddingIndexSettings(_Null()), device=ctx.dev)
    vb.adopt_device_corpus(corpus)
    queries = host_queries(max(64, nq * BATCH_ROTATION), dim, 4242)
    # the subset form: ONE Python list handed in again and again, as tools/benchmark_vectorbase.py:133-163 and the memory provider do
    sub_list = np.random.default_rng(99).choice(int(corpus.shape[0]), size=int(wl["subset"]), replace=False).tolist() if wl.get("subset") else None
    out = {}
    for label, kw in (("scored_int_lists", {}), ("as_arrays", {"as_arrays": True})):
        def call(i):
            b0 = (i % BATCH_ROTATION) * nq if nq > 1 else i % len(queries)
            if sub_list is not None:
                return vb.fuzzy_lookup_embedding_in_subset(queries[b0], sub_list, max_hits=k, min_score=min_score)
            if nq == 1:
                return vb.fuzzy_lookup_embedding(queries[b0], max_hits=k, min_score=min_score)
            return vb.fuzzy_lookup_embeddings(queries[b0 : b0 + nq], max_hits=k, min_score=min_score, **kw)
        if nq == 1 and kw:
            continue
        call(0)
        import gc

        gc.collect()  # (the collector stays ON here: a consumer of the class runs with it; start from a clean slate so that no old garbage is billed to these calls)
        t0 = time.perf_counter()
        for i in range(steps):
            call(1 + i)
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[label] = {"ms_per_step": ms, "queries_per_sec": nq / (ms * 1e-3)}
    if sub_list is not None:  # ... and handed a FRESH list every call (nothing to recognise: list -> ndarray, range check, upload, every time)
        call(0)
        n_fresh = max(3, min(steps, 20))
        t0 = time.perf_counter()
        for i in range(n_fresh):
            vb.fuzzy_lookup_embedding_in_subset(queries[i % len(queries)], list(sub_list), max_hits=k, min_score=min_score)
        out["fresh_list_every_call"] = {"ms_per_step": (time.perf_counter() - t0) / n_fresh * 1e3}
    del vb
    return out


def sharded_class_api_rates(ctx: Ctx, wl: dict, corpus, shard_lo: int, min_score: float, steps: int) -> dict:
    """COLLECTIVE (every rank calls it): the same batch through the class-shaped front end of the row-sharded path,
    `ShardedVectorBase.fuzzy_lookup_embeddings` -- host queries in (staged through a reused pinned buffer), one `tavb_search_allgather`,
    `list[list[ScoredInt]]` out (built in C) -- what a consumer of the sharded class pays on top of the engine step of the headline."""
    from typeagent_py_amd.sharded import ShardedSearcher, ShardedVectorBase

    nq, k, dim = wl["nq"], wl["k"], wl["dim"]
    svb = ShardedVectorBase(ctx.backend, shard_lo, int(corpus.shape[0]), wl["rows_total"])
    if ctx.dry_run:
        svb.searcher = ShardedSearcher(ctx.backend, gather_fn=ctx.gather_fn())
    queries = host_queries(max(64, nq * BATCH_ROTATION), dim, 4242)

    def call(i):
        b0 = (i % BATCH_ROTATION) * nq
        return svb.fuzzy_lookup_embeddings(queries[b0 : b0 + nq], max_hits=k, min_score=min_score)

    call(0)
    import gc

    gc.collect()
    ctx.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        call(1 + i)
    ctx.barrier()
    ms = ctx.max_over_ranks(time.perf_counter() - t0) / steps * 1e3
    return {"sharded_scored_int_lists": {"ms_per_step": ms, "queries_per_sec": nq / (ms * 1e-3)}}


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    from typeagent_py_amd.sharded import shard_range

    return shard_range(total, world, rank)


def headline_line(ctx: Ctx, rec: dict, name: str, wl: dict, scaling: str, sub: dict | None) -> dict:
    out = {
        "metric": BASELINE_METRIC,
        "value": rec["queries_per_sec"],
        "unit": "queries/s",
        "n_gpus": ctx.world,
        "steps": rec["steps"],
        "warmup": rec["warmup"],
        "ms_per_step": rec["ms_per_step"],
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": rec["dtype"],
        "data": "synthetic",
        "config": {
            "workload": rec["workload"],
            "total_rows": wl["rows_total"],
            "queries_per_step": wl["nq"],
            "k": wl["k"],
            "parallelism": (f"row-sharded x{ctx.world} ({scaling} scaling), RCCL all-gather of per-shard top-k keys + merge kernel on every rank"
                            if ctx.world > 1 else "single GPU"),
        },
        "p50_latency_us": rec["p50_latency_us"],
        "p99_latency_us": rec["p99_latency_us"],
        "roofline": {k: v for k, v in rec["roofline"].items() if k not in ("pipe", "ms_per_step_with_events")},
        "cpu_baseline": rec.get("cpu_baseline"),
    }
    if getattr(ctx, "dry_run", False):
        out["dry_run"] = (f"{ctx.world} ranks sharing ONE GPU over {ctx.dist_backend}: this line proves that bench.py's N > 1 branches run "
                          "(rank != 0, per-rank timing gather, whole-corpus parity over every rank's rows, cfg4_weak); its rates are not measurements")
    if "parity" in rec:
        out["parity"] = {k: v for k, v in rec["parity"].items() if k not in ("rows", "max_inverted_gap_gpu", "max_inverted_gap_ref")}
    if "host_buffer_form" in rec:
        out["host_buffer_form"] = rec["host_buffer_form"]  # PCIe-inclusive rate of the same batch (never `value`)
    for key in ("query_batches_in_rotation", "flagged_fraction", "class_api", "exchange"):
        if key in rec:
            out[key] = rec[key]
    if scaling == "weak" and ctx.world >= 1:
        out["row_queries_per_sec"] = rec["queries_per_sec"] * wl["rows_total"]
    if sub:
        # the driver's record keeps the top-level contract fields and the last 2000 characters of the line: the records a reader is most likely
        # to look for there (this round's: the mid-batch tiles, cfg5's variants) go last
        last = [k for k in ("cfg5", "cfg3_dup", "cfg1_1k_d384", "cfg1_d384", "cfg1_subset", "cfg1_terms32", "cfg3_subset", "cfg2_d3072", "cfg3_d3072_q1", "cfg3_d3072", "cfg3_aniso_q1", "cfg3_aniso",
                            "cfg4_weak") if k in sub]
        out["sub"] = {k: slim_sub(sub[k]) for k in [k for k in sub if k not in last] + last}
    return out


def run_cfg5(args, wl, emit: bool = True, steps: int | None = None, warmup: int | None = None, variants: bool = False):
    """BASELINE config 5: user queries/s of the fused multi-index submission (typeagent_py_amd/fused.py).  `emit=False`: return the
    record (a sub-record of the north-star suite) instead of printing it."""
    import torch

    from typeagent_py_amd import TextEmbeddingIndexSettings, VectorBase, _native
    from typeagent_py_amd.fused import FusedIndexQuery

    rows, dim = wl["rows"], wl["dim"]
    steps = steps if steps is not None else (args.steps if args.steps is not None else 30)
    warmup = warmup if warmup is not None else (args.warmup if args.warmup is not None else 3)
    fq = FusedIndexQuery(0)
    eng = fq.engine
    with torch.cuda.stream(fq.stream):
        terms = make_device_corpus(eng, rows, dim, 50_043, wl["dtype"])
        msgs = make_device_corpus(eng, rows, dim, 50_044, wl["dtype"])
        threads = make_device_corpus(eng, 1000, dim, 50_045, wl["dtype"])
    fq.set_corpus("terms", terms)
    fq.set_corpus("messages", msgs)
    fq.set_corpus("threads", threads)
    for item in args.opt:
        name, val = item.split("=")
        eng.set_option(name, int(val))
    rng = np.random.default_rng(5)
    # queries near real rows so that the thresholds keep a few hits (gaussian data is otherwise all below 0.7)
    def near(t, r, eps):
        v = t[r].float().cpu().numpy() + eps * rng.standard_normal(dim).astype(np.float32) / np.sqrt(dim)
        return (v / np.linalg.norm(v)).astype(np.float32)
    user_queries = []
    for u in range(8):
        tq = np.stack([near(terms, int(rng.integers(rows)), 0.3) for _ in range(4)])
        user_queries.append((tq, near(msgs, int(rng.integers(rows)), 0.6), near(threads, int(rng.integers(1000)), 0.6)))
    def make_one(separate: bool, subset):
        if separate:
            def one(i):
                tq, mq, hq = user_queries[i % len(user_queries)]
                out = [vbs[0].fuzzy_lookup_embedding(q, max_hits=50, min_score=0.85) for q in tq]
                if subset is None:
                    out.append(vbs[1].fuzzy_lookup_embedding(mq, max_hits=25, min_score=0.7))
                else:
                    out.append(vbs[1].fuzzy_lookup_embedding_in_subset(mq, subset, max_hits=25, min_score=0.7))
                out.append(vbs[2].fuzzy_lookup_embedding(hq, max_hits=10, min_score=0.7))
                return out
        else:
            def one(i):
                tq, mq, hq = user_queries[i % len(user_queries)]
                return fq.run(tq, mq, hq, message_subset=subset)
        return one

    def subset_of(n: int):
        return np.random.default_rng(99).choice(rows, size=n, replace=False).tolist() if n else None

    vbs = []
    if args.cfg5_separate or variants:
        class _Null:
            model_name = "bench"
        for t in (terms, msgs, threads):
            vb = VectorBase(TextEmbeddingIndexSettings(_Null()), device=0)
            vb.adopt_device_corpus(t)
            vbs.append(vb)

    import gc

    def time_mode(one, n_steps: int, n_warm: int):
        for i in range(n_warm):
            one(i)
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()  # (as in run_record: no generation-2 collection inside the timed region)
        lat = []
        t0 = time.perf_counter()
        for i in range(n_steps):  # the timed region: un-instrumented
            s0 = time.perf_counter_ns()
            one(n_warm + i)
            lat.append((time.perf_counter_ns() - s0) / 1e3)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        gc.enable()
        return el, lat

    subset = subset_of(args.cfg5_subset)
    one = make_one(args.cfg5_separate, subset)
    elapsed, lat = time_mode(one, steps, warmup)
    if not args.cfg5_separate:  # the same steps again with HIP event pairs around the launches (kernel times)
        eng.profile_enable(True)
        eng.profile_reset()
        for i in range(steps):
            one(warmup + i)
        torch.cuda.synchronize()
    esize = 2 if wl["dtype"] == "fp16" else 4

    def alg_bytes(sub_):
        return rows * dim * esize + (len(sub_) if sub_ else rows) * dim * esize + 1000 * dim * esize

    alg = alg_bytes(subset)
    out = {
        "metric": "user queries/sec + p50 latency, fused multi-index VectorBase lookups (cfg5)",
        "value": steps / elapsed, "unit": "user-queries/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 storage, f32 accumulate", "data": "synthetic",
        "config": {"workload": f"cfg5: 4 term lookups k=50@0.85 + message re-rank k=25@0.7 ({'subset of ' + str(len(subset)) if subset else 'full scan'}), "
                               f"each on {rows}x{dim}, + thread lookup k=10@0.7 on 1000x{dim}; {'six separate calls' if args.cfg5_separate else 'one fused submission'}"},
        "p50_latency_us": float(np.percentile(lat, 50)), "p99_latency_us": float(np.percentile(lat, 99)),
        "roofline": {"bound": "hbm", "achieved": alg / (elapsed / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, "traffic": None},  # whole user query (kernels + copies + one sync) against the bytes the three passes must read
        "cpu_baseline": None,
    }
    if not args.cfg5_separate:
        ms, n = eng.profile_read(_native.KERNEL_SCAN)
        out["roofline"]["scan_kernel_ms_per_user_query"] = ms / steps
        out["roofline"]["scan_launches_per_user_query"] = n / steps
        eng.profile_enable(False)
    if variants:
        # SURVEY 8d cfg5, the other forms on the same corpora: the memory provider's message re-rank over a subset of 1000 ordinals
        # (tools/benchmark_vectorbase.py:133-136), and the six lookups as separate synchronous calls through the drop-in class
        sub1000 = subset_of(1000)
        one_s = make_one(False, sub1000)
        el_s, lat_s = time_mode(one_s, steps, warmup)
        a_s = alg_bytes(sub1000)
        out["variants"] = {"subset1000": {"value": steps / el_s, "ms_per_step": el_s / steps * 1e3, "hbm_frac": a_s / (el_s / steps) / 1e9 / HBM_PEAK_GBS}}
        if not args.no_parity:
            pa = cfg5_parity(eng, one_s, user_queries[:1], {"terms": (terms, 50_043), "messages": (msgs, 50_044), "threads": (threads, 50_045)}, dim, wl["dtype"], sub1000,
                             only=("messages",))
            out["variants"]["subset1000"]["parity"] = {k: pa[k] for k in ("ok", "error", "lookups_checked", "hits_returned") if k in pa}
        one_x = make_one(True, None)
        el_x, _ = time_mode(one_x, max(5, steps // 2), 2)
        n_x = max(5, steps // 2)
        out["variants"]["separate_calls"] = {"value": n_x / el_x, "ms_per_step": el_x / n_x * 1e3, "fused_speedup": (steps / elapsed) / (n_x / el_x)}
    if not args.no_parity:
        out["parity"] = cfg5_parity(eng, one, user_queries[:2], {"terms": (terms, 50_043), "messages": (msgs, 50_044), "threads": (threads, 50_045)},
                                    dim, wl["dtype"], subset)
    del terms, msgs, threads
    fq.close() if hasattr(fq, "close") else None
    torch.cuda.empty_cache()
    if not emit:
        return out
    emit_result(out)
    if out.get("parity") and not out["parity"]["ok"]:
        sys.stderr.write("bench.py: PARITY CHECK FAILED (see the `parity` object in the line above)\n")
        raise SystemExit(3)


def cfg5_parity(eng, one, user_queries, corpora: dict, dim: int, dtype: str, subset, only=None) -> dict:
    """Two user queries (12 lookups) of the cfg5 submission against the oracle over the WHOLE corpora (10M-row passes in
    ORACLE_CHUNK-row chunks): term lookups k=50 @0.85, message re-rank k=25 @0.7 (full scan or subset), thread lookup k=10 @0.7."""
    from oracle import vectorbase_oracle as vo

    t0 = time.perf_counter()
    checked = hits = 0
    tally = ParityTally()
    try:
        results = []
        for ui in range(len(user_queries)):
            res = one(ui)
            results.append((res.terms, res.messages, res.threads) if hasattr(res, "terms") else (res[:4], res[4], res[5]))
        # one oracle pass per corpus for all the user queries' lookups on it
        legs = [("terms", np.concatenate([u[0] for u in user_queries]), [h for r in results for h in r[0]], 50, 0.85),
                ("messages", np.stack([u[1] for u in user_queries]), [r[1] for r in results], 25, 0.7),
                ("threads", np.stack([u[2] for u in user_queries]), [r[2] for r in results], 10, 0.7)]
        for cname, qs, got_lists, k, ms in legs:
            if only is not None and cname not in only:
                continue
            tensor, seed = corpora[cname]
            thr = float(_f32_threshold(ms))
            sub = np.asarray(subset, dtype=np.int64) if (cname == "messages" and subset is not None) else None
            extra = [np.concatenate([np.asarray([r.item for r in got], dtype=np.int64), sub if sub is not None else np.zeros(0, dtype=np.int64)]) for got in got_lists]
            ref, referee = vo.scores_full_chunked_refereed(oracle_chunks(eng, tensor, 0, int(tensor.shape[0]), dim, seed, dtype), np.asarray(qs, dtype=np.float32),
                                                           extra, keep=k + 256)
            for j, got in enumerate(got_lists):
                items, scs = [r.item for r in got], [r.score for r in got]
                truth = referee.for_query(j)
                if sub is not None:
                    pos = {int(o): i for i, o in enumerate(subset)}
                    def sub_truth(p_, t=truth):
                        return t(sub[np.asarray(p_)])
                    sub_truth.dim = dim
                    rep, n_near = vo.check_topk_parity_large(ref[j][sub], [pos[i] for i in items], scs, k, thr, referee=sub_truth)
                else:
                    rep, n_near = vo.check_topk_parity_large(ref[j], items, scs, k, thr, referee=truth)
                    if items:
                        tally.worst = max(tally.worst, float(np.max(np.abs(ref[j][np.asarray(items)] - np.asarray(scs, dtype=np.float32)))))
                checked += 1
                hits += len(items)
                tally.add(rep, n_near)
    except AssertionError as exc:
        return {"ok": False, "error": str(exc)[:300], "lookups_checked": checked}
    return {"ok": True, "lookups_checked": checked, "hits_returned": hits, **tally.fields(), "seconds": round(time.perf_counter() - t0, 1)}


def _f32_threshold(x: float):
    from typeagent_py_amd import _native

    return _native.f32_threshold(x)


_RESULT_FD = None


def quiet_stdout() -> None:
    """Point fd 1 at stderr for the whole run (RCCL / HIP runtime banners, stray prints of imported code) and keep the real
    stdout aside: the one JSON line is the only thing the driver finds on stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def slim_sub(rec: dict) -> dict:
    """A sub-record of the suite without what the headline already says or profiles/README.md explains: the line has to stay under ~6 KB for
    the driver's record to hold every sub-record (round 3's 15 KB line lost `sub.cfg3_q1`)."""
    out = {}
    for key in ("workload", "queries_per_sec", "value", "unit", "ms_per_step", "p50_latency_us", "flagged_fraction", "vs_gaussian", "class_api",
                "variants", "exchange", "row_queries_per_sec"):
        if key in rec:
            out[key] = rec[key]
    if "row_queries_per_sec" in rec and "scaling" in rec:
        out["scaling"] = rec["scaling"]
    if "config" in rec and "workload" not in out:
        out["workload"] = rec["config"]["workload"]
    if "workload" in out:
        out["workload"] = out["workload"].split(": ", 1)[-1]  # (the key already names the workload)
        if out["workload"].endswith(", min_score 0"):
            out["workload"] = out["workload"][: -len(", min_score 0")]
    if rec.get("query_batches_in_rotation") or out.get("ms_per_step", 0) > 0.2:
        out.pop("p50_latency_us", None)  # a batch's latency is its ms_per_step; so is a long single lookup's (kept for the launch-bound cfg1)
    ro = rec.get("roofline") or {}
    # (`peak` is the headline's for the same `bound`: 8000 GB/s hbm, 2500 TFLOP/s fp16 mfma)
    keep = {k: ro[k] for k in ("bound", "achieved", "frac", "traffic", "kernel", "kernel_ms_per_step", "scan_kernel_ms_per_user_query", "scanned",
                               "kernel_us_per_step", "frac_at_clock", "sclk_mhz", "power_w") if k in ro}
    if keep.get("bound") != "mfma":
        for key in ("sclk_mhz", "power_w"):
            keep.pop(key, None)
    if keep.get("traffic") is None:
        keep.pop("traffic", None)
    if "exact" not in str(keep.get("kernel", "exact")):
        keep.pop("kernel")  # (named only where it is not the workload's usual kernel)
    if ro.get("other_kernels_ms_per_step"):
        keep["other_ms"] = ro["other_kernels_ms_per_step"]  # the kernels of a step that are not the roofline's (merge / select, rescoring)
    out["roofline"] = keep
    pa = rec.get("parity")
    if pa:
        out["parity"] = {k: pa[k] for k in ("ok", "error", "lookups_checked", "positions_exact", "positions_permuted", "max_permuted_gap",
                                            "gpu_inversions_vs_f64", "reference_inversions_vs_f64", "noise_gpu", "noise_ref", "rows", "seconds") if k in pa}
        if out["parity"].get("rows", 0) <= 12_500_000:  # (how long the oracle's pass took matters where it is long: the N > 1 records)
            out["parity"].pop("rows", None)
            out["parity"].pop("seconds", None)
        if out["parity"].get("positions_permuted") == 0:  # every position exact: nothing permuted, nothing inverted (on either side)
            for key in ("max_permuted_gap", "gpu_inversions_vs_f64", "reference_inversions_vs_f64"):
                out["parity"].pop(key, None)
    cb = rec.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "cores", "kind", "p50_ms_per_query_on_sample") if k in cb}
    return compact(out, 4)


def compact(x, digits: int = 5):
    """Floats to `digits` significant digits."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: compact(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [compact(v, digits) for v in x]
    return x


def emit_result(line: dict) -> None:
    data = (json.dumps(compact(line), separators=(",", ":")) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def respawn_under_torchrun(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no rank environment: become the launcher (one rank per GPU)."""
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def selftest(args) -> int:
    """`python bench.py --selftest`: prove N = 2 wherever a second GPU shows up.  Runs the headline workload (cfg3, strong scaling) at N = 1
    and N = 2 as child processes (N = 2: one rank per GPU, the exchange = tavb_search_allgather over RCCL), requires both parity objects
    to be green and prints one line with both records and the measured speed-up.  With one GPU: a `skipped` line, exit 0."""
    import torch

    n_gpu = torch.cuda.device_count()
    if n_gpu < 2:
        print(json.dumps({"selftest": "skipped", "reason": f"{n_gpu} GPU(s) visible; the N = 2 path needs two", "n_gpus_visible": n_gpu}))
        return 0
    lines = {}
    for n in (1, 2):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--workload", "cfg3", "--no-cpu-baseline", "--steps", str(args.steps or 10)]
        if args.rows:
            cmd += ["--rows", str(args.rows)]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        proc = subprocess.run(cmd, env=env, capture_output=True, text=True)
        out = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or not out:
            print(json.dumps({"selftest": "failed", "n_gpus": n, "rc": proc.returncode, "stderr_tail": proc.stderr[-1500:]}))
            return 1
        lines[n] = json.loads(out[-1])
    ok = all(lines[n].get("parity", {}).get("ok") for n in (1, 2)) and lines[2]["n_gpus"] == 2
    print(json.dumps({"selftest": "ok" if ok else "failed", "speedup_n2_over_n1": lines[2]["value"] / lines[1]["value"],
                      "n1": {k: lines[1][k] for k in ("value", "ms_per_step", "parity", "roofline")},
                      "n2": {k: lines[2][k] for k in ("value", "ms_per_step", "parity", "roofline", "config")}}))
    return 0 if ok else 1


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="one workload only (default: the north-star suite, headline cfg3)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="--gpus N > 1: split the same corpus (strong) or 12.5M rows per GPU (weak, cfg4)")
    ap.add_argument("--rows", type=int, default=None, help="override the row count (debugging)")
    ap.add_argument("--queries", type=int, default=None, help="override the queries per step (debugging)")
    ap.add_argument("--dim", type=int, default=None, help="override the row width (debugging: widths that are not a multiple of 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-sub", action="store_true", help="north-star suite: headline only")
    ap.add_argument("--budget-seconds", type=float, default=420.0, help="north-star suite: sub-record groups that would START after this many seconds of process time are skipped (the whole default suite takes ~280 s on a box of the pool)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cfg5-subset", type=int, default=0, help="cfg5: message re-rank over a subset of this many ordinals (0 = full scan)")
    ap.add_argument("--cfg5-separate", action="store_true", help="cfg5: issue the six lookups as separate synchronous calls")
    ap.add_argument("--min-score", type=float, default=0.0, help="score threshold of the lookups (0.0 = every row survives: worst case for selection; 0.85 = the reference's related-terms default)")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (e.g. scan_unroll=4)")
    ap.add_argument("--no-calibration", action="store_true", help="skip roofline.sustained (the MFMA-only ablation + vendor GEMM after the headline): counter passes want the lookups only")
    ap.add_argument("--class-api", action="store_true", help="also time the workload through VectorBase.fuzzy_lookup_embedding(s) (always on in the default suite)")
    ap.add_argument("--selftest", action="store_true", help="on a box with >= 2 GPUs: cfg3 strong scaling at N = 2 over RCCL with full parity, next to N = 1 (one JSON line); skipped on one GPU")
    args = ap.parse_args()
    if args.selftest:
        raise SystemExit(selftest(args))
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn_under_torchrun(args))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")

    quiet_stdout()
    if args.workload == "cfg5":
        wl = dict(WORKLOADS["cfg5"])
        if args.rows:
            wl["rows"] = args.rows
        run_cfg5(args, wl)
        return

    ctx = Ctx(args)
    suite = args.workload is None
    name = args.workload or ("cfg4" if (ctx.world > 1 and args.scaling == "weak") else "cfg3")
    wl = dict(WORKLOADS[name])
    if args.rows:
        wl["rows"] = args.rows
    if args.queries:
        wl["nq"] = args.queries
    if args.dim:
        wl["dim"] = args.dim
    weak = (name == "cfg4") or (ctx.world > 1 and args.scaling == "weak")
    scaling = "weak" if weak else "strong"
    if ctx.world == 1:
        scaling = "weak" if name == "cfg4" else "strong"
    if weak:
        wl["rows_total"] = wl["rows"] * ctx.world
        lo, hi = ctx.rank * wl["rows"], (ctx.rank + 1) * wl["rows"]
    else:
        wl["rows_total"] = wl["rows"]
        lo, hi = shard_bounds(wl["rows"], ctx.world, ctx.rank)
    steps = args.steps if args.steps is not None else (200 if wl["nq"] == 1 else 10)
    warmup = args.warmup if args.warmup is not None else (20 if wl["nq"] == 1 else 2)

    torch = ctx.torch
    stream_ctx = torch.cuda.stream(ctx.backend.stream) if ctx.backend is not None else torch.cuda.stream(torch.cuda.current_stream(ctx.dev))
    with stream_ctx:
        corpus = gen_rows(ctx.eng, lo, hi, wl["dim"], wl["seed"], wl["dtype"], wl.get("kind", "gaussian"), wl["rows_total"], wl.get("cluster_rows", CLUSTER_ROWS))
    rec = run_record(ctx, name, wl, corpus, lo, steps, warmup, with_cpu=(ctx.world == 1))
    if ctx.world == 1 and not ctx.distributed and ctx.rank == 0 and not any(o.startswith("mfma_ablate") for o in args.opt):
        if rec["roofline"]["bound"] == "mfma" and rec["roofline"]["kernel"] == "mfma_scan_kernel" and wl["dtype"] == "fp16" and not args.no_calibration:
            rec["roofline"]["sustained"] = sustained_calibration(ctx, wl, corpus, rec["roofline"]["kernel_ms_per_step"])
        if suite or args.class_api:
            rec["class_api"] = class_api_rates(ctx, wl, corpus, args.min_score, 5 if wl["nq"] > 1 else 50)

    if ctx.distributed and wl["nq"] > 1 and (suite or args.class_api):
        api = sharded_class_api_rates(ctx, wl, corpus, lo, args.min_score, 5)
        if ctx.rank == 0:
            rec["class_api"] = api

    sub = None
    sub_skipped = []
    if suite and not ctx.distributed and not args.no_sub:
        sub = {}
        # the single-query target on the same 10M-row corpus
        w2 = dict(WORKLOADS["cfg3_q1"], rows_total=wl["rows_total"], rows=wl["rows"])
        sub["cfg3_q1"] = run_record(ctx, "cfg3_q1", w2, corpus, 0, 40, 5, with_cpu=False)
        # middle batch sizes on the same corpus (what a batched `lookup_terms` lands on: storage/sqlite/reltermsindex.py:259-271): one HBM pass
        # serves the whole batch -- 32 queries on the 32/64-query tile, 128 on the 128-query tile
        for mid in ("cfg3_b32", "cfg3_b128"):
            wm = dict(WORKLOADS[mid], rows_total=wl["rows_total"], rows=wl["rows"])
            sub[mid] = run_record(ctx, mid, wm, corpus, 0, 20, 3, with_cpu=False)
        # the subset form at bench scale on the same corpus: 1M random ordinals of the 10M rows, the row list resident (+ through the class)
        ws = dict(WORKLOADS["cfg3_subset"], rows_total=wl["rows_total"], rows=wl["rows"])
        ws["subset"] = min(ws["subset"], max(1, wl["rows"] // 10))
        sub["cfg3_subset"] = run_record(ctx, "cfg3_subset", ws, corpus, 0, 40, 5, with_cpu=False)
        sub["cfg3_subset"]["class_api"] = class_api_rates(ctx, ws, corpus, args.min_score, 20)
        del corpus
        torch.cuda.empty_cache()

        def group(label, fn):
            # the sub-records in the order of their weight (north-star targets and BASELINE configs first, then the regime checks): a box so slow that
            # the suite would not end within --budget-seconds drops the LAST groups and says so (`sub_skipped`) instead of running into whatever
            # limit the caller has.  The headline and the records on its corpus (above) always run.
            spent = time.perf_counter() - _PROCESS_T0
            if spent > args.budget_seconds:
                sub_skipped.append(label)
                sys.stderr.write(f"bench.py: {label}: skipped, {spent:.0f} s spent of --budget-seconds {args.budget_seconds:g}\n")
                return
            fn()
            torch.cuda.empty_cache()
            sys.stderr.write(f"bench.py: {label} done, {time.perf_counter() - _PROCESS_T0:.0f} s since start\n")

        def g_cfg2():
            # (the two small corpora next: measured right after big ones have come and gone, cfg2 reads 6 % slower -- where its 6 GB land in HBM)
            w3 = dict(WORKLOADS["cfg2"])
            w3["rows_total"] = w3["rows"]
            c2 = gen_rows(ctx.eng, 0, w3["rows"], w3["dim"], w3["seed"], w3["dtype"])
            sub["cfg2"] = run_record(ctx, "cfg2", w3, c2, 0, 100, 10, with_cpu=True)
            # the same lookup at the reference's related-terms threshold (min_score 0.85, knowpro/convsettings.py:61-63): nothing of a gaussian
            # corpus survives it -- the scan without any selection work
            sub["cfg2_ms085"] = run_record(ctx, "cfg2", dict(w3, min_score=0.85), c2, 0, 100, 10, with_cpu=False)
            wb = dict(WORKLOADS["cfg2_b32"], rows_total=w3["rows"])
            sub["cfg2_b32"] = run_record(ctx, "cfg2_b32", wb, c2, 0, 40, 5, with_cpu=False)
            del c2

        def g_cfg1():
            # BASELINE config 1: the reference's own scale (10k x 1536 fp32, one query, top-10) -- launch-bound; with the class-level rate
            w1 = dict(WORKLOADS["cfg1"])
            w1["rows_total"] = w1["rows"]
            c1 = gen_rows(ctx.eng, 0, w1["rows"], w1["dim"], w1["seed"], w1["dtype"])
            sub["cfg1"] = run_record(ctx, "cfg1", w1, c1, 0, 500, 50, with_cpu=True)
            sub["cfg1"]["class_api"] = class_api_rates(ctx, w1, c1, args.min_score, 500)
            # the reference script's third row on the same corpus: fuzzy_lookup_embedding_in_subset, 1000 of 10k (tools/benchmark_vectorbase.py:133-163)
            w1s = dict(WORKLOADS["cfg1_subset"], rows_total=w1["rows"], cpu_seconds=6.0)
            sub["cfg1_subset"] = run_record(ctx, "cfg1_subset", w1s, c1, 0, 500, 50, with_cpu=True)
            sub["cfg1_subset"]["class_api"] = class_api_rates(ctx, w1s, c1, args.min_score, 500)
            del c1
            # batched related-term lookups at the reference's scale (32 terms, 1294 rows); `variants`: the same call with the grouped form off
            # (the 32-query tile: the routing until the end of round 6) and as 32 sequential single lookups
            wt = dict(WORKLOADS["cfg1_terms32"])
            wt.update(rows_total=wt["rows"], cpu_seconds=4.0)
            ct = gen_rows(ctx.eng, 0, wt["rows"], wt["dim"], wt["seed"], wt["dtype"], "aniso", wt["rows"])
            sub["cfg1_terms32"] = run_record(ctx, "cfg1_terms32", wt, ct, 0, 300, 30, with_cpu=True)
            sub["cfg1_terms32"]["variants"] = terms_variants(ctx, wt, ct)
            del ct
            # ... and the width that script defaults to (--dim 384, :55-76)
            w1d = dict(WORKLOADS["cfg1_d384"])
            w1d.update(rows_total=w1d["rows"], cpu_seconds=6.0)
            c1d = gen_rows(ctx.eng, 0, w1d["rows"], w1d["dim"], w1d["seed"], w1d["dtype"])
            sub["cfg1_d384"] = run_record(ctx, "cfg1_d384", w1d, c1d, 0, 500, 50, with_cpu=True)
            del c1d
            # ... and that script's first row, 1k vectors (1.5 MB): the CPU answers out of its cache in less time than a kernel launch + synchronise
            # takes -- reported for what it is (the product has no CPU path to fall back to)
            w1k = dict(WORKLOADS["cfg1_1k_d384"])
            w1k.update(rows_total=w1k["rows"], cpu_seconds=4.0)
            c1k = gen_rows(ctx.eng, 0, w1k["rows"], w1k["dim"], w1k["seed"], w1k["dtype"])
            sub["cfg1_1k_d384"] = run_record(ctx, "cfg1_1k_d384", w1k, c1k, 0, 500, 50, with_cpu=True)
            del c1k
            torch.cuda.empty_cache()

        def g_cfg4():
            # one GPU's shard of cfg4 (100M rows over 8 GPUs = 12.5M rows each): the per-GPU work of the weak-scaling config, as a corpus of its own
            w4 = dict(WORKLOADS["cfg4"])
            w4["rows_total"] = w4["rows"]
            c4 = gen_rows(ctx.eng, 0, w4["rows"], w4["dim"], w4["seed"], w4["dtype"])
            sub["cfg4_shard"] = run_record(ctx, "cfg4", w4, c4, 0, 10, 2, with_cpu=False)
            # (rank 0's 12.5M-row shard of cfg4 searched on its own: no exchange)
            del c4
            torch.cuda.empty_cache()

        def g_cfg5():
            # cfg5: the fused multi-index user query (4 term lookups k=50@0.85 + message re-rank k=25@0.7 + thread lookup k=10@0.7; convsettings.py:61-67)
            # + the memory provider's 1000-ordinal message subset and the six lookups as separate synchronous calls, on the same corpora
            sub["cfg5"] = run_cfg5(args, dict(WORKLOADS["cfg5"]), emit=False, steps=20, warmup=3, variants=True)

        def g_d3072():
            # the 3072-wide model of the reference's table (text-embedding-3-large, vectorbase.py:31-35): cfg2's and cfg3's shapes at that width
            wd = dict(WORKLOADS["cfg2_d3072"])
            wd["rows_total"] = wd["rows"]
            cw = gen_rows(ctx.eng, 0, wd["rows"], wd["dim"], wd["seed"], wd["dtype"])
            sub["cfg2_d3072"] = run_record(ctx, "cfg2_d3072", wd, cw, 0, 60, 8, with_cpu=False)
            del cw
            torch.cuda.empty_cache()
            wd = dict(WORKLOADS["cfg3_d3072"])
            wd["rows_total"] = wd["rows"]
            cw = gen_rows(ctx.eng, 0, wd["rows"], wd["dim"], wd["seed"], wd["dtype"])
            sub["cfg3_d3072"] = run_record(ctx, "cfg3_d3072", wd, cw, 0, 10, 2, with_cpu=False)
            sub["cfg3_d3072_q1"] = run_record(ctx, "cfg3_d3072_q1", dict(WORKLOADS["cfg3_d3072_q1"], rows_total=wd["rows"]), cw, 0, 40, 5, with_cpu=False)
            del cw
            torch.cuda.empty_cache()

        def g_aniso():
            # a real-embedding-like corpus (one common direction, mean pairwise cosine 0.75) at the reference's threshold 0.85: most rows survive
            wa = dict(WORKLOADS["cfg3_aniso"])
            wa["rows_total"] = wa["rows"]
            ca = gen_rows(ctx.eng, 0, wa["rows"], wa["dim"], wa["seed"], wa["dtype"], "aniso", wa["rows"])
            sub["cfg3_aniso"] = run_record(ctx, "cfg3_aniso", wa, ca, 0, 10, 2, with_cpu=False)
            sub["cfg3_aniso"]["vs_gaussian"] = sub["cfg3_aniso"]["queries_per_sec"] / rec["queries_per_sec"]
            sub["cfg3_aniso_q1"] = run_record(ctx, "cfg3_aniso_q1", dict(WORKLOADS["cfg3_aniso_q1"], rows_total=wa["rows"]), ca, 0, 40, 5, with_cpu=False)
            del ca
            torch.cuda.empty_cache()

        def g_clustered():
            # the headline shape on a clustered corpus (near-duplicate clusters + exact duplicates, queries next to cluster centres): what the
            # wide tile's band selection is there for; `vs_gaussian` = its rate over the headline's
            wc = dict(WORKLOADS["cfg3_clustered"])
            wc["rows_total"] = wc["rows"]
            cc = gen_rows(ctx.eng, 0, wc["rows"], wc["dim"], wc["seed"], wc["dtype"], "clustered", wc["rows"])
            sub["cfg3_clustered"] = run_record(ctx, "cfg3_clustered", wc, cc, 0, 10, 2, with_cpu=False)
            sub["cfg3_clustered"]["vs_gaussian"] = sub["cfg3_clustered"]["queries_per_sec"] / rec["queries_per_sec"]
            del cc
            torch.cuda.empty_cache()

        def g_dup():
            # the duplication cliff: 1500-row clusters -- more near-duplicates than a band holds, EVERY query ends up on the 256-query tile's exact
            # split-plane form (flagged_fraction 1.0).  The library finds that out before the filter's last phase and skips it (early_exact):
            # vs_gaussian ~1/2 = the filter's first phases + an exact pass of twice the MFMAs (round 4 before that: ~1/3, round 3: 1/22)
            wd = dict(WORKLOADS["cfg3_dup"])
            wd["rows_total"] = wd["rows"]
            cd = gen_rows(ctx.eng, 0, wd["rows"], wd["dim"], wd["seed"], wd["dtype"], "clustered", wd["rows"], wd["cluster_rows"])
            sub["cfg3_dup"] = run_record(ctx, "cfg3_dup", wd, cd, 0, 4, 1, with_cpu=False)
            sub["cfg3_dup"]["vs_gaussian"] = sub["cfg3_dup"]["queries_per_sec"] / rec["queries_per_sec"]
            del cd
            torch.cuda.empty_cache()

        for label, fn in (("cfg2 group", g_cfg2), ("cfg1 group", g_cfg1), ("cfg4_shard", g_cfg4), ("cfg5", g_cfg5), ("3072-wide group", g_d3072),
                          ("cfg3_aniso group", g_aniso), ("cfg3_clustered", g_clustered), ("cfg3_dup", g_dup)):
            group(label, fn)
    elif suite and ctx.distributed and not args.no_sub:  # (also the one-rank dry run of this code, TAVB_BENCH_FORCE_DIST=1)
        # N > 1: beside the strong-scaling headline, BASELINE configs[3] as the north star words it -- 12.5M rows PER GPU (100M rows at N = 8),
        # the same 1024-query batches; `row_queries_per_sec` is the weak-scaling figure (rows x queries per second over all ranks)
        del corpus
        torch.cuda.empty_cache()
        w4 = dict(WORKLOADS["cfg4"])
        if args.rows:
            w4["rows"] = args.rows
        w4["rows_total"] = w4["rows"] * ctx.world
        lo4 = ctx.rank * w4["rows"]
        with stream_ctx:
            c4 = gen_rows(ctx.eng, lo4, lo4 + w4["rows"], w4["dim"], w4["seed"], w4["dtype"], "gaussian", w4["rows_total"])
        r4 = run_record(ctx, "cfg4", w4, c4, lo4, 10, 2, with_cpu=False)
        if ctx.rank == 0:
            r4["row_queries_per_sec"] = r4["queries_per_sec"] * w4["rows_total"]
            r4["scaling"] = "weak"
            sub = {"cfg4_weak": r4}
        del c4
        torch.cuda.empty_cache()
    ok = True
    if ctx.rank == 0:
        line = headline_line(ctx, rec, name, wl, scaling, sub)
        if sub_skipped:
            line["sub_skipped"] = {"groups": sub_skipped, "why": f"--budget-seconds {args.budget_seconds:g} spent before they would have started"}
        emit_result(line)
        checks = [rec.get("parity")] + [r.get("parity") for r in (sub or {}).values()]
        ok = all(c is None or c.get("ok") for c in checks)
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()
    if not ok:
        sys.stderr.write("bench.py: PARITY CHECK FAILED (see the `parity` objects in the line above)\n")
        raise SystemExit(3)


if __name__ == "__main__":
    main()
