#!/usr/bin/env python3
"""Benchmark of the VectorBase kNN hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg2_f16|cfg2_b32|cfg2_b1024|cfg3|cfg3_q1|cfg4|cfg5|cfg1]

Contract (one JSON line on stdout from rank 0):
  * a "step" is one lookup pass of the hot path over the resident corpus:
      cfg2 (default, BASELINE.json configs[1]): 1M x 1536 fp32 corpus, ONE query, top-32,
           through the synchronous C-ABI call `tavb_search` (query H2D + scan + merge +
           result D2H + sync) -- what a `VectorBase.fuzzy_lookup_embedding` caller sees;
      cfg3 (configs[2]): 10M x 1536 fp16 corpus, a 1024-query batch, top-32 (256-query MFMA tile);
      cfg4 (configs[3]): cfg3 at 12.5M rows per GPU (100M rows when run with --gpus 8);
      cfg2_b32 / cfg2_b1024: 32 / 1024-query batches on the cfg2 fp32 corpus (32-query fp32 MFMA tile);
      cfg5 (configs[4]): fused multi-index user query; cfg1 (configs[0]): the reference's own 10k-row case.
  * value = queries/sec over the timed K steps (wall clock, barrier + synchronize on both
    sides, max over ranks).  With --gpus N the corpus is row-sharded, 1M (cfg2) / 10M (cfg3)
    rows PER GPU (weak scaling: total rows = N x that), per-shard top-k lists are
    all-gathered over RCCL and merged on every rank; `value` then counts shard-scans,
    i.e. queries/s x N (rows scanned per second / rows per shard), and the plain
    end-to-end rate is reported beside it as `queries_per_sec`.
  * roofline: algorithmic bytes (cfg2: N*D*4 per query) or flops (cfg3: 2*Q*N*D per batch)
    divided by the scan kernel's mean duration measured with HIP events on the stream the
    kernel runs on (libtavb's profile API), against 8 TB/s HBM / 2.5 PFLOP/s dense fp16 MFMA.
  * cpu_baseline: the numpy oracle (a restatement of the reference's VectorBase arithmetic,
    oracle/vectorbase_oracle.py) timed on this box's host cores on the same corpus.
Synthetic data: gaussian rows, L2-normalised on the device (no dataset exists for this path).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3   # fp32 matrix rate (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md peak table

WORKLOADS = {
    #            rows/GPU    dim   dtype   queries/step  k
    "cfg2": dict(rows=1_000_000, dim=1536, dtype="fp32", nq=1, k=32, bound="hbm"),
    "cfg2_f16": dict(rows=1_000_000, dim=1536, dtype="fp16", nq=1, k=32, bound="hbm"),
    "cfg3": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=1024, k=32, bound="mfma"),
    # the north star's single-query target on the cfg3 corpus: HBM-bound, 30.72 GB per query
    "cfg3_q1": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=1, k=32, bound="hbm"),
    "cfg4": dict(rows=12_500_000, dim=1536, dtype="fp16", nq=1024, k=32, bound="mfma"),  # x8 GPUs = 100M rows
    "cfg1": dict(rows=10_000, dim=1536, dtype="fp32", nq=1, k=10, bound="hbm"),
    # batches on the reference's own dtype (fp32): the 32-query MFMA tile.  32 queries ride one HBM pass; 1024 are bound by
    # the fp32 matrix rate.
    "cfg2_b32": dict(rows=1_000_000, dim=1536, dtype="fp32", nq=32, k=32, bound="hbm"),
    "cfg2_b1024": dict(rows=1_000_000, dim=1536, dtype="fp32", nq=1024, k=32, bound="mfma"),
    # fused multi-index user query: 4 term lookups k=50@0.85 on a 10M-row terms corpus + 1 message
    # re-rank k=25@0.7 on a 10M-row message corpus (full scan, or --cfg5-subset 1000) + 1 thread lookup k=10@0.7 on 1k rows
    "cfg5": dict(rows=10_000_000, dim=1536, dtype="fp16", nq=6, k=50, bound="hbm"),
}


def make_device_corpus(eng, rows: int, dim: int, seed: int, dtype: str, chunk: int = 262_144):
    """Gaussian rows generated on the device, normalised by our K1 kernel (and rounded to
    fp16 by our convert kernel).  The host never holds more than it asks for."""
    import torch

    dev = torch.device("cuda", eng.device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    out = torch.empty((rows, dim), dtype=torch.float16 if dtype == "fp16" else torch.float32, device=dev)
    for lo in range(0, rows, chunk):
        hi = min(rows, lo + chunk)
        if dtype == "fp16":
            tmp = torch.empty((hi - lo, dim), dtype=torch.float32, device=dev)
            tmp.normal_(generator=gen)
            eng.normalize_rows_(tmp)
            out[lo:hi].copy_(eng.to_f16(tmp))
            del tmp
        else:
            view = out[lo:hi]
            view.normal_(generator=gen)
            eng.normalize_rows_(view)
    torch.cuda.synchronize(dev)
    return out


def host_queries(count: int, dim: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((count, dim)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def cpu_baseline(corpus_host: np.ndarray, queries: np.ndarray, k: int, budget_s: float = 15.0) -> dict:
    from oracle import vectorbase_oracle as vo

    cores = len(os.sched_getaffinity(0))
    vo.lookup(corpus_host, queries[0], k, 0.0)  # warm-up (BLAS thread pool, page-in)
    times = []
    t_end = time.perf_counter() + budget_s
    i = 0
    while (time.perf_counter() < t_end and i < 400) or i < 3:
        q = queries[i % len(queries)]
        t0 = time.perf_counter_ns()
        vo.lookup(corpus_host, q, k, 0.0)
        times.append((time.perf_counter_ns() - t0) / 1e9)
        i += 1
    med = float(np.median(times))
    one_thread = None
    try:  # also report the reference arithmetic on one core
        from threadpoolctl import threadpool_limits

        with threadpool_limits(limits=1, user_api="blas"):
            t1 = []
            for j in range(3):
                t0 = time.perf_counter_ns()
                vo.lookup(corpus_host, queries[j % len(queries)], k, 0.0)
                t1.append((time.perf_counter_ns() - t0) / 1e9)
        one_thread = {"value": 1.0 / float(np.median(t1)), "unit": "queries/s", "cores": 1, "sample": f"3 lookups, median {np.median(t1) * 1e3:.1f} ms"}
    except Exception:
        pass
    return {
        "value": 1.0 / med,
        "unit": "queries/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{len(times)} sequential single-query lookups on the same {corpus_host.shape[0]}x{corpus_host.shape[1]} fp32 corpus "
                  f"(numpy {np.__version__} / OpenBLAS sgemv, default threads), median {med * 1e3:.2f} ms, min {min(times) * 1e3:.2f} ms",
        "p50_ms": med * 1e3,
        "one_thread": one_thread,
    }


def run_cfg5(args, wl) -> None:
    """BASELINE config 5: user queries/s of the fused multi-index submission (typeagent_py_amd/fused.py)."""
    import torch

    from typeagent_py_amd import TextEmbeddingIndexSettings, VectorBase, _native
    from typeagent_py_amd.fused import FusedIndexQuery

    rows, dim = wl["rows"], wl["dim"]
    steps = args.steps if args.steps is not None else 30
    warmup = args.warmup if args.warmup is not None else 3
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
    subset = np.random.default_rng(99).choice(rows, size=args.cfg5_subset, replace=False).tolist() if args.cfg5_subset else None

    if args.cfg5_separate:
        class _Null:
            model_name = "bench"
        vbs = []
        for t in (terms, msgs, threads):
            vb = VectorBase(TextEmbeddingIndexSettings(_Null()), device=0)
            vb.adopt_device_corpus(t)
            vbs.append(vb)

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

    for i in range(warmup):
        one(i)
    if not args.cfg5_separate:
        eng.profile_enable(True)
        eng.profile_reset()
    torch.cuda.synchronize()
    lat = []
    t0 = time.perf_counter()
    for i in range(steps):
        s0 = time.perf_counter_ns()
        one(warmup + i)
        lat.append((time.perf_counter_ns() - s0) / 1e3)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    esize = 2 if wl["dtype"] == "fp16" else 4
    alg = rows * dim * esize + (len(subset) if subset else rows) * dim * esize + 1000 * dim * esize
    out = {
        "metric": "user queries/sec + p50 latency, fused multi-index VectorBase lookups (cfg5)",
        "value": steps / elapsed, "unit": "user-queries/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 storage, f32 accumulate", "data": "synthetic",
        "config": {"workload": f"cfg5: 4 term lookups k=50@0.85 on {rows}x{dim} + message re-rank k=25@0.7 "
                               f"({'subset of ' + str(len(subset)) if subset else 'full scan'}) on {rows}x{dim} + thread lookup k=10@0.7 on 1000x{dim}; "
                               f"{'six separate synchronous calls' if args.cfg5_separate else 'one fused submission'}"},
        "p50_latency_us": float(np.percentile(lat, 50)), "p99_latency_us": float(np.percentile(lat, 99)),
        "roofline": {"bound": "hbm", "achieved": alg / (elapsed / steps) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (elapsed / steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                     "note": "whole user query (all kernels + copies + one sync) against the bytes the three corpora passes must read"},
        "cpu_baseline": None,
    }
    if not args.cfg5_separate:
        ms, n = eng.profile_read(_native.KERNEL_SCAN)
        out["roofline"]["scan_kernel_ms_per_user_query"] = ms / steps
        out["roofline"]["scan_launches_per_user_query"] = n / steps
    print(json.dumps(out))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--rows", type=int, default=None, help="override rows per GPU (debugging)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--tiled", action="store_true", help="cfg3: also build the K-blocked fp16 image and feed the MFMA kernel from it (measured: no gain)")
    ap.add_argument("--cfg5-subset", type=int, default=0, help="cfg5: message re-rank over a subset of this many ordinals (0 = full scan)")
    ap.add_argument("--cfg5-separate", action="store_true", help="cfg5: issue the six lookups as separate synchronous calls")
    ap.add_argument("--min-score", type=float, default=0.0, help="score threshold of the lookups (0.0 = every row survives: worst case for selection; 0.85 = the reference's related-terms default)")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (e.g. scan_unroll=4)")
    args = ap.parse_args()

    import torch

    from typeagent_py_amd import _native

    wl = dict(WORKLOADS[args.workload])
    if args.rows:
        wl["rows"] = args.rows
    if args.workload == "cfg5":
        run_cfg5(args, wl)
        return
    steps = args.steps if args.steps is not None else (200 if wl["nq"] == 1 else 10)
    warmup = args.warmup if args.warmup is not None else (20 if wl["nq"] == 1 else 2)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("TAVB_BENCH_FORCE_DIST") == "1"  # the latter: 1-rank dry run of the N>1 code
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and args.gpus > 1:
        sys.stderr.write(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run\n")
    dev = local_rank if distributed else 0
    torch.cuda.set_device(dev)

    rows, dim, k, nq = wl["rows"], wl["dim"], wl["k"], wl["nq"]
    min_score = args.min_score
    thr = float(_native.f32_threshold(min_score))
    queries = host_queries(max(64, nq), dim, 4242)  # identical on every rank

    if distributed:
        from typeagent_py_amd.sharded import DeviceShardBackend, ShardedSearcher

        backend = DeviceShardBackend(dev)
        eng = backend.engine
        with torch.cuda.stream(backend.stream):
            corpus = make_device_corpus(eng, rows, dim, 100_043 + rank, wl["dtype"])
        backend.set_shard(corpus, row_offset=rank * rows)
        if wl["bound"] == "mfma" and args.tiled:
            with torch.cuda.stream(backend.stream):
                eng.build_tiled()
        searcher = ShardedSearcher(backend, always_collective=True)
    else:
        eng = _native.Engine(dev)
        corpus = make_device_corpus(eng, rows, dim, 1043, wl["dtype"])
        eng.set_corpus_tensor(corpus)
        if wl["bound"] == "mfma" and args.tiled:
            eng.build_tiled()  # K-blocked fp16 image beside the row-major corpus: what the MFMA kernel streams
        searcher = None
    for item in args.opt:
        name, val = item.split("=")
        eng.set_option(name, int(val))

    dq_all = torch.from_numpy(queries).to(torch.device("cuda", dev))
    torch.cuda.synchronize(dev)

    def one_step(i: int):
        if nq == 1:
            qi = i % len(queries)
            if searcher is None:
                return eng.search(queries[qi], k, np.float32(thr))
            return searcher.search(dq_all[qi:qi + 1], k, min_score)
        if searcher is None:
            return eng.search_batch(queries[:nq], k, np.float32(thr))
        return searcher.search(dq_all[:nq], k, min_score)

    for i in range(warmup):
        one_step(i)
    eng.profile_enable(True)
    eng.profile_reset()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    lat = []
    t0 = time.perf_counter()
    for i in range(steps):
        s0 = time.perf_counter_ns()
        one_step(warmup + i)
        lat.append((time.perf_counter_ns() - s0) / 1e3)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the kernel family that served the steps: 256-query MFMA tile, 32-query MFMA tile, or a streaming tier
    kid = _native.KERNEL_SCAN
    for cand in (_native.KERNEL_MFMA, _native.KERNEL_SKINNY):
        if eng.profile_read(cand)[1]:
            kid = cand
    kern_ms, kern_n = eng.profile_read(kid)
    if kid != _native.KERNEL_SCAN:  # the earlier phases of the threshold ladder are part of the same job: charge their time
        s_ms, _ = eng.profile_read(_native.KERNEL_MFMA_SAMPLE)
        kern_ms += s_ms
    merge_ms, merge_n = eng.profile_read(_native.KERNEL_MERGE)
    eng.profile_enable(False)

    if rank == 0:
        qps = steps * nq / elapsed
        esize = 2 if wl["dtype"] == "fp16" else 4
        avg_kernel_s = (kern_ms / max(kern_n, 1)) * 1e-3
        launches_per_step = kern_n / steps
        if wl["bound"] == "hbm":
            alg = rows * dim * esize  # bytes one launch must read: the shard once
            achieved = alg / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None}
        else:
            nq_per_launch = nq / max(launches_per_step, 1e-9)
            alg = 2.0 * nq_per_launch * rows * dim
            achieved = alg / avg_kernel_s / 1e12 if avg_kernel_s > 0 else 0.0
            peak = MFMA_F16_PEAK_TFLOPS if wl["dtype"] == "fp16" else MFMA_F32_PEAK_TFLOPS
            roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None}
        try:  # HBM traffic comes from a separate rocprofv3 --pmc pass (bench.py cannot count it itself)
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f).get(args.workload)
            if pmc and pmc.get("traffic_bytes_per_launch") and not args.rows:
                roof["traffic"] = pmc["traffic_bytes_per_launch"] / (1e9 if wl["bound"] == "hbm" else 1.0)
                roof["traffic_unit"] = "GB per launch" if wl["bound"] == "hbm" else "B per launch"
                roof["traffic_source"] = pmc["source"]
        except Exception:
            pass
        roof["kernel"] = {0: "scan (tavb::scan_*_kernel)", 2: "mfma (tavb::mfma_scan_kernel_v3 for the small ladder phases, _v5 for the big ones)",
                          6: "skinny (tavb::skinny_scan_kernel)"}.get(kid, str(kid))
        roof["kernel_avg_ms"] = avg_kernel_s * 1e3
        roof["kernel_launches"] = kern_n
        roof["algorithmic_per_launch"] = alg
        roof["merge_avg_us"] = (merge_ms / max(merge_n, 1)) * 1e3

        out = {
            "metric": "queries/sec + p50 lookup latency, 1536-d top-32 kNN",
            "value": qps * world,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if wl["dtype"] == "fp32" else "f16 storage, f32 accumulate",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {rows}x{dim} {wl['dtype']} rows per GPU, {nq} quer{'y' if nq == 1 else 'ies'}/step, top-{k}, min_score {min_score}",
                "rows_per_gpu": rows,
                "total_rows": rows * world,
                "queries_per_step": nq,
                "k": k,
                "parallelism": f"row-sharded x{world}, RCCL all-gather of per-shard top-k + merge" if world > 1 else "single GPU",
                "value_counts": "queries/s x n_gpus (shard scans per second)" if world > 1 else "queries/s",
            },
            "queries_per_sec": qps,
            "p50_latency_us": float(np.percentile(lat, 50)),
            "p99_latency_us": float(np.percentile(lat, 99)),
            "min_latency_us": float(np.min(lat)),
            "roofline": roof,
        }
        if not args.no_cpu_baseline and not distributed:
            n_host = min(rows, 1_000_000)
            host = corpus[:n_host].float().cpu().numpy()
            # the GPU answer for query 0 must be the oracle's answer on the same bytes
            from oracle import vectorbase_oracle as vo

            o, s = eng.search(queries[0], k, np.float32(thr)) if n_host == rows else (None, None)
            if o is not None:
                vo.check_topk_parity(vo.scores_full(host, queries[0]), o.tolist(), s.tolist(), k, min_score)
                out["parity_check"] = "query 0: top-k ordinals/scores match the oracle on the same corpus bytes"
            base = cpu_baseline(host, queries, k, args.cpu_seconds)
            if n_host != rows:
                base["sample"] += f"; first {n_host} of {rows} rows only"
            out["cpu_baseline"] = base
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
