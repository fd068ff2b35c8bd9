#!/usr/bin/env python3
"""Drive the C ABI without torch (torch's CUDA init does not survive the ASan runtime): device buffers come from hipMalloc through
ctypes.  Meant for `libtavb_debug.so` (host side under AddressSanitizer + UBSan, `make -C typeagent_py_amd/csrc debug`), see
`tools/gpu.sh LABEL asan`.  Checks answers against numpy on the way (test infrastructure, not product)."""
import ctypes
import os
import sys
from ctypes import POINTER, byref, c_float, c_int, c_int32, c_int64, c_void_p

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from typeagent_py_amd import _native  # noqa: E402

lib = _native.load_library(preload_torch=False)
hip = ctypes.CDLL("libamdhip64.so", mode=ctypes.RTLD_GLOBAL)
hip.hipMalloc.argtypes = [POINTER(c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [c_void_p]
hip.hipMemcpy.argtypes = [c_void_p, c_void_p, ctypes.c_size_t, c_int]


def ok(rc):
    if rc:
        raise RuntimeError(lib.tavb_last_error().decode())


def dmalloc(n):
    p = c_void_p()
    assert hip.hipMalloc(byref(p), n) == 0
    return p


def ptr(a):
    return a.ctypes.data_as(c_void_p)


def scores(v, q):
    return np.clip((v @ q + np.float32(1)) * np.float32(0.5), 0, 1).astype(np.float32)


def main():
    rng = np.random.default_rng(7)
    n, d, k = 70_003, 1536, 32
    v = rng.standard_normal((n, d)).astype(np.float32)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    h = c_void_p()
    bad = []
    ok(lib.tavb_create(0, None, byref(h)))
    for dtype, esz in ((_native.TAVB_F32, 4), (_native.TAVB_F16, 2)):
        vv = v if dtype == _native.TAVB_F32 else v.astype(np.float16).astype(np.float32)
        dev = dmalloc(n * d * esz)
        ok(lib.tavb_upload_rows(h, ptr(v), n, d, dev, dtype))  # pinned ring + staging threads + on-device conversion
        ok(lib.tavb_set_corpus(h, dev, n, d, dtype, 0))
        ok(lib.tavb_corpus_modified(h, 0))
        # one query, every tier of batch (streaming, 32/64-query tile, 128/256-query tile + rescoring)
        for nq in (1, 3, 20, 64, 100, 300):
            q = rng.standard_normal((nq, d)).astype(np.float32)
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            o = np.empty((nq, k), np.int64); s = np.empty((nq, k), np.float32); c = np.empty(nq, np.int32)
            thr = np.zeros(nq, np.float32)
            ok(lib.tavb_search_batch(h, ptr(q), nq, k, ptr(thr), ptr(o), ptr(s), ptr(c)))
            for qi in (0, nq - 1):
                ref = np.argsort(-scores(vv, q[qi]), kind="stable")[:k]
                if not (c[qi] == k and (o[qi] == ref).mean() > 0.9):
                    tier = c_int64(); lib.tavb_get_option(h, b"last_tier", byref(tier))
                    print("MISMATCH", dtype, nq, qi, "count", c[qi], "tier", tier.value, "\n got", o[qi][:8], s[qi][:8], "\n ref", ref[:8], scores(vv, q[qi])[ref[:8]], flush=True)
                    bad.append((dtype, nq, qi))
            keys = np.empty((nq, k), np.uint64)
            ok(lib.tavb_search_begin(h, ptr(q), nq, k, ptr(thr), None))
            ok(lib.tavb_search_end(h, nq, k, ptr(keys)))
            two = np.stack([keys, keys])
            merged = np.empty((nq, k), np.uint64)
            ok(lib.tavb_merge_keys_host(ptr(two), 2, nq, k, ptr(merged)))
            o2 = np.empty((nq, k), np.int64); s2 = np.empty((nq, k), np.float32); c2 = np.empty(nq, np.int32)
            ok(lib.tavb_decode_keys(ptr(keys), nq, k, ptr(o2), ptr(s2), ptr(c2)))
            if not (o2 == o).all(): print("begin/end differs from search_batch at nq", nq, "counts", c2[:4], flush=True)
        # round 5: per-query thresholds and k beyond 64 on the tiles (k = 100: the wide tile's split-plane fallback on fp16), a 1024-query
        # batch (the query staging copy on several threads)
        for nq, kk in ((40, 32), (130, 100), (1024, 32)):
            if kk > 64 and dtype != _native.TAVB_F16:
                continue
            q = rng.standard_normal((nq, d)).astype(np.float32)
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            o = np.empty((nq, kk), np.int64); s = np.empty((nq, kk), np.float32); c = np.empty(nq, np.int32)
            thr = np.where(np.arange(nq) % 3 == 0, 0.0, 0.52).astype(np.float32)
            thr[1] = np.nan
            ok(lib.tavb_search_batch(h, ptr(q), nq, kk, ptr(thr), ptr(o), ptr(s), ptr(c)))
            for qi in (0, 2, nq - 1):
                sc = scores(vv, q[qi])
                want = min(kk, int((sc >= thr[qi]).sum()))
                ref = np.argsort(-sc, kind="stable")[:want]
                if not (c[qi] == want and (want == 0 or (o[qi][:want] == ref).mean() > 0.9)):
                    print("MISMATCH per-query thresholds", dtype, nq, kk, qi, "count", c[qi], "want", want, flush=True)
                    bad.append((dtype, "thr", nq, qi))
            if c[1] != 0:
                print("MISMATCH: a NaN threshold let rows pass", flush=True); bad.append((dtype, "nan", nq))
        print("per-query thresholds / k = 100 / 1024-query batch: ok", flush=True)
        q = v[5].copy()
        o = np.empty(k, np.int64); s = np.empty(k, np.float32); cnt = c_int32()
        ok(lib.tavb_search(h, ptr(q), k, c_float(0.0), ptr(o), ptr(s), byref(cnt)))
        print("single search: count", cnt.value, "first", o[0], flush=True)
        # the one-launch form of the same lookup (round 4: per-workgroup lists into pinned memory, host merge) against the two-launch form
        two_launch = o.copy()
        ok(lib.tavb_set_option(h, b"small_direct_bytes", 1 << 30))
        for kk in (k, 1, 200):
            o3 = np.empty(kk, np.int64); s3 = np.empty(kk, np.float32)
            ok(lib.tavb_search(h, ptr(q), kk, c_float(0.0), ptr(o3), ptr(s3), byref(cnt)))
            dflag = c_int64(); ok(lib.tavb_get_option(h, b"last_direct", byref(dflag)))
            # (k = 200: the 8192-key list budget would cut the grid to 40 of 256 workgroups for 70k rows -- less than half: the one-launch path
            #  steps aside, round 5)
            assert dflag.value in ((0,) if kk == 200 else (1, 2)) and cnt.value == kk and (o3[: min(k, kk)] == two_launch[: min(k, kk)]).all(), (kk, dflag.value, cnt.value)
        # ... and a FEW queries at once through it (the multi-query scan's lists merged per query on the host), every k and count, per-query thresholds
        for nq3, kk in ((2, k), (3, 1), (4, 200), (8, k)):
            q3 = v[5 : 5 + nq3].copy()
            o3 = np.empty((nq3, kk), np.int64); s3 = np.empty((nq3, kk), np.float32); c3 = np.empty(nq3, np.int32)
            thr3 = np.linspace(0.0, 0.5, nq3).astype(np.float32)
            ok(lib.tavb_search_batch(h, ptr(q3), nq3, kk, ptr(thr3), ptr(o3), ptr(s3), ptr(c3)))
            dflag = c_int64(); ok(lib.tavb_get_option(h, b"last_direct", byref(dflag)))
            assert (o3[:, 0] == np.arange(5, 5 + nq3)).all() and (c3 >= 1).all(), (nq3, kk, dflag.value, o3[:, 0], c3)
        ok(lib.tavb_set_option(h, b"small_direct_bytes", 128 << 20))
        # end of round 6: batches of 2 .. 64 queries on a SMALL corpus (the first 3000 rows) as ONE grouped streaming launch -- lists of every
        # (query, row workgroup) into pinned memory, k-way merged on the host with the next query's lists prefetched -- every group size, per-query
        # thresholds, the answers of the single lookups bit for bit; and the device-resident form (scan + one merge launch)
        n_small = 3000
        ok(lib.tavb_set_corpus(h, dev, n_small, d, dtype, 0))
        for nq4, kk in ((5, 10), (24, 50), (64, 10), (64, 64), (33, 1)):
            q4 = rng.standard_normal((nq4, d)).astype(np.float32)
            q4 /= np.linalg.norm(q4, axis=1, keepdims=True)
            thr4 = np.linspace(0.0, 0.51, nq4).astype(np.float32)
            singles = []
            for qi in range(nq4):
                o1 = np.empty(kk, np.int64); s1 = np.empty(kk, np.float32); c1 = c_int32()
                ok(lib.tavb_search(h, ptr(q4[qi]), kk, c_float(float(thr4[qi])), ptr(o1), ptr(s1), byref(c1)))
                singles.append((o1[: c1.value].copy(), s1[: c1.value].copy()))
            for group in (0, 1, 2, 4, 8):
                ok(lib.tavb_set_option(h, b"direct_group", group))
                o4 = np.empty((nq4, kk), np.int64); s4 = np.empty((nq4, kk), np.float32); c4 = np.empty(nq4, np.int32)
                ok(lib.tavb_search_batch(h, ptr(q4), nq4, kk, ptr(thr4), ptr(o4), ptr(s4), ptr(c4)))
                dflag = c_int64(); ok(lib.tavb_get_option(h, b"last_direct", byref(dflag)))
                assert dflag.value == 3 or (group == 0 and kk == 64), (nq4, kk, group, dflag.value)  # (unforced: the cost model decides; 64 x 64 keys per workgroup tip it to the tiles)
                for qi in range(nq4):
                    m = c4[qi]
                    if dflag.value == 3 and not (m == len(singles[qi][0]) and (o4[qi, :m] == singles[qi][0]).all() and (s4[qi, :m] == singles[qi][1]).all()):
                        print("MISMATCH grouped batch", dtype, nq4, kk, group, qi, flush=True); bad.append((dtype, "grouped", nq4, kk, group, qi))
            ok(lib.tavb_set_option(h, b"direct_group", 0))
            dq4 = dmalloc(q4.nbytes); dk4 = dmalloc(nq4 * kk * 8)
            assert hip.hipMemcpy(dq4, ptr(q4), q4.nbytes, 1) == 0
            ok(lib.tavb_search_device(h, dq4, nq4, kk, c_float(0.0), dk4))
            ok(lib.tavb_synchronize(h))
            dflag = c_int64(); ok(lib.tavb_get_option(h, b"last_direct", byref(dflag)))
            keys4 = np.empty((nq4, kk), np.uint64)
            assert hip.hipMemcpy(ptr(keys4), dk4, keys4.nbytes, 2) == 0 and dflag.value in (0, 4), dflag.value
            o5 = np.empty((nq4, kk), np.int64); s5 = np.empty((nq4, kk), np.float32); c5 = np.empty(nq4, np.int32)
            ok(lib.tavb_decode_keys(ptr(keys4), nq4, kk, ptr(o5), ptr(s5), ptr(c5)))
            ref0 = np.argsort(-scores(vv[:n_small], q4[0]), kind="stable")[:kk]
            if not (c5[0] == min(kk, n_small) and (o5[0] == ref0).mean() > 0.9):
                print("MISMATCH grouped device form", dtype, nq4, kk, flush=True); bad.append((dtype, "grouped_dev", nq4, kk))
            hip.hipFree(dq4); hip.hipFree(dk4)
        ok(lib.tavb_set_corpus(h, dev, n, d, dtype, 0))
        print("grouped one-launch batches: ok", flush=True)
        bounds = (c_int64 * 16)()
        phases = lib.tavb_plan_ladder(10_000_000, 1024, 256, bounds, 16)
        assert phases >= 2 and bounds[0] == 0 and bounds[phases] == 10_000_000 and lib.tavb_plan_ladder(-1, 1024, 256, None, 0) < 0
        print("one-launch lookup + ladder plan: ok,", phases, "phases", flush=True)
        rows = rng.integers(0, n, 5000).astype(np.int64)
        ok(lib.tavb_search_subset(h, ptr(q), ptr(rows), rows.size, k, c_float(0.0), ptr(o), ptr(s), byref(cnt)))
        print("subset search: count", cnt.value, flush=True)
        # the same subset with its row list resident on the device (round 6: tavb_search_subset_resident)
        r32 = rows.astype(np.int32); drows = dmalloc(r32.size * 4)
        assert hip.hipMemcpy(drows, ptr(r32), r32.size * 4, 1) == 0
        o_r = np.empty(k, np.int64); s_r = np.empty(k, np.float32); cnt_r = c_int32()
        ok(lib.tavb_search_subset_resident(h, ptr(q), drows, r32.size, k, c_float(0.0), ptr(o_r), ptr(s_r), byref(cnt_r)))
        if not (cnt_r.value == cnt.value and (o_r[: cnt.value] == o[: cnt.value]).all() and (s_r[: cnt.value] == s[: cnt.value]).all()):
            print("MISMATCH resident subset", dtype, flush=True); bad.append((dtype, "subset_resident"))
        assert lib.tavb_search_subset_resident(h, ptr(q), None, r32.size, k, c_float(0.0), ptr(o_r), ptr(s_r), byref(cnt_r)) != 0
        hip.hipFree(drows)
        cap = 4096
        oa = np.empty(cap, np.int64); sa = np.empty(cap, np.float32); got = c_int64(); tot = c_int64()
        ok(lib.tavb_search_all(h, ptr(q), c_float(0.52), cap, ptr(oa), ptr(sa), byref(got), byref(tot)))
        print("search_all: got", got.value, "expected", int((scores(vv, q) >= np.float32(0.52)).sum()), flush=True)
        ok(lib.tavb_search_subset_all(h, ptr(q), ptr(rows), rows.size, c_float(0.5), cap, ptr(oa), ptr(sa), byref(got), byref(tot)))
        # message re-rank
        r2m = (np.arange(n) // 3).astype(np.int32)
        dmap = dmalloc(n * 4)
        assert hip.hipMemcpy(dmap, ptr(r2m), n * 4, 1) == 0
        ok(lib.tavb_set_row_messages(h, dmap, n, int(r2m.max()) + 1))
        om = np.empty(k, np.int64); sm = np.empty(k, np.float32)
        acc = np.arange(0, int(r2m.max()) + 1, 2).astype(np.int32)
        ok(lib.tavb_search_messages(h, ptr(q), k, c_float(0.0), ptr(acc), acc.size, 10, ptr(om), ptr(sm), byref(cnt)))
        print("messages: count", cnt.value, flush=True)
        ok(lib.tavb_search_messages_subset(h, ptr(q), ptr(rows), rows.size, k, c_float(0.0), 10, ptr(om), ptr(sm), byref(cnt)))
        # the captured-graph form of the small single-query lookup (plain, capture, replay, replay) -- and back
        ok(lib.tavb_set_option(h, b"graph_max_bytes", 1 << 30))
        first = None
        for rep in range(4):
            ok(lib.tavb_search(h, ptr(q), k, c_float(0.0), ptr(o), ptr(s), byref(cnt)))
            first = o.copy() if first is None else first
            assert (o == first).all()
        g = c_int64(); ok(lib.tavb_get_option(h, b"last_graph", byref(g)))
        print("graph replay: last_graph", g.value, flush=True)
        ok(lib.tavb_set_option(h, b"graph_max_bytes", 0))
        # row shards: the library's own RCCL communicator, a world of one with the collective forced (round 3)
        if os.environ.get("TAVB_ASAN_SKIP_RCCL") != "1":
            uid = ctypes.create_string_buffer(128)
            ok(lib.tavb_comm_unique_id(uid))
            ok(lib.tavb_set_option(h, b"comm_reserve_keys", 2048))  # (round 6) 200 queries x k keys travel in chunks through the reserved buffers
            ok(lib.tavb_comm_init(h, uid, 0, 1))
            ok(lib.tavb_set_option(h, b"comm_force", 1))
            ok(lib.tavb_set_option(h, b"comm_timeout_ms", 20000))  # the polling form of tavb_synchronize
            for nq in (1, 40, 200):
                qq = rng.standard_normal((nq, d)).astype(np.float32)
                qq /= np.linalg.norm(qq, axis=1, keepdims=True)
                dq = dmalloc(nq * d * 4); dk = dmalloc(nq * k * 8); dk2 = dmalloc(nq * k * 8)
                assert hip.hipMemcpy(dq, ptr(qq), nq * d * 4, 1) == 0
                ok(lib.tavb_search_allgather(h, dq, nq, k, c_float(0.0), dk))
                ok(lib.tavb_allgather_merge(h, dk, nq, k, dk2))
                ok(lib.tavb_synchronize(h))
                keys = np.empty((nq, k), np.uint64); keys2 = np.empty((nq, k), np.uint64)
                assert hip.hipMemcpy(ptr(keys), dk, nq * k * 8, 2) == 0 and hip.hipMemcpy(ptr(keys2), dk2, nq * k * 8, 2) == 0
                oo = np.empty((nq, k), np.int64); ss = np.empty((nq, k), np.float32); cc = np.empty(nq, np.int32)
                ok(lib.tavb_decode_keys(ptr(keys), nq, k, ptr(oo), ptr(ss), ptr(cc)))
                ref = np.argsort(-scores(vv, qq[0]), kind="stable")[:k]
                if not ((keys == keys2).all() and cc[0] == k and (oo[0] == ref).mean() > 0.9):
                    print("MISMATCH allgather", dtype, nq, flush=True); bad.append((dtype, "allgather", nq))
                # positions -> map[position] (identity + 7)
                pmap = (np.arange(n, dtype=np.int32) + 7); dm = dmalloc(n * 4)
                assert hip.hipMemcpy(dm, ptr(pmap), n * 4, 1) == 0
                ok(lib.tavb_remap_key_positions(h, dk2, nq * k, dm, n)); ok(lib.tavb_synchronize(h))
                assert hip.hipMemcpy(ptr(keys2), dk2, nq * k * 8, 2) == 0
                ok(lib.tavb_decode_keys(ptr(keys2), nq, k, ptr(oo), ptr(ss), ptr(cc)))
                for fr in (dq, dk, dk2, dm): hip.hipFree(fr)
            # (round 6) a list allocation that fails: the rank's own error, the exchange still runs chunk by chunk with the failure key
            nq = 200
            dq = dmalloc(nq * d * 4); dk = dmalloc(nq * k * 8)
            assert hip.hipMemset(dq, 0, nq * d * 4) == 0
            ok(lib.tavb_set_option(h, b"comm_fail_alloc", 1))
            assert lib.tavb_search_allgather(h, dq, nq, k, c_float(0.0), dk) != 0 and b"injected failure" in lib.tavb_last_error()
            ok(lib.tavb_synchronize(h))
            keys = np.empty((nq, k), np.uint64)
            assert hip.hipMemcpy(ptr(keys), dk, nq * k * 8, 2) == 0 and (keys == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
            oo = np.empty((nq, k), np.int64); ss = np.empty((nq, k), np.float32); cc = np.empty(nq, np.int32)
            assert lib.tavb_decode_keys(ptr(keys), nq, k, ptr(oo), ptr(ss), ptr(cc)) == -6
            ok(lib.tavb_set_option(h, b"comm_fail_alloc", 0))
            # (round 6) the timeout: an exchange held up for 300 ms against a 30 ms limit -> TAVB_E_TIMEOUT, communicator aborted, context usable
            ok(lib.tavb_set_option(h, b"comm_timeout_ms", 30))
            ok(lib.tavb_set_option(h, b"comm_stall_ms", 300))
            ok(lib.tavb_search_allgather(h, dq, nq, k, c_float(0.0), dk))
            assert lib.tavb_synchronize(h) == -7 and b"did not complete within" in lib.tavb_last_error()
            w = c_int64(); ok(lib.tavb_get_option(h, b"comm_world", byref(w))); assert w.value == 0
            ok(lib.tavb_synchronize(h))
            for fr in (dq, dk): hip.hipFree(fr)
            ok(lib.tavb_comm_destroy(h))
            ok(lib.tavb_set_option(h, b"comm_force", 0))
            ok(lib.tavb_set_option(h, b"comm_timeout_ms", 0))
            print("rccl exchange (chunks, failed allocation, timeout): ok", flush=True)
        # bad arguments must come back as error codes, not as crashes
        assert lib.tavb_search(h, ptr(q), 100000, c_float(0.0), ptr(o), ptr(s), byref(cnt)) != 0
        assert lib.tavb_set_option(h, b"no_such_option", 1) != 0
        ok(lib.tavb_set_corpus(h, dev, 0, d, dtype, 0))
        hip.hipFree(dev); hip.hipFree(dmap)
        print("dtype", dtype, "ok", flush=True)
    # round 5: a width that is not a multiple of 64 on the wide tile (zero-padded copy of the rows, padded queries), with an append in between
    d2, n2 = 1000, 30_000
    v2 = rng.standard_normal((n2 + 500, d2)).astype(np.float32)
    v2 /= np.linalg.norm(v2, axis=1, keepdims=True)
    for dtype, esz in ((_native.TAVB_F16, 2), (_native.TAVB_F32, 4)):
        dd = d2 if dtype == _native.TAVB_F16 else 1008  # (fp32: multiples of 16)
        rows = np.ascontiguousarray(v2 if dd == d2 else np.pad(v2, ((0, 0), (0, dd - d2))))
        seen = rows.astype(np.float16).astype(np.float32) if dtype == _native.TAVB_F16 else rows
        dev2 = dmalloc((n2 + 500) * dd * esz)
        for upto in (n2, n2 + 500):  # the second round: 500 appended rows extend the padded copy
            ok(lib.tavb_upload_rows(h, ptr(rows), upto, dd, dev2, dtype))
            ok(lib.tavb_set_corpus(h, dev2, upto, dd, dtype, 0))
            nq = 70
            q = rng.standard_normal((nq, dd)).astype(np.float32)
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            q[4] = rows[upto - 1]
            o = np.empty((nq, k), np.int64); s = np.empty((nq, k), np.float32); c = np.empty(nq, np.int32)
            thr = np.zeros(nq, np.float32)
            ok(lib.tavb_search_batch(h, ptr(q), nq, k, ptr(thr), ptr(o), ptr(s), ptr(c)))
            tier = c_int64(); ok(lib.tavb_get_option(h, b"last_tier", byref(tier)))
            for qi in (0, 4, nq - 1):
                ref = np.argsort(-scores(seen[:upto], q[qi]), kind="stable")[:k]
                if not (tier.value == 4 and c[qi] == k and (o[qi] == ref).mean() > 0.9):
                    print("MISMATCH odd width", dtype, dd, upto, qi, "tier", tier.value, "count", c[qi], flush=True); bad.append((dtype, "odd", upto, qi))
            if o[4][0] != upto - 1:
                print("MISMATCH odd width: appended row not found", dtype, upto, o[4][:3], flush=True); bad.append((dtype, "odd-append", upto))
        ok(lib.tavb_set_corpus(h, dev2, 0, dd, dtype, 0))
        hip.hipFree(dev2)
    print("odd widths on the wide tile: ok", flush=True)
    ok(lib.tavb_destroy(h))
    print("asan exercise:", "all good" if not bad else f"MISMATCHES {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
