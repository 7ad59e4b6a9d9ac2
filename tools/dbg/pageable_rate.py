#!/usr/bin/env python3
"""cfr_classify_batch from PAGEABLE host buffers that are allocated once and touched (what a caller that recycles its batch
buffers hands over): reads/s, single-end -k 1 and pairs -k 5, with and without the SDUST pre-step.
Measured (r3r, two boxes): 1.8e8 and 3.0e8 reads/s pageable, 3.3e8 pinned.  A staging layer of the library's own (pinned
chunks filled by a crew of copy threads) was tried against the runtime's and changed nothing; the 8.7e7 that earlier bench
lines reported for "pageable" was the page faults of freshly allocated result arrays.  HL_PIN=1 (set by the driver mode for
its "pinned" legs) runs the same calls from cfr_host_alloc memory; CFR_DUST_INLINE was an experiment that is no longer in the library.
Usage: python tools/dbg/pageable_rate.py [reads]"""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def child(reads, pairs):
    import argparse, numpy as np, torch
    import bench
    from centrifuger_amd import capi
    sys.argv = ["bench.py", "--reads", str(reads)]
    ap_args = None
    # bench.main parses its own arguments: reuse its parser by calling the pieces directly
    import types
    args = types.SimpleNamespace(species=50, strains=5, genome_len=4_000_000, divergence_step=0.01, reads=reads, read_len=150, seed=20260928,
                                 build_threads=128, builder="own", index_gbp=0.0, cache=os.environ.get("CFR_BENCH_CACHE", "/tmp/cfr_bench"), workload="cfg2", mode="se")
    device = torch.device("cuda", 0)
    bench.bind_to_gpu_numa_node(torch, 0)
    cache = os.path.join(args.cache, bench.cache_key(args))
    prefix = bench.build_index(args, cache, device)
    cat = np.load(os.path.join(cache, "genome_cat.npy"), mmap_mode="r")
    starts = np.load(os.path.join(cache, "genome_starts.npy"))
    cat_d = torch.from_numpy(np.ascontiguousarray(cat)).to(device)
    k = 5 if pairs else 1
    if pairs:
        r1, r2 = bench.make_pairs_gpu(torch, cat_d, starts, reads, 150, 1, device)
        b2 = r2.reshape(-1).cpu().numpy().copy()
    else:
        r1 = bench.make_reads_gpu(torch, cat_d, starts, reads, 150, 1, device)
        b2 = None
    b1 = r1.reshape(-1).cpu().numpy().copy()
    offs = (np.arange(reads + 1, dtype=np.uint64) * 150)
    if os.environ.get("HL_PIN"):                      # the same legs from cfr_host_alloc memory
        keep = []
        def pin(a):
            pa = capi.PinnedArray(a.size, a.dtype); pa.array[:] = a; keep.append(pa); return pa.array
        b1 = pin(b1); offs = pin(offs)
        if b2 is not None: b2 = pin(b2)
    idx = capi.Index(prefix, capi.default_params(max_result=k))
    dev = capi.DeviceIndex(idx, 0)
    if os.environ.get("HL_PIN"):
        pr, pm = capi.PinnedArray(reads, capi.RESULT_DTYPE), capi.PinnedArray(reads * k, capi.MATCH_DTYPE)
        results, matches = pr.array, pm.array
    else:
        results = np.empty(reads, dtype=capi.RESULT_DTYPE); results.view(np.uint8)[:] = 0            # touched: no page faults in the timed calls
        matches = np.empty(reads * k, dtype=capi.MATCH_DTYPE); matches.view(np.uint8)[:] = 0
    out = {}
    for dust in (False, True):
        dev.set_dust(dust)
        best = 1e9
        for it in range(4):
            t0 = time.perf_counter()
            if pairs: res, m = dev.classify(b1, offs, b2, offs, results=results, matches=matches)
            else: res, m = dev.classify(b1, offs, results=results, matches=matches)
            dt = time.perf_counter() - t0
            if it: best = min(best, dt)
        out["dust" if dust else "plain"] = reads / best
        import hashlib
        out["md5_" + ("dust" if dust else "plain")] = hashlib.md5(res.tobytes() + m.tobytes()).hexdigest()
    print("RESULT " + json.dumps(out), flush=True)

if __name__ == "__main__":
    if os.environ.get("HL_CHILD"):
        child(int(sys.argv[1]), sys.argv[2] == "pe")
        sys.exit(0)
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    for mode in ("se", "pe"):
        for name, env in [("pageable", {}), ("pinned", {"HL_PIN": "1"})]:
            e = dict(os.environ, HL_CHILD="1", **env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), str(reads if mode == "se" else reads // 2), mode], env=e, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            print(mode, name, line[0] if line else ("FAILED\n" + p.stdout[-1500:] + p.stderr[-3000:]), flush=True)
