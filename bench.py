#!/usr/bin/env python3
"""bench.py — classified reads/s of the MI355X hot path (BASELINE.json metric).

A "step" = one pass of the hot path (Classifier::Query for every read: search + locate on the
device, scoring/taxonomy tail) over one batch of synthetic reads that are ALREADY RESIDENT in HBM.
Workload at N=1 = BASELINE.json configs[1]: ~1 Gbp synthetic bacterial index (250 genomes x 4 Mbp,
50 species x 5 strains), 10 M synthetic 150 bp single-end reads, default options (-k 1).
N>1: one process per GPU, index replicated, each rank classifies its own 10 M reads (weak scaling,
no data-path collective); value = all reads / max-over-ranks time.

Also reported on the same JSON line:
  roofline     — dominant kernel (k_search_chains): algorithmic bytes (counted exactly by
                 the C oracle on a sample and scaled per read) / kernel time from HIP events inside the library.
  cpu_baseline — the REAL reference binary (oracle/_ref/centrifuger, compiled from /root/reference in the dev
                 container) on this box's host cores, on a bounded sample of the same reads, and a byte-level
                 TSV parity check GPU vs reference on that sample.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


COMPACT_LIMIT = 4096      # bytes of the ONE line the driver parses (round 5's 29 KB line was not parsed: BENCH_r05 `parsed: null`)


def _r(v, nd=4):
    """a number short enough for the compact line (4 significant digits), None for anything that is not a finite number"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, int):
        return v
    try:
        f = float(v)
    except (TypeError, ValueError):
        return None
    if f != f or f in (float("inf"), float("-inf")):
        return None
    return float(f"{f:.{nd}g}")


def compact_line(out):
    """The line the driver reads: the contract's keys, `roofline` and `cpu_baseline` as flat objects of scalars, the parity booleans and ONE scalar
    per sub-result - a few hundred bytes each, <= COMPACT_LIMIT in all.  Everything else (notes, per-kernel counters, iteration mixes) goes to the
    detail file / stderr (`emit_line`).  tests/test_bench_static.py builds this from stored detail objects and checks size, keys and finiteness."""
    g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config") or {}
    line["config"] = {k: (v[:200] if isinstance(v, str) else v) for k, v in cfg.items() if k in ("workload", "index_bp", "reads_per_step_per_gpu", "read_len", "parallelism", "timed_entry")}
    roof = out.get("roofline") or {}
    line["roofline"] = {"bound": roof.get("bound"), "kernel": roof.get("kernel"), "kernel_ms": _r(roof.get("kernel_ms")), "unit": roof.get("unit"), "peak": _r(roof.get("peak")),
                        "achieved": _r(roof.get("achieved")), "frac": _r(roof.get("frac")),
                        "achieved_counter_traffic": _r(roof.get("achieved_counter_traffic")), "frac_counter_traffic": _r(roof.get("frac_counter_traffic")),
                        "frac_useful_bytes": _r(roof.get("frac_useful_bytes")), "gather_frac": _r(g(roof, "gather", "frac") if roof.get("gather") else roof.get("gather_ceiling_frac")),
                        "l2_hit": _r(roof.get("l2_hit")), "x_reference_algorithm": _r(roof.get("x_reference_algorithm")), "traffic": _r(roof.get("traffic"), 6),
                        "requests_per_read": _r(roof.get("fabric_read_requests_per_read")), "fetched_over_useful": _r(roof.get("fetched_over_useful")),
                        "valu_per_wave_iteration": _r(g(roof, "instruction_stream", "valu_wave_instructions_per_wave_iteration")),
                        "valu_issue_frac": _r(g(roof, "instruction_stream", "valu_issue_frac")), "valu_time_frac": _r(g(roof, "instruction_stream", "valu_time_frac")),
                        "kernel_ms_alone": _r(roof.get("kernel_ms_alone"))}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb.get("value"), 6), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": (cb.get("sample") or "")[:160],
                                "classify_only_value": _r(g(cb, "classify_only", "value"), 6), "classify_only_cores": g(cb, "classify_only", "cores")}
    par = {}
    for k, v in (("tsv_identical_to_reference", g(out, "parity", "tsv_identical_to_reference")),
                 ("timed_entry_tsv_identical_to_reference_no_dust", g(out, "parity", "timed_entry_tsv_identical_to_reference_no_dust")),
                 ("reads_vs_reference", g(out, "parity", "reads")), ("equals_oracle", g(out, "parity_oracle", "equals_oracle")),
                 ("reads_vs_oracle", g(out, "parity_oracle", "reads")),
                 ("host_entries_equal_resident", g(out, "pcie_inclusive", "host_entry_equals_resident_entry")),
                 ("cli_tsv_identical_to_reference", g(out, "e2e_cli", "tsv_identical_to_reference")),
                 ("cli_100m_md5_equals_reference_rows", g(out, "e2e_cli_100m", "md5_equals_reference_rows")),
                 ("cli_100m_fastq_md5_equals_reference_rows", g(out, "e2e_cli_100m_fastq", "md5_equals_reference_rows"))):
        if v is not None:
            par[k] = v
    line["parity"] = par
    if g(out, "index_on_ranks", "ranks") is not None:
        line["index_on_ranks"] = {k: out["index_on_ranks"].get(k) for k in ("ranks", "every_rank_maps_the_file", "copied_bytes_max")}
    st = out.get("stage_ms") or {}
    line["stage_ms"] = {k: _r(st.get(k)) for k in ("search_ms", "tail_ms", "total_ms") if k in st}
    if out.get("classified_fraction") is not None:
        line["classified_fraction"] = _r(out["classified_fraction"], 7)
    sub = {}
    if g(out, "with_device_sdust", "value") is not None:
        sub["value_default_options"] = _r(out["with_device_sdust"]["value"], 6)      # SDUST on (the reference's default), on the device
        sub["k_dust_ms"] = _r(g(out, "with_device_sdust", "roofline", "pre_step_ms_per_step_measured"))
    for k in ("pinned_value", "packed_pinned_value"):
        if g(out, "pcie_inclusive", k) is not None:
            sub["pcie_" + k] = _r(out["pcie_inclusive"][k], 6)
    if g(out, "e2e_cli_100m", "value") is not None:
        sub["e2e_cli_100m"] = _r(out["e2e_cli_100m"]["value"], 6)
    if g(out, "e2e_cli_100m_fastq", "value") is not None:
        sub["e2e_cli_100m_fastq"] = _r(out["e2e_cli_100m_fastq"]["value"], 6)
    if g(out, "e2e_cli", "speedup") is not None:
        sub["e2e_cli_speedup_vs_reference"] = _r(out["e2e_cli"]["speedup"])
    ps = g(out, "post_stage", "roofline") or {}
    if ps:
        sub["post_stage"] = {"ms_alone": _r(ps.get("ms_alone_per_step")), "ms_in_step": _r(ps.get("ms_inside_the_step_overlapped")), "frac_counter_traffic": _r(ps.get("frac_counter_traffic"))}
    for name, o in (out.get("other_configs") or {}).items():
        if not isinstance(o, dict):
            continue
        if "value" in o:
            e = {"value": _r(o["value"], 6), "ms_per_step": _r(o.get("ms_per_step"))}
            ok = o.get("equals_oracle")
            if ok is not None:
                e["equals_oracle"] = ok
            pv = o.get("parity_vs_reference_binary") or {}
            if pv.get("timed_entry_tsv_identical_to_reference_no_dust") is not None:
                e["tsv_identical_to_reference"] = bool(pv.get("timed_entry_tsv_identical_to_reference_no_dust")) and pv.get("tsv_identical_to_reference") is not False
            rf = o.get("roofline") or {}
            if rf.get("fabric_read_requests_per_read") is not None:
                e["requests_per_read"] = _r(rf["fabric_read_requests_per_read"])
            if isinstance(rf.get("gather"), dict) and rf["gather"].get("frac") is not None:
                e["gather_frac"] = _r(rf["gather"]["frac"])
            if rf.get("frac_counter_traffic") is not None:
                e["frac_counter_traffic"] = _r(rf["frac_counter_traffic"])
            pst = g(o, "post_stage", "roofline") or {}
            if pst.get("ms_alone_per_step") is not None:
                e["post_ms_alone"], e["post_ms_in_step"] = _r(pst.get("ms_alone_per_step")), _r(pst.get("ms_inside_the_step_overlapped"))
            sub[name] = e
        elif "skipped" in o:
            sub[name] = {"skipped": str(o["skipped"])[:120]}
        else:
            sub[name] = {"error": str(o.get("error"))[:120]}
    line["sub_results"] = sub
    if out.get("detail"):
        line["detail"] = out["detail"]
    s = json.dumps(line, allow_nan=False, separators=(",", ":"))
    shrink = [lambda: [e.update({k: v[:32] for k, v in e.items() if isinstance(v, str)}) for e in sub.values() if isinstance(e, dict)],     # shorter error texts
              lambda: [sub.pop(k) for k in [k for k, e in sub.items() if isinstance(e, dict) and "value" not in e and "ms_alone" not in e]],   # no error texts
              lambda: [sub.update({k: e["value"]}) for k, e in list(sub.items()) if isinstance(e, dict) and "value" in e],                   # one scalar per sub-result
              lambda: line.pop("sub_results", None), lambda: line.pop("stage_ms", None), lambda: line.pop("parity", None)]
    while len(s.encode()) > COMPACT_LIMIT - 64 and shrink:       # never with today's fields on a healthy run; a run full of errors loses their texts, not the line
        shrink.pop(0)()
        s = json.dumps(line, allow_nan=False, separators=(",", ":"))
    return s


def emit_line(out, args=None):
    """Children of the default run (`--inner`, `--sub-result`) and `CFR_BENCH_FULL_LINE=1` print the full object (their parent parses it).  Everything
    else prints the compact line LAST on stdout, after the full object has gone to stderr and to a detail file ($CFR_BENCH_DETAIL, else
    gpurun_out/bench_detail_latest.json beside this file when that directory exists, else the cache directory)."""
    full = json.dumps(out)
    if (args is not None and (getattr(args, "inner", False) or getattr(args, "sub_result", False))) or os.environ.get("CFR_BENCH_FULL_LINE") == "1":
        print(full, flush=True)
        return
    path = os.environ.get("CFR_BENCH_DETAIL")
    if not path:
        gdir = os.path.join(ROOT, "gpurun_out")
        path = os.path.join(gdir if os.path.isdir(gdir) else (getattr(args, "cache", None) or "/tmp"), "bench_detail_latest.json")
    try:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        with open(path, "w") as f:
            f.write(full + "\n")
        out = dict(out, detail=path if not path.startswith(ROOT) else os.path.relpath(path, ROOT))
    except OSError as e:
        log(f"[bench] detail file {path}: {e}")
    sys.stderr.write("[bench] detail " + full + "\n")
    sys.stderr.flush()
    sys.stdout.flush()
    print(compact_line(out), flush=True)


def build_index(args, cache, device=None):
    """Index generation is OUT of the timed path.  --builder own: the native writer behind cfr_build_index (suffix array in
    HBM, csrc/cfr_build_sa.hip; writes the same .cfr fields as the reference's builder, tests/test_gpu_build.py);
    --builder reference: the reference's own centrifuger-build (oracle/_ref); --builder python: centrifuger_amd/indexbuild.py
    (the torch restatement of the writer, texts below 2.1 Gbp)."""
    from centrifuger_amd import synth
    prefix = os.path.join(cache, "idx")
    if os.path.exists(prefix + ".done"):
        return prefix
    os.makedirs(cache, exist_ok=True)
    t0 = time.time()
    if args.index_gbp:      # multi-Gbp texts: threaded generator, one buffer (not the random stream of the cfg2 text)
        g, cat = synth.make_genomes_fast(args.species, args.strains, args.genome_len, seed=args.seed, divergence_step=args.divergence_step,
                                         threads=min(os.cpu_count() or 1, 64))
    else:
        g = synth.make_genomes(args.species, args.strains, args.genome_len, seed=args.seed, divergence_step=args.divergence_step,
                               model=getattr(args, "divergence_model", "star"))
        cat = np.concatenate(g.seqs)
    np.save(os.path.join(cache, "genome_cat.npy"), cat)
    lens = np.array([len(s) for s in g.seqs], dtype=np.uint64)
    if args.index_gbp:      # the writer reads the text once, front to back: hand it the file mapping and let go of the 1 byte/base copy
        g.seqs = [None] * len(g.seqs)
        del cat
        cat = np.load(os.path.join(cache, "genome_cat.npy"), mmap_mode="r")
    np.save(os.path.join(cache, "genome_starts.npy"), np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
    log(f"genomes: {int(lens.sum())/1e6:.0f} Mbp generated in {time.time()-t0:.1f}s")
    t0 = time.time()
    if args.builder == "own":
        from centrifuger_amd import capi
        rep = capi.build_index(g.names, g.taxids, (cat, lens), g.nodes, g.tax_names, prefix, device=device.index or 0 if device is not None else 0, verbose=True)
        log(f"index built by the native writer (cfr_build_index) in {time.time()-t0:.1f}s: suffix array {rep['seconds_sa']:.1f}s, {rep['rounds']} doubling rounds, b = {rep['b']}")
        json.dump({"builder": "cfr_build_index", "n": rep["n"], "seconds_total": rep["seconds_total"], "seconds_sa": rep["seconds_sa"],
                   "rounds": rep["rounds"], "block_size": rep["b"]}, open(prefix + ".build.json", "w"))
    elif args.builder == "python":
        from centrifuger_amd import indexbuild
        indexbuild.build_index(g.names, g.taxids, g.seqs, g.nodes, g.tax_names, prefix, device=device, log=log)
        log(f"index built by centrifuger_amd.indexbuild in {time.time()-t0:.1f}s")
    else:
        synth.write_reference_inputs(g, cache)
        builder = os.path.join(ROOT, "oracle", "_ref", "centrifuger-build")
        if not os.path.exists(builder):
            raise SystemExit("oracle/_ref/centrifuger-build missing: run __graft_entry__.build() where /root/reference exists")
        subprocess.run([builder, "-t", str(min(os.cpu_count() or 1, args.build_threads)), "-r", os.path.join(cache, "ref.fa"),
                        "--taxonomy-tree", os.path.join(cache, "nodes.dmp"), "--name-table", os.path.join(cache, "names.dmp"),
                        "--conversion-table", os.path.join(cache, "seqid.map"), "-o", prefix],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        log(f"index built by reference centrifuger-build in {time.time()-t0:.1f}s")
        os.remove(os.path.join(cache, "ref.fa"))
    open(prefix + ".done", "w").close()
    return prefix


def bind_to_gpu_numa_node(torch, local_rank):
    """One process per GPU: keep the process (and therefore the pinned result buffers it allocates and the copies into them)
    on the NUMA node the GPU hangs off.  Best effort; CFR_BENCH_NO_NUMA=1 turns it off.  Returns the node or None."""
    if os.environ.get("CFR_BENCH_NO_NUMA") == "1":
        return None
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            log(f"rank on GPU {local_rank} ({bdf}) bound to NUMA node {node} ({len(cpus)} cpus)")
            return node
    except Exception as e:      # containers without sysfs, older torch: run unbound
        log(f"NUMA binding skipped: {e}")
    return None


def make_reads_gpu(torch, cat_d, starts, n_reads, read_len, seed, device, sub_rate=0.01, n_rate=0.001):
    """Synthetic reads generated directly in HBM (uniform over genomes/positions/strands, 1 % subs, 0.1 % N)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    starts_d = torch.as_tensor(starts, device=device)
    lens = starts_d[1:] - starts_d[:-1]
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=device)
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    code = torch.zeros(256, dtype=torch.int64, device=device)
    for i, a in enumerate(b"ACGT"):
        code[a] = i
    ar = torch.arange(read_len, device=device)
    chunk = 1 << 20
    for lo in range(0, n_reads, chunk):
        m = min(chunk, n_reads - lo)
        gi = torch.randint(0, len(lens), (m,), generator=gen, device=device)
        pos = (torch.rand(m, generator=gen, device=device, dtype=torch.float64) * (lens[gi] - read_len).double()).long()
        r = cat_d[(starts_d[gi] + pos)[:, None] + ar[None, :]]
        rc = torch.rand(m, generator=gen, device=device) < 0.5
        r = torch.where(rc[:, None], comp[r.long()].flip(1), r)
        sub = torch.rand((m, read_len), generator=gen, device=device) < sub_rate
        shift = torch.randint(1, 4, (m, read_len), generator=gen, device=device)
        r = torch.where(sub, acgt[(code[r.long()] + shift) & 3], r)
        isn = torch.rand((m, read_len), generator=gen, device=device) < n_rate
        r = torch.where(isn, torch.full_like(r, ord("N")), r)
        out[lo:lo + m] = r
    return out


def make_pairs_gpu(torch, cat_d, starts, n_pairs, read_len, seed, device, ins_lo=250, ins_hi=500, sub_rate=0.01, n_rate=0.001):
    """FR pairs, insert U[250,500] (BASELINE configs[2]); mate 2 is the reverse complement of the fragment's far end."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    starts_d = torch.as_tensor(starts, device=device)
    lens = starts_d[1:] - starts_d[:-1]
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    code = torch.zeros(256, dtype=torch.int64, device=device)
    for i, a in enumerate(b"ACGT"):
        code[a] = i
    ar = torch.arange(read_len, device=device)
    out1 = torch.empty((n_pairs, read_len), dtype=torch.uint8, device=device)
    out2 = torch.empty((n_pairs, read_len), dtype=torch.uint8, device=device)

    def mutate(r, m):
        sub = torch.rand((m, read_len), generator=gen, device=device) < sub_rate
        shift = torch.randint(1, 4, (m, read_len), generator=gen, device=device)
        r = torch.where(sub, acgt[(code[r.long()] + shift) & 3], r)
        isn = torch.rand((m, read_len), generator=gen, device=device) < n_rate
        return torch.where(isn, torch.full_like(r, ord("N")), r)

    chunk = 1 << 20
    for lo in range(0, n_pairs, chunk):
        m = min(chunk, n_pairs - lo)
        gi = torch.randint(0, len(lens), (m,), generator=gen, device=device)
        ins = torch.randint(ins_lo, ins_hi + 1, (m,), generator=gen, device=device)
        pos = (torch.rand(m, generator=gen, device=device, dtype=torch.float64) * (lens[gi] - ins).double()).long()
        base = starts_d[gi] + pos
        left = cat_d[base[:, None] + ar[None, :]]
        right = comp[cat_d[(base + ins - read_len)[:, None] + ar[None, :]].long()].flip(1)
        flip = torch.rand(m, generator=gen, device=device) < 0.5
        r1 = torch.where(flip[:, None], right, left)
        r2 = torch.where(flip[:, None], left, right)
        out1[lo:lo + m] = mutate(r1, m)
        out2[lo:lo + m] = mutate(r2, m)
    return out1, out2


def make_long_reads_gpu(torch, cat_d, starts, n_reads, seed, device, len_lo=5000, len_hi=20000, sub_rate=0.04, ins_rate=0.03, del_rate=0.03):
    """ONT-like reads: U[5k,20k] bp, i.i.d. substitutions / insertions / deletions; flat buffer + offsets, built in HBM."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    starts_d = torch.as_tensor(starts, device=device)
    lens = starts_d[1:] - starts_d[:-1]
    comp = torch.full((256,), ord("N"), dtype=torch.uint8, device=device)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    code = torch.zeros(256, dtype=torch.int64, device=device)
    for i, a in enumerate(b"ACGT"):
        code[a] = i
    pieces, piece_lens = [], []
    chunk = 4096
    for lo in range(0, n_reads, chunk):
        m = min(chunk, n_reads - lo)
        L = torch.randint(len_lo, len_hi + 1, (m,), generator=gen, device=device)
        gi = torch.randint(0, len(lens), (m,), generator=gen, device=device)
        pos = (torch.rand(m, generator=gen, device=device, dtype=torch.float64) * (lens[gi] - L).double()).long()
        off = torch.zeros(m + 1, dtype=torch.int64, device=device)
        off[1:] = torch.cumsum(L, 0)
        tot = int(off[-1].item())
        rid = torch.repeat_interleave(torch.arange(m, device=device), L)
        within = torch.arange(tot, device=device) - off[rid]
        rcflag = (torch.rand(m, generator=gen, device=device) < 0.5)[rid]
        # reverse-strand reads walk the genome segment backwards, complemented
        src = starts_d[gi][rid] + pos[rid] + torch.where(rcflag, L[rid] - 1 - within, within)
        r = cat_d[src]
        r = torch.where(rcflag, comp[r.long()], r)
        sub = torch.rand(tot, generator=gen, device=device) < sub_rate
        r = torch.where(sub, acgt[(code[r.long()] + torch.randint(1, 4, (tot,), generator=gen, device=device)) & 3], r)
        keep = torch.rand(tot, generator=gen, device=device) >= del_rate
        ins = (torch.rand(tot, generator=gen, device=device) < ins_rate) & keep
        emit = keep.long() + ins.long()                       # 0 (deleted), 1, or 2 (base + inserted base)
        out_len = torch.zeros(m, dtype=torch.int64, device=device).index_add_(0, rid, emit)
        rep = torch.repeat_interleave(torch.arange(tot, device=device), emit)
        o = r[rep]
        first = torch.ones_like(rep, dtype=torch.bool)
        first[1:] = rep[1:] != rep[:-1]
        rnd = acgt[torch.randint(0, 4, (len(rep),), generator=gen, device=device)]
        o = torch.where(first, o, rnd)                        # the second copy of a base is the inserted random base
        pieces.append(o)
        piece_lens.append(out_len)
    bases = torch.cat(pieces)
    ll = torch.cat(piece_lens)
    offs = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
    offs[1:] = torch.cumsum(ll, 0)
    return bases, offs


BYTES_PER_RANDOM_REQUEST = 128.0     # profiles/r2a_gather_calib.json


def kernel_source_sha():
    h = hashlib.sha1()
    for f in ("cfr_kernels.hip.inc", "cfr_device.hip", "cfr_device.hpp"):
        h.update(open(os.path.join(ROOT, "centrifuger_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def live_pmc(args, cache, gpu=0):
    """HBM-side counters of k_search_chains_v2 measured in THIS run: the bench re-executes itself (--inner: warm-up + one
    2 M-read step, a single launch) under `rocprofv3 --pmc ... --kernel-trace` - one pass per counter group, as the guide
    prescribes - plus one un-profiled pass with the kernel's diagnostic instantiation for its iteration mix.
    Returns None when rocprofv3 is not usable here."""
    import csv
    import glob
    import shutil
    import tempfile
    if args.mode not in ("se", "long"):
        return None
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    n_inner = min(2_000_000, args.reads)
    inner = [sys.executable, os.path.abspath(__file__), "--inner", "--reads", str(n_inner), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
             "--species", str(args.species), "--strains", str(args.strains), "--genome-len", str(args.genome_len), "--divergence-step", str(args.divergence_step),
             "--read-len", str(args.read_len), "--seed", str(args.seed), "--builder", args.builder, "--cache", args.cache] + (["--index-gbp", str(args.index_gbp)] if args.index_gbp else [])
    inner += ["--divergence-model", getattr(args, "divergence_model", "star")] + (["--mode", "long"] if args.mode == "long" else [])
    env = dict(os.environ, CFR_DEBUG_ENV="1", CFR_SUBBATCH=str(n_inner), CFR_TAPER_FLOOR="0", TMPDIR="/tmp")
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k_, None)           # the child is a plain one-process run on this rank's GPU
    if gpu:
        env["HIP_VISIBLE_DEVICES"] = str(gpu)
    vals, dur = {}, None
    work = tempfile.mkdtemp(prefix="cfr_pmc_", dir="/tmp")
    try:
        # (third pass, round 5: the kernel's instruction stream - the search is bound by it as much as by its gathers)
        for gi, group in enumerate((["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "FETCH_SIZE"], ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"],
                                    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"])):
            if gi == 2 and getattr(args, "sub_result", False):
                break                  # (the main line only: a pass over a 40 Gbp sub-result costs a minute and a half)
            d = os.path.join(work, f"pass{gi}")
            r = subprocess.run([rocprof, "--pmc"] + group + ["--kernel-trace", "--output-format", "csv", "--kernel-include-regex", "k_search_chains_v2",
                                "-d", d, "--"] + inner, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            if r.returncode != 0:
                log("live PMC pass failed:", r.stderr.decode()[-400:])
                if gi == 2:
                    break              # (the traffic counters are in: the line goes out without the instruction-stream object)
                return None
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_search_chains_v2" in row["Kernel_Name"]:
                        vals[row["Counter_Name"]] = float(row["Counter_Value"])          # last dispatch = the timed step
                        dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        if "TCC_EA0_RDREQ_sum" not in vals or "WRITE_SIZE" not in vals:
            return None
        prof = None
        r = subprocess.run(inner, cwd="/tmp", env=dict(env, CFR_SEARCH_PROF="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        for line in r.stderr.decode().splitlines():
            if line.startswith("[search prof]"):
                tok = line.split(":", 1)[1].replace("(per read)", "").split()
                prof = {tok[i]: float(tok[i + 1]) for i in range(0, len(tok) - 1, 2)}      # last launch wins
        return {"source": f"live in this run: rocprofv3 --pmc (2 passes) around one {n_inner}-read launch of the same build, scaled per read",
                "reads": n_inner, "rdreq": vals["TCC_EA0_RDREQ_sum"], "rdreq_32b": vals.get("TCC_EA0_RDREQ_32B_sum", 0.0),
                "fetch_size_kib": vals.get("FETCH_SIZE", 0.0), "write_size_kib": vals["WRITE_SIZE"], "kernel_ms_profiled": dur, "prof": prof,
                "l2_hit": (vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])) if vals.get("TCC_HIT_sum") is not None and (vals.get("TCC_HIT_sum", 0) + vals.get("TCC_MISS_sum", 0)) > 0 else None,
                "sq": {k_: vals[k_] for k_ in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE") if k_ in vals},
                "kernel_source_sha": kernel_source_sha()}
    except Exception as e:
        log("live PMC unavailable:", repr(e))
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def live_pmc_dust(args, cache, gpu=0):
    """Issue-side counters of the SDUST kernels (k_dust<true>: reads of A, C, G, T only - the kernel that carries the pre-step),
    measured like live_pmc: this script re-executed (--inner --inner-dust: one 2 M-read step with cfr_device_index_set_dust(1))
    under rocprofv3 --pmc, one pass per counter group.  Returns {counter: value, "ms": duration of that kernel} or None."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof) or args.mode != "se":
        return None
    n_inner = min(2_000_000, args.reads)
    inner = [sys.executable, os.path.abspath(__file__), "--inner", "--inner-dust", "--reads", str(n_inner), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
             "--species", str(args.species), "--strains", str(args.strains), "--genome-len", str(args.genome_len), "--divergence-step", str(args.divergence_step),
             "--read-len", str(args.read_len), "--seed", str(args.seed), "--builder", args.builder, "--cache", args.cache]
    env = dict(os.environ, CFR_DEBUG_ENV="1", CFR_SUBBATCH=str(n_inner), CFR_TAPER_FLOOR="0", TMPDIR="/tmp")
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k_, None)
    if gpu:
        env["HIP_VISIBLE_DEVICES"] = str(gpu)
    groups = (["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"],
              ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_INSTS_SMEM"])
    out = {}
    work = tempfile.mkdtemp(prefix="cfr_pmc_dust_", dir="/tmp")
    try:
        for gi, group in enumerate(groups):
            d = os.path.join(work, f"pass{gi}")
            r = subprocess.run([rocprof, "--pmc"] + group + ["--kernel-trace", "--output-format", "csv", "--kernel-include-regex", "k_dust",
                                "-d", d, "--"] + inner, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            if r.returncode != 0:
                log("live PMC pass (dust) failed:", r.stderr.decode()[-300:])
                continue
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "k_dust<true>" in row["Kernel_Name"].replace(" ", ""):
                        out[row["Counter_Name"]] = float(row["Counter_Value"])          # last dispatch = the timed step
                        out["ms"] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        return dict(out, reads=n_inner) if "SQ_INSTS_VALU" in out else None
    except Exception as e:
        log("live PMC (dust) unavailable:", repr(e))
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def dust_roofline(pm, reads, dust_ms):
    """The SDUST kernel is bound by its own instruction stream (profiles/HISTORY.md section 4): its roofline is the VALU issue rate.  A wave64
    VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md, wave scheduling), so peak = 1024 SIMDs x clock / 2
    wave-instructions per second; clock = GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time when that counter came back."""
    if not pm:
        return None
    ms = pm["ms"]
    clock = (pm["GRBM_GUI_ACTIVE"] / 8.0 / (ms / 1e3)) if pm.get("GRBM_GUI_ACTIVE") else 2.4e9
    peak = 1024.0 * clock / 2.0
    ach = pm["SQ_INSTS_VALU"] / (ms / 1e3)
    r = {"bound": "valu", "kernel": "k_dust<true>", "unit": "G wave-instructions/s", "achieved": ach / 1e9, "peak": peak / 1e9, "frac": ach / peak,
         "kernel_ms_profiled": ms, "reads_in_profiled_launch": pm["reads"], "kernel_ms_per_step_scaled": ms / pm["reads"] * reads,
         "pre_step_ms_per_step_measured": dust_ms,
         "clock_GHz": clock / 1e9, "valu_instructions_per_wave_and_base": pm["SQ_INSTS_VALU"] * 64.0 / (pm["reads"] * 150.0),
         "waves_per_simd": (pm["SQ_WAVE_CYCLES"] * 4.0 / ((ms / 1e3) * clock * 1024.0)) if pm.get("SQ_WAVE_CYCLES") else None,
         "wait_fraction_of_wave_cycles": (pm["SQ_WAIT_ANY"] / pm["SQ_WAVE_CYCLES"]) if pm.get("SQ_WAVE_CYCLES") and pm.get("SQ_WAIT_ANY") is not None else None,
         "lds_instructions_per_valu": (pm["SQ_INSTS_LDS"] / pm["SQ_INSTS_VALU"]) if pm.get("SQ_INSTS_LDS") is not None else None,
         "lds_bank_conflict_over_lds_active": (pm["SQ_LDS_BANK_CONFLICT"] / pm["SQ_ACTIVE_INST_LDS"]) if pm.get("SQ_ACTIVE_INST_LDS") else None,
         "counters": {k_: v for k_, v in pm.items() if k_ not in ("ms", "reads")},
         "note": "k_dust<true> under rocprofv3 --pmc (2 passes, one 2 M-read launch): VALU wave-instructions issued per second over what 1024 SIMD-32 units "
                 "can issue (one wave64 VALU instruction per 2 cycles); SQ_* cycle counters are quad-cycles (x 4)"}
    return r


def useful_bytes_per_read(pr, hits_per_read=0.0):
    """bytes of the fetched lines the search kernel consumes, from its iteration mix (CFR_SEARCH_PROF): 16 B per K-mer entry / text window /
    SA fetch / read-block pair, 24-48 B per BWT extend, 64 B per wide SA fetch, 32 B per hit written"""
    return (16 * (pr.get("table", 0) + pr.get("table10", 0)) + 48 * pr.get("ext_two_records", 0) + 24 * (pr.get("ext", 0) - pr.get("ext_two_records", 0))
            + 16 * pr.get("text_rows", 0) + 16 * pr.get("sa", 0) + 64 * pr.get("saw", 0) + 32 * pr.get("text_hits", 0) + 16 * pr.get("block_loads", 0) + 32 * hits_per_read)


POST_KERNELS = ("k_adjust_tail", "k_post_fast", "k_tail_heavy")
_POST_PMC_BROKEN = False


def live_pmc_post(args, cache, gpu=0):
    """Counters of the POST-STAGE kernels (k_adjust_tail; k_tail_heavy's two tiers; k_post_fast when it is switched on), measured like
    live_pmc: this script re-executed (--inner: one 2 M-read step = one launch of each) under rocprofv3 --pmc, one pass per counter group.
    Returns {kernel: {counter: value, "ms": its duration in that pass}} or None."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof) or args.mode != "se":
        return None
    n_inner = min(2_000_000, args.reads)
    inner = [sys.executable, os.path.abspath(__file__), "--inner", "--reads", str(n_inner), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
             "--species", str(args.species), "--strains", str(args.strains), "--genome-len", str(args.genome_len), "--divergence-step", str(args.divergence_step),
             "--read-len", str(args.read_len), "--seed", str(args.seed), "--builder", args.builder, "--cache", args.cache] + (["--index-gbp", str(args.index_gbp)] if args.index_gbp else [])
    inner += ["--divergence-model", getattr(args, "divergence_model", "star")]
    env = dict(os.environ, CFR_DEBUG_ENV="1", CFR_SUBBATCH=str(n_inner), CFR_TAPER_FLOOR="0", CFR_TAIL_STREAM="0", TMPDIR="/tmp")    # the post stage behind its search: alone
    for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k_, None)
    if gpu:
        env["HIP_VISIBLE_DEVICES"] = str(gpu)
    # (three passes: the four TCC counters in ONE pass never finish on this stack - rocprofv3 sat in the first dispatch until the timeout,
    #  tools/dbg/pmc_post_try.sh - two at a time do)
    groups = (["TCC_EA0_RDREQ_sum", "WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"],
              ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"])
    global _POST_PMC_BROKEN
    if _POST_PMC_BROKEN:
        return None
    out = {}
    work = tempfile.mkdtemp(prefix="cfr_pmc_post_", dir="/tmp")
    try:
        for gi, group in enumerate(groups):
            d = os.path.join(work, f"pass{gi}")
            r = subprocess.run([rocprof, "--pmc"] + group + ["--kernel-trace", "--output-format", "csv", "--kernel-include-regex", "|".join(POST_KERNELS),
                                "-d", d, "--"] + inner, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
            if r.returncode != 0:
                log("live PMC pass (post stage) failed:", r.stderr.decode()[-300:])
                continue
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row["Kernel_Name"].replace(" ", "")
                    short = next((k_ for k_ in POST_KERNELS if k_ in name), None)
                    if short is None:
                        continue
                    if short == "k_tail_heavy":
                        short = "k_tail_heavy_large_teams" if ",32,256," in name else "k_tail_heavy_small_teams"
                    e = out.setdefault(short, {})
                    e[row["Counter_Name"]] = float(row["Counter_Value"])                 # last dispatch = the timed step
                    e["ms"] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        return dict(out, reads=n_inner) if out else None
    except Exception as e:
        log("live PMC (post stage) unavailable:", repr(e)[:300])
        _POST_PMC_BROKEN = True                 # (a pass that timed out is not tried again in this run: the default line must finish in minutes)
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def post_stage_roofline(pm, reads, tail_ms, search_requests_per_read=None, hits_per_read=None, rows_per_read=None):
    """The post stage (boundary adjustment, strand choice, locate, scoring, LCA) is a short chain of dependent gathers per read: what
    bounds it is the latency of those gathers at the occupancy its registers allow, and what it costs the step is the requests it adds
    to the fabric the search beside it is bound by.  So the object reports, per kernel and summed: time alone (the one-launch counter
    pass), fabric requests and written bytes per read, L2 hit rate, VALU instructions per read and the share of the VALU issue rate."""
    if not pm:
        return None
    n = pm["reads"]
    kernels, tot_ms, tot_req, tot_wr, tot_valu = {}, 0.0, 0.0, 0.0, 0.0
    for name, e in pm.items():
        if not isinstance(e, dict):
            continue
        ms = e.get("ms", 0.0)
        clock = (e["GRBM_GUI_ACTIVE"] / 8.0 / (ms / 1e3)) if e.get("GRBM_GUI_ACTIVE") and ms else 2.4e9
        req = e.get("TCC_EA0_RDREQ_sum", 0.0)
        kernels[name] = {
            "ms_alone_per_step": ms / n * reads, "fabric_read_requests_per_read": req / n, "write_bytes_per_read": e.get("WRITE_SIZE", 0.0) * 1024 / n,
            "l2_hit": (e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])) if e.get("TCC_HIT_sum") is not None and (e.get("TCC_HIT_sum", 0) + e.get("TCC_MISS_sum", 0)) > 0 else None,
            "valu_wave_instructions_per_read": (e["SQ_INSTS_VALU"] / n) if e.get("SQ_INSTS_VALU") is not None else None,
            "valu_issue_frac": (e["SQ_INSTS_VALU"] / (ms / 1e3) / (1024.0 * clock / 2.0)) if e.get("SQ_INSTS_VALU") is not None and ms else None,
            "waves_per_simd": (e["SQ_WAVE_CYCLES"] * 4.0 / ((ms / 1e3) * clock * 1024.0)) if e.get("SQ_WAVE_CYCLES") and ms else None,
            "wait_fraction_of_wave_cycles": (e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"]) if e.get("SQ_WAVE_CYCLES") and e.get("SQ_WAIT_ANY") is not None else None}
        tot_ms += ms / n * reads
        tot_req += req / n
        tot_wr += e.get("WRITE_SIZE", 0.0) * 1024 / n
        tot_valu += (e.get("SQ_INSTS_VALU") or 0.0) / n
    traffic = (tot_req * BYTES_PER_RANDOM_REQUEST + tot_wr) * reads
    out = {"bound": "hbm", "kernel": "+".join(sorted(kernels)), "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernels": kernels,
           "ms_alone_per_step": tot_ms, "ms_inside_the_step_overlapped": tail_ms,
           "fabric_read_requests_per_read": tot_req, "write_bytes_per_read": tot_wr, "valu_wave_instructions_per_read": tot_valu,
           "traffic": traffic, "achieved_counter_traffic": traffic / (tot_ms / 1e3) / 1e9 if tot_ms else None,
           "frac_counter_traffic": traffic / (tot_ms / 1e3) / 1e9 / HBM_PEAK_GBS if tot_ms else None,
           "requests_per_s_alone": tot_req * reads / (tot_ms / 1e3) if tot_ms else None,
           "gather_ceiling_frac_alone": tot_req * reads / (tot_ms / 1e3) / 48e9 if tot_ms else None,
           "note": "one 2 M-read launch of each post-stage kernel under rocprofv3 --pmc (2 passes), scaled per read; ms_alone = the kernels with the chip to themselves, "
                   "ms_inside_the_step = first start to last end beside the next sub-batch's search (mostly waiting for wave slots); frac = useful bytes "
                   "(hits read, suffix-array entries, results written) over the time alone - a latency-bound stage: its cost to the step is the requests it adds"}
    if search_requests_per_read:
        out["share_of_the_step's_fabric_requests"] = tot_req / (tot_req + search_requests_per_read)
    if hits_per_read is not None and rows_per_read is not None:
        useful = 32.0 * hits_per_read + 4.0 * rows_per_read + 24.0 + 32.0       # hits, SA entries, the read's offsets / chain counts / hit offset, its result + match slot (compact layout)
        out["useful_bytes_per_read"] = useful
        out["achieved"] = useful * reads / (tot_ms / 1e3) / 1e9 if tot_ms else None
        out["frac"] = out["achieved"] / HBM_PEAK_GBS if tot_ms else None
        out["fetched_over_useful"] = (tot_req * BYTES_PER_RANDOM_REQUEST + tot_wr) / useful
    else:
        out["achieved"] = None
        out["frac"] = None
    return out


def instruction_stream(pmc):
    """The search kernel's own instruction stream from the SQ pass of live_pmc (one launch alone on the chip): wave64 vector instructions
    per read and per lane iteration, the VALU issue rate they amount to (a wave64 VALU instruction occupies its SIMD-32 for 2 cycles:
    peak = 1024 SIMDs x clock / 2), and where the waves' cycles go.  SQ_*_CYCLES are quad-cycles."""
    sq = (pmc or {}).get("sq") or {}
    ms = (pmc or {}).get("kernel_ms_profiled")
    if "SQ_INSTS_VALU" not in sq or not ms:
        return None
    n = pmc["reads"]
    clock = (sq["GRBM_GUI_ACTIVE"] / 8.0 / (ms / 1e3)) if sq.get("GRBM_GUI_ACTIVE") else 2.4e9
    iters = (pmc.get("prof") or {}).get("lane_iterations")
    wc = sq.get("SQ_WAVE_CYCLES")
    return {"valu_wave_instructions_per_read": sq["SQ_INSTS_VALU"] / n, "salu_wave_instructions_per_read": (sq["SQ_INSTS_SALU"] / n) if sq.get("SQ_INSTS_SALU") is not None else None,
            "vmem_read_wave_instructions_per_read": (sq["SQ_INSTS_VMEM_RD"] / n) if sq.get("SQ_INSTS_VMEM_RD") is not None else None,
            "valu_issue_frac": sq["SQ_INSTS_VALU"] / (ms / 1e3) / (1024.0 * clock / 2.0), "clock_GHz": clock / 1e9,
            # (round 6) SQ_ACTIVE_INST_VALU stands at 1.00 QUAD-cycles per vector instruction on this chip (the SDUST pass collects it): a wave64
            # instruction holds its SIMD for four cycles, so the share of the chip's VALU time this kernel uses is twice the figure above
            "valu_time_frac": sq["SQ_INSTS_VALU"] / (ms / 1e3) / (1024.0 * clock / 4.0),
            "valu_wave_instructions_per_wave_iteration": (sq["SQ_INSTS_VALU"] * 64.0 / (n * iters)) if iters else None,
            "waves_per_simd": (wc * 4.0 / ((ms / 1e3) * clock * 1024.0)) if wc else None,
            "wave_cycles_waiting": (sq["SQ_WAIT_ANY"] / wc) if wc and sq.get("SQ_WAIT_ANY") is not None else None,
            "wave_cycles_executing": (sq["SQ_ACTIVE_INST_ANY"] / wc) if wc and sq.get("SQ_ACTIVE_INST_ANY") is not None else None,
            "note": "lanes of a wave are in different states of the search, so a wave runs nearly every branch of the loop in every iteration: "
                    "valu_wave_instructions_per_wave_iteration is taken as if the lanes' iterations were spread evenly over full waves.  Round 5 cut "
                    "the static instruction count (-18 %), the lane iterations (-12 %) and the fabric requests (-6 %; block loads -33 % with a dense "
                    "read form) of this kernel in turn: none of it shortened the kernel on one box (profiles/r5r_ab_libs.txt, r5l_ab_dense.txt)"}


def mini_roofline(pmc, reads, search_ms):
    """roofline object of a sub-result (same counter arithmetic as the main line's: 128-byte requests, WRITE_SIZE, 8 TB/s)"""
    if not pmc:
        return {"bound": "hbm", "kernel": "k_search_chains_v2", "peak": HBM_PEAK_GBS, "unit": "GB/s", "achieved": None, "frac": None, "traffic": None,
                "note": "live PMC passes unavailable"}
    per_read_rd = pmc["rdreq"] * BYTES_PER_RANDOM_REQUEST / pmc["reads"]
    per_read_wr = pmc["write_size_kib"] * 1024 / pmc["reads"]
    traffic = (per_read_rd + per_read_wr) * reads
    ach = traffic / (search_ms / 1e3) / 1e9
    req_s = pmc["rdreq"] / pmc["reads"] * reads / (search_ms / 1e3)
    useful = useful_bytes_per_read(pmc["prof"]) if pmc.get("prof") else None
    ach_useful = (useful * reads / (search_ms / 1e3) / 1e9) if useful else None
    return {"bound": "hbm", "kernel": "k_search_chains_v2", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_ms": search_ms,
            "achieved": ach_useful, "frac": (ach_useful / HBM_PEAK_GBS) if ach_useful else None, "frac_useful_bytes": (ach_useful / HBM_PEAK_GBS) if ach_useful else None,
            "achieved_counter_traffic": ach, "frac_counter_traffic": ach / HBM_PEAK_GBS, "useful_bytes_per_read": useful,
            "fetched_over_useful": ((per_read_rd + per_read_wr) / useful) if useful else None,
            "traffic": traffic, "fabric_read_requests_per_read": pmc["rdreq"] / pmc["reads"], "l2_hit": pmc.get("l2_hit"),
            "gather": {"requests_per_s": req_s, "ceiling_per_s": 48e9, "frac": req_s / 48e9},
            "gather_frac_of_48G_requests_per_s": req_s / 48e9, "iteration_mix_per_read": pmc.get("prof"), "traffic_source": pmc["source"],
            "instruction_stream": instruction_stream(pmc),
            "yardsticks": "frac = frac_useful_bytes (bytes of the fetched lines the kernel consumes / kernel time / 8 TB/s); frac_counter_traffic = fabric bytes by PMC / kernel time / 8 TB/s"}


def live_pmc_protein(args, n):
    """The protein leg's counters, measured like live_pmc: this script re-executed (--mode protein --inner) under rocprofv3 --pmc, one
    pass per counter group, k_search_prot and k_translate_prot.  Returns {kernel: {counter: value, "ms": duration}} or None."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    inner = [sys.executable, os.path.abspath(__file__), "--mode", "protein", "--inner", "--reads", str(n), "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
             "--prot-species", str(args.prot_species), "--seed", str(args.seed), "--cache", args.cache] + (["-k", str(args.k)] if args.k is not None else [])
    env = dict(os.environ, CFR_DEBUG_ENV="1", CFR_SUBBATCH=str(n), CFR_TAPER_FLOOR="0", TMPDIR="/tmp")      # the step as ONE launch of each kernel
    out = {}
    work = tempfile.mkdtemp(prefix="cfr_pmc_", dir="/tmp")
    try:
        for gi, group in enumerate((["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "FETCH_SIZE"], ["WRITE_SIZE"])):
            d = os.path.join(work, f"pass{gi}")
            r = subprocess.run([rocprof, "--pmc"] + group + ["--kernel-trace", "--output-format", "csv", "--kernel-include-regex", "k_search_prot|k_translate_prot",
                                "-d", d, "--"] + inner, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            if r.returncode != 0:
                log("live PMC pass (protein) failed:", r.stderr.decode()[-400:])
                return None
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    for kname in ("k_search_prot", "k_translate_prot"):
                        if kname in row["Kernel_Name"]:
                            e = out.setdefault(kname, {})
                            e[row["Counter_Name"]] = float(row["Counter_Value"])        # last dispatch = the timed step
                            e["ms"] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
        if "TCC_EA0_RDREQ_sum" not in out.get("k_search_prot", {}) or "WRITE_SIZE" not in out.get("k_search_prot", {}):
            return None
        return out
    except Exception as e:
        log("live PMC (protein) unavailable:", repr(e))
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cache_key(args):
    return hashlib.md5((f"{args.species}-{args.strains}-{args.genome_len}-{args.seed}-{args.builder}" + (f"-{args.divergence_step}" if args.divergence_step != 0.01 else "")
                        + ("-fastgen" if args.index_gbp else "") + ("-" + args.divergence_model if getattr(args, "divergence_model", "star") != "star" else "")).encode()).hexdigest()[:10]


STRAIN_WORKLOADS = {
    # name: (species, strains per species, genome length, divergence step, genome model, description)
    "strains20": (25, 20, 2_000_000, 0.001, "star", "25 species x 20 strains 0.1 % apart"),
    "strains200": (10, 200, 500_000, 0.0005, "tree", "10 species x 200 strains, a binary phylogeny with 0.05 % new substitutions per edge (neighbours 0.05-0.1 % apart)"),
}


def strains_config(torch, capi, ora, args, device, name="strains20"):
    """The data-sensitivity cases as sub-results: many near-identical strains per species (1 Gbp), the metric's reads (10 M x 150 bp SE,
    -k 1).  strains20: ranges of ~20 rows per hit (wide text mode in the search, k_tail_heavy in the tail).  strains200: what a
    redundant database (RefSeq / GTDB species with hundreds of genomes, /root/reference/README.md:13) looks like - ranges of ~200 rows,
    strided locate (Classifier.hpp:640-666).  Checked against the C oracle's TSV lines; roofline + iteration mix from live PMC passes."""
    a2 = argparse.Namespace(**vars(args))
    a2.species, a2.strains, a2.genome_len, a2.divergence_step, a2.divergence_model, desc = STRAIN_WORKLOADS[name]
    a2.index_gbp = 0.0
    cache = os.path.join(args.cache, cache_key(a2))
    prefix = build_index(a2, cache, device)
    torch.cuda.empty_cache()
    cat = np.load(os.path.join(cache, "genome_cat.npy"), mmap_mode="r")
    starts = np.load(os.path.join(cache, "genome_starts.npy"))
    cat_d = torch.from_numpy(np.ascontiguousarray(cat)).to(device)
    n = args.reads
    reads_d = make_reads_gpu(torch, cat_d, starts, n, args.read_len, args.seed + 4000, device)
    offs_d = torch.arange(n + 1, device=device, dtype=torch.int64) * args.read_len
    del cat_d, cat
    torch.cuda.empty_cache()
    idx = capi.Index(prefix, capi.default_params(max_result=1))
    dev = capi.DeviceIndex(idx, device.index or 0)
    total = n * args.read_len
    compact = args.results == "compact"
    res_pin = capi.PinnedArray(n, capi.RESULT_COMPACT_DTYPE if compact else capi.RESULT_DTYPE)
    mat_pin = capi.PinnedArray(n, capi.MATCH_COMPACT_DTYPE if compact else capi.MATCH_DTYPE)

    def step():
        f = dev.classify_resident_compact if compact else dev.classify_resident
        return f(reads_d.data_ptr(), offs_d.data_ptr(), n, total, results=res_pin.array, matches=mat_pin.array)
    step()          # (the scratch pool of the single-lane folds grows on the way, and the library picks its schedule - post stage
    step()          #  beside the next search or behind it - from what the previous call saw: two untimed steps)
    torch.cuda.synchronize()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = dev.last_stats()
    nchk = 5000
    b1 = reads_d.reshape(-1)[:nchk * args.read_len].cpu().numpy()
    offs_h = (np.arange(nchk + 1, dtype=np.uint64) * np.uint64(args.read_len))
    oo = ora.OracleIndex(prefix, max_result=1)
    ores = oo.classify(b1, offs_h, dust=False, threads=min(os.cpu_count() or 1, 64))
    res, mat = capi.expand_compact(res_pin.array[:nchk], mat_pin.array[:nchk], 1) if compact else (res_pin.array, mat_pin.array)
    same = all(idx.format_tsv("r", res[i], mat) == oo.format("r", ores[i]) for i in range(nchk))
    out = {"value": n * steps / el, "unit": "reads/s", "ms_per_step": 1000 * el / steps, "steps": steps,
           "workload": f"{idx.info().n/1e9:.2f} Gbp index of {desc}, {n} x {args.read_len} bp SE reads, -k 1, inputs resident in HBM",
           "search_ms": st.search_ms, "tail_ms": st.tail_ms, "classified_fraction": float((res_pin.array["n_match"] > 0).mean()),
           "tsv_lines_equal_oracle_on_first": nchk, "equals_oracle": bool(same)}
    oo.close()
    res_pin.free()
    mat_pin.free()
    dev.close()
    del reads_d, offs_d
    torch.cuda.empty_cache()
    if not args.no_pmc:
        a2.mode, a2.reads = "se", n
        out["roofline"] = mini_roofline(live_pmc(a2, cache, device.index or 0), n, st.search_ms)
        out["post_stage"] = {"roofline": post_stage_roofline(live_pmc_post(a2, cache, device.index or 0), n, st.tail_ms, out["roofline"].get("fabric_read_requests_per_read"))}
    return out


def extra_config(torch, capi, ora, args, mode, prefix, cache, device):
    """One of the other BASELINE configs on the same index: timed steps with resident inputs + a check against the C oracle."""
    paired = mode == "pe"
    k = 5 if paired else 1
    n = args.reads if paired else (200_000 if args.index_gbp else 1_000_000)   # (cfg5's share per GPU is 2.5 M long reads: 9.0e6 reads/s there, 8.8e6 at 1 M, 7.2e6 at 200 k - chains per lane)
    cat = np.load(os.path.join(cache, "genome_cat.npy"), mmap_mode="r")
    starts = np.load(os.path.join(cache, "genome_starts.npy"))
    cat_d = torch.from_numpy(np.ascontiguousarray(cat)).to(device)       # (reads first, image second: see main)
    if paired:
        r1, r2 = make_pairs_gpu(torch, cat_d, starts, n, args.read_len, args.seed + 2000, device)
        offs_d = torch.arange(n + 1, device=device, dtype=torch.int64) * args.read_len
    else:
        r1, offs_d = make_long_reads_gpu(torch, cat_d, starts, n, args.seed + 3000, device)
        r2 = None
    del cat_d, cat
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    idx = capi.Index(prefix, capi.default_params(max_result=k))
    dev = capi.DeviceIndex(idx, device.index or 0)
    offs_h = offs_d.cpu().numpy().astype(np.uint64)
    total = int(offs_h[-1])
    compact = args.results == "compact"
    res_pin = capi.PinnedArray(n, capi.RESULT_COMPACT_DTYPE if compact else capi.RESULT_DTYPE)
    mat_pin = capi.PinnedArray(n * k, capi.MATCH_COMPACT_DTYPE if compact else capi.MATCH_DTYPE)

    def step():
        f = dev.classify_resident_compact if compact else dev.classify_resident
        if paired:
            return f(r1.data_ptr(), offs_d.data_ptr(), n, total, r2.data_ptr(), offs_d.data_ptr(), total, results=res_pin.array, matches=mat_pin.array)
        return f(r1.data_ptr(), offs_d.data_ptr(), n, total, results=res_pin.array, matches=mat_pin.array)
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 3
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = dev.last_stats()
    # the first AND the last reads of the batch against the C oracle (score / second score / hit length / match count of every read): the
    # last reads are the last chains the lanes take - a hand-out that loses its tail (round 6 had such a variant for an hour) shows there
    nchk = 20_000 if paired else 500
    oo = ora.OracleIndex(prefix, max_result=k)
    same = True
    for lo_, hi_ in ((0, nchk // 2), (n - nchk // 2, n)):
        a_, b_ = int(offs_h[lo_]), int(offs_h[hi_])
        b1 = r1.reshape(-1)[a_:b_].cpu().numpy()
        b2 = r2.reshape(-1)[a_:b_].cpu().numpy() if paired else None
        oh = (offs_h[lo_:hi_ + 1] - offs_h[lo_]).astype(np.uint64)
        ores = oo.classify(b1, oh.copy(), b2, oh.copy() if paired else None, dust=False, threads=min(os.cpu_count() or 1, 64))
        res = capi.expand_compact(res_pin.array[lo_:hi_], mat_pin.array[lo_ * k:hi_ * k], k)[0] if compact else res_pin.array[lo_:hi_]
        same = same and all((int(res[i]["score"]), int(res[i]["secondary_score"]), int(res[i]["hit_length"]), int(res[i]["n_match"])) ==
                            (ores[i].score, ores[i].secondaryScore, ores[i].hitLength, ores[i].nmatch) for i in range(hi_ - lo_))
    out = {"value": n * steps / el, "unit": "read pairs/s" if paired else "reads/s", "ms_per_step": 1000 * el / steps, "steps": steps,
           "workload": (f"{n} x 2x{args.read_len} bp pairs, insert 250-500, -k 5 (BASELINE configs[2])" if paired else
                        f"{n} long reads, 5-20 kbp (mean {total / n:.0f} bp), 3% del / 3% ins / 4% sub (BASELINE configs[4]-style reads on this index)"),
           "bases_per_s": total * steps / el, "search_ms": st.search_ms, "classified_fraction": float((res_pin.array["n_match"] > 0).mean()),
           "entry": "cfr_classify_batch_resident_compact" if compact else "cfr_classify_batch_resident",
           "equals_oracle_on_first": nchk, "equals_oracle_sample": "first and last %d reads of the batch" % (nchk // 2), "equals_oracle": bool(same)}
    res_pin.free()
    mat_pin.free()
    dev.close()
    return out


AA_LIST = "ARNDCEQGHILKMFPSTWYV"
AA_CODON = {"A": "GCT", "R": "CGT", "N": "AAT", "D": "GAT", "C": "TGT", "E": "GAA", "Q": "CAA", "G": "GGT", "H": "CAT", "I": "ATT",
            "L": "CTT", "K": "AAA", "M": "ATG", "F": "TTT", "P": "CCT", "S": "TCT", "T": "ACT", "W": "TGG", "Y": "TAT", "V": "GTT"}


def protein_mode(torch, capi, args, device):
    """translated search of 150 bp DNA reads against a protein index (FMIndex<Sequence_RunBlockOneTree>,
    Classifier::TranslatedSearch, Classifier.hpp:463-506).  The index is written by the REFERENCE's `centrifuger-build --protein`
    (oracle/_ref) from a synthetic proteome: --prot-species species x 5 strains (2 % substitutions per strain step) x 400 proteins of
    ~300 aa.  Reads: 50-codon windows of random proteins, one codon per amino acid, 1 % substitutions, either strand.  One JSON line:
    reads/s of cfr_classify_batch_resident (inputs resident in HBM), TSV parity and CPU baseline = the reference binary on a sample."""
    from centrifuger_amd import synth
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    cache = os.path.join(args.cache, f"prot-{args.prot_species}-{args.seed}")
    prefix = os.path.join(cache, "idx")
    rng = np.random.default_rng(args.seed)
    n_sp, n_st, n_pr = args.prot_species, 5, 400
    os.makedirs(cache, exist_ok=True)
    if not os.path.exists(prefix + ".done"):
        t0 = time.time()
        lens = rng.integers(150, 450, size=(n_sp, n_pr))
        nodes, names = [(1, 1, "no rank")], [(1, "root")]
        with open(os.path.join(cache, "prot.fa"), "wb") as fa, open(os.path.join(cache, "seqid.map"), "w") as mp:
            aa = np.frombuffer(AA_LIST.encode(), dtype=np.uint8)
            starts, total = [], 0
            flat = []
            for sp in range(n_sp):
                sp_tid = 1000 + sp * 10
                nodes.append((sp_tid, 1, "species")); names.append((sp_tid, f"species {sp}"))
                base = [aa[rng.integers(0, 20, size=int(l))] for l in lens[sp]]
                for k in range(n_st):
                    st_tid = sp_tid + 1 + k
                    nodes.append((st_tid, sp_tid, "strain")); names.append((st_tid, f"species {sp} strain {k}"))
                    for pi, b in enumerate(base):
                        q = b.copy()
                        nm = int(len(q) * 0.02 * k)
                        if nm:
                            q[rng.integers(0, len(q), size=nm)] = aa[rng.integers(0, 20, size=nm)]
                        name = f"P{sp}_{k}_{pi}"
                        fa.write(b">" + name.encode() + b"\n" + q.tobytes() + b"\n")
                        mp.write(f"{name}\t{st_tid}\n")
                        flat.append(q); starts.append(total); total += len(q)
        np.save(os.path.join(cache, "prot_cat.npy"), np.concatenate(flat))
        np.save(os.path.join(cache, "prot_starts.npy"), np.array(starts + [total], dtype=np.int64))
        with open(os.path.join(cache, "nodes.dmp"), "w") as f:
            for t, par, rank in nodes:
                f.write(f"{t}\t|\t{par}\t|\t{rank}\t|\n")
        with open(os.path.join(cache, "names.dmp"), "w") as f:
            for t, nm in names:
                f.write(f"{t}\t|\t{nm}\t|\t\t|\tscientific name\t|\n")
        log(f"proteome: {total/1e6:.1f} M amino acids in {len(starts)} proteins, {time.time()-t0:.1f}s")
        # the index is written by the native writer (cfr_build_index, protein = 1: suffix array of the byte text on this GPU);
        # --prot-ref-build also runs the reference's centrifuger-build --protein on the same files and compares what the two wrote
        t0 = time.time()
        seq_names = [f"P{sp}_{k}_{pi}" for sp in range(n_sp) for k in range(n_st) for pi in range(n_pr)]
        seq_tids = [1000 + sp * 10 + 1 + k for sp in range(n_sp) for k in range(n_st) for pi in range(n_pr)]
        rep = capi.build_index(seq_names, seq_tids, (np.concatenate(flat), np.array([len(q) for q in flat], dtype=np.uint64)), nodes, names, prefix,
                               ftab_chars=4, protein=True, device=device.index or 0)
        build_info = {"builder": "cfr_build_index (protein)", "n": int(rep["n"]), "seconds_total": time.time() - t0, "seconds_sa": rep["seconds_sa"],
                      "rounds": rep["rounds"], "block_size": int(rep["b"])}
        log(f"protein index ({rep['n']/1e6:.1f} M symbols) written by the native writer in {build_info['seconds_total']:.1f}s (suffix array {rep['seconds_sa']:.2f}s, {rep['rounds']} rounds)")
        if args.prot_ref_build and os.path.exists(os.path.join(ref_dir, "centrifuger-build")):
            t0 = time.time()
            subprocess.run([os.path.join(ref_dir, "centrifuger-build"), "--protein", "-t", str(min(os.cpu_count() or 1, args.build_threads)), "-r", os.path.join(cache, "prot.fa"),
                            "--taxonomy-tree", os.path.join(cache, "nodes.dmp"), "--name-table", os.path.join(cache, "names.dmp"),
                            "--conversion-table", os.path.join(cache, "seqid.map"), "-o", prefix + "_ref"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            build_info["reference_builder_seconds"] = time.time() - t0
            a, b = capi.Index(prefix), capi.Index(prefix + "_ref")
            build_info["digest_equals_reference_built_index"] = a.digest() == b.digest()
            build_info["taxonomy_file_identical"] = open(prefix + ".2.cfr", "rb").read() == open(prefix + "_ref.2.cfr", "rb").read()
            a.close(); b.close()
            log(f"reference's centrifuger-build --protein: {build_info['reference_builder_seconds']:.1f}s; parsed indexes equal: {build_info['digest_equals_reference_built_index']}")
        with open(prefix + ".build.json", "w") as f:
            json.dump(build_info, f)
        open(prefix + ".done", "w").close()
    cat = np.load(os.path.join(cache, "prot_cat.npy"))
    starts = np.load(os.path.join(cache, "prot_starts.npy"))
    plen = np.diff(starts)
    n = args.reads if args.reads != 10_000_000 else 4_000_000
    codon = np.zeros((256, 3), dtype=np.uint8)
    for a, c in AA_CODON.items():
        codon[ord(a)] = np.frombuffer(c.encode(), dtype=np.uint8)
    pi = rng.integers(0, len(plen), size=n)
    off = (rng.random(n) * (plen[pi] - 50)).astype(np.int64)
    aa_win = cat[(starts[pi] + off)[:, None] + np.arange(50)[None, :]]
    reads = codon[aa_win].reshape(n, 150)
    mut = rng.random((n, 150)) < 0.01
    reads[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(mut.sum()))]
    rc = rng.random(n) < 0.5
    reads[rc] = synth.revcomp(reads[rc])
    reads = np.ascontiguousarray(reads.reshape(-1))
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)
    reads_d = torch.from_numpy(reads).to(device)
    offs_d = torch.from_numpy(offs.astype(np.int64)).to(device)
    k = args.k if args.k is not None else 1
    t0 = time.time()
    idx = capi.Index(prefix, capi.default_params(max_result=k))
    dev = capi.DeviceIndex(idx, device.index or 0)
    info = dev.info()
    log(f"protein index n={info.n} loaded; device image {info.device_bytes/1e6:.0f} MB in {time.time()-t0:.1f}s")
    res_pin = capi.PinnedArray(n, capi.RESULT_DTYPE)
    mat_pin = capi.PinnedArray(n * k, capi.MATCH_DTYPE)

    def step():
        return dev.classify_resident(reads_d.data_ptr(), offs_d.data_ptr(), n, n * 150, results=res_pin.array, matches=mat_pin.array)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kst = []
    for _ in range(args.steps):
        step()
        kst.append(dev.last_stats())
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    results, matches = res_pin.array, mat_pin.array
    out = {"metric": "classified reads/sec (150 bp, translated search against a protein index)", "value": n * args.steps / el, "unit": "reads/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "u64", "data": "synthetic",
           "config": {"workload": f"{info.n/1e6:.0f} M-symbol protein index ({n_sp} species x {n_st} strains x {n_pr} proteins, written by cfr_build_index, "
                                  f"protein = 1), {n} x 150 bp DNA reads per step, -k {k}, inputs resident in HBM", "index_symbols": int(info.n)},
           "classified_fraction": float((results["n_match"] > 0).mean()),
           "stage_ms": {kk: float(np.mean([getattr(s_, kk) for s_ in kst])) for kk in ("search_ms", "adjust_ms", "rows_ms", "locate_ms", "tail_ms", "total_ms")}}
    if os.path.exists(prefix + ".build.json"):
        out["index"] = json.load(open(prefix + ".build.json"))
    if getattr(args, "inner", False):
        emit_line(out, args)
        dev.close()
        return
    if not args.no_pmc:
        dev.close()                          # the child builds the same image on the same GPU
        pm = live_pmc_protein(args, n)
        dev = capi.DeviceIndex(idx, device.index or 0)
        if pm:
            # k_search_prot is the dominant kernel: dependent gathers into the K-mer table, the 128-byte occurrence records, the
            # suffix array and the byte text - HBM / fabric bound like the nucleotide search (a request = one 128-byte line,
            # profiles/r2a_gather_calib.json).  Its duration: the kernel trace of the counter pass (the in-library events
            # bracket caps + scan + translate + search together: stage_ms.search_ms).
            sp, tp = pm["k_search_prot"], pm.get("k_translate_prot", {})
            rd, wr = sp["TCC_EA0_RDREQ_sum"] * 128.0, sp["WRITE_SIZE"] * 1024.0
            out["roofline"] = {"bound": "hbm", "kernel": "k_search_prot", "peak": 8000.0, "unit": "GB/s", "kernel_ms": sp["ms"],
                               "achieved": None, "frac": None, "achieved_counter_traffic": (rd + wr) / (sp["ms"] * 1e-3) / 1e9,
                               "frac_counter_traffic": (rd + wr) / (sp["ms"] * 1e-3) / 1e9 / 8000.0, "traffic": rd + wr,
                               "yardsticks": "frac (useful bytes) is not instrumented for the translated search: frac_counter_traffic = fabric bytes by PMC / kernel time / 8 TB/s",
                               "fabric_read_requests_per_read": sp["TCC_EA0_RDREQ_sum"] / n, "read_bytes_per_read": rd / n, "write_bytes_per_read": wr / n,
                               "requests_per_s": sp["TCC_EA0_RDREQ_sum"] / (sp["ms"] * 1e-3),
                               "gather_ceiling_frac": sp["TCC_EA0_RDREQ_sum"] / (sp["ms"] * 1e-3) / 48e9,
                               "translate_kernel": {"kernel": "k_translate_prot", "ms": tp.get("ms"),
                                                    "read_bytes": tp.get("TCC_EA0_RDREQ_sum", 0.0) * 128.0, "write_bytes": tp.get("WRITE_SIZE", 0.0) * 1024.0},
                               "traffic_source": f"live in this run: rocprofv3 --pmc (2 passes, --kernel-trace) around one {n}-read step of the same build; "
                                                 "kernel_ms is that trace's duration of the kernel",
                               "note": "achieved = (TCC_EA0_RDREQ x 128 B + WRITE_SIZE) of k_search_prot / its duration; 6 chains per read (strand x frame), one lane each"}
        else:
            out["roofline"] = {"bound": "hbm", "kernel": "k_search_prot", "peak": 8000.0, "unit": "GB/s", "achieved": None, "frac": None, "traffic": None,
                               "note": "rocprofv3 --pmc not usable here"}
    refbin = os.path.join(ref_dir, "centrifuger")
    if not args.no_cpu_baseline and os.path.exists(refbin):
        nb = min(args.cpu_sample if args.cpu_sample != 2_000_000 else 400_000, n)
        rs = synth.ReadSet(reads[:nb * 150].copy(), offs[:nb + 1].copy())
        fa = os.path.join(cache, "sample.fa")
        synth.write_fasta(rs, fa)
        one = os.path.join(cache, "one.fa")
        synth.write_fasta(rs.slice(0, 1), one)
        ncpu = os.cpu_count() or 1

        def ref_run(t, f):
            t0 = time.time()
            o_ = subprocess.run([refbin, "-x", prefix, "-t", str(t), "-k", str(k), "-u", f], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
            return time.time() - t0, o_
        t_load = ref_run(min(ncpu, 64), one)[0]
        t_full, ref_tsv = ref_run(min(ncpu, 64), fa)
        own = capi.tsv_header() + b"".join(idx.format_tsv(f"r{i}", results[i], matches) for i in range(nb))
        out["cpu_baseline"] = {"value": nb / max(t_full - t_load, 1e-9), "unit": "reads/s", "cores": min(ncpu, 64), "kind": "reference",
                               "sample": f"first {nb} reads, oracle/_ref/centrifuger -t {min(ncpu, 64)} -k {k} on the same protein index, wall {t_full:.1f}s minus index-load run {t_load:.1f}s"}
        out["parity"] = {"reads": nb, "tsv_identical_to_reference": own == ref_tsv, "md5": hashlib.md5(own).hexdigest()}
    emit_line(out, args)
    dev.close()


def sub_config_40gbp(args, cfg):
    """BASELINE configs[3] (cfg4) / configs[4] (cfg5) on this one GPU, as a sub-result of the default line: `bench.py --config cfg4|cfg5
    --sub-result` in a process of its own (the 249 GB image needs the GPU to itself; the index - 40 Gbp, written once by the native
    writer in ~4 min - is shared through the cache).  What comes back is that run's own JSON line, cut down to the fields that matter
    here.  Skipped, with the reason, on a box that cannot hold it."""
    import shutil
    try:
        import torch
        need = []
        free_b, total_b = torch.cuda.mem_get_info()
        if total_b < 270e9:
            need.append(f"HBM {total_b/1e9:.0f} GB < 270 GB (the 40 Gbp image takes 249 GB)")
        avail = None
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                avail = int(ln.split()[1]) * 1024
        lim = None
        for f in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            try:
                v = open(f).read().strip()
                if v.isdigit():
                    lim = int(v)
                    break
            except OSError:
                pass
        a2 = argparse.Namespace(**vars(args))
        a2.index_gbp, a2.divergence_model = 40.0, "star"
        a2.species = max(1, int(round(a2.index_gbp * 1e9 / (a2.strains * a2.genome_len))))
        have_index = os.path.exists(os.path.join(args.cache, cache_key(a2), "idx.done"))
        host_need = 60e9 if have_index else 250e9      # building keeps the suffix array in host memory (186 GiB) beside the text's file mapping
        host_have = min(x for x in (avail, lim) if x is not None) if (avail is not None or lim is not None) else None
        if host_have is not None and host_have < host_need:
            need.append(f"host memory {host_have/1e9:.0f} GB < {host_need/1e9:.0f} GB")
        os.makedirs(args.cache, exist_ok=True)
        disk = shutil.disk_usage(args.cache).free
        if disk < (6e9 if have_index else 64e9):
            need.append(f"{args.cache}: {disk/1e9:.0f} GB free (text 40 GB + index files 17 GB + samples)")
        if need:
            return {"skipped": "; ".join(need)}
        cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg, "--sub-result", "--cache", args.cache, "--seed", str(args.seed)] + (["--no-pmc"] if args.no_pmc else [])
        if cfg == "cfg5":
            cmd.append("--no-cpu-baseline")      # (20 000 long reads through the reference binary at 40 Gbp take minutes; parity of this leg = the C oracle on 4000 reads)
        env = dict(os.environ)
        for k_ in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
            env.pop(k_, None)
        t0 = time.time()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=2400 if not have_index else 1500)
        wall = time.time() - t0
        line = None
        for ln in r.stdout.decode().splitlines():
            if ln.startswith("{") and '"metric"' in ln:
                line = ln
        if r.returncode != 0 or line is None:
            return {"error": f"rc {r.returncode}", "stderr_tail": r.stderr.decode()[-600:], "seconds": wall}
        d = json.loads(line)
        roof = d.get("roofline") or {}
        keep = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "workload": d["config"]["workload"],
                "index_bp": d["config"].get("index_bp"), "timed_entry": d["config"].get("timed_entry"), "stage_ms": d.get("stage_ms"),
                "classified_fraction": d.get("classified_fraction"), "bases_per_s": d.get("bases_per_s"),
                "equals_oracle": (d.get("parity_oracle") or {}).get("equals_oracle"), "equals_oracle_on_first": (d.get("parity_oracle") or {}).get("reads"),
                "parity_vs_reference_binary": {k_: (d.get("parity") or {}).get(k_) for k_ in ("reads", "timed_entry_tsv_identical_to_reference_no_dust", "tsv_identical_to_reference")} if d.get("parity") else None,
                "cpu_baseline": {k_: (d.get("cpu_baseline") or {}).get(k_) for k_ in ("value", "unit", "cores", "kind", "sample")} if d.get("cpu_baseline") else None,
                "roofline": {k_: roof.get(k_) for k_ in ("bound", "kernel", "peak", "unit", "achieved", "frac", "achieved_counter_traffic", "frac_counter_traffic", "fetched_over_useful", "traffic", "kernel_ms", "fabric_read_requests_per_read", "l2_hit",
                                                        "frac_useful_bytes", "x_reference_algorithm", "gather", "iteration_mix_per_read", "traffic_source")},
                "index": d.get("index"), "multi_rank_load": d.get("multi_rank_load"), "seconds": wall,
                "note": f"`bench.py --config {cfg}` in its own process on this GPU: BASELINE {'configs[3]' if cfg == 'cfg4' else 'configs[4]'} per rank (the 8-GPU job is 8 such ranks, index replicated)"}
        return keep
    except subprocess.TimeoutExpired:
        return {"error": "timed out"}
    except Exception as e:
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--species", type=int, default=50)
    ap.add_argument("--strains", type=int, default=5)
    ap.add_argument("--genome-len", type=int, default=4_000_000)
    ap.add_argument("--divergence-step", type=float, default=0.01, help="strain k of a species differs from its base by k x this fraction of substitutions")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per step per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--seed", type=int, default=20260928)
    ap.add_argument("--cpu-sample", type=int, default=2_000_000, help="reads given to the CPU reference baseline")
    ap.add_argument("--count-sample", type=int, default=200_000, help="reads the C oracle counts operations on")
    ap.add_argument("--build-threads", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic then comes from profiles/pmc_latest.json)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the paired-end / long-read legs reported under other_configs")
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)   # child of the live PMC passes: warm-up + timed steps only
    ap.add_argument("--builder", choices=["own", "reference", "python"], default="own", help="who writes the .cfr index (outside the timed path)")
    ap.add_argument("--index-gbp", type=float, default=0.0, help="size of the synthetic index in Gbp (sets --species; same 5-strain model); "
                                                                  "0 = BASELINE configs[1] (1 Gbp, the metric's config)")
    ap.add_argument("--cache", default=os.environ.get("CFR_BENCH_CACHE", "/tmp/cfr_bench"))
    ap.add_argument("--prot-species", type=int, default=200, help="--mode protein: species of the synthetic proteome (x 5 strains x 400 proteins x ~300 aa)")
    ap.add_argument("--prot-ref-build", action="store_true", help="--mode protein: also build the index with oracle/_ref/centrifuger-build --protein and compare")
    ap.add_argument("--mode", choices=["se", "pe", "long", "protein"], default="se",
                    help="se = BASELINE configs[1] (default, the metric's config); pe = configs[2]: 2x150 bp pairs, -k 5; "
                         "long = configs[4]-style reads (5-20 kbp, 3%% del / 3%% ins / 4%% sub) on the 1 Gbp index")
    ap.add_argument("-k", type=int, default=None, help="max_result (default 1 for se, 5 for pe)")
    ap.add_argument("--results", choices=["compact", "wide"], default="compact",
                    help="result layout of the timed entry: compact = cfr_classify_batch_resident_compact (20 + 12 bytes per read / match slot "
                         "cross PCIe), wide = cfr_classify_batch_resident (40 + 24); the other one is reported under other_result_layout")
    ap.add_argument("--workload", choices=["cfg2", "strains20", "strains200"], default="cfg2",
                    help="cfg2 = the metric's index (50 species x 5 strains 1 %% apart); strains20 = the data-sensitivity case: "
                         "25 species x 20 strains 0.1 %% apart x 2 Mbp (ranges of ~20 rows per hit: wide text mode, team fold)")
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4", "cfg5"], default=None,
                    help="BASELINE.json presets, per rank (so that --gpus 8 runs the configuration as written): cfg2 = configs[1] (default); "
                         "cfg3 = configs[2] (10 M pairs, -k 5); cfg4 = configs[3]: 40 Gbp index, 12.5 M x 150 bp reads per rank per step (100 M over 8 ranks); "
                         "cfg5 = configs[4]: 40 Gbp index, long reads 5-20 kbp, 2.5 M per rank over the timed steps (312 500 per step: the bases of one "
                         "step are 3.9 GB beside a 245 GB image)")
    ap.add_argument("--sdust-steps", type=int, default=5, help="timed steps of the legs with the SDUST pre-step on the device")
    ap.add_argument("--divergence-model", choices=["star", "tree"], default="star", help="synth.make_genomes model (tree: strain k descends from strain (k-1)//2)")
    ap.add_argument("--inner-dust", action="store_true", help=argparse.SUPPRESS)   # child of the SDUST PMC passes: the inner step with the pre-step on
    ap.add_argument("--sub-result", action="store_true", help=argparse.SUPPRESS)   # child of the default run: one BASELINE config as a compact sub-result
    ap.add_argument("--no-40gbp", action="store_true", help="skip the cfg4 / cfg5 legs (40 Gbp index on this GPU) of the default run")
    args = ap.parse_args()
    if args.config == "cfg3":
        args.mode = "pe"
    elif args.config == "cfg4":        # (an explicit --index-gbp / --reads wins: the multi-rank test runs this preset on a 1 Gbp index)
        args.index_gbp = args.index_gbp or 40.0
        if args.reads == 10_000_000:
            args.reads = 12_500_000
    elif args.config == "cfg5":
        args.index_gbp, args.mode = args.index_gbp or 40.0, "long"
        if args.reads == 10_000_000:
            args.reads = 312_500
        if args.steps == 3:
            args.steps = 8          # 8 x 312 500 = cfg5's 2.5 M long reads per rank
    if args.index_gbp >= 4 and args.cpu_sample == 2_000_000:
        args.cpu_sample = 400_000       # the reference loads a 40 Gbp index for minutes per run: a smaller sample, no thread sweep
    if args.config in ("cfg4", "cfg5"):
        args.no_extra_configs = True
    if args.workload in STRAIN_WORKLOADS:
        args.species, args.strains, args.genome_len, args.divergence_step, args.divergence_model = STRAIN_WORKLOADS[args.workload][:5]
    if args.index_gbp:
        args.species = max(1, int(round(args.index_gbp * 1e9 / (args.strains * args.genome_len))))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly the way the driver does
        # (one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1), and hand its exit status back.
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("spawning", args.gpus, "ranks:", " ".join(cmd))
        raise SystemExit(subprocess.run(cmd).returncode)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in the product path)")
    # CFR_BENCH_SHARE_GPU=1: every rank uses device 0 and the process group is gloo - only for exercising the
    # multi-rank code path on a 1-GPU box; the driver's real runs are one rank per GPU over RCCL.
    share_gpu = os.environ.get("CFR_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if share_gpu else "nccl")

    all_cpus = os.sched_getaffinity(0)
    numa = bind_to_gpu_numa_node(torch, local_rank)      # host threads + pinned buffers next to this rank's GPU
    from centrifuger_amd import capi
    if args.mode == "protein":
        if world != 1:
            raise SystemExit("bench.py --mode protein is a one-GPU leg")
        os.sched_setaffinity(0, all_cpus)
        return protein_mode(torch, capi, args, device)
    key = cache_key(args)
    cache = os.path.join(args.cache, key)
    if rank == 0:
        prefix = build_index(args, cache, device)
        torch.cuda.empty_cache()
    if dist is not None:
        dist.barrier()
    prefix = os.path.join(cache, "idx")

    paired = args.mode == "pe"
    k = args.k if args.k is not None else (5 if paired else 1)
    # reads first, image second: at 40 Gbp the genomes (40 GB, only needed to draw the reads) and the device image (250 GB) do not
    # fit HBM together
    cat = np.load(os.path.join(cache, "genome_cat.npy"), mmap_mode="r")
    starts = np.load(os.path.join(cache, "genome_starts.npy"))
    cat_d = torch.from_numpy(np.ascontiguousarray(cat)).to(device)
    longmode = args.mode == "long"
    if longmode and args.reads == 10_000_000:
        args.reads = 312_500 if args.index_gbp else 1_000_000
    if longmode and args.cpu_sample == 2_000_000:
        args.cpu_sample = 20_000
    if longmode and args.count_sample == 200_000:
        args.count_sample = 4_000
    if paired:
        reads_d, reads2_d = make_pairs_gpu(torch, cat_d, starts, args.reads, args.read_len, args.seed + 1000 + rank, device)
    elif longmode:
        reads_d, offs_d = make_long_reads_gpu(torch, cat_d, starts, args.reads, args.seed + 1000 + rank, device)
        reads2_d = None
    else:
        reads_d = make_reads_gpu(torch, cat_d, starts, args.reads, args.read_len, args.seed + 1000 + rank, device)
        reads2_d = None
    del cat_d, cat
    torch.cuda.empty_cache()
    if not longmode:
        offs_d = (torch.arange(args.reads + 1, device=device, dtype=torch.int64) * args.read_len)
    torch.cuda.synchronize()
    offs_h = offs_d.cpu().numpy().astype(np.uint64)
    total_bases = int(offs_h[-1])
    t0 = time.time()
    idx = capi.Index(prefix, capi.default_params(max_result=k))
    if share_gpu and world > 1:
        # every rank on ONE device (a test of the multi-rank code path, not a configuration): the images are built one rank after the
        # other and without the tables that are sized from the free HBM (each rank would otherwise see the memory the others are about to take)
        opts = capi.default_device_options(ftabx_width=13, loc_memo_gb=0.0)
        dev = None
        for r_ in range(world):
            if r_ == rank:
                dev = capi.DeviceIndex(idx, local_rank, opts)
            dist.barrier()
    else:
        dev = capi.DeviceIndex(idx, local_rank)
    info = dev.info()
    log(f"rank {rank}: index n={info.n} b={info.block_size} loaded; device image {info.device_bytes/1e6:.0f} MB in {time.time()-t0:.1f}s")

    def sample_of(t, nsel):
        """first nsel reads of flat device buffer t as (bases, offsets) numpy"""
        hi = int(offs_h[nsel])
        return t.reshape(-1)[:hi].cpu().numpy(), offs_h[:nsel + 1].copy()
    res_pin = capi.PinnedArray(args.reads, capi.RESULT_DTYPE)     # cfr_host_alloc: D2H at PCIe rate
    mat_pin = capi.PinnedArray(args.reads * k, capi.MATCH_DTYPE)
    results, matches = res_pin.array, mat_pin.array
    # the timed entry: cfr_classify_batch_resident_compact (same results, 20 + 12 bytes per read / match slot instead of
    # 40 + 24: the step's D2H halves) unless --results wide; the other one is timed as a sub-result
    entry = {"compact": args.results == "compact"}
    cres_pin = capi.PinnedArray(args.reads, capi.RESULT_COMPACT_DTYPE)
    cmat_pin = capi.PinnedArray(args.reads * k, capi.MATCH_COMPACT_DTYPE)

    def step():
        f = dev.classify_resident_compact if entry["compact"] else dev.classify_resident
        r, m = (cres_pin.array, cmat_pin.array) if entry["compact"] else (results, matches)
        if paired:
            return f(reads_d.data_ptr(), offs_d.data_ptr(), args.reads, total_bases, reads2_d.data_ptr(), offs_d.data_ptr(), total_bases, results=r, matches=m)
        return f(reads_d.data_ptr(), offs_d.data_ptr(), args.reads, total_bases, results=r, matches=m)

    def widen():
        """the wide arrays every check below reads, from what the timed entry delivered"""
        if entry["compact"]:
            r, m = capi.expand_compact(cres_pin.array, cmat_pin.array, k)
            results[:] = r
            matches[:len(m)] = m

    if args.inner and args.inner_dust:
        dev.set_dust(True)                               # (child of the SDUST counter passes: the step with the pre-step on the device)
    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kstats = []
    for _ in range(args.steps):
        step()
        kstats.append(dev.last_stats())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    from centrifuger_amd import shard
    elapsed = shard.max_over_ranks(elapsed, dist=dist, device=None if share_gpu else device)   # the job is as slow as its slowest rank
    widen()
    classified = int((results["n_match"] > 0).sum())
    if args.inner:      # run under rocprofv3 by the parent bench: the timed launches are all it is for
        if rank == 0:
            print(json.dumps({"inner": True, "reads": args.reads, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps,
                              "search_ms": float(np.mean([s.search_ms for s in kstats])), "tail_ms": float(np.mean([s.tail_ms for s in kstats])),
                              "device_total_ms": float(np.mean([s.total_ms for s in kstats]))}), flush=True)
        return
    # the same step with the SDUST pre-step on the device (private copy of the reads + k_dust); never `value`
    dev.set_dust(True)
    step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.sdust_steps):
        step()
    torch.cuda.synchronize()
    ms_with_dust = 1000.0 * (time.perf_counter() - t1) / args.sdust_steps
    dev.set_dust(False)
    # the other result layout, timed alike (sub-result)
    entry["compact"] = not entry["compact"]
    step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms_other = 1000.0 * (time.perf_counter() - t1) / args.steps
    other_is = "compact" if entry["compact"] else "wide"
    entry["compact"] = not entry["compact"]
    step()                                               # results of the plain step again (the checks below read them)
    torch.cuda.synchronize()
    widen()
    os.sched_setaffinity(0, all_cpus)                    # the CPU legs below (oracle counters, reference baseline) get every core back

    # the ranks part here: nothing below needs the process group (rank 0 goes on alone with the CPU baseline, the live PMC
    # passes and the sub-results - minutes during which the other ranks must not be held, nor rank 0 wait for them at exit)
    # how every rank holds the index's bit strings: mapped from the one file (the ranks of a node share its pages) or copied
    mapped_b, copied_b = idx.mapped_bytes()
    ranks_index = [(int(mapped_b), int(copied_b), float(info.device_bytes))]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks_index[0])
        ranks_index = gathered
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank != 0:
        return

    total_reads = args.reads * args.steps * world
    value = total_reads / elapsed
    search_ms = float(np.mean([s.search_ms for s in kstats]))
    out = {
        "metric": "classified reads/sec (150 bp)", "value": value, "unit": "read pairs/s" if paired else "reads/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "index_on_ranks": {"ranks": len(ranks_index), "mapped_bytes_min": min(r_[0] for r_ in ranks_index), "copied_bytes_max": max(r_[1] for r_ in ranks_index),
                           "device_image_bytes": ranks_index[0][2],
                           "every_rank_maps_the_file": all(r_[0] > 0 and r_[1] == 0 for r_ in ranks_index)},
        "config": {"workload": f"{info.n/1e9:.2f} Gbp synthetic index ({args.species}x{args.strains}x{args.genome_len/1e6:g} Mbp), " +
                               (f"{args.reads} long reads (5-20 kbp, mean {total_bases/args.reads:.0f} bp) per step per GPU, -k {k}, inputs resident in HBM" if longmode else
                                f"{args.reads} x {'2x' if paired else ''}{args.read_len} bp {'PE' if paired else 'SE'} reads per step per GPU, -k {k}, inputs resident in HBM"),
                   "index_bp": int(info.n), "reads_per_step_per_gpu": args.reads, "read_len": args.read_len,
                   "parallelism": f"reads sharded over {world} GPU(s), index replicated, no collective",
                   "numa_node_of_rank0": numa},
        "classified_fraction": classified / args.reads,
        "stage_ms": {k: float(np.mean([getattr(s, k) for s in kstats])) for k in
                     ("pack_ms", "search_ms", "adjust_ms", "rows_ms", "locate_ms", "tail_ms", "total_ms")},
    }

    out["config"]["timed_entry"] = ("cfr_classify_batch_resident_compact (20 + 12 bytes per read / match slot to the host)" if entry["compact"]
                                    else "cfr_classify_batch_resident (40 + 24 bytes per read / match slot to the host)")
    out["other_result_layout"] = {"layout": other_is, "value": args.reads * world / (ms_other / 1e3), "unit": out["unit"], "ms_per_step": ms_other,
                                  "note": "the same step through the other entry (rank 0's own clock): the results are the same fields in the other widths"}
    out["with_device_sdust"] = {"value": args.reads / (ms_with_dust / 1e3), "unit": out["unit"], "ms_per_step": ms_with_dust,
                                "note": "the same step with the reference's default pre-step (SDUST, CentrifugerClass.cpp:276-316) done on the device: "
                                        f"unmasked reads resident in HBM in, private masked copy + k_dust + the step; rank 0's clock over {args.sdust_steps} steps", "steps": args.sdust_steps}

    # ---- roofline of the dominant kernel
    import oracle_lib as ora
    ns = min(args.count_sample, args.reads)
    sample, soffs = sample_of(reads_d, ns)
    sample2 = sample_of(reads2_d, ns)[0] if paired else None
    o = ora.OracleIndex(prefix, max_result=k)
    threads = min(os.cpu_count() or 1, 64)
    ores, cnt = o.classify(sample, soffs, sample2, soffs if paired else None, threads=threads, counters=True)
    c = cnt.as_dict()
    # the timed entry's results against the C oracle on that sample (every field of every read; the first 5000 also as TSV lines)
    same_fields = all((int(results[i]["score"]), int(results[i]["secondary_score"]), int(results[i]["hit_length"]), int(results[i]["query_length"]), int(results[i]["n_match"])) ==
                      (ores[i].score, ores[i].secondaryScore, ores[i].hitLength, ores[i].queryLength, ores[i].nmatch) for i in range(ns))
    same_tsv = all(idx.format_tsv("r", results[i], matches) == o.format("r", ores[i]) for i in range(min(ns, 5000)))
    # ... and the LAST reads of the batch (the last chains the lanes take: a hand-out that loses its tail shows here, not in the first reads)
    nt = min(20_000, args.reads)
    lo_ = args.reads - nt
    a_, b_ = int(offs_h[lo_]), int(offs_h[args.reads])
    tb = reads_d.reshape(-1)[a_:b_].cpu().numpy()
    tb2 = reads2_d.reshape(-1)[a_:b_].cpu().numpy() if paired else None
    toffs = (offs_h[lo_:args.reads + 1] - offs_h[lo_]).astype(np.uint64)
    tres = o.classify(tb, toffs.copy(), tb2, toffs.copy() if paired else None, threads=threads)
    same_tail = all((int(results[lo_ + i]["score"]), int(results[lo_ + i]["secondary_score"]), int(results[lo_ + i]["hit_length"]), int(results[lo_ + i]["query_length"]), int(results[lo_ + i]["n_match"])) ==
                    (tres[i].score, tres[i].secondaryScore, tres[i].hitLength, tres[i].queryLength, tres[i].nmatch) for i in range(nt))
    out["parity_oracle"] = {"reads": ns, "tail_reads": nt, "tsv_lines": min(ns, 5000), "equals_oracle": bool(same_fields and same_tsv and same_tail),
                            "note": "timed entry (no pre-step) vs oracle/liboracle.so Query on the first reads of the step batch and on its last 20 000: score, second score, hit length, query length, match count of every read; TSV lines of the first 5000"}
    try:
        out["index"] = json.load(open(prefix + ".build.json"))
    except Exception:
        pass
    # reference-algorithm bytes of the search part = everything except the locate part
    bytes_search = cnt.search_bytes() / ns
    bytes_locate = cnt.locate_bytes() / ns
    ref_alg_gbs = bytes_search * args.reads / (search_ms / 1e3) / 1e9
    copy_gbs = None
    try:   # what a plain device copy sustains on this box
        xa = torch.empty(1 << 31, dtype=torch.uint8, device=device)
        xb = torch.empty_like(xa)
        xb.copy_(xa)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            xb.copy_(xa)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * xa.numel() / (e0.elapsed_time(e1) / 1e3) / 1e9      # read + write
        del xa, xb
    except Exception:
        pass
    def build_roofline(pmc):
      roof = {"bound": "hbm", "kernel": "k_search_chains_v2", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel_ms": search_ms,
              "per": "one step = the launches of the step's sub-batches (traffic and kernel_ms are summed over them)",
              "measured_copy_GBs": copy_gbs}
      if pmc is None and args.mode == "se" and args.read_len == 150:
          try:    # no live passes: the committed profile of the same kernel, flagged as such - and only when it was taken on this
              # very index size with this very kernel source (another image has another request mix: traffic stays null then)
              pmj = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
              if abs(float(pmj.get("index_bp", 1e9)) - float(info.n)) > 0.02 * float(info.n) or pmj.get("kernel_source_sha") != kernel_source_sha():
                  raise ValueError("committed profile is of another index size or kernel source")
              pm = pmj["k_search_chains_v2"]
              pmc = {"source": f"COMMITTED profile profiles/pmc_latest.json ({pmj.get('round')}), not this run", "reads": pmj["reads_in_profiled_launch"],
                     "rdreq": pm["TCC_EA0_RDREQ"], "rdreq_32b": pm.get("TCC_EA0_RDREQ_32B") or 0.0, "fetch_size_kib": pm["FETCH_SIZE_KiB"],
                     "write_size_kib": pm["WRITE_SIZE_KiB"], "kernel_ms_profiled": None, "prof": None,
                     "kernel_source_sha": pmj.get("kernel_source_sha")}
          except Exception:
              pmc = None
      if pmc:
          # calibration (profiles/r2a_gather_calib.json, tools/gather_calib.sh): a fabric read request of a random gather is one
          # 128-byte line (two loads in the two halves of one line: 1.04 requests; FETCH_SIZE tallies it at 64 bytes)
          per_read_rd = pmc["rdreq"] * BYTES_PER_RANDOM_REQUEST / pmc["reads"]
          per_read_wr = pmc["write_size_kib"] * 1024 / pmc["reads"]
          traffic = (per_read_rd + per_read_wr) * args.reads
          ach = traffic / (search_ms / 1e3) / 1e9
          req_s = pmc["rdreq"] / pmc["reads"] * args.reads / (search_ms / 1e3)
          roof.update({
              "achieved": None, "frac": None, "achieved_counter_traffic": ach, "frac_counter_traffic": ach / HBM_PEAK_GBS, "traffic": traffic,
              "traffic_source": pmc["source"], "kernel_source_sha_now": kernel_source_sha(), "kernel_source_sha_of_profile": pmc.get("kernel_source_sha"),
              "fabric_read_requests_per_read": pmc["rdreq"] / pmc["reads"], "bytes_per_read_request": BYTES_PER_RANDOM_REQUEST,
              "calibration": "profiles/r2a_gather_calib.json: random 16-byte gathers from a 32 GB table, 1.00 request per touched 128-byte line "
                             "(both halves of a line: 1.04), 48 G requests/s = 6.2 TB/s; FETCH_SIZE counts a request as 64 bytes",
              "read_bytes_per_read": per_read_rd, "write_bytes_per_read": per_read_wr,
              "fetch_size_counter_bytes_per_read": pmc["fetch_size_kib"] * 1024 / pmc["reads"],
              "kernel_ms_under_profiler_per_read_x_reads": (pmc["kernel_ms_profiled"] / pmc["reads"] * args.reads) if pmc.get("kernel_ms_profiled") else None,
              "gather": {"requests_per_s": req_s, "ceiling_per_s": 48e9, "frac": req_s / 48e9,
                         "note": "fabric read requests per second over tools/gather_bench's ceiling for dependent random gathers at this footprint"},
              "note": "achieved = bytes the kernel moves over the fabric (PMC TCC_EA0_RDREQ x 128 B calibrated + WRITE_SIZE, one 2 M-read launch "
                      "under rocprofv3 --pmc, scaled per read) / kernel time of the timed steps (HIP events on the library stream)"})
          if pmc.get("kernel_ms_profiled"):
              # the same traffic over the kernel's duration when it has the chip to itself (the one-launch counter pass: no post stage
              # of a previous sub-batch beside it) - `frac` is taken over the time it needs inside the step, sharing the chip with that post stage
              alone_ms = pmc["kernel_ms_profiled"] / pmc["reads"] * args.reads
              roof["frac_kernel_alone"] = traffic / (alone_ms / 1e3) / 1e9 / HBM_PEAK_GBS
              roof["kernel_ms_alone"] = alone_ms
          roof["l2_hit"] = pmc.get("l2_hit")
          roof["instruction_stream"] = instruction_stream(pmc)
          if pmc.get("prof"):
              pr = pmc["prof"]
              useful = useful_bytes_per_read(pr, c.get("hits", 0) / ns)
              roof["useful_bytes_per_read"] = useful
              roof["iteration_mix_per_read"] = pr
              roof["fetched_over_useful"] = (per_read_rd + per_read_wr) / useful if useful else None
              roof["achieved"] = useful * args.reads / (search_ms / 1e3) / 1e9          # the plain pair achieved / frac = USEFUL bytes
              roof["frac_useful_bytes"] = roof["achieved"] / HBM_PEAK_GBS
              roof["frac"] = roof["frac_useful_bytes"]
      else:
          roof.update({"achieved": None, "frac": None, "achieved_counter_traffic": None, "frac_counter_traffic": None, "traffic": None,
                       "note": "no PMC source available (rocprofv3 failed and no committed profile for this workload)"})
      # ONE place for the yardsticks: achieved / frac = frac_useful_bytes = the bytes of the fetched lines the kernel
      # consumes over its time; frac_counter_traffic = bytes the counters saw cross the fabric (128-byte lines); x_reference_algorithm =
      # SURVEY.md section 8(d)'s algorithmic bytes of the REFERENCE's data structures over this kernel's time, as a multiple of the
      # peak (> 1: the kernel does not perform that memory work - K-mer table, text mode, step function replace it; results identical)
      roof["x_reference_algorithm"] = ref_alg_gbs / HBM_PEAK_GBS
      roof["yardsticks"] = ("achieved / frac = frac_useful_bytes: bytes of the fetched lines the kernel consumes / kernel time (/ 8 TB/s); achieved_counter_traffic / "
                            "frac_counter_traffic: fabric bytes by PMC (TCC_EA0_RDREQ x 128 B + WRITE_SIZE) / kernel time; gather.frac: fabric requests per second / "
                            "the 48 G/s this chip sustains for dependent random gathers; l2_hit: TCC_HIT / (TCC_HIT + TCC_MISS); x_reference_algorithm: "
                            "SURVEY 8(d) bytes of the reference algorithm / kernel time / 8 TB/s (an algorithmic speed-up, not a bandwidth)")
      roof["reference_algorithm"] = {
          "bytes_per_read": bytes_search, "GBs_if_the_reference_traffic_were_moved": ref_alg_gbs, "x_of_hbm_peak": ref_alg_gbs / HBM_PEAK_GBS,
          "note": "SURVEY.md section 8(d) figure: bytes the REFERENCE algorithm reads for the same searches (24 B/bit-rank, 8 B/bit-access, 16 B/ftab, "
                  "read bytes, 32 B/hit; counted exactly by the C oracle on a sample) / this kernel's time.  Above 1 because the flat occurrence "
                  "image, the K-mer table, text mode and the locate memo do not perform that traffic - an algorithmic speed-up, not a bandwidth",
          "locate_part_bytes_per_read": bytes_locate, "whole_query_bytes_per_read": cnt.algorithmic_bytes() / ns,
          "ops_per_read": {kk: v / ns for kk, v in c.items()}}
      return roof

    # ---- CPU baselines + parity
    cli_job = None
    refbin = os.path.join(ROOT, "oracle", "_ref", "centrifuger")
    if not args.no_cpu_baseline and os.path.exists(refbin):    # reported baseline: rank 0 (the other ranks have left; at N > 1 on a smaller sample)
        from centrifuger_amd import synth
        ncpu = os.cpu_count() or 1
        nb = min(args.cpu_sample if world == 1 else max(1, args.cpu_sample // 4), args.reads)
        rs = synth.ReadSet(*sample_of(reads_d, nb))
        fa = os.path.join(cache, f"sample_{rank}.fa")
        synth.write_fasta(rs, fa)
        one = os.path.join(cache, "one.fa")
        synth.write_fasta(rs.slice(0, 1), one)
        files, files_one = ["-u", fa], ["-u", one]
        rs2 = None
        if paired:
            rs2 = synth.ReadSet(sample_of(reads2_d, nb)[0], rs.offsets.copy())
            fa2 = os.path.join(cache, f"sample_{rank}_2.fa")
            synth.write_fasta(rs2, fa2)
            one2 = os.path.join(cache, "one_2.fa")
            synth.write_fasta(rs2.slice(0, 1), one2)
            files, files_one = ["-1", fa, "-2", fa2], ["-1", one, "-2", one2]

        def ref_run(t, fl, extra=()):
            t0 = time.time()
            o_ = subprocess.run([refbin, "-x", prefix, "-t", str(t), "-k", str(k)] + list(extra) + fl, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
            return time.time() - t0, o_
        t_load = ref_run(min(ncpu, 64), files_one)[0]
        # thread sweep on a quarter of the sample: the reference starts its threads per batch and reads with one thread, so
        # its best thread count is not necessarily all cores
        nq = max(1, nb // 4)
        sweep = {}
        if not paired and info.n < 4_000_000_000:      # (every run of the reference loads the index again: no sweep on large ones)
            fq = os.path.join(cache, f"sample_{rank}_q.fa")
            synth.write_fasta(rs.slice(0, nq), fq)
            for t in sorted({32, 64, 128, ncpu}):
                if t <= ncpu:
                    sweep[t] = nq / max(ref_run(t, ["-u", fq])[0] - t_load, 1e-9)
        best_t = max(sweep, key=sweep.get) if sweep else min(ncpu, 64)
        t_full, ref_tsv = ref_run(best_t, files)
        cpu_rate = nb / max(t_full - t_load, 1e-9)
        t_nodust, ref_tsv_nodust = ref_run(best_t, files, ["--no-dust"])
        # (1) parity of the TIMED entry: cfr_classify_batch_resident without the pre-step == the reference run with --no-dust
        timed_tsv = capi.tsv_header() + b"".join(idx.format_tsv(f"r{i}", results[i], matches) for i in range(nb))
        # (2) default options: unmasked reads in, SDUST on the device (cfr_device_index_set_dust) like the reference's pre-step
        b = rs.bases.copy()
        dev.set_dust(True)
        if paired:
            bb2 = rs2.bases.copy()
            r2, m2 = dev.classify(b, rs.offsets, bb2, rs2.offsets)
            dev.set_dust(False)
        else:
            r2, m2 = dev.classify(b, rs.offsets)
            dev.set_dust(False)
            capi.dust_mask(b, rs.offsets, threads=min(ncpu, 64))      # the timed legs below take masked reads (the Query contract)
            dev.classify(b, rs.offsets)                                   # untimed: page in, size the staging buffers
            t0 = time.perf_counter()
            dev.classify(b, rs.offsets)
            fresh_rate = nb / (time.perf_counter() - t0)
            # ordinary (pageable) host memory, the way a caller that recycles its batch buffers hands it over: allocated once and
            # touched.  (With FRESH result arrays per call the same entry spends most of its time in the page faults of those
            # arrays: that figure is kept as fresh_arrays_value - earlier rounds reported it as the pageable rate.)
            hb = reads_d.reshape(-1).cpu().numpy().copy()
            res_pg = np.empty(args.reads, dtype=capi.RESULT_DTYPE); res_pg.view(np.uint8)[:] = 0
            mat_pg = np.empty(args.reads * max(1, k), dtype=capi.MATCH_DTYPE); mat_pg.view(np.uint8)[:] = 0
            dev.classify(hb, offs_h, results=res_pg, matches=mat_pg)
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                dev.classify(hb, offs_h, results=res_pg, matches=mat_pg)
                best = min(best, time.perf_counter() - t0)
            out["pcie_inclusive"] = {"value": args.reads / best, "unit": "reads/s",
                                     "note": f"cfr_classify_batch on {args.reads} reads from pageable host memory (numpy arrays allocated once and touched) to pageable host memory "
                                             f"(H2D 158 B/read, D2H 64 B/read), best of 2",
                                     "fresh_arrays_value": fresh_rate,
                                     "fresh_arrays_note": f"{nb} reads, result / match arrays allocated inside the call (numpy.zeros: first-touch page faults in the timed region)"}
            pageable_equal = bool(res_pg[:nb].tobytes() == results[:nb].tobytes())
            del hb, res_pg, mat_pg
            # the whole step batch with every host buffer pinned (cfr_host_alloc): the bases of sub-batch k+1 go up while sub-batch k computes
            pb = capi.PinnedArray(total_bases, np.uint8)
            po = capi.PinnedArray(args.reads + 1, np.uint64)
            pb.array[:] = reads_d.reshape(-1).cpu().numpy()
            po.array[:] = offs_h
            res_keep = results[:nb].copy()
            dev.classify(pb.array, po.array, results=results, matches=matches)
            t0 = time.perf_counter()
            dev.classify(pb.array, po.array, results=results, matches=matches)
            out["pcie_inclusive"]["pinned_value"] = args.reads / (time.perf_counter() - t0)
            out["pcie_inclusive"]["pinned_note"] = f"{args.reads} reads, bases / offsets / results / matches all in cfr_host_alloc memory"
            out["pcie_inclusive"]["host_entry_equals_resident_entry"] = bool(res_keep.tobytes() == results[:nb].tobytes()) and pageable_equal
            # the same batch handed over PACKED (cfr_classify_batch_packed: 0.5 byte per base instead of 1 over the link); packing on host threads timed apart
            pk = capi.PinnedArray((total_bases + 15) // 16, np.uint64)
            capi.pack_reads(pb.array, threads=min(ncpu, 64), out=pk.array)
            t0 = time.perf_counter()
            capi.pack_reads(pb.array, threads=min(ncpu, 64), out=pk.array)
            t_pack = time.perf_counter() - t0
            dev.classify_packed(pk.array, po.array, results=results, matches=matches)
            bestp = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                dev.classify_packed(pk.array, po.array, results=results, matches=matches)
                bestp = min(bestp, time.perf_counter() - t0)
            out["pcie_inclusive"]["packed_pinned_value"] = args.reads / bestp
            out["pcie_inclusive"]["packed_pinned_note"] = (f"cfr_classify_batch_packed: {args.reads} reads as 2-bit blocks + validity bits (75 B/read + 8 B offset up, 64 B/read down), "
                                                           f"every buffer in cfr_host_alloc memory, best of 3; cfr_pack_reads of the batch on {min(ncpu, 64)} host threads: {t_pack*1e3:.1f} ms = {args.reads/t_pack:.3g} reads/s (not inside)")
            out["pcie_inclusive"]["packed_entry_equals_resident_entry"] = bool(res_keep.tobytes() == results[:nb].tobytes())
            out["pcie_inclusive"]["host_pack_reads_per_s"] = args.reads / t_pack
            # what a host caller with the reference's default options gets: unmasked reads in pinned host memory, SDUST on the device
            dev.set_dust(True)
            dev.classify(pb.array, po.array, results=results, matches=matches)
            t0 = time.perf_counter()
            for _ in range(args.sdust_steps):
                dev.classify(pb.array, po.array, results=results, matches=matches)
            out["pcie_inclusive"]["pinned_with_sdust_value"] = args.reads * args.sdust_steps / (time.perf_counter() - t0)
            out["pcie_inclusive"]["pinned_with_sdust_note"] = (f"cfr_classify_batch with cfr_device_index_set_dust(1): {args.reads} unmasked reads from cfr_host_alloc memory, "
                                                               f"results to cfr_host_alloc memory, {args.sdust_steps} steps")
            dev.classify_packed(pk.array, po.array, results=results, matches=matches)
            t0 = time.perf_counter()
            for _ in range(args.sdust_steps):
                dev.classify_packed(pk.array, po.array, results=results, matches=matches)
            out["pcie_inclusive"]["packed_pinned_with_sdust_value"] = args.reads * args.sdust_steps / (time.perf_counter() - t0)
            dev.set_dust(False)
            pk.free()
            pb.free()
            po.free()
        gpu_tsv = capi.tsv_header() + b"".join(idx.format_tsv(f"r{i}", r2[i], m2) for i in range(nb))
        out["cpu_baseline"] = {"value": cpu_rate, "unit": out["unit"], "cores": best_t, "kind": "reference",
                               "sample": f"first {nb} {'pairs' if paired else 'reads'} of the step batch, oracle/_ref/centrifuger -t {best_t} -k {k} "
                                         f"(best of the thread sweep), wall {t_full:.1f}s minus index-load run {t_load:.1f}s: END TO END (FASTA parse, dust, classify, TSV)",
                               "thread_sweep_reads_per_s": {str(t): v for t, v in sweep.items()}, "host_cores": ncpu,
                               "no_dust_value": nb / max(t_nodust - t_load, 1e-9)}
        # classify-only on the host cores: the C restatement (oracle, same data structures as the reference) without parse / dust / output
        try:
            nco = min(nb, 1_000_000)
            cs, co = sample_of(reads_d, nco)
            cs2 = sample_of(reads2_d, nco)[0] if paired else None
            capi.dust_mask(cs, co, threads=min(ncpu, 64))
            t0 = time.perf_counter()
            o.classify(cs, co, cs2, co if paired else None, dust=False, threads=ncpu)
            out["cpu_baseline"]["classify_only"] = {"value": nco / (time.perf_counter() - t0), "kind": "port", "cores": ncpu,
                                                    "sample": f"{nco} masked reads through oracle/liboracle.so (Query only: no parse, no dust, no TSV)"}
        except Exception as e:
            out["cpu_baseline"]["classify_only"] = {"error": str(e)}
        out["parity"] = {"reads": nb,
                         "timed_entry_tsv_identical_to_reference_no_dust": timed_tsv == ref_tsv_nodust,
                         "tsv_identical_to_reference": gpu_tsv == ref_tsv,
                         "md5": hashlib.md5(gpu_tsv).hexdigest(), "md5_timed_entry": hashlib.md5(timed_tsv).hexdigest(),
                         "note": "timed entry = cfr_classify_batch_resident, the call `value` times (reads as they are, no pre-step) vs `centrifuger --no-dust`; "
                                 "the second check hands unmasked reads to cfr_classify_batch with SDUST on the device vs the reference's default run"}
        out["kernel_vs_e2e_cpu"] = value / world / cpu_rate      # resident-input device step over the END-TO-END reference: not like for like (see e2e_cli.speedup)
        cli_job = (files, ref_tsv, t_full, nb, ncpu)
    if args.sub_result and args.config == "cfg4":
        # what 8 ranks of one node do to each other at load time on the HOST side: 8 processes open this index at once (cfr_index_open:
        # the 15 GB .1.cfr through the page cache into 8 private copies) against one process alone.  (The device half - 8 images
        # derived side by side - needs 8 GPUs.)
        try:
            # (round 5: cfr_index_open and the digest timed apart - the open leaves the bit strings in the file's mapping and is what a
            #  rank waits for before it starts uploading; the digest reads all 15 GB and was most of the figure rounds 3-4 reported)
            code = ("import sys,time; sys.path.insert(0, %r); from centrifuger_amd import capi; t0=time.time(); i=capi.Index(%r); t1=time.time(); d=i.digest(); print(t1-t0, time.time()-t0, d)" % (ROOT, prefix))
            def opens(k_):
                t0_ = time.time()
                ps = [subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(k_)]
                outs = [p_.communicate()[0].decode().split() for p_ in ps]
                outs = [o_ for o_ in outs if len(o_) == 3]
                return time.time() - t0_, [float(o_[0]) for o_ in outs], [float(o_[1]) for o_ in outs], {o_[2] for o_ in outs}
            w1, t1, td1, d1 = opens(1)
            w8, t8, td8, d8 = opens(8)
            out["multi_rank_load"] = {"index_open_alone_s": t1[0] if t1 else None, "index_open_8_at_once_s_each": t8,
                                      "open_plus_digest_alone_s": td1[0] if td1 else None, "open_plus_digest_8_at_once_s_each": td8, "wall_8_at_once_s": w8,
                                      "digests_equal": len(d1 | d8) == 1 and len(t8) == 8,
                                      "note": "cfr_index_open of this index in 1 process and in 8 processes started together; open_plus_digest adds cfr_index_digest, which reads "
                                              "every array of the 15 GB file (the figure rounds 3-4 reported as the open); process start and imports are in the wall figure only"}
        except Exception as e:
            out["multi_rank_load"] = {"error": repr(e)}
    # ---- the live PMC passes need the GPU to themselves (the K-mer table is sized from the free HBM): this process lets go of
    # its image and reads first, so the child builds exactly the image that was timed
    dev.close()
    del reads_d, reads2_d
    torch.cuda.empty_cache()
    out["roofline"] = build_roofline(live_pmc(args, cache, local_rank) if not args.no_pmc else None)    # (rank 0's GPU; the other ranks have left)
    if not args.no_pmc and args.mode == "se":
        out["post_stage"] = {"roofline": post_stage_roofline(live_pmc_post(args, cache, local_rank), args.reads, out["stage_ms"].get("tail_ms"),
                                                             out["roofline"].get("fabric_read_requests_per_read"), c.get("hits", 0) / ns, c.get("locates", 0) / ns)}
    if not args.no_pmc and args.mode == "se" and not args.sub_result:
        out["with_device_sdust"]["roofline"] = dust_roofline(live_pmc_dust(args, cache, local_rank), args.reads, ms_with_dust - 1000.0 * elapsed / args.steps)
    if cli_job is not None:
        # ---- end-to-end wall clock of the drop-in command line on the same file (parse + dust + device + TSV, index load included);
        # runs with the GPU to itself, like a user's run
        files, ref_tsv, t_full, nb, ncpu = cli_job
        cli = os.path.join(ROOT, "centrifuger_amd", "bin", "centrifuger")
        if os.path.exists(cli):
            try:
                t0 = time.time()
                cli_tsv = subprocess.run([cli, "-x", prefix, "-t", str(min(ncpu, 64)), "-k", str(k)] + files, check=True,
                                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
                t_cli = time.time() - t0
                out["e2e_cli"] = {"reads": nb, "seconds": t_cli, "reference_seconds": t_full, "speedup": t_full / t_cli,
                                  "tsv_identical_to_reference": cli_tsv == ref_tsv,
                                  "note": "wall clock of `centrifuger -x idx ...` on the sample file, index load / device image included on both sides: "
                                          "the like-for-like ratio"}
            except Exception as e:
                out["e2e_cli"] = {"error": repr(e)}
            # ---- steady state: cfg4's read count (100 M) through the command line - the sample file back to back (the ids repeat, so the
            # rows are the reference's rows of the sample, repeated); a run of a few seconds in which the index load no longer dominates
            if world == 1 and args.mode == "se" and not paired and not args.sub_result and len(files) == 2 and not args.no_extra_configs:
                try:
                    reps = max(1, 100_000_000 // nb)
                    big = os.path.join(cache, "e2e_100m.fa")
                    with open(big, "wb") as fo:
                        blob = open(files[1], "rb").read()
                        for _ in range(reps):
                            fo.write(blob)
                    del blob
                    tsv_path = os.path.join(cache, "e2e_100m.tsv")
                    t0 = time.time()
                    with open(tsv_path, "wb") as fo:
                        subprocess.run([cli, "-x", prefix, "-t", str(min(ncpu, 64)), "-k", str(k), "-u", big], check=True, stdout=fo, stderr=subprocess.DEVNULL)
                    t_big = time.time() - t0
                    h_got = hashlib.md5()
                    with open(tsv_path, "rb") as fi:
                        for chunk in iter(lambda: fi.read(1 << 24), b""):
                            h_got.update(chunk)
                    head, _, rows = ref_tsv.partition(b"\n")
                    h_want = hashlib.md5(head + b"\n")
                    for _ in range(reps):
                        h_want.update(rows)
                    out["e2e_cli_100m"] = {"reads": nb * reps, "seconds": t_big, "value": nb * reps / t_big, "unit": "reads/s",
                                           "md5_equals_reference_rows": h_got.hexdigest() == h_want.hexdigest(),
                                           "reference_reads_per_s": nb / t_full, "speedup": (nb * reps / t_big) / (nb / t_full),
                                           "note": "wall clock of `centrifuger -x idx -u 100M.fa -t 64 > file` (process start, index load, device image, parse, SDUST on the "
                                                   "device, classify, TSV); the reference's rate is its run on the sample (it scales linearly in reads)"}
                    os.unlink(big)
                    os.unlink(tsv_path)
                except Exception as e:
                    out["e2e_cli_100m"] = {"error": repr(e)}
                # ---- the same 100 M reads as FASTQ (what sequencers write: twice the bytes, four lines per record)
                try:
                    bigq = os.path.join(cache, "e2e_100m.fq")
                    lines = open(files[1], "rb").read().split(b"\n")
                    recs = []
                    for j in range(0, len(lines) - 1, 2):
                        recs.append(b"@" + lines[j][1:] + b"\n" + lines[j + 1] + b"\n+\n" + b"I" * len(lines[j + 1]) + b"\n")
                    blob = b"".join(recs)
                    del lines, recs
                    with open(bigq, "wb") as fo:
                        for _ in range(reps):
                            fo.write(blob)
                    del blob
                    t0 = time.time()
                    with open(tsv_path, "wb") as fo:
                        subprocess.run([cli, "-x", prefix, "-t", str(min(ncpu, 64)), "-k", str(k), "-u", bigq], check=True, stdout=fo, stderr=subprocess.DEVNULL)
                    t_bigq = time.time() - t0
                    h_got = hashlib.md5()
                    with open(tsv_path, "rb") as fi:
                        for chunk in iter(lambda: fi.read(1 << 24), b""):
                            h_got.update(chunk)
                    out["e2e_cli_100m_fastq"] = {"reads": nb * reps, "seconds": t_bigq, "value": nb * reps / t_bigq, "unit": "reads/s",
                                                 "md5_equals_reference_rows": h_got.hexdigest() == h_want.hexdigest(),
                                                 "note": "the same reads as a plain FASTQ file (30 GB) through `centrifuger -x idx -u 100M.fq -t 64 > file`, wall clock of the process"}
                    os.unlink(bigq)
                    os.unlink(tsv_path)
                except Exception as e:
                    out["e2e_cli_100m_fastq"] = {"error": repr(e)}
    # ---- the other BASELINE configs on the same index, as sub-results (configs[2] paired-end -k 5, configs[4]-style long reads)
    if world == 1 and args.mode == "se" and not args.no_extra_configs and not args.inner:
        out["other_configs"] = {}
        for m in ("pe", "long"):
            try:
                out["other_configs"][m] = extra_config(torch, capi, ora, args, m, prefix, cache, device)
            except Exception as e:
                out["other_configs"][m] = {"error": repr(e)}
        if args.workload == "cfg2" and not args.index_gbp:
            for wname in ("strains20", "strains200"):
                try:
                    out["other_configs"][wname] = strains_config(torch, capi, ora, args, device, wname)
                except Exception as e:
                    out["other_configs"][wname] = {"error": repr(e)}
            if not args.no_40gbp:
                # BASELINE configs[3] / configs[4] on ONE GPU: the 40 Gbp index (written once by the native writer, ~4 min) and this
                # rank's share of the reads - 12.5 M x 150 bp per step, 312 500 long reads per step - each in a process of its own
                for cfg in ("cfg4", "cfg5"):
                    out["other_configs"][cfg + "_1gpu"] = sub_config_40gbp(args, cfg)
    emit_line(out, args)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
