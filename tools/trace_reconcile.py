#!/usr/bin/env python3
"""Per-STEP kernel times of a rocprofv3 --kernel-trace of `bench.py --inner` (warm-up + timed steps of the plain entry, nothing else
in the process) next to what the run itself measured: every dispatch divided by the steps the process ran, the copy / fill blits
counted, the sum checked against ms_per_step.  usage: trace_reconcile.py <trace.db> <inner.json> [out.txt]"""
import json
import sqlite3
import sys


def main(db_path, inner_path, out=None):
    inner = None
    for ln in open(inner_path):
        if ln.startswith("{") and '"inner"' in ln:
            inner = json.loads(ln)
    steps = inner["steps"] + inner["warmup"]
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), max(vgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc").fetchall()
    lines = [f"# rocprofv3 --kernel-trace of `bench.py --inner --reads {inner['reads']} --steps {inner['steps']} --warmup {inner['warmup']}`: {steps} steps in the process",
             f"# the run's own clock (under the profiler): ms_per_step {inner['ms_per_step']:.3f}, HIP events per step: search {inner['search_ms']:.3f}, post stage {inner['tail_ms']:.3f}, first to last event {inner['device_total_ms']:.3f}",
             f"{'calls':>6} {'per step':>9} {'ms per step':>12} {'avg us':>10} {'vgpr':>5} {'lds':>6} {'scratch':>7}  name"]
    per_step_total = 0.0
    blit_calls = blit_ms = 0.0
    for name, calls, tot, avg, vg, lds, scr in rows:
        ms = tot / 1e6 / steps
        per_step_total += ms
        if "rocclr" in name or "fillBuffer" in name or "copyBuffer" in name:
            blit_calls += calls / steps
            blit_ms += ms
        lines.append(f"{calls:>6} {calls/steps:>9.1f} {ms:>12.4f} {avg/1e3:>10.2f} {vg or 0:>5} {lds or 0:>6} {scr or 0:>7}  {name[:110]}")
    lines.append(f"# sum of all dispatch durations per step: {per_step_total:.3f} ms (dispatches on different streams overlap - the post stage of a sub-batch runs BESIDE the search of the next one, and k_adjust_tail's duration here is mostly its wait for wave slots under that search - so the sum exceeds the step's wall time; under the profiler the step itself is ~35 % longer than unprofiled)")
    lines.append(f"# runtime blits (copyBuffer / fillBuffer: the small device-side copies and memsets the library enqueues): {blit_calls:.1f} dispatches, {blit_ms:.4f} ms of dispatch time per step")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
