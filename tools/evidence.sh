#!/bin/bash
export CFR_DEBUG_ENV=1   # the gate behind which the library reads its CFR_* A/B switches
# One evidence pass on the GPU box.  usage: tools/evidence.sh <tag> [big]     -> gpurun_out/<tag>_*
#   <tag>_bench.json            the default bench line as the driver reads it (compact, <= 4 KB); <tag>_bench_detail.json: the full object (cfg2 SE + CPU baseline + parity
#                               + live PMC roofline + PE / long / strains / 40 Gbp sub-results)
#   <tag>_kernel_trace_stats    rocprofv3 --kernel-trace --stats of the same timed steps (no CPU legs)
#   <tag>_pmc_summary / _pmc_latest.json   per-kernel PMC passes (separate runs per counter group)
#   <tag>_bench_2ranks.json     bench.py --gpus 2 on this one GPU (CFR_BENCH_SHARE_GPU=1, gloo): the multi-rank code path executed
#   <tag>_bench_8gbp.json       (with "big") the same bench on an 8 Gbp index: n > 2^32, written on the box by the native writer
set -u
TAG=$1
BIG=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd $ROOT
CFR_BENCH_DETAIL=$O/${TAG}_bench_detail.json python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.log      # the compact line; the full object beside it
# kernel trace of the plain entry alone: warm-up + timed steps and nothing else in the process, so that every dispatch divides by the steps
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -- python $ROOT/bench.py --inner --no-cpu-baseline --no-pmc --no-extra-configs --steps 10 --warmup 2 > $O/${TAG}_trace.json 2> $O/${TAG}_trace.log )
db=$(find $O/${TAG}_trace -name "*.db" | head -1)
[ -n "$db" ] && python tools/trace_reconcile.py $db $O/${TAG}_trace.json $O/${TAG}_kernel_trace_stats.txt > /dev/null
rm -rf $O/${TAG}_trace
# the same command WITHOUT the profiler, for the cross-check the trace alone cannot give (kernels run ~25 % longer under rocprofv3's
# kernel trace on this box: a trace's per-step sum may exceed the unprofiled step): HIP-event durations of the unprofiled run go on top of the file
python $ROOT/bench.py --inner --no-cpu-baseline --no-pmc --no-extra-configs --steps 10 --warmup 2 > $O/${TAG}_untraced.json 2> /dev/null
python - <<PY
import json
try:
    d = json.loads(open("$O/${TAG}_untraced.json").read().strip().splitlines()[-1])
    st = d.get("stage_ms") or d
    line = ("# the same command without the profiler: ms_per_step %.3f, HIP events per step: search %.3f (= %.1f us per launch over 10 launches), post stage %.3f\n"
            % (d["ms_per_step"], st.get("search_ms", float("nan")), 100.0 * st.get("search_ms", float("nan")), st.get("tail_ms", float("nan"))))
    p = "$O/${TAG}_kernel_trace_stats.txt"
    body = open(p).read().splitlines(True)
    open(p, "w").write("".join(body[:2]) + line + "".join(body[2:]))
except Exception as e:
    print("untraced run:", e)
PY
# PMC passes: one 2 M-read launch per kernel (a single sub-batch, so per-launch counters divide by 2 M reads)
CFR_SUBBATCH=2000000 CFR_TAPER_FLOOR=0 tools/pmc_passes.sh $O/${TAG}_pmc --no-pmc --no-extra-configs > $O/${TAG}_pmc.log 2>&1
python tools/pmc_summary.py $O/${TAG}_pmc > $O/${TAG}_pmc_summary.txt 2>> $O/${TAG}_pmc.log
python tools/pmc_latest.py $O/${TAG}_pmc $TAG $O/${TAG}_pmc_latest.json >> $O/${TAG}_pmc.log 2>&1
rm -rf $O/${TAG}_pmc/pmc*/
CFR_BENCH_SHARE_GPU=1 CFR_BENCH_DETAIL=$O/${TAG}_bench_2ranks_detail.json python bench.py --gpus 2 --steps 3 --no-cpu-baseline 2> $O/${TAG}_bench_2ranks.log | grep '^{"metric"' > $O/${TAG}_bench_2ranks.json   # (gloo prints its own lines on stdout)
if [ -n "$BIG" ]; then
  CFR_BENCH_DETAIL=$O/${TAG}_bench_8gbp_detail.json python bench.py --index-gbp 8 --steps 3 --cpu-sample 500000 > $O/${TAG}_bench_8gbp.json 2> $O/${TAG}_bench_8gbp.log
fi
python - <<PY
import json
for f in ("bench", "bench_2ranks", "bench_8gbp"):
    try:
        d = json.loads(open("$O/${TAG}_%s.json" % f).read().strip().splitlines()[-1])
        assert len(json.dumps(d)) < 4096 or f != "bench", "the driver's line must stay under 4 KB"
        try:
            d = json.load(open("$O/${TAG}_%s_detail.json" % f))
        except Exception:
            pass
    except Exception as e:
        print(f, "missing:", e); continue
    r = d.get("roofline", {})
    print(f, "value %.4g n_gpus %d ms/step %.2f" % (d["value"], d["n_gpus"], d["ms_per_step"]), "roofline.frac", r.get("frac"), "parity", d.get("parity", {}).get("tsv_identical_to_reference"),
          d.get("parity", {}).get("timed_entry_tsv_identical_to_reference_no_dust"), "dust", d.get("with_device_sdust", {}).get("value"),
          {k: (v.get("value"), v.get("equals_oracle")) for k, v in d.get("other_configs", {}).items()})
PY
head -14 $O/${TAG}_kernel_trace_stats.txt
