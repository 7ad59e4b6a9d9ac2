#!/bin/bash
# the many-strain workload (25 species x 20 strains 0.1 % apart): kernel times + search phase profile
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$ROOT/gpurun_out/strain}; mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd); shift
ARGS="--species 25 --strains 20 --genome-len 2000000 --divergence-step 0.001 --no-cpu-baseline --no-pmc --no-extra-configs --steps 3 --warmup 1"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.log
python - <<PY
import csv, glob, json
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:10]:
        print("%-60s calls %5s avg %8.3f ms total %8.2f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
try:
    d = json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
    print("value %.4g ms_per_step %.2f parity %s" % (d["value"], d["ms_per_step"], d.get("parity")))
except Exception as e:
    print("no bench line", e)
PY
