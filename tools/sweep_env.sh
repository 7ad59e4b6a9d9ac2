#!/bin/bash
# A/B sweep of the library's tuning switches on the bench workload (one gpurun call; results under gpurun_out/sweep.txt).
# usage: tools/sweep_env.sh "VAR=val VAR2=val" "VAR=val" ...   (each argument = one configuration; "" = defaults)
mkdir -p gpurun_out
out=gpurun_out/sweep.txt
[ -n "${APPEND:-}" ] || : > $out
python bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > /dev/null 2>&1   # builds + caches the index
for cfg in "$@"; do
  [ -n "${SLEEP:-}" ] && sleep $SLEEP
  line=$(env $cfg python bench.py --steps 5 --warmup 2 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | tail -1)
  echo "$cfg => $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f M/s  ms/step %.2f  stages %s parity=%s' % (d['value']/1e6, d['ms_per_step'], {k: round(v,2) for k,v in d.get('stage_ms',{}).items()}, d.get('parity')))" 2>&1 | tail -1)" >> $out
done
cat $out
