#!/bin/bash
# several builds of the library against each other on one box, alternating twice: LIBS="name=path ..." (the shipped one is "shipped").
# legs: cfg2, pairs, the 36-bit kernel on an 8 Gbp index, long reads
export CFR_DEBUG_ENV=1
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
LIBS=${LIBS:-"shipped old=tools/dbg/libcfr_hip_old.so"}
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.4g /s  search %.2f ms  post %.2f ms  total %.2f ms  step %.2f ms' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'], d['stage_ms']['total_ms'], d['ms_per_step']))"; }
legs() {
  echo -n "$1 | cfg2 se: "; run; echo -n "$1 | cfg3 pe: "; run --mode pe
  echo -n "$1 | 8 Gbp (36-bit kernel): "; run --index-gbp 8
  echo -n "$1 | long: "; run --mode long
}
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so
for rep in 1 2; do
  for L in $LIBS; do
    name=${L%%=*}; path=${L#*=}
    if [ "$name" = "shipped" ]; then cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so; else cp $path centrifuger_amd/libcfr_hip.so; fi
    legs $name
  done
done
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
