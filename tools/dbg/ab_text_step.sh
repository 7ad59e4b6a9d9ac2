#!/bin/bash
# sensitivity of the search to the number of text steps: kTextStep = 24 (tools/dbg/libcfr_hip_ts24.so) against the shipped 48
set -e
run() { CFR_DEBUG_ENV=1 CFR_SEARCH_PROF=$1 python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/tmp/ts.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms']))"; grep "search prof" /tmp/ts.err | tail -1 | cut -c1-400; }
echo "== kTextStep 48 (shipped)"; run 0; run 1
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so; cp tools/dbg/libcfr_hip_ts24.so centrifuger_amd/libcfr_hip.so
echo "== kTextStep 24"; run 0; run 1
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
