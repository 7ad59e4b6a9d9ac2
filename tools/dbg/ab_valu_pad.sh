#!/bin/bash
# Is k_search_chains_v2 bound by its VALU instruction stream?  The same kernel with 64 / 128 extra v_add_u32 per loop iteration
# (tools/dbg/libcfr_hip_pad{64,128}.so, -DCFR_VALU_PAD=N: +10 % / +20 % of its ~650 VALU instructions per wave-iteration).
set -e
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms'])"; }
echo -n "shipped: "; run
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so
for n in 64 128; do cp tools/dbg/libcfr_hip_pad$n.so centrifuger_amd/libcfr_hip.so; echo -n "pad $n: "; run; done
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
