#!/bin/bash
# where k_dust spends its clocks (tools/dbg/libcfr_hip_dustprof.so, -DCFR_DUST_PROF=1): rounds, lanes per round and clocks per wave of the five phases
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so; cp tools/dbg/libcfr_hip_dustprof.so centrifuger_amd/libcfr_hip.so
python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 1 --warmup 0 --sdust-steps 1 2>&1 | grep "^\[dust\]" | tail -12
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
