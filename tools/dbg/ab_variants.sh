#!/bin/bash
# A/B of build variants (tools/dbg/libcfr_hip_<name>.so swapped in for the shipped library): the small teams of k_tail_heavy with 128 slots /
# 96 entries (16 teams per block) on the strain workloads; the SDUST ring at 68 bytes per lane (a whole number of words) instead of 65
strain() { python bench.py --workload $1 --no-cpu-baseline --no-pmc --no-extra-configs --steps 4 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f  tail %.2f' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['stage_ms']['tail_ms']))"; }
dust() { python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 4 --warmup 2 --sdust-steps 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); w=d['with_device_sdust']; print('plain %.2f ms   with SDUST %.2f ms' % (d['ms_per_step'], w['ms_per_step']))"; }
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so
echo "== shipped"; echo -n "strains20: "; strain strains20; echo -n "strains200: "; strain strains200; echo -n "dust: "; dust
cp tools/dbg/libcfr_hip_team128.so centrifuger_amd/libcfr_hip.so
echo "== small teams with 128 slots / 96 entries"; echo -n "strains20: "; strain strains20; echo -n "strains200: "; strain strains200
cp tools/dbg/libcfr_hip_ring68.so centrifuger_amd/libcfr_hip.so
echo "== SDUST ring stride 68"; echo -n "dust: "; dust
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
