#!/bin/bash
# k_dust's quorum (window steps run while that many lanes of a wave are ready for one) and resident blocks, now that the screen leaves it the 31 % of reads with repeat structure
export CFR_DEBUG_ENV=1
dust() { CFR_BENCH_FULL_LINE=1 python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 4 --warmup 2 --sdust-steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); w=d['with_device_sdust']; print('plain %.2f ms   with SDUST %.2f ms (%.4g reads/s)  pre-step %.2f' % (d['ms_per_step'], w['ms_per_step'], w['value'], w['ms_per_step']-d['ms_per_step']))"; }
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so
echo -n "shipped (quorum 32): "; dust
for v in q8 q16 q48; do cp tools/dbg/libcfr_hip_$v.so centrifuger_amd/libcfr_hip.so; echo -n "$v: "; dust; done
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
echo -n "shipped again: "; dust
for b in 4 5 7 8; do echo -n "shipped, blocks per CU $b: "; CFR_DUST_BLOCKS=$b dust; done
