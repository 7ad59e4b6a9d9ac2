#!/bin/bash
# k_dust<true> without its ring: resident blocks per CU (9 fit), against the step without SDUST
dust() { python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 4 --warmup 2 --sdust-steps 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); w=d['with_device_sdust']; print('plain %.2f ms   with SDUST %.2f ms (%.4g reads/s)' % (d['ms_per_step'], w['ms_per_step'], w['value']))"; }
export CFR_DEBUG_ENV=1
for b in 9 8 7 6 5; do echo -n "blocks per CU $b: "; CFR_DUST_BLOCKS=$b dust; done
