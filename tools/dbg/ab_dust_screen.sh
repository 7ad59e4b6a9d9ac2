#!/bin/bash
# A/B of the SDUST screen (k_dust_screen in front of k_dust<true>): CFR_DUST_SCREEN=0 / 1, same library, same box; the figure is bench.py's with_device_sdust leg
export CFR_DEBUG_ENV=1
run() { CFR_BENCH_FULL_LINE=1 python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); w=d['with_device_sdust']; print('step %.2f ms  with SDUST on the device %.2f ms = %.4g reads/s  (pre-step %.2f ms)' % (d['ms_per_step'], w['ms_per_step'], w['value'], w['ms_per_step']-d['ms_per_step']))"; }
for rep in 1 2; do
  for m in 0 1; do echo -n "CFR_DUST_SCREEN=$m cfg2: "; CFR_DUST_SCREEN=$m run; done
done
for m in 0 1; do echo -n "CFR_DUST_SCREEN=$m pairs: "; CFR_DUST_SCREEN=$m run --mode pe; done
