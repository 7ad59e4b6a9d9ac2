export CFR_DEBUG_ENV=1
run() { CFR_BENCH_FULL_LINE=1 python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); w=d['with_device_sdust']; print('step %.2f ms  with SDUST on the device %.2f ms = %.4g reads/s  (pre-step %.2f ms)' % (d['ms_per_step'], w['ms_per_step'], w['value'], w['ms_per_step']-d['ms_per_step']))"; }
for m in 0 1 0 1; do echo -n "CFR_DUST_PIECES=$m: "; CFR_DUST_PIECES=$m run; done
