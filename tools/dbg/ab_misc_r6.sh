export CFR_DEBUG_ENV=1
run() { CFR_BENCH_FULL_LINE=1 python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.4g /s  step %.2f ms  search %.2f  post %.2f  device total %.2f' % (d['value'], d['ms_per_step'], s['search_ms'], s['tail_ms'], s['total_ms']))"; }
echo -n "cfg2 default: "; run
for b in 3 5; do echo -n "cfg2 CFR_BLOCKS_PER_CU=$b: "; CFR_BLOCKS_PER_CU=$b run; done
for sb in 833334 1000000 1666667 2500000; do echo -n "cfg2 CFR_SUBBATCH=$sb: "; CFR_SUBBATCH=$sb run; done
echo -n "cfg2 CFR_TAPER_FLOOR=0: "; CFR_TAPER_FLOOR=0 run
echo -n "cfg2 CFR_SEARCH_DYN=2 (wave tiles): "; CFR_SEARCH_DYN=2 run
echo -n "cfg2 default: "; run
