export CFR_DEBUG_ENV=1
run() { CFR_BENCH_FULL_LINE=1 python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.4g /s  step %.2f ms  search %.2f  post %.2f  device total %.2f' % (d['value'], d['ms_per_step'], s['search_ms'], s['tail_ms'], s['total_ms']))"; }
for r in 1 2; do
echo -n "cfg2 default (post stage of sub-batch k beside the search of k+1): "; run
echo -n "cfg2 CFR_TAIL_STREAM=0 (post stage behind its search, search at 5 blocks per CU): "; CFR_TAIL_STREAM=0 run
done
echo -n "strains20 default: "; run --workload strains20
echo -n "strains20 CFR_TAIL_STREAM=0: "; CFR_TAIL_STREAM=0 run --workload strains20
