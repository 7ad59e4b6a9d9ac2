export CFR_DEBUG_ENV=1
run() { CFR_BENCH_FULL_LINE=1 python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('%.4g /s  step %.2f ms  search %.2f  post %.2f  device total %.2f  oracle %s' % (d['value'], d['ms_per_step'], s['search_ms'], s['tail_ms'], s['total_ms'], d['parity_oracle']['equals_oracle']))"; }
for r in 1 2; do
echo -n "cfg2 default: "; run
echo -n "cfg2 CFR_POST_FAST=1: "; CFR_POST_FAST=1 run
echo -n "cfg2 CFR_POST_FAST=1 CFR_BLOCKS_PER_CU=5: "; CFR_POST_FAST=1 CFR_BLOCKS_PER_CU=5 run
done
echo -n "pe default: "; run --mode pe
echo -n "pe CFR_POST_FAST=1: "; CFR_POST_FAST=1 run --mode pe
