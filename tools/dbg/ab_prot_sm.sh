#!/bin/bash
# protein leg: lanes as state machines (k_search_prot_sm) against one chain per lane (k_search_prot), register budgets of both
export CFR_DEBUG_ENV=1
line() { python bench.py --mode protein --no-pmc "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  stages %s  parity %s' % (d['value'], d['ms_per_step'], {k: round(v, 2) for k, v in d['stage_ms'].items()}, d.get('parity')))"; }
echo -n "default, TSV against the reference binary:        "; line --steps 3 --warmup 1
echo -n "state machines (CFR_PROT_SM=1):                    "; CFR_PROT_SM=1 line --steps 3 --warmup 1 --no-cpu-baseline
echo -n "state machines, 80 registers (6 blocks per CU):    "; CFR_PROT_SM=1 CFR_PROT_SM_MINB=6 line --steps 3 --warmup 1 --no-cpu-baseline
echo -n "one chain per lane (CFR_PROT_SM=0):                "; CFR_PROT_SM=0 line --steps 3 --warmup 1 --no-cpu-baseline
echo -n "one chain per lane, 80 registers (6 blocks per CU):"; CFR_PROT_SM=0 CFR_PROT_MINB=6 line --steps 3 --warmup 1 --no-cpu-baseline
echo -n "one chain per lane, 64 registers (8 blocks per CU):"; CFR_PROT_SM=0 CFR_PROT_MINB=8 line --steps 3 --warmup 1 --no-cpu-baseline
