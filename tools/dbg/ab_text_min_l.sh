#!/bin/bash
# from how many matched characters on a small range continues on the text (text_min_l, default log4(n) + 2): 8 Gbp (K = 17 table: default 18) and 40 Gbp (K = 16: default 19)
export CFR_DEBUG_ENV=1
go() { local t=$1; shift; if [ "$t" = default ]; then run "$@"; else CFR_TEXT_MIN_L=$t run "$@"; fi; }
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 2 2>/tmp/tml.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f ms  oracle %s' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['parity'].get('equals_oracle')))" || tail -3 /tmp/tml.err; }
for t in default 17 16; do echo -n "8 Gbp  CFR_TEXT_MIN_L=$t: "; go "$t" --index-gbp 8; done
for t in default 17; do echo -n "8 Gbp long  CFR_TEXT_MIN_L=$t: "; go "$t" --index-gbp 8 --mode long; done
for t in default 18 17 16; do echo -n "cfg4  CFR_TEXT_MIN_L=$t: "; go "$t" --config cfg4; done
for t in default 17; do echo -n "cfg5  CFR_TEXT_MIN_L=$t: "; go "$t" --config cfg5; done
