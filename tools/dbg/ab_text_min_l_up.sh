#!/bin/bash
# text_min_l RAISED above its default log4(n) + 2 (lower lost: profiles/r6zz_text_min_l.txt)
export CFR_DEBUG_ENV=1
go() { local t=$1; shift; if [ "$t" = default ]; then run "$@"; else CFR_TEXT_MIN_L=$t run "$@"; fi; }
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 2 2>/tmp/tml.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  step %.2f ms  search %.2f ms  oracle %s' % (d['value'], d['ms_per_step'], d['stage_ms']['search_ms'], d['parity'].get('equals_oracle')))" || tail -3 /tmp/tml.err; }
for t in default 18 19 default; do echo -n "cfg2 (1 Gbp)  CFR_TEXT_MIN_L=$t: "; go "$t"; done
for t in default 19 20; do echo -n "8 Gbp  CFR_TEXT_MIN_L=$t: "; go "$t" --index-gbp 8; done
for t in default 20 21; do echo -n "cfg4  CFR_TEXT_MIN_L=$t: "; go "$t" --config cfg4; done
for t in default 20; do echo -n "cfg5  CFR_TEXT_MIN_L=$t: "; go "$t" --config cfg5; done
