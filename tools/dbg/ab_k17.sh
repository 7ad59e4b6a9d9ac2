#!/bin/bash
# cfg2 / pairs with the derived K-mer table at K = 16 (16-byte entries with text positions: CFR_K17=0, the default before round 6), K = 17 with 8-byte entries
# (the default where it fits) without / with the text positions in place of single rows (CFR_FTABX_TEXTPOS=0 / default); same library, same box, alternating
export CFR_DEBUG_ENV=1
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 6 --warmup 2 2>/tmp/k17.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4g reads/s  search %.2f ms  total %.2f ms  oracle %s' % (d['value'], d['stage_ms']['search_ms'], d['stage_ms']['total_ms'], d['parity'].get('equals_oracle')))" || tail -3 /tmp/k17.err; grep -E "device image" /tmp/k17.err | cut -c1-120; }
for rep in 1 2; do
  echo "== K = 16, 16-byte entries"; CFR_K17=0 run
  echo "== K = 17, 8-byte entries, rows only"; CFR_FTABX_TEXTPOS=0 run
  echo "== K = 17, 8-byte entries, text positions in place of single rows (default)"; run
done
echo "== pairs: K = 16 / K = 17 rows only / K = 17 default"; CFR_K17=0 run --mode pe; CFR_FTABX_TEXTPOS=0 run --mode pe; run --mode pe
echo "== long: K = 16 / K = 17 rows only / K = 17 default"; CFR_K17=0 run --mode long; CFR_FTABX_TEXTPOS=0 run --mode long; run --mode long
