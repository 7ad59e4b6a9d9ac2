#!/bin/bash
# SDUST of resident reads: the whole batch masked up front (default) against sub-batch k + 1 masked on its own stream beside the search of
# sub-batch k (CFR_DUST_PIECES=1; measured slower in round 2, before the search was held to 4 blocks per CU beside the post stage)
run() { python bench.py "$@" --no-cpu-baseline --no-pmc --no-extra-configs --steps 5 --warmup 2 --sdust-steps 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); w=d['with_device_sdust']; print('plain %.4g reads/s (%.2f ms)   with SDUST %.4g reads/s (%.2f ms)' % (d['value'], d['ms_per_step'], w['value'], w['ms_per_step']))"; }
export CFR_DEBUG_ENV=1
echo -n "up front:              "; run
echo -n "by sub-batch (pieces): "; CFR_DUST_PIECES=1 run
echo -n "pieces, 3 blocks/CU:   "; CFR_DUST_PIECES=1 CFR_BLOCKS_PER_CU=3 run
echo -n "pairs up front:        "; run --mode pe
echo -n "pairs pieces:          "; CFR_DUST_PIECES=1 run --mode pe
