#!/bin/bash
# why is the command line slow on 10 M pairs?  (5 x the 2 M-pair sample) -k 5 under a few library switches
export CFR_DEBUG_ENV=1
python bench.py --mode pe --steps 1 --warmup 1 --no-pmc --no-extra-configs > /dev/null 2> /dev/null
f1=$(ls /tmp/cfr_bench/*/sample_0.fa 2>/dev/null | head -1); f2=${f1%.fa}_2.fa; idx=$(dirname $f1)/idx
for i in 1 2 3 4 5; do cat $f1; done > /tmp/big10m_1.fa
for i in 1 2 3 4 5; do cat $f2; done > /tmp/big10m_2.fa
t() { echo -n "$1: "; shift; ( time env "$@" CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -1 /tmp/big10m_1.fa -2 /tmp/big10m_2.fa -k 5 -t 64 > /tmp/o.tsv ) 2>&1 | grep -E "parse|classify|format|wall|real" | tr '\n' ' '; md5sum /tmp/o.tsv | cut -c1-8; }
t default X=1
t default_again X=1
t tail_stream_0 CFR_TAIL_STREAM=0
t direct_rows_0 CFR_HEAVY_DIRECT_ROWS=0
t team_tail_0 CFR_TEAM_TAIL=0
echo -n "gpu-batch 1M: "; ( time CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -1 /tmp/big10m_1.fa -2 /tmp/big10m_2.fa -k 5 -t 64 --gpu-batch 1000000 > /tmp/o.tsv ) 2>&1 | grep -E "parse|classify|wall|real" | tr '\n' ' '; echo
echo -n "pairs -k 1: "; ( time CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -1 /tmp/big10m_1.fa -2 /tmp/big10m_2.fa -k 1 -t 64 > /tmp/o.tsv ) 2>&1 | grep -E "parse|classify|wall|real" | tr '\n' ' '; echo
