#!/bin/bash
export CFR_DEBUG_ENV=1
python bench.py --steps 2 --warmup 1 --no-pmc --no-extra-configs --cpu-sample 2000000 > gpurun_out/cli_bench.json 2> gpurun_out/cli_bench.err
idx=$(ls /tmp/cfr_bench/*/idx.1.cfr 2>/dev/null | head -1); idx=${idx%.1.cfr}
fa=$(ls /tmp/cfr_bench/*/sample_0.fa 2>/dev/null | head -1)
big=/tmp/big10m.fa
for i in 1 2 3 4 5; do cat $fa; done > $big
for gb in 262144 1048576 2500000; do for pt in 4 8; do for prof in "" "--gpu-fast-load"; do
  echo "== gpu-batch $gb parse-threads $pt $prof"
  ( time CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -u $big -t 64 --parse-threads $pt --gpu-batch $gb $prof > /tmp/cli_big.tsv ) 2>&1 | grep -E "timing|real" | tr '\n' ' '; echo
  md5sum /tmp/cli_big.tsv
done; done; done
