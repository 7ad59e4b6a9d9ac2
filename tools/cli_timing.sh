#!/bin/bash
export CFR_DEBUG_ENV=1   # the gate behind which the library reads its CFR_* A/B switches
# Stage timing of the drop-in command line on the bench sample (run on the GPU box after bench.py has filled its cache).
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 > gpurun_out/cli_bench.json 2> gpurun_out/cli_bench.err
idx=$(ls /tmp/cfr_bench/*.1.cfr /tmp/cfr_bench/*/*.1.cfr 2>/dev/null | head -1); idx=${idx%.1.cfr}
fa=$(ls /tmp/cfr_bench/sample_0.fa /tmp/cfr_bench/*/sample_0.fa 2>/dev/null | head -1)
echo "index $idx reads $fa" > gpurun_out/cli_timing.txt
for t in 16 64; do
  echo "== -t $t" >> gpurun_out/cli_timing.txt
  CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -u $fa -t $t 2>> gpurun_out/cli_timing.txt > /tmp/cli.tsv
  md5sum /tmp/cli.tsv >> gpurun_out/cli_timing.txt
  echo "== -t $t throughput profile" >> gpurun_out/cli_timing.txt
  CFR_PROFILE=throughput CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -u $fa -t $t 2>> gpurun_out/cli_timing.txt > /tmp/cli.tsv
  md5sum /tmp/cli.tsv >> gpurun_out/cli_timing.txt
done
cat gpurun_out/cli_timing.txt
python -c "import json;d=json.loads(open('gpurun_out/cli_bench.json').read().strip().splitlines()[-1]);print(d['value'],d.get('parity'),d.get('e2e_cli'))"
# 10 M reads (the sample five times): the stages overlap, so the wall clock approaches the slowest stage
big=/tmp/big10m.fa
for i in 1 2 3 4 5; do cat $fa; done > $big
echo "== 10 M reads, -t 64" | tee -a gpurun_out/cli_timing.txt
( time CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -u $big -t 64 > /tmp/cli_big.tsv ) 2>&1 | grep -E "timing|real" | tee -a gpurun_out/cli_timing.txt
md5sum /tmp/cli_big.tsv | tee -a gpurun_out/cli_timing.txt
if [ -x oracle/_ref/centrifuger ]; then
  echo "== 10 M reads, reference -t $(nproc)" | tee -a gpurun_out/cli_timing.txt
  ( time oracle/_ref/centrifuger -x $idx -u $big -t $(nproc) > /tmp/ref_big.tsv 2>/dev/null ) 2>&1 | grep real | tee -a gpurun_out/cli_timing.txt
  md5sum /tmp/ref_big.tsv | tee -a gpurun_out/cli_timing.txt
fi
