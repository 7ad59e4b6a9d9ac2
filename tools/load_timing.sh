#!/bin/bash
export CFR_DEBUG_ENV=1   # the gate behind which the library reads its CFR_* A/B switches
# Load-stage timing of the device index on the bench index (GPU box).
mkdir -p gpurun_out
python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
idx=$(ls /tmp/cfr_bench/*.1.cfr /tmp/cfr_bench/*/*.1.cfr 2>/dev/null | head -1); idx=${idx%.1.cfr}
printf ">r0\nACGTACGTAGCTAGCTAGCTAGCATCGATCGATCGATCAGCTAGCTAGCTAGCTAGC\n" > /tmp/one.fa
for i in 1 2; do CFR_LOAD_TIMING=1 CFR_CLI_TIMING=1 centrifuger_amd/bin/centrifuger -x $idx -u /tmp/one.fa 2>&1 >/dev/null | grep -E "load|timing"; done | tee gpurun_out/load_timing.txt
