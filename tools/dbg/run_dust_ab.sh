#!/bin/bash
VARIANTS="ringless oldring" BLOCKS="0" bash tools/dbg/ab_dust_r5.sh 2>&1 | tee gpurun_out/ab_dust_r5_2.txt
BLOCKS=8 VARIANTS="ringless" bash tools/dbg/ab_dust_r5.sh 2>&1 | grep -A2 ringless | tee -a gpurun_out/ab_dust_r5_2.txt
python bench.py --no-40gbp --no-cpu-baseline --no-extra-configs --steps 5 > gpurun_out/bench_r5b.json 2> gpurun_out/bench_r5b.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_r5b.json"))
r=d["with_device_sdust"].get("roofline") or {}
print("dust", d["with_device_sdust"]["ms_per_step"]-d["ms_per_step"], {k:r.get(k) for k in ("frac","waves_per_simd","wait_fraction_of_wave_cycles","lds_bank_conflict_over_lds_active","kernel_ms_per_step_scaled")})
print("post", json.dumps(d.get("post_stage"))[:1800])
PY
