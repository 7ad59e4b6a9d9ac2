#!/bin/bash
# round 5: k_dust<true> with its window in registers (-DCFR_DUST_RINGLESS=1: 128 bytes of LDS per lane, 9 blocks per CU) against the ring in LDS
export CFR_DEBUG_ENV=1
dust() { python bench.py --no-cpu-baseline --no-pmc --no-extra-configs --steps 4 --warmup 2 --sdust-steps 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); w=d['with_device_sdust']; print('plain %.2f ms   with SDUST %.2f ms (%.4g reads/s)  pre-step %.2f ms' % (d['ms_per_step'], w['ms_per_step'], w['value'], w['ms_per_step'] - d['ms_per_step']))"; }
cp centrifuger_amd/libcfr_hip.so /tmp/shipped.so
for v in shipped ${VARIANTS:-ringless}; do
  if [ $v = shipped ]; then cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so; else cp tools/dbg/libcfr_hip_$v.so centrifuger_amd/libcfr_hip.so; fi
  echo "== $v"; python -m pytest tests/test_gpu_dust.py -m gpu -x -q 2>&1 | tail -1
  for b in ${BLOCKS:-"0"}; do echo -n "   blocks per CU $b: "; if [ $b = 0 ]; then dust; else CFR_DUST_BLOCKS=$b dust; fi; done
done
cp /tmp/shipped.so centrifuger_amd/libcfr_hip.so
