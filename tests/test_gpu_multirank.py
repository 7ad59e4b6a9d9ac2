"""The 8-rank launch of bench.py on ONE GPU (CFR_BENCH_SHARE_GPU=1: every rank on device 0, gloo instead of RCCL): what the driver's
`torch.distributed.run --nproc-per-node 8 bench.py --gpus 8` exercises apart from the devices themselves - rank 0 writes the index
while the others wait at the barrier, every rank loads its own replica and classifies its own reads, the max-over-ranks clock, the
ranks leaving while rank 0 goes on alone, one JSON line.  BASELINE configs[3]'s preset (--config cfg4) on a 1 Gbp index.  -m gpu."""
import json
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_eight_ranks_share_one_gpu(tmp_path):
    env = dict(os.environ, CFR_BENCH_SHARE_GPU="1")
    env.pop("CFR_BENCH_FULL_LINE", None)
    t0 = time.time()
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "cfg4", "--index-gbp", "1", "--reads", "1000000",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pmc", "--count-sample", "20000", "--cache", str(tmp_path / "cache")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, "\n".join(ln for ln in r.stderr.decode().splitlines() if "rror" in ln or "Traceback" in ln or "bench" in ln)[-4000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout.decode()[-2000:]            # rank 0 alone prints
    assert len(lines[0]) < 4096                                  # the compact line, the same for any number of ranks
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["config"]["reads_per_step_per_gpu"] == 1000000 and abs(d["config"]["index_bp"] - 1e9) < 2e7
    assert d["value"] > 0 and abs(d["value"] - 8 * 1000000 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]      # whole-job aggregate over the 8 ranks
    assert d["parity"]["equals_oracle"] is True and d["parity"]["reads_vs_oracle"] == 20000
    assert d["classified_fraction"] > 0.99
    # every rank reads the index's bit strings from the one file mapping (cfr_index_mapped_bytes): nothing copied per rank
    assert d["index_on_ranks"] == {"ranks": 8, "every_rank_maps_the_file": True, "copied_bytes_max": 0}
    assert time.time() - t0 < 300                                # (the whole 8-rank run of this preset)
