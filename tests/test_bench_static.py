"""bench.py is the driver's contract and only runs end to end on the GPU box, so what can be checked without one is checked here: it
parses, its option parser builds, and no function re-imports a name the module imports (a function-local `import hashlib` made every
earlier use of `hashlib` in main() an UnboundLocalError and the default run printed no line for two commits of round 5)."""
import ast
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def test_no_function_shadows_a_module_level_import():
    tree = ast.parse(open(BENCH).read())
    top = {a.asname or a.name.split(".")[0] for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) for a in n.names}
    bad = []
    for f in ast.walk(tree):
        if isinstance(f, (ast.FunctionDef, ast.AsyncFunctionDef)):
            for n in ast.walk(f):
                if isinstance(n, (ast.Import, ast.ImportFrom)):
                    bad += [(f.name, n.lineno, a.name) for a in n.names if (a.asname or a.name.split(".")[0]) in top]
    assert not bad, bad


def test_names_assigned_in_main_are_not_used_as_module_globals_before():
    """the same failure through an assignment: a module-level name that main() also assigns is local to ALL of main()"""
    tree = ast.parse(open(BENCH).read())
    top = {a.asname or a.name.split(".")[0] for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) for a in n.names}
    top |= {t.id for n in tree.body if isinstance(n, ast.Assign) for t in n.targets if isinstance(t, ast.Name)}
    top |= {n.name for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
    bad = []
    for f in tree.body:
        if isinstance(f, ast.FunctionDef):
            declared = {g for n in ast.walk(f) if isinstance(n, ast.Global) for g in n.names}
            stores = {n.id for n in ast.walk(f) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store)}
            bad += [(f.name, name) for name in sorted((stores & top) - declared)]
    assert not bad, bad


def test_option_parser_builds_and_names_the_contract_flags():
    r = subprocess.run([sys.executable, BENCH, "--help"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out


# ---- the ONE line the driver parses (round 5: a 29 KB line was not parsed and the round went unmeasured) ----
import copy
import glob
import importlib.util
import json
import math

import pytest


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _stored_details():
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench*.json"))):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        if isinstance(d, dict) and "metric" in d:
            out.append((os.path.basename(f), d))
    return out


def _check_line(s, want_cpu_baseline=True):
    assert "\n" not in s
    assert len(s.encode()) < 4096, len(s)
    d = json.loads(s, parse_constant=lambda c: pytest.fail(f"non-finite constant {c} in the line"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    if want_cpu_baseline:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in d["cpu_baseline"], k

    def walk(o):
        if isinstance(o, float):
            assert math.isfinite(o)
        elif isinstance(o, dict):
            for v in o.values():
                walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)
    walk(d)
    return d


def test_compact_line_of_every_stored_detail_object_is_small_and_complete():
    b = _bench_module()
    stored = _stored_details()
    assert any(len(json.dumps(d)) > 20000 for _, d in stored)      # the 29 KB object of round 5 is among them
    for name, d in stored:
        line = _check_line(b.compact_line(d), want_cpu_baseline="cpu_baseline" in d)
        assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"], name


def test_compact_line_survives_hostile_detail():
    """NaN / inf values, kilobyte-long notes and error texts, every sub-result present: still one parseable line under the limit"""
    b = _bench_module()
    d = copy.deepcopy(max((d for _, d in _stored_details()), key=lambda x: len(json.dumps(x))))
    d["roofline"]["frac"] = float("nan")
    d["roofline"]["l2_hit"] = float("inf")
    d["cpu_baseline"]["sample"] = "x" * 5000
    d["config"]["workload"] = "w" * 5000
    oc = d.setdefault("other_configs", {})
    for i in range(8):
        oc[f"extra{i}"] = {"error": "e" * 3000}
    oc["skippy"] = {"skipped": "s" * 3000}
    oc["nanny"] = {"value": float("nan"), "ms_per_step": float("inf"), "equals_oracle": False}
    d["e2e_cli_100m_fastq"] = {"value": 3.3e7, "md5_equals_reference_rows": True}
    line = _check_line(b.compact_line(d))
    assert line["roofline"]["frac"] is None and line["sub_results"]["nanny"]["value"] is None
    assert line["sub_results"]["e2e_cli_100m_fastq"] == 3.3e7 and line["parity"]["cli_100m_fastq_md5_equals_reference_rows"] is True


def test_emit_prints_the_compact_line_last_and_the_detail_elsewhere(tmp_path):
    name, d = max(_stored_details(), key=lambda x: len(json.dumps(x[1])))
    src = tmp_path / "detail_in.json"
    src.write_text(json.dumps(d))
    code = ("import importlib.util, json, sys; spec = importlib.util.spec_from_file_location('b', %r); b = importlib.util.module_from_spec(spec); "
            "spec.loader.exec_module(b); print('[noise] something a library printed'); b.emit_line(json.load(open(%r)), None)" % (BENCH, str(src)))
    env = dict(os.environ, CFR_BENCH_DETAIL=str(tmp_path / "detail_out.json"))
    env.pop("CFR_BENCH_FULL_LINE", None)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = r.stdout.decode().splitlines()
    line = _check_line(lines[-1])
    assert line["detail"] == str(tmp_path / "detail_out.json")
    assert len(r.stdout) < 8192                                              # nothing but the line (and the noise above) on stdout
    full = json.loads((tmp_path / "detail_out.json").read_text())
    assert full["other_configs"].keys() == d["other_configs"].keys()          # the detail is whole
    assert b"[bench] detail {" in r.stderr
    # children keep the full object on stdout (their parent parses it)
    r2 = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, CFR_BENCH_FULL_LINE="1"), timeout=300)
    assert json.loads(r2.stdout.decode().splitlines()[-1]).keys() == d.keys()
