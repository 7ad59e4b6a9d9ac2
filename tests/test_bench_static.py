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
