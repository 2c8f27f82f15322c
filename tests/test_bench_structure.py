"""bench.py runs its secondary legs as functions under a guard; a leg that is defined but never
called, or that forgets to return its figures, leaves a null in the line without any error.
Checked on the source, without running the benchmark."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_leg_returns_its_figures_and_is_called_under_the_guard():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    legs = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name.startswith("leg_")]
    assert {n.name for n in legs} == {"leg_sharded", "leg_batched", "leg_calibration", "leg_extras"}
    guarded = {c.args[1].id for c in ast.walk(tree)
               if isinstance(c, ast.Call) and getattr(c.func, "id", None) == "guard" and len(c.args) == 2
               and isinstance(c.args[1], ast.Name)}
    for leg in legs:
        assert any(isinstance(x, ast.Return) and x.value is not None for x in ast.walk(leg)), leg.name
        assert leg.name in guarded, leg.name
    # the line is written exactly once, through write_line
    writes = [c for c in ast.walk(tree) if isinstance(c, ast.Call) and getattr(c.func, "attr", None) == "write"
              and getattr(getattr(c.func, "value", None), "id", None) == "os"]
    assert len(writes) == 1
