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
              and getattr(getattr(c.func, "value", None), "id", None) == "os"
              and getattr(c.args[0], "id", None) == "json_fd"]
    assert len(writes) == 1


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # (main() runs only under __main__)
    return mod


def test_a_run_that_gave_up_on_a_leg_does_not_exit_zero():
    bench = _bench_module()
    assert bench.exit_status({}, None) == 0
    assert bench.exit_status({"sharded": "RuntimeError: x"}, None) != 0
    assert bench.exit_status({}, "sharded: no return within 480 s") != 0
    assert bench.exit_status({"batched": "e"}, "sharded: no return") != 0
    # the two ways of giving up can be told apart by the status alone
    assert bench.exit_status({"a": "e"}, None) != bench.exit_status({}, "a: no return")


def test_the_exit_status_is_what_the_process_ends_with():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    # no path out of the process hands the system a literal status: the watchdog's os._exit and the
    # sys.exit at the end both take what exit_status() computes from the failed / unfinished legs
    exits = [c for c in ast.walk(tree) if isinstance(c, ast.Call)
             and getattr(c.func, "attr", None) in ("_exit", "exit")
             and getattr(getattr(c.func, "value", None), "id", None) in ("os", "sys")]
    assert len(exits) == 2
    for c in exits:
        arg = c.args[0]
        ok = (isinstance(arg, ast.Call) and getattr(arg.func, "id", None) == "exit_status") or \
             (isinstance(arg, ast.Name) and arg.id == "status")
        assert ok, ast.dump(arg)
    # nothing in the line is a number typed into the source as if it had been measured
    assert "where_the_time_goes_us" not in src
