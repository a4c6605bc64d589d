"""Run the reference's OWN Python test-suite against the CPU oracle (build container only).

Test infrastructure.  This pins the oracle: the reference's pure-Python half
(``/root/reference/src/bourse``: runner, agents, data_processing) and its tests
(``/root/reference/tests``) are imported from where they lie, with the PyO3
extension ``bourse.core`` (Rust, unbuildable here) supplied by ``oracle/pyoracle.py``.
Nothing is copied; ``/root/reference`` does not exist on the GPU box, so this
script is never part of ``pytest tests/`` -- its result is recorded in DESIGN.md and
re-checked by ``tests/test_oracle_conformance.py`` only when the reference is present.

Usage:  python oracle/run_reference_pytests.py [extra pytest args]
"""
import sys

sys.dont_write_bytecode = True  # never write into /root/reference
import importlib.abc
import importlib.machinery
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("BOURSE_REFERENCE", "/root/reference")


def install_core_shim():
    """Make ``import bourse.core`` resolve to the oracle-backed classes."""
    sys.path.insert(0, HERE)
    import pyoracle

    core = types.ModuleType("bourse.core")
    core.StepEnv = pyoracle.StepEnv
    core.StepEnvNumpy = pyoracle.StepEnvNumpy
    core.OrderBook = pyoracle.OrderBook
    core.order_book_from_json = pyoracle.order_book_from_json

    class _Loader(importlib.abc.Loader):
        def create_module(self, spec):
            return core

        def exec_module(self, module):
            pass

    class _Finder(importlib.abc.MetaPathFinder):
        def find_spec(self, name, path, target=None):
            if name == "bourse.core":
                return importlib.machinery.ModuleSpec(name, _Loader())
            return None

    sys.meta_path.insert(0, _Finder())
    sys.path.insert(0, os.path.join(REF, "src"))
    return core


def main(argv):
    if not os.path.isdir(REF):
        print(f"reference not present at {REF}; nothing to run")
        return 0
    install_core_shim()
    import pytest

    args = [
        os.path.join(REF, "tests", "test_order_book.py"),
        os.path.join(REF, "tests", "test_step_sim", "test_env.py"),
        os.path.join(REF, "tests", "test_step_sim", "test_numpy_api.py"),
        os.path.join(REF, "tests", "test_step_sim", "test_agents.py"),
        # test_benchmarks.py needs the pytest-benchmark plugin (absent) and asserts nothing.
        "-p", "no:cacheprovider", "-q", "--rootdir", "/tmp", "-c", "/dev/null",
    ] + list(argv)
    return pytest.main(args)


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
