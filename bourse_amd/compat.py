"""``install_as_bourse()``: make ``import bourse`` resolve to this package, so that agent code written against the
reference's Python package (``import bourse``, ``from bourse.step_sim.agents import BaseAgent``, ``bourse.core.StepEnv``,
``bourse.step_sim.run`` - ref src/bourse/__init__.py, src/bourse/step_sim/runner.py:103-118) runs UNMODIFIED on the GPU.

The alias is a module object of its own exposing exactly the reference's public surface (``core``, ``step_sim``,
``data_processing``, ``MAX_PRICE``); its submodules ARE this package's modules, so ``bourse.step_sim.agents.BaseAgent`` is
``bourse_amd.step_sim.agents.BaseAgent`` and the runner's ``isinstance`` checks hold for user agents."""
import sys
import types

_SUBMODULES = ("core", "data_processing", "step_sim", "step_sim.agents", "step_sim.agents.base_agent",
               "step_sim.agents.random_agent", "step_sim.runner")


def install_as_bourse(force: bool = False) -> types.ModuleType:
    """Register ``bourse`` (and ``bourse.core``, ``bourse.step_sim``, ``bourse.step_sim.agents``, ...) in
    ``sys.modules``.  Refuses to shadow a real ``bourse`` that is already imported unless ``force``; idempotent."""
    import importlib

    import bourse_amd

    cur = sys.modules.get("bourse")
    if cur is not None and getattr(cur, "__bourse_amd_alias__", False):
        return cur
    if cur is not None and not force:
        raise ImportError("a module named 'bourse' is already imported (the reference package?): "
                          "install_as_bourse(force=True) replaces it for imports made from now on")
    alias = types.ModuleType("bourse", "bourse_amd installed under the reference's package name (bourse_amd.install_as_bourse)")
    alias.__bourse_amd_alias__ = True
    alias.__path__ = []  # a package: `import bourse.core` consults sys.modules first and finds the entries below
    alias.MAX_PRICE = bourse_amd.MAX_PRICE
    if force:
        for name in [n for n in sys.modules if n == "bourse" or n.startswith("bourse.")]:
            del sys.modules[name]
    sys.modules["bourse"] = alias
    for sub in _SUBMODULES:
        mod = importlib.import_module("bourse_amd." + sub)
        sys.modules["bourse." + sub] = mod
        if "." not in sub:
            setattr(alias, sub, mod)
    return alias


def uninstall_bourse_alias() -> None:
    """Remove the alias again (tests)."""
    cur = sys.modules.get("bourse")
    if cur is not None and getattr(cur, "__bourse_amd_alias__", False):
        for name in [n for n in sys.modules if n == "bourse" or n.startswith("bourse.")]:
            del sys.modules[name]
