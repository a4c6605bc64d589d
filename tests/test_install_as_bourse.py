"""bourse_amd.install_as_bourse(): code written against the reference's Python package (`import bourse`) runs unmodified
(ref src/bourse/__init__.py, src/bourse/step_sim/runner.py:103-118)."""
import importlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


@pytest.fixture()
def alias():
    import bourse_amd

    assert "bourse" not in sys.modules or getattr(sys.modules["bourse"], "__bourse_amd_alias__", False), \
        "the reference package is importable here: this test wants the alias, not the reference"
    mod = bourse_amd.install_as_bourse()
    yield mod
    bourse_amd.uninstall_bourse_alias()
    sys.modules.pop("agents_written_against_bourse", None)


def _user_module():
    sys.path.insert(0, os.path.join(HERE, "user_code"))
    try:
        return importlib.import_module("agents_written_against_bourse")
    finally:
        sys.path.pop(0)


def test_alias_exposes_the_reference_surface(alias):
    import bourse_amd
    import bourse  # noqa: F401  (the alias)
    from bourse.step_sim.agents import BaseAgent, BaseNumpyAgent, InstructionArrays, NumpyRandomAgents, RandomAgent  # noqa: F401
    from bourse.step_sim.runner import run
    import bourse.core as core
    import bourse.data_processing as dp

    assert bourse is alias and bourse.MAX_PRICE == 2**32 - 1
    assert BaseAgent is bourse_amd.step_sim.agents.BaseAgent and run is bourse_amd.step_sim.run
    assert core is bourse_amd.core and bourse.step_sim.run is run
    for name in ("StepEnv", "StepEnvNumpy", "OrderBook", "order_book_from_json"):
        assert hasattr(core, name), name
    assert bourse_amd.install_as_bourse() is alias  # idempotent
    df = dp.trades_to_dataframe([(5, True, 10, 3, 1, 0), (6, False, 12, 1, 2, 1)])
    assert list(df.columns) == ["time", "side", "price", "vol", "active_id", "passive_id"] and list(df["side"]) == ["bid", "ask"]
    od = dp.orders_to_dataframe([(True, 2, 0, 5, 0, 3, 10, 7, 0)])
    assert list(od.columns) == ["side", "status", "arr time", "end_time", "vol", "start_vol", "price", "trader_id", "order_id"]
    assert od["status"][0] == "filled" and od["side"][0] == "bid"


def test_alias_refuses_to_shadow_a_foreign_bourse():
    import types

    import bourse_amd

    bourse_amd.uninstall_bourse_alias()
    sys.modules["bourse"] = types.ModuleType("bourse")
    try:
        with pytest.raises(ImportError, match="already imported"):
            bourse_amd.install_as_bourse()
        mod = bourse_amd.install_as_bourse(force=True)
        assert sys.modules["bourse"] is mod and sys.modules["bourse.core"] is bourse_amd.core
    finally:
        bourse_amd.uninstall_bourse_alias()
        sys.modules.pop("bourse", None)


@pytest.mark.gpu
def test_user_code_written_against_bourse_runs_unmodified_on_the_gpu(alias):
    """The user module imports only `bourse`; its C1 run must reproduce the fixture generated from the REFERENCE's Python
    package (tests/golden/make_golden.py c1_random_trades: market data, trades, orders)."""
    user = _user_module()
    src = open(os.path.join(HERE, "user_code", "agents_written_against_bourse.py")).read()
    assert "import bourse_amd" not in src and "from bourse_amd" not in src  # really written against `bourse`
    env, data = user.random_trades()
    fx = np.load(os.path.join(HERE, "golden", "c1_random_trades.npz"))
    for key, v in data.items():
        assert np.array_equal(v, fx[f"md_{key}"]), key
    assert np.array_equal(np.array(env.get_trades(), dtype=np.uint64), fx["trades"])
    assert np.array_equal(np.array(env.get_orders(), dtype=np.uint64), fx["orders"])
    # a user-defined BaseAgent subclass passes the runner's interface check (runner.py:103-106)
    env2, data2, frame = user.pingers()
    assert len(data2["bid_price"]) == 12 and int(data2["bid_price"][-1]) == 44
    assert list(frame["status"]).count("active") == 2 and list(frame["status"]).count("cancelled") == 22
