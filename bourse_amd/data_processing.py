"""Output helpers with the reference's names (ref src/bourse/data_processing.py:10-105): the tuples ``get_trades()`` /
``get_orders()`` return, as pandas frames.  Not on the step path - here so that code written against
``bourse.data_processing`` keeps working after ``bourse_amd.install_as_bourse()``."""
import typing

_SIDE = {True: "bid", False: "ask"}
_STATUS = {0: "new", 1: "active", 2: "filled", 3: "cancelled", 4: "rejected"}  # types.rs:51-63
# column names exactly as the reference's frames carry them ("arr time" with a blank is the reference's, :86-96)
TRADE_COLUMNS = ("time", "side", "price", "vol", "active_id", "passive_id")
ORDER_COLUMNS = ("side", "status", "arr time", "end_time", "vol", "start_vol", "price", "trader_id", "order_id")


def _frame(records, columns, maps):
    import pandas as pd

    df = pd.DataFrame.from_records(list(records), columns=list(columns))
    for col, table in maps.items():
        df[col] = df[col].map(table)
    return df


def trades_to_dataframe(trades: typing.List[typing.Tuple]):
    """``(time, side, price, vol, active_id, passive_id)`` tuples -> DataFrame, side as "bid" / "ask"."""
    return _frame(trades, TRADE_COLUMNS, {"side": _SIDE})


def orders_to_dataframe(order_history: typing.List[typing.Tuple]):
    """``(side, status, arr_time, end_time, vol, start_vol, price, trader_id, order_id)`` tuples -> DataFrame, side as
    "bid" / "ask", status as "new" / "active" / "filled" / "cancelled" / "rejected"."""
    return _frame(order_history, ORDER_COLUMNS, {"side": _SIDE, "status": _STATUS})
