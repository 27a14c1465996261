"""Parity ledger: every GPU parity test that compares an engine output with the oracle under the yardstick rule
(max|engine - fp32 oracle| <= 2 x max|bf16 oracle - fp32 oracle|) also RECORDS error / yardstick here, so that the
budget a round spends is visible round over round.  The session writes the rows to $ZE_PARITY_LEDGER (default
gpurun_out/parity_ledger.json); the builder commits the file as profiles/parity_rNN.json.  Test infrastructure only."""
import json
import os

_ROWS = []


def record(max_err, yardstick, what="", sub_margin_steps=None, bar=2.0):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    max_err, yardstick = float(max_err), float(yardstick)
    _ROWS.append(dict(test=test, what=what, max_err=round(max_err, 6), yardstick=round(yardstick, 6),
                      ratio=round(max_err / yardstick, 4) if yardstick > 0 else None, bar=bar,
                      sub_margin_steps=sub_margin_steps))


def rows():
    return list(_ROWS)


def flush():
    if not _ROWS:
        return None
    path = os.environ.get("ZE_PARITY_LEDGER") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_ledger.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    old = []
    if os.environ.get("ZE_PARITY_LEDGER_APPEND") == "1" and os.path.exists(path):
        with open(path) as f:
            old = json.load(f).get("rows", [])
    worst = max((r["ratio"] or 0.0) / r["bar"] for r in old + _ROWS)
    with open(path, "w") as f:
        json.dump(dict(rule="max|engine - fp32 oracle| <= bar x yardstick (yardstick = the oracle's own bf16-vs-fp32 error on the same inputs, "
                            "or the stated stand-in); ratio = max_err / yardstick", worst_ratio_over_bar=round(worst, 4), rows=old + _ROWS), f, indent=1)
    return path
