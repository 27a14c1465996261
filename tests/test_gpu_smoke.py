"""The driver's smoke() as a test, so that its error / yardstick ratio lands in the parity ledger."""
import pytest

import parity_ledger

pytestmark = pytest.mark.gpu


def test_smoke_entry_point_and_its_ratio():
    import __graft_entry__ as g
    err, yard = g.smoke()
    parity_ledger.record(err, yard, "__graft_entry__.smoke(): tiny model, view + prefill logits vs the fp32 oracle")
    assert err <= 2.0 * yard
