"""Runs LAST (file name): re-checks the allowance ledger of the session (tests/tolerances.py).

The parity bounds have two principled allowances -- the "+ 2|ref - f64|" widening next to the reference's own rounding
error and the exclusion of L1-tie pixels -- and every use of either is counted.  The per-assertion caps already fail a
test that uses too many; this test looks at the session as a whole: no comparison against a FIXTURE (data frozen from
the reference, tests/golden/) may have needed the widening for more than 4 rendering pixels or 8 gradient elements,
no record may sit above its own cap, and the at-size comparisons (which pass their own, larger, printed caps for
ties) may use the widening for at most 2e-6 of their elements.  conftest.py writes the ledger itself to
gpurun_out/tolerance_uses.txt at the end of the session."""
import pytest

import tolerances

FIXTURE_MAX_WIDENED_RENDER = 4
FIXTURE_MAX_WIDENED_GRAD = 8


def _check():
    used = list(tolerances.ALLOWANCES_USED)
    worst = {"widened": 0, "ties": 0}
    for what, kind, count, total, cap in used:
        assert count <= cap, "%s: %s used %d > cap %d" % (what, kind, count, cap)
        if kind.startswith("widened"):
            limit = max(FIXTURE_MAX_WIDENED_GRAD, int(2e-6 * total))
            assert count <= limit, "%s: widening used by %d of %d elements (limit %d)" % (what, count, total, limit)
            worst["widened"] = max(worst["widened"], count)
        else:
            worst["ties"] = max(worst["ties"], count)
    return len(used), worst


@pytest.mark.gpu
def test_allowance_ledger_of_the_gpu_session():
    n, worst = _check()
    assert n > 0, "the GPU parity tests recorded no allowance use at all: did they run before this file?"
    print("[tolerance] ledger: %d records, most widened elements in one comparison %d, most tie pixels %d" % (
        n, worst["widened"], worst["ties"]))


def test_allowance_ledger_of_the_cpu_session():
    n, worst = _check()
    print("[tolerance] ledger: %d records, most widened elements in one comparison %d" % (n, worst["widened"]))
