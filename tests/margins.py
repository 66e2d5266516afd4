"""Observed parity margins of the GPU suite, one line per comparison: what was compared, on which
kernel flavour, at which pass, how far apart the two sides were and how far they were allowed to be.
tests/conftest.py writes the table to gpurun_out/parity_margins.txt when a `-m gpu` session ends
(copied to profiles/rNN_parity_margins.txt)."""
from __future__ import annotations

import numpy as np

ROWS: list = []


def rel(a, b, rtol: float, atol: float = 0.0) -> float:
    """The largest |a - b| in units of what np.testing.assert_allclose(a, b, rtol, atol) allows per unit of
    rtol: max |a - b| / (|b| + atol / rtol).  `<= rtol` is exactly that assertion."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b) / (np.abs(b) + (atol / rtol if rtol > 0 else 0.0) + 1e-300)))


def check(fixture: str, flavour: str, what: str, a, b, rtol: float, atol: float = 0.0, against: str = "reference"):
    """Record the observed margin, then assert it."""
    got = rel(a, b, rtol, atol)
    ROWS.append((fixture, flavour, what, against, got, rtol, atol))
    assert got <= rtol, f"{fixture} [{flavour}] {what} vs {against}: observed {got:.3e} > allowed {rtol:.3e} (atol {atol:g})"
    return got


def table() -> str:
    head = f"{'fixture':30s} {'kernel flavour':22s} {'quantity':14s} {'against':18s} {'observed':>10s} {'allowed':>10s} {'atol':>8s}"
    lines = [head, "-" * len(head)]
    for f, fl, w, ag, got, rtol, atol in ROWS:
        lines.append(f"{f:30s} {fl:22s} {w:14s} {ag:18s} {got:10.3e} {rtol:10.3e} {atol:8.1e}")
    return "\n".join(lines) + "\n"
