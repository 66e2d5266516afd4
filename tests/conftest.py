import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    o = oracle.Oracle()
    o.set_threads(1)
    return o


@pytest.fixture(scope="session")
def lib():
    from bammmotif2_amd import abi, build
    build.build_library()
    return abi.load()


@pytest.fixture(scope="session")
def gpu_ctx(lib):
    import bammmotif2_amd as bm
    ctx = bm.Context(0)          # raises loudly without a gfx950 device: no fallback
    # set-sized scratch (dense r, lists, logs) travels from handle to handle through the context: every block is
    # filled with 0xFF (NaNs / huge indices) when it is handed out, so that a kernel reading what it did not write shows
    ctx.set_tuning(scratch_poison=1)
    yield ctx
    ctx.close()


def pytest_sessionfinish(session, exitstatus):
    """The observed parity margins of a GPU session (tests/margins.py) as a table under gpurun_out/."""
    from tests import margins
    if not margins.ROWS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_margins.txt"), "w") as fh:
        fh.write("max |a - b| / (|b| + atol / rtol) per comparison of the -m gpu suite: `observed <= allowed` is the assertion\n")
        fh.write(margins.table())
