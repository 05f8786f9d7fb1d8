import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


# ---- measured margins: every tolerance check that goes through chk() is recorded (test, line, value, bar) and the list
# is written to gpurun_out/parity_margins.json at the end of a GPU session, so that the bars in the tests can be kept at
# about twice what the kernels actually deliver (profiles/*_parity_margins.json holds the committed copy)
_MARGINS = []


def chk(value, bar, name=None):
    """name: a stable key for rows that something else reads back (bench.parity_statement) -- line numbers move"""
    import inspect
    fr = inspect.stack()[1]
    row = {"test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0], "line": fr.lineno, "value": float(value), "bar": float(bar)}
    if name is not None:
        row["name"] = name
    _MARGINS.append(row)
    return value < bar


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        try:
            import bench
            h = bench.kernel_sources_hash()      # which kernel sources these margins were measured on (bench.parity_statement)
        except Exception:
            h = None
        with open(os.path.join(out, "parity_margins.json"), "w") as f:
            json.dump({"kernel_sources_hash": h, "rows": _MARGINS}, f, indent=0)
    except OSError:
        pass
