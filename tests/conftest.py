import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robust-segmentation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# Suite budget (round 5): the driver gives `pytest -m gpu` 1200 s.  No single GPU test of the default selection may take more
# than SEA_TEST_MAX_SECONDS (default 180; the builder's full-set runs use SEA_MIOU_FULL=1, which lifts it): a test that creeps past
# it fails HERE, by name, instead of taking the whole suite's evidence with it by timeout at the end of a round.  (Slowest test
# of the suite: 53 s.  The cap was 75 s until a two-rank test that normally takes 21 s took 65 s as the FIRST process-spawning
# test on a fresh box -- cold page cache for the children's `import torch` --, and the 57 s test took 80 s on one lease: the
# margin is for the box, not for the code.)
_MAX_S = float(os.environ.get("SEA_TEST_MAX_SECONDS", "0" if os.environ.get("SEA_MIOU_FULL") == "1" else "180"))
DURATIONS = {}


@pytest.hookimpl(wrapper=True)
def pytest_runtest_call(item):
    import time
    t0 = time.perf_counter()
    res = yield
    dt = time.perf_counter() - t0
    DURATIONS[item.nodeid] = dt
    if _MAX_S > 0 and "gpu" in item.keywords and dt > _MAX_S:
        pytest.fail(f"{item.nodeid} took {dt:.0f} s: over the {_MAX_S:.0f} s budget of a GPU test (tests/conftest.py)", pytrace=False)
    return res


def pytest_terminal_summary(terminalreporter):
    if DURATIONS:
        total = sum(DURATIONS.values())
        worst = sorted(DURATIONS.items(), key=lambda kv: -kv[1])[:5]
        terminalreporter.write_line(f"[suite budget] {total:.0f} s in test bodies; slowest: "
                                    + "; ".join(f"{k.split('::')[-1]} {v:.0f} s" for k, v in worst))


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in d.files:
        v = d[k]
        out[k] = torch.from_numpy(v) if v.ndim > 0 else v.item()
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden
