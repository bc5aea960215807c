import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-engine_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")
    config.addinivalue_line("markers", "slow: minutes of CPU; runs only with RT_SLOW=1")
    # make sure the libraries exist (no-op when up to date; hipcc cross-compiles without a GPU)
    import __graft_entry__ as g
    g.build(quiet=True)


def pytest_collection_modifyitems(config, items):
    if os.environ.get("RT_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="set RT_SLOW=1 to run")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json
    gdir = os.path.join(ROOT, "tests", "golden")
    return dict(dir=gdir, manifest=json.load(open(os.path.join(gdir, "manifest.json"))),
                vectors=json.load(open(os.path.join(gdir, "ref_vectors.json"))))
