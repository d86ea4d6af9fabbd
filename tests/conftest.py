import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """import jpeg-decoder_amd/ (the directory name is not an identifier) as jpeg_decoder_amd."""
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def manifest():
    return json.load(open(os.path.join(GOLD, "manifest.json")))["entries"]


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


def golden_stream(name):
    return open(os.path.join(GOLD, "streams", name + ".jpg"), "rb").read()


def golden_arrays(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return z["coef"], z["ycbcr"], z["rgb"]


def all_golden_names():
    m = json.load(open(os.path.join(GOLD, "manifest.json")))["entries"]
    return sorted(m)


def plain_names():
    m = json.load(open(os.path.join(GOLD, "manifest.json")))["entries"]
    return sorted(n for n in m if not m[n]["dri"])


def dri_names():
    m = json.load(open(os.path.join(GOLD, "manifest.json")))["entries"]
    return sorted(n for n in m if m[n]["dri"])
