import os
import sys

import numpy as np
import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
PKG = os.path.join(ROOT, "car-racing_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def AB():
    A = np.genfromtxt(os.path.join(ROOT, "data/sys/LTI/matrix_A.csv"), delimiter=",")
    B = np.genfromtxt(os.path.join(ROOT, "data/sys/LTI/matrix_B.csv"), delimiter=",")
    return A, B


@pytest.fixture(scope="session")
def orc():
    import oracle

    return oracle.load()


class Golden:
    def __init__(self, path):
        self.z = np.load(path, allow_pickle=False)
        self.names = [str(n) for n in self.z["names"]]

    def case(self, name):
        pre = name + "/"
        return {k[len(pre):]: self.z[k] for k in self.z.files if k.startswith(pre)}


@pytest.fixture(scope="session")
def golden_mpccbf():
    return Golden(os.path.join(GOLDEN, "mpccbf.npz"))


@pytest.fixture(scope="session")
def golden_planner():
    return Golden(os.path.join(GOLDEN, "planner.npz"))


@pytest.fixture(scope="session")
def golden_racing_game():
    return np.load(os.path.join(GOLDEN, "racing_game.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden_path():
    return Golden(os.path.join(GOLDEN, "path_planner.npz"))
