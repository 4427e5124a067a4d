import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def tmp_models(tmp_path_factory):
    """Session cache dir for seed-regenerated .bin models."""
    return str(tmp_path_factory.mktemp("models"))


@pytest.fixture(scope="session")
def tiny_model_path(tmp_models):
    g = golden("tiny.npz")
    p = os.path.join(tmp_models, "tiny.bin")
    with open(p, "wb") as f:
        f.write(g["model_bin"].tobytes())
    return p


@pytest.fixture(scope="session")
def mid_model_path(tmp_models):
    from fast_dnn_amd import formats as F

    p = os.path.join(tmp_models, "mid.bin")
    F.write_model_bin(p, F.synth_net([432, 256, 256, 256, 1000], seed=5))
    assert F.sha256_file(p) == str(golden("mid_lazy.npz")["model_sha256"])
    return p


@pytest.fixture(scope="session")
def sat_model_path(tmp_models):
    from fast_dnn_amd import formats as F

    p = os.path.join(tmp_models, "sat.bin")
    F.write_model_bin(p, F.synth_net([432, 128, 128, 128, 200], seed=8))
    assert F.sha256_file(p) == str(golden("sat.npz")["model_sha256"])
    return p


@pytest.fixture(scope="session")
def net_model_path():
    """The full 432->7x2048->8000 net, regenerated from its seed (170 MB, cached in /tmp)."""
    from fast_dnn_amd import formats as F

    p = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fdnn_net_seed1_gauss.bin")
    F.ensure_model_file(p, F.NET_TOPOLOGY, seed=1, mode="gauss")
    return p
