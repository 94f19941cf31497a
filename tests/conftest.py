import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def tree_spec(fixture):
    """Decode the tree description stored in a fixture (topology, time, G, K)."""
    d = json.loads(str(fixture["tree"]) if not isinstance(fixture, str) else fixture)
    if d["int_labels"]:
        d["time"] = {int(k): v for k, v in d["time"].items()}
    return d


def label_of(spec, text):
    return int(text) if spec["int_labels"] else str(text)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session", autouse=True)
def device_hw_tables():
    """On a GPU box: give the scalar model (oracle/nb_model.c) the device's own tables of the three hardware
    functions of the inversion class (v_rcp_f32, v_log_f32, v_exp_f32 -- written by the product's probe kernel,
    prosstt_amd_hw_math) and the function that answers the gamma-Poisson class's questions (prosstt_amd_hw_math_at:
    y[i] = op(x[i]) on the device), so that every comparison of device counts with the model is bit for bit.  Without
    a GPU the model stays on its libm stand-ins (the same law; no device result is compared then)."""
    if not has_gpu():
        yield False
        return
    from oracle import nb_model
    nb_model.install_hw_tables_from_device()
    assert nb_model.hw_mode()
    yield True
