import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_meta(g):
    return json.loads(str(g["meta/cfg"]))


def rel_l2(a, b):
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b)).double().reshape(-1)
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0)


@pytest.fixture(scope="session")
def gpu_available():
    return torch.cuda.is_available()
