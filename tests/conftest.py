import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_schema(net):
    from collections import OrderedDict
    with open(os.path.join(GOLD, f"schema_{net}.json")) as f:
        raw = json.load(f, object_pairs_hook=OrderedDict)
    return OrderedDict((k, (tuple(v[0]), v[1])) for k, v in raw.items())


def load_golden(name):
    with np.load(os.path.join(GOLD, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


_SD_CACHE = {}


def synth_sd(net, seed=0):
    """Synthetic state_dict for `net`, cached per session (CPU tensors)."""
    from future_urban_scene_generation_amd.synth import synth_state_dict
    key = (net, seed)
    if key not in _SD_CACHE:
        _SD_CACHE[key] = synth_state_dict(net, load_schema(net), seed)
    return _SD_CACHE[key]


@pytest.fixture(autouse=True)
def _no_grad():
    with torch.no_grad():
        yield
