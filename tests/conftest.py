import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True, scope="session")
def _memoized_synthetic_weights():
    """synth.make_state_dict of a ViT-L costs 5-15 s of host time and the suite asks for the same few (config, seed) pairs two dozen times: hand out
    one set of tensors per pair (a fresh dict each time; the tests only read the tensors -- load_state_dict copies, the oracle never writes).
    Bounded: the four most recent pairs (1.2 GB each at ViT-L)."""
    import json
    from collections import OrderedDict
    from toc3d_amd import synth
    real, cache = synth.make_state_dict, OrderedDict()

    def cached(cfg, seed=0, device="cpu", stress=False):
        key = (json.dumps(cfg, sort_keys=True, default=str), int(seed), str(device), bool(stress))
        if key in cache:
            cache.move_to_end(key)
        else:
            cache[key] = real(cfg, seed=seed, device=device, stress=stress)
            while len(cache) > 4:
                cache.popitem(last=False)
        return type(cache[key])(cache[key])

    synth.make_state_dict = cached
    yield
    synth.make_state_dict = real
