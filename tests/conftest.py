"""pytest configuration: `gpu` marker + shared synthetic-checkpoint fixtures."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_SD_CACHE = {}


def cached_state_dict(geom_name: str, variant: str = "xavier", end_to_end: bool = True,
                      img_feature_dim=None, eos_idx=None):
    """Synthetic checkpoints are deterministic; build each once per session."""
    from on_device_image_captioning_amd import weights as W
    g = getattr(W, geom_name)
    if eos_idx is None:
        eos_idx = 77 if geom_name == "FULL" else 2
    key = (geom_name, variant, end_to_end, img_feature_dim, eos_idx)
    if key not in _SD_CACHE:
        _SD_CACHE[key] = W.synth_state_dict(g, variant=variant, end_to_end=end_to_end,
                                            img_feature_dim=img_feature_dim, eos_idx=eos_idx)
    return _SD_CACHE[key]
