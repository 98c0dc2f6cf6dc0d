import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))   # tests may import the oracle (the checker)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def device_status_clean_at_session_end():
    """After the last test of a GPU session: no kernel of ANY test gave up one of its bounded waits (kpr_device_status; the
    tests that raise the word on purpose clear it themselves)."""
    yield
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        has_gpu = False
    if has_gpu:
        from kapre_amd import _ffi
        assert _ffi.device_status(raise_on_error=False) == 0


class Golden:
    """tests/golden/kapre_ref_cases.{npz,json}: outputs of the reference's own LAYER code (kapre/*.py imported
    unmodified by oracle/make_golden.py) executed on numpy stand-ins for the tf.* / librosa.* symbols it calls
    (oracle/ref_stubs/).  Kapre-level behaviour (defaults, padding, transposes, axes, config keys) therefore comes
    from the reference; the L0 arithmetic underneath (tf.signal.stft / windows, librosa.filters.mel ...) is the
    oracle's restatement, which tests/test_oracle_published.py pins to published third-party values."""

    def __init__(self):
        self.arrays = np.load(os.path.join(GOLDEN, "kapre_ref_cases.npz"))
        with open(os.path.join(GOLDEN, "kapre_ref_cases.json")) as f:
            meta = json.load(f)
        self.cases = meta["cases"]
        self.errors = meta["errors"]

    def names(self, kind):
        return sorted(n for n, c in self.cases.items() if c["kind"] == kind)

    def get(self, name):
        c = self.cases[name]
        return c["kwargs"], self.arrays[name + "__x"], self.arrays[name + "__y"], c["extra"]


_GOLDEN = Golden()


@pytest.fixture(scope="session")
def golden():
    return _GOLDEN


def golden_names(kind):
    return _GOLDEN.names(kind)


def speech(length=8000, offset=0):
    src = np.load(os.path.join(GOLDEN, "speech_test_file.npz"))["audio_data"].astype(np.float32)
    return src[offset:offset + length]


def rel_err(a, b):
    """max |a-b| relative to the scale of the reference b (north_star: <= 1e-4)."""
    a = np.asarray(a)
    b = np.asarray(b)
    scale = np.max(np.abs(b)) if b.size else 1.0
    return float(np.max(np.abs(a - b)) / max(scale, 1e-30)) if b.size else 0.0
