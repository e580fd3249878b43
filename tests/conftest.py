import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def vg_switch(monkeypatch):
    """Set one of the library's optional VG_* kernel-selection switches for the duration of a test.  The library reads
    them ONCE, when it is loaded (include/vaegan_hip.h: vg_reload_switches), so every change is followed by a reload,
    and the defaults are restored (and re-read) when the test ends."""
    from importlib import import_module
    ops = import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.ops")

    def set_switch(name, value):
        monkeypatch.setenv(name, str(value))
        ops.reload_switches()

    yield set_switch
    monkeypatch.undo()
    ops.reload_switches()
