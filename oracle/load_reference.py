"""Container-only helper: import the reference's hot-path classes for golden-vector generation.

TEST INFRASTRUCTURE -- never imported by the product package, bench.py's GPU leg or
anything that runs on the GPU box (``/root/reference`` does not exist there).

The reference (``/root/reference``, read-only) imports torchvision / torchmetrics at module
scope (main_vae.py:11-18, gan_code.py:4-6, dataset_code.py:4-6); neither package is
installed here and neither is touched by the hot path, so they are replaced by inert
``MagicMock`` modules (recipe of SURVEY.md section 8(c)).  Only ``torch`` is exercised by
the classes we use: ``main_vae.ConvBlock/Encoder`` (main_vae.py:20-58),
``gan_code.Generator/Discriminator/weights_init`` (gan_code.py:16-97) and
``utils.configure_seed`` (utils.py:6-14).
"""
import importlib
import importlib.machinery
import os
import sys
import types
from unittest import mock

REFERENCE_DIR = os.environ.get("VAEGAN_REFERENCE_DIR", "/root/reference")

_STUBS = [
    "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
    "torchvision.datasets", "torchvision.datasets.folder", "torchvision.utils",
    "torchmetrics", "torchmetrics.image", "torchmetrics.image.inception",
    "torchmetrics.image.fid",
]


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_DIR, "vaegan_code.py"))


def load_reference() -> types.SimpleNamespace:
    """Return a namespace with the reference's real classes (only torch is executed)."""
    if not reference_available():
        raise RuntimeError(f"reference not present at {REFERENCE_DIR} (expected on the GPU box)")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    for name in _STUBS:
        if name not in sys.modules:
            m = mock.MagicMock(name=name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
            m.__path__ = []
            sys.modules[name] = m
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    main_vae = importlib.import_module("main_vae")
    gan_code = importlib.import_module("gan_code")
    utils = importlib.import_module("utils")
    return types.SimpleNamespace(
        ConvBlock=main_vae.ConvBlock,
        Encoder=main_vae.Encoder,
        Generator=gan_code.Generator,
        Discriminator=gan_code.Discriminator,
        weights_init=gan_code.weights_init,
        configure_seed=utils.configure_seed,
    )
