"""Import alias: ``import vaegan_amd`` -> the package directory
``vae-gan-based-model-for-image-generation-and-denoising_amd`` (not a valid identifier)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("vae-gan-based-model-for-image-generation-and-denoising_amd")
# submodules under the alias too (``from vaegan_amd.data import get_dataset_loaders``): the SAME module objects
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_pkg.__name__ + "."):
        sys.modules[__name__ + _name[len(_pkg.__name__):]] = _mod
sys.modules[__name__] = _pkg
