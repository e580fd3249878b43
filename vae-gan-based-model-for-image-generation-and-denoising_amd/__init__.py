"""MI355X-native VAE-GAN training path.  Importable as ``vaegan_amd`` via the
alias module at the repository root (the directory name is not a Python identifier)."""
from . import geometry  # noqa: F401

__all__ = ["geometry"]
