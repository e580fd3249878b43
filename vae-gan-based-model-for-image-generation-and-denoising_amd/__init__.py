"""MI355X-native VAE-GAN training path.  Importable as ``vaegan_amd`` through the alias
module at the repository root (this directory's name is not a Python identifier).

Drop-in surface: ``Encoder``, ``ConvBlock`` (main_vae.py:20-58), ``Generator``,
``Discriminator``, ``weights_init`` (gan_code.py:16-97), ``Adam`` (torch.optim.Adam as used at
vaegan_code.py:42-44), ``BCELoss`` / ``MSELoss`` (vaegan_code.py:46-47), ``configure_seed``
(utils.py:6-14), ``VAEGANTrainer`` (the loop body of vaegan_code.py:65-135) and ``graphed`` (hipGraph replay of a
reference-shaped step function).
"""
from . import data  # noqa: F401
from . import geometry  # noqa: F401
from .ddp import GradReducer
from .denoise import denoise_eval, validation_epoch
from .graphed import graphed
from .losses import BCELoss, MSELoss
from .nets import ConvBlock, Discriminator, Encoder, Generator, weights_init
from .optim import Adam
from .siblings import DCGANTrainer, VAETrainer, WGANTrainer
from .trainer import LOSS_NAMES, VAEGANTrainer
from .utils import configure_seed

__all__ = ["ConvBlock", "Encoder", "Generator", "Discriminator", "weights_init", "Adam", "BCELoss", "MSELoss",
           "VAEGANTrainer", "LOSS_NAMES", "configure_seed", "geometry", "denoise_eval", "validation_epoch", "GradReducer", "data", "VAETrainer", "DCGANTrainer", "WGANTrainer", "graphed"]
