"""Data path of the reference (dataset_code.py:137-178: ``CelebAHQDataset`` + ``get_dataset_loaders``), MI355X-first.

The reference preloads every decoded image into host RAM as float tensors and lets a DataLoader copy one batch per
step over PCIe.  Here the decoded images stay **u8 and resident in HBM** (CelebA-HQ 30 000 x 256 x 256 x 3 = 5.9 GB
of the 288 GB); a batch is assembled ON THE DEVICE from the sampler's indices by one HIP kernel that also applies
``ToTensor`` + ``Normalize((0.5,), (0.5,))`` (dataset_code.py:147-150) -- no per-step host work, no PCIe traffic,
4x less resident memory than f32, and the batch equals the reference's bit for bit.

What is reproduced exactly (checked against torch.utils.data in tests/test_data_cpu.py):
  * file order ``glob.iglob(folder/*.jpg)`` and ``dataset_size`` truncation (:141-145),
  * the 90/10 ``random_split`` (:175) -- ``randperm(n)`` from torch's default generator,
  * ``DataLoader(shuffle=True)`` order, batch boundaries and the ragged last batch (``drop_last=False``, the
    variable last-batch size of vaegan_code.py:67), including how much of the default RNG stream an epoch consumes
    (one int64 draw for the iterator's base seed, one for the sampler's private generator),
  * ``DataLoader(shuffle=False)`` for the validation split.
New (the reference is single-process): ``rank`` / ``world`` shard every global batch of ``world * batch_size``
samples contiguously over the ranks (same sample order as a single process with the global batch).
JPEG decoding is host work (PIL, a process pool as dataset_code.py:153-155); there is no device JPEG decoder here.
"""
import glob
import os
from multiprocessing import Pool, cpu_count
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

from . import ops


def _decode(path: str) -> np.ndarray:
    from PIL import Image                      # torchvision's default_loader: Image.open(f).convert("RGB")
    with open(path, "rb") as f:
        img = Image.open(f)
        return np.asarray(img.convert("RGB"), dtype=np.uint8)


def list_images(image_folder: str, dataset_size: Optional[int] = None):
    paths = list(glob.iglob(os.path.join(image_folder, "*.jpg"), recursive=False))      # dataset_code.py:141-142
    if dataset_size is not None:
        paths = paths[:dataset_size]                                                     # :144-145
    return paths


def decode_folder(image_folder: str, dataset_size: Optional[int] = None, workers: Optional[int] = None) -> torch.Tensor:
    """-> uint8 tensor [N, H, W, 3] on the host (all images must share one size, as CelebA-HQ does)."""
    paths = list_images(image_folder, dataset_size)
    if not paths:
        raise RuntimeError(f"no *.jpg files in {image_folder}")
    if workers is None:
        workers = max(1, cpu_count() - 2)                                                # :153
    if workers > 1 and len(paths) > 64:
        with Pool(workers) as pool:
            arrs = pool.map(_decode, paths, chunksize=64)
    else:
        arrs = [_decode(p) for p in paths]
    shape = arrs[0].shape
    for p, a in zip(paths, arrs):
        if a.shape != shape:
            raise RuntimeError(f"{p}: image size {a.shape} differs from {shape} (the resident layout is one [N,H,W,3] array)")
    return torch.from_numpy(np.stack(arrs))


class ResidentImages:
    """The decoded image set in HBM: u8 [N, H, W, C].  ``ds[i]`` returns what ``CelebAHQDataset[i]`` returns
    (f32 [C,H,W] in [-1,1]) -- as a device tensor."""

    def __init__(self, images_u8: torch.Tensor, device="cuda"):
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4:
            raise RuntimeError("ResidentImages needs a uint8 [N,H,W,C] tensor")
        self.images = images_u8.to(device).contiguous()
        if not self.images.is_cuda:
            raise RuntimeError("ResidentImages lives in MI355X HBM ('cuda'); there is no CPU path")

    @classmethod
    def from_folder(cls, image_folder: str, dataset_size: Optional[int] = None, device="cuda", workers=None):
        return cls(decode_folder(image_folder, dataset_size, workers), device)

    def __len__(self) -> int:
        return self.images.shape[0]

    @property
    def image_shape(self) -> Tuple[int, int, int]:
        n, h, w, c = self.images.shape
        return (c, h, w)                                           # dataset[0].numpy().shape (:178)

    def batch(self, idx: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return ops.gather_normalize_u8(self.images, idx, out)

    def __getitem__(self, i: int) -> torch.Tensor:
        n = len(self)
        if not -n <= i < n:
            raise IndexError(i)
        return self.batch(torch.tensor([i % n], dtype=torch.int64, device=self.images.device))[0]


def random_split_indices(n: int, train_p: float = 0.9):
    """torch.utils.data.random_split(dataset, [train, test]) of dataset_code.py:172-175: one randperm(n) from
    torch's DEFAULT generator (so utils.configure_seed governs it), first ``round(train_p*n)`` indices train."""
    train_size = round(train_p * n)
    perm = torch.randperm(n)
    return perm[:train_size].clone(), perm[train_size:].clone()


class DeviceLoader:
    """Iterates batches of a subset of a ResidentImages set like ``DataLoader(Subset, batch_size, shuffle,
    num_workers=0, drop_last=False)`` would, yielding device tensors [b,C,H,W] f32."""

    def __init__(self, dataset: ResidentImages, indices: torch.Tensor, batch_size: int = 64, shuffle: bool = False,
                 rank: int = 0, world: int = 1):
        if batch_size <= 0 or world <= 0 or not 0 <= rank < world:
            raise ValueError("bad batch_size / rank / world")
        idx = torch.as_tensor(indices, dtype=torch.int64).cpu()
        if idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= len(dataset)):
            raise IndexError("subset index outside the dataset")
        self.dataset, self.indices = dataset, idx
        self.batch_size, self.shuffle, self.rank, self.world = batch_size, shuffle, rank, world
        self.last_order: Optional[torch.Tensor] = None           # the epoch's sample order (host), for inspection

    def __len__(self) -> int:
        return sum(1 for _ in self.global_batches(self.indices.numel()))

    def global_batches(self, n: int):
        """(lo, hi) bounds of THIS rank's shard of every global batch of an epoch of n samples.
        world == 1: exactly DataLoader(drop_last=False): the ragged last batch is kept (vaegan_code.py:67).
        world  > 1: every rank must take the same number of steps with the SAME number of rows -- a rank without
        rows would miss the gradient all-reduce its peers wait in, and unequal shards make SUM/world differ from the
        global-batch gradient (ddp.py) and break the SyncBN element count.  The ragged last global batch is therefore
        cut down to the largest multiple of `world` (at most world-1 samples of an epoch are skipped; they come back
        in the next epoch's shuffle) and dropped when it has fewer samples than ranks."""
        B, W, r = self.batch_size, self.world, self.rank
        for start in range(0, n, B * W):
            stop = min(n, start + B * W)
            if W == 1:
                yield start, stop
                continue
            per = (stop - start) // W                                      # equal shards, remainder skipped
            if per == 0:
                return
            yield start + r * per, start + (r + 1) * per

    def epoch_order(self) -> torch.Tensor:
        """Consumes the default RNG exactly as one ``iter(DataLoader)`` + first ``next()`` does."""
        torch.empty((), dtype=torch.int64).random_()                       # _BaseDataLoaderIter: base seed
        n = self.indices.numel()
        if self.shuffle:                                                   # RandomSampler.__iter__, generator=None
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            g = torch.Generator()
            g.manual_seed(seed)
            perm = torch.randperm(n, generator=g)
        else:
            perm = torch.arange(n)
        return self.indices[perm]

    def bind_output(self, out: Optional[torch.Tensor]) -> None:
        """Assemble every FULL batch into `out` ([B, C, H, W] f32 on the device, e.g. VAEGANTrainer.graph_input()) and
        yield that same tensor: the consumer must be done with a batch before asking for the next (a training loop
        is).  A ragged last batch gets its own tensor.  None unbinds."""
        self._out = out

    def __iter__(self) -> Iterator[torch.Tensor]:
        order = self.epoch_order()
        self.last_order = order
        dev_order = order.to(self.dataset.images.device)                   # one small H2D copy per epoch
        out = getattr(self, "_out", None)
        for lo, hi in self.global_batches(order.numel()):
            idx = dev_order[lo:hi]
            if out is not None and out.shape[0] == idx.numel():
                yield self.dataset.batch(idx, out)
            else:
                yield self.dataset.batch(idx)


def get_dataset_loaders(path, batch_size=64, train_p=0.9, dataset_size=None, device="cuda", rank=0, world=1,
                        workers=None):
    """dataset_code.py:165-178 for dataset_type 'HQ' -> (train_loader, test_loader, image_shape)."""
    ds = path if isinstance(path, ResidentImages) else ResidentImages.from_folder(path, dataset_size, device, workers)
    train_idx, test_idx = random_split_indices(len(ds), train_p)
    train = DeviceLoader(ds, train_idx, batch_size, shuffle=True, rank=rank, world=world)
    test = DeviceLoader(ds, test_idx, batch_size, shuffle=False, rank=rank, world=world)
    return train, test, ds.image_shape
