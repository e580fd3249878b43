"""CPU oracle for the reference's data path (dataset_code.py:137-178, dataset_type 'HQ').

TEST INFRASTRUCTURE (same rules as vaegan_ref.py).  ``torchvision`` is not installed in this image, so the three
torchvision pieces the reference uses are restated from their documented behaviour -- parity of THOSE restatements
is unpinned (no reference fixture covers them); everything else is stock ``torch.utils.data``:

  * ``default_loader``  (dataset_code.py:6,134)  -> PIL ``Image.open(f).convert("RGB")``
  * ``transforms.ToTensor()``   (:148)  -> uint8 HWC -> CHW, ``.to(torch.float32).div(255)``
  * ``transforms.Normalize((0.5,), (0.5,))``  (:149)  -> ``tensor.sub_(mean).div_(std)`` per channel
"""
import glob
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


def load_and_transform(path: str) -> torch.Tensor:
    """_load_and_transform (dataset_code.py:132-135) with clean_transform of :147-150."""
    from PIL import Image
    with open(path, "rb") as f:
        img = Image.open(f).convert("RGB")
    t = torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).contiguous()
    t = t.to(dtype=torch.float32).div(255)                                  # ToTensor
    mean = torch.tensor([0.5]).view(-1, 1, 1)
    std = torch.tensor([0.5]).view(-1, 1, 1)
    return t.sub_(mean).div_(std)                                           # Normalize((0.5,), (0.5,))


class RefCelebAHQDataset(Dataset):
    """dataset_code.py:137-163 with preload=True (serial instead of a process pool: same result)."""

    def __init__(self, image_folder, dataset_size=None):
        pattern = os.path.join(image_folder, "*.jpg")
        self.image_paths = list(glob.iglob(pattern, recursive=False))      # :141-142
        if dataset_size is not None:
            self.image_paths = self.image_paths[:dataset_size]              # :144-145
        self.cached_data = [load_and_transform(p) for p in self.image_paths]   # :152-155

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, idx):
        return self.cached_data[idx]


def get_dataset_loaders(path, batch_size=64, train_p=0.9, dataset_size=None):
    """dataset_code.py:165-178 (dataset_type 'HQ', num_workers 0; pin_memory dropped: no device here)."""
    dataset = RefCelebAHQDataset(path, dataset_size)
    n = len(dataset)
    train_size = round(train_p * n)
    test_size = n - train_size
    train_dataset, test_dataset = torch.utils.data.dataset.random_split(dataset, [train_size, test_size])
    train_loader = DataLoader(train_dataset, batch_size=batch_size, shuffle=True, num_workers=0)
    test_loader = DataLoader(test_dataset, batch_size=batch_size, shuffle=False, num_workers=0)
    return train_loader, test_loader, dataset[0].numpy().shape
