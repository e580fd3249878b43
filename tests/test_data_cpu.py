"""CPU: the host logic of the data path (split, shuffle order, batch boundaries, RNG consumption, rank sharding)
against stock torch.utils.data used the way dataset_code.py:165-178 uses it.  The device gather itself is checked
in tests/test_gpu_data.py."""
import importlib

import pytest
import torch
from torch.utils.data import DataLoader, Dataset, random_split

D = importlib.import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.data")


class _Idx(Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return torch.tensor(i)


class _FakeResident:
    """Stands in for ResidentImages on the CPU: DeviceLoader needs len(), .images.device and .batch(indices); batch()
    hands the index slice back, so the REAL DeviceLoader.__iter__ can be driven without a GPU."""

    def __init__(self, n):
        self.n = n
        self.images = torch.empty(0)

    def __len__(self):
        return self.n

    def batch(self, idx):
        return idx.tolist()


@pytest.mark.parametrize("n,bs", [(103, 8), (64, 64), (10, 3), (1, 4)])
def test_split_and_epoch_orders_equal_torch_dataloader(n, bs):
    # the reference: random_split then two epochs of (train loop, validation loop)
    torch.manual_seed(42)
    ds = _Idx(n)
    ts = round(0.9 * n)
    tr, te = random_split(ds, [ts, n - ts])
    tl = DataLoader(tr, batch_size=bs, shuffle=True, num_workers=0)
    vl = DataLoader(te, batch_size=bs, shuffle=False, num_workers=0)
    ref = []
    for _ in range(2):
        ref.append(([b.tolist() for b in tl], [b.tolist() for b in vl]))
    ref_tail = torch.rand(3)

    torch.manual_seed(42)
    tri, tei = D.random_split_indices(n, 0.9)
    assert tri.tolist() == list(tr.indices) and tei.tolist() == list(te.indices)
    fake = _FakeResident(n)
    mt = D.DeviceLoader(fake, tri, bs, shuffle=True)
    mv = D.DeviceLoader(fake, tei, bs, shuffle=False)
    assert len(mt) == len(tl) and len(mv) == len(vl)
    for epoch in range(2):
        for loader, want in ((mt, ref[epoch][0]), (mv, ref[epoch][1])):
            order = loader.epoch_order().tolist()
            got = [order[i:i + bs] for i in range(0, len(order), bs)]
            assert got == want
    assert torch.equal(torch.rand(3), ref_tail)          # the default RNG stream advanced by exactly as much


@pytest.mark.parametrize("n,bs,world", [(103, 8, 4), (4 * 8 * 3 + 9, 4, 8), (64, 8, 8), (5, 4, 8), (8 * 4 + 8, 4, 8)])
def test_rank_sharding_gives_every_rank_the_same_steps_and_rows(n, bs, world):
    """Drives the real DeviceLoader.__iter__ for every rank.  (B=4, W=8 with 9 left-over samples is the case that
    used to produce shards [2,2,2,2,1,0,0,0]: a rank without rows never reaches the gradient all-reduce.)"""
    torch.manual_seed(1)
    idx = torch.randperm(n)
    fake = _FakeResident(n)
    shards = []
    for r in range(world):
        torch.manual_seed(7)
        ld = D.DeviceLoader(fake, idx, bs, shuffle=True, rank=r, world=world)
        shards.append(list(ld))
        assert len(ld) == len(shards[-1])
    torch.manual_seed(7)
    whole = D.DeviceLoader(fake, idx, bs * world, shuffle=True).epoch_order().tolist()
    steps = len(shards[0])
    assert all(len(s) == steps for s in shards)
    full, rem = divmod(n, bs * world)
    assert steps == full + (1 if rem >= world else 0)
    seen = []
    for k in range(steps):
        rows = {len(shards[r][k]) for r in range(world)}
        assert len(rows) == 1 and rows.pop() >= 1                 # equal, non-empty shards on every rank
        merged = sum((shards[r][k] for r in range(world)), [])
        assert merged == whole[k * bs * world:k * bs * world + len(merged)]
        seen += merged
    assert n - len(seen) == (rem % world if rem >= world else rem)   # only the unsplittable tail is skipped


def test_single_process_loader_keeps_the_ragged_last_batch():
    fake = _FakeResident(10)
    got = list(D.DeviceLoader(fake, torch.arange(10), 4, shuffle=False))
    assert got == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]]


def test_bad_arguments():
    fake = _FakeResident(4)
    with pytest.raises(IndexError):
        D.DeviceLoader(fake, torch.tensor([0, 4]), 2)
    with pytest.raises(ValueError):
        D.DeviceLoader(fake, torch.tensor([0, 1]), 0)
    with pytest.raises(ValueError):
        D.DeviceLoader(fake, torch.tensor([0, 1]), 2, rank=2, world=2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        D.ResidentImages(torch.zeros(2, 4, 4, 3, dtype=torch.uint8), device="cpu")
