"""GPU: the data path (HBM-resident u8 image set + on-device batch assembly) against the CPU oracle of
dataset_code.py:137-178 (oracle/data_ref.py: PIL decode, ToTensor, Normalize, stock random_split / DataLoader)."""
import os

import numpy as np
import pytest
import torch

import data_ref as DR
import vaegan_amd as V
from test_gpu_parity import build

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def jpeg_folder(tmp_path_factory):
    from PIL import Image
    d = tmp_path_factory.mktemp("celeba_like")
    rng = np.random.default_rng(5)
    for i in range(45):
        # smooth content + noise so that the JPEG round trip leaves a wide range of byte values
        yy, xx = np.mgrid[0:64, 0:64]
        base = np.stack([(yy * 4 + i * 3) % 256, (xx * 4 + i * 7) % 256, ((yy + xx) * 2 + i * 11) % 256], -1)
        img = np.clip(base + rng.integers(-20, 20, base.shape), 0, 255).astype(np.uint8)
        Image.fromarray(img, "RGB").save(os.path.join(d, f"{i:05d}.jpg"), quality=92)
    (d / "notes.txt").write_text("not an image")
    return str(d)


def test_resident_batches_equal_reference_transform_bitwise(jpeg_folder):
    ref = DR.RefCelebAHQDataset(jpeg_folder)
    ds = V.data.ResidentImages.from_folder(jpeg_folder, device=DEV, workers=1)
    assert len(ds) == len(ref) == 45 and ds.image_shape == tuple(ref[0].numpy().shape) == (3, 64, 64)
    idx = torch.tensor([3, 0, 44, 17, 17, 9], device=DEV)
    got = ds.batch(idx).cpu()
    want = torch.stack([ref[i] for i in idx.tolist()])
    assert got.dtype == torch.float32 and torch.equal(got, want)
    assert torch.equal(ds[7].cpu(), ref[7]) and torch.equal(ds[-1].cpu(), ref[44])
    assert float(got.min()) >= -1.0 and float(got.max()) <= 1.0
    # every byte value maps exactly like ToTensor + Normalize
    ramp = torch.arange(256, dtype=torch.uint8).view(1, 16, 16, 1).expand(1, 16, 16, 3).contiguous()
    out = V.data.ResidentImages(ramp, DEV).batch(torch.zeros(1, dtype=torch.int64, device=DEV)).cpu()
    exp = ramp.permute(0, 3, 1, 2).to(torch.float32).div(255).sub(0.5).div(0.5)
    assert torch.equal(out, exp)
    with pytest.raises(IndexError):
        ds[45]


def test_loaders_yield_the_reference_batches_bitwise(jpeg_folder):
    """get_dataset_loaders (dataset_code.py:165-178): same split, same shuffled order, same ragged last batch,
    same values -- two epochs of train + validation, default RNG seeded as utils.configure_seed does."""
    torch.manual_seed(42)
    rtl, rvl, rshape = DR.get_dataset_loaders(jpeg_folder, batch_size=8)
    ref = [([b.clone() for b in rtl], [b.clone() for b in rvl]) for _ in range(2)]
    torch.manual_seed(42)
    tl, vl, shape = V.data.get_dataset_loaders(jpeg_folder, batch_size=8, device=DEV, workers=2)
    assert tuple(shape) == tuple(rshape)
    for epoch in range(2):
        for loader, want in ((tl, ref[epoch][0]), (vl, ref[epoch][1])):
            got = [b.cpu() for b in loader]
            assert len(got) == len(want) == len(loader)
            for a, b in zip(got, want):
                assert a.shape == b.shape and torch.equal(a, b)
    assert ref[0][0][-1].shape[0] == 40 % 8 or ref[0][0][-1].shape[0] == 8      # 40 train images: 5 full batches
    assert ref[0][1][-1].shape[0] == 5                                           # 5 validation images: ragged


def test_two_rank_shards_concatenate_to_the_single_process_batches(jpeg_folder):
    ds = V.data.ResidentImages.from_folder(jpeg_folder, device=DEV, workers=1)
    idx = torch.arange(len(ds))
    torch.manual_seed(3)
    whole = [b.cpu() for b in V.data.DeviceLoader(ds, idx, 16, shuffle=True)]
    parts = []
    for r in range(2):
        torch.manual_seed(3)
        parts.append([b.cpu() for b in V.data.DeviceLoader(ds, idx, 8, shuffle=True, rank=r, world=2)])
    assert len(parts[0]) == len(parts[1]) == len(whole) == 3
    for k, w in enumerate(whole):
        both = torch.cat([parts[0][k], parts[1][k]])
        assert parts[0][k].shape[0] == parts[1][k].shape[0]                 # equal shards on every rank, every step
        assert torch.equal(both, w[:both.shape[0]])
        assert w.shape[0] - both.shape[0] < 2                               # only an unsplittable remainder is skipped
    assert whole[-1].shape[0] == 13 and parts[0][-1].shape[0] == 6          # 45 = 16 + 16 + 13 -> shards of 6


def test_training_from_the_resident_dataset_runs_the_variable_last_batch(jpeg_folder):
    """vaegan_code.py:65-67: the loop takes whatever batch size the loader yields (drop_last=False)."""
    torch.manual_seed(42)
    tl, _, shape = V.data.get_dataset_loaders(jpeg_folder, batch_size=16, device=DEV, workers=1)
    e, g, d, tr = build(shape[1])
    sizes, losses = [], []
    for real in tl:
        sizes.append(real.shape[0])
        losses.append(tr.train_step(real, 60)[:5].clone())
    torch.cuda.synchronize()
    assert sizes == [16, 16, 8]                                   # 40 training images
    assert all(bool(torch.isfinite(l).all()) for l in losses)


def test_validation_epoch_over_the_resident_loader_vs_oracle(jpeg_folder):
    """vaegan_code.py:147-191: eval-mode E -> G over the validation loader, val_loss = sum(mse + 0.1*KL) / samples,
    SSIM over all images, the ragged last batch included -- against the oracle's loop on the reference data path's
    batches (oracle/data_ref.py), identical weights and injected noise."""
    import vaegan_ref as R
    from _inputs import make_inputs
    from test_gpu_parity import sync_from_oracle
    S = 64
    torch.manual_seed(42)
    _, rvl, _ = DR.get_dataset_loaders(jpeg_folder, batch_size=2)
    ref_batches = [b.clone() for b in rvl]                                   # 5 validation images: 2 + 2 + 1
    torch.manual_seed(42)
    _, vl, _ = V.data.get_dataset_loaders(jpeg_folder, batch_size=2, device=DEV, workers=1)
    e, g, d, tr = build(S)
    o = R.RefVAEGAN(img_size=S, seed=42)
    real, ez, er, ec = make_inputs(4, S, 4711)
    o.train_step(real, ez, er, ec, 60)                                       # non-trivial BatchNorm running statistics
    sync_from_oracle(o, e, g, d, tr)
    gen = torch.Generator().manual_seed(99)
    noises = [(torch.randn(b.shape, generator=gen), torch.randn(b.shape[0], 100, generator=gen)) for b in ref_batches]
    want = R.validation_epoch(o, ref_batches, [(0.05 * n, z) for n, z in noises])
    got = V.validation_epoch(e, g, vl, noise_fn=lambda i, img: (noises[i][0].to(DEV), noises[i][1].to(DEV)))
    assert not e.training and not g.training and d.training                  # :147-148 leaves the discriminator alone
    assert got["samples"] == want["samples"] == 5 and got["batches"] == 3
    assert abs(got["val_loss"] - want["val_loss"]) <= 1e-4 * abs(want["val_loss"])
    assert abs(got["ssim"] - want["ssim"]) < 1e-4 and abs(got["psnr"] - want["psnr"]) < 1e-3
    # device-generated noise: finite, reproducible under utils.configure_seed (which restarts the HIP noise stream)
    V.configure_seed(7)
    a = V.validation_epoch(e, g, vl)
    c = V.validation_epoch(e, g, vl)
    V.configure_seed(7)
    b = V.validation_epoch(e, g, vl)
    assert a == b and a != c and all(np.isfinite(v) for v in a.values())


def test_loader_bound_to_the_graph_input_trains_identically_without_the_copy(jpeg_folder):
    """Zero-copy hand-off: DeviceLoader.bind_output(trainer.graph_input()) assembles each full batch straight into the
    buffer the captured iteration reads; the losses equal those of the unbound loader bit for bit."""
    runs = []
    for bound in (False, True):
        V.configure_seed(42)
        tl, _, shape = V.data.get_dataset_loaders(jpeg_folder, batch_size=16, device=DEV, workers=1)
        e, g, d, tr = build(shape[1])
        losses = []
        for epoch in range(3):                                    # epoch 0: eager warm-up + capture; then replays
            for real in tl:
                if real.shape[0] != 16:
                    continue                                      # the ragged last batch would need its own graph
                if bound and tr.graph_input() is not None and getattr(tl, "_out", None) is None:
                    tl.bind_output(tr.graph_input())
                out = tr.train_step_graphed(real, 60)
                losses.append(out[:5].clone())
                if bound and getattr(tl, "_out", None) is not None and epoch == 2:
                    assert real.data_ptr() == tr.graph_input().data_ptr()
        torch.cuda.synchronize()
        runs.append(torch.stack(losses).cpu())
    assert torch.equal(runs[0], runs[1])
