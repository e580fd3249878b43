#!/usr/bin/env python3
"""Benchmark of the VAE-GAN training iteration (vaegan_code.py:65-135) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full iteration: E fwd, G fwd, 5 D fwd, all backward passes, 4 Adam steps (+ gradient
all-reduces over RCCL when N > 1).  Workload (BASELINE.json configs[1], SURVEY.md 8(d) "C2"): CelebA-shaped
synthetic 64x64 images, batch 128 PER GPU (weak scaling), inputs (images and the three noise tensors) resident
in HBM before the timed region, seed-42 random-init weights of the A0 size family.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

# the host driver of this pool only supports dmabuf IPC: without this RCCL's peer-memory exchange fails with
# "hipIpcGetMemHandle: invalid argument" (already exported on the GPU boxes; set here in case a launcher drops it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# ALGORITHMIC conv/convT/linear FLOP per image per step (fwd + wgrad + dgrad, D x5)
ALG_GFLOP_PER_IMAGE = {64: 6.45, 128: 8.77, 256: 12.561}
PEAK = {"fp32": 157.3, "bf16": 2500.0, "fp8": 5000.0}          # dense MFMA TFLOP/s, MI355X_MICROARCH.md (fp8: block-scaled K=128 forms)
# HBM-side bytes per gather-GEMM launch from the rocprofv3 PMC passes committed under profiles/ (separate
# --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command; FETCH_SIZE doubled per the gfx950 correction of
# MI355X_MICROARCH.md section HBM, counters in KiB): measured offline, NOT re-measured by this run.
def pmc_traffic(dtype, family="gather_gemm"):
    """(HBM-side bytes per launch of `family`, source file) from the newest committed PMC summary
    (profiles/rNN_<dtype>_pmc_traffic.json): measured offline by separate rocprofv3 --pmc passes, not by this run."""
    for rnd in ("r04", "r03", "r02", "r01"):
        f = os.path.join(ROOT, "profiles", f"{rnd}_{dtype}_pmc_traffic.json")
        try:
            return round(json.load(open(f))["families"][family]["hbm_bytes_per_launch"]), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def pmc_mfma_busy(dtype):
    """(MFMA-busy fraction of the gather-GEMM family, of the whole run, source file) from the newest committed
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass (profiles/rNN_<dtype>_mfma_busy.json, made by
    profiles/summarize_mfma.py): measured offline by that separate pass, not by this run."""
    for rnd in ("r04", "r03"):
        f = os.path.join(ROOT, "profiles", f"{rnd}_{dtype}_mfma_busy.json")
        try:
            j = json.load(open(f))
            return j["families"]["gather_gemm"]["mfma_busy_frac"], j["whole_run"]["mfma_busy_frac"], os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None, None


def cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return model, avail, os.cpu_count() or avail


HBM_PEAK_GBS = 8000.0


def make_inputs(B, S, seed, latent=100):
    g = torch.Generator().manual_seed(seed)
    real = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    return real, torch.randn(B, latent, generator=g), torch.randn(B, 3, S, S, generator=g), \
        torch.randn(B, 3, S, S, generator=g)


def cpu_baseline(S, B, budget_s=20.0):
    """The CPU oracle (bit-checked against the reference classes, tests/golden) timed on this box's host cores:
    a bounded sample of the bench workload, plus C1 (S=64, B=16: the reference's own CPU-runnable configuration,
    BASELINE.md section 3).  Reported baseline only."""
    import vaegan_ref as R
    model, avail, total = cpu_info()
    # Threads: measured on the GPU box (tools/cpu_cores_probe.py, round 3; EPYC 9575F, 256 hardware threads visible, S=64
    # B=128 fp32 step of this oracle): 8 threads 103.6 img/s, 16: 101.8, 32: 87.2, 64: 48.2, 128: 22.3, 256: 1.3 -- ATen's CPU
    # conv / BatchNorm kernels stop scaling at ~8-16 threads at these sizes and collapse when oversubscribed, so
    # torch.set_num_threads(os.cpu_count()) would report 1 % of what the host can do.  16 is used;
    # `cores` says so and `cores_available` what the box has.
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)

    def sample(S_, B_, budget):
        o = R.RefVAEGAN(img_size=S_, seed=42)
        inp = make_inputs(B_, S_, 1234)
        t0 = time.time()
        o.train_step(*inp, 60)                                   # warm-up (also sizes the sample)
        warm = time.time() - t0
        steps = max(1, min(5, int(budget / max(warm, 1e-3))))
        t0 = time.time()
        for _ in range(steps):
            o.train_step(*inp, 60)
        return (time.time() - t0) / steps, steps

    dt, steps = sample(S, B, budget_s)
    out = {"value": round(B / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
           "cpu_model": model, "cores_available": avail, "cores_total": total,
           "cores_note": "thread count with the best measured throughput of this oracle on this host class "
                         "(8: 103.6, 16: 101.8, 32: 87.2, 64: 48.2, 128: 22.3, 256: 1.3 img/s; tools/cpu_cores_probe.py)",
           "sample": f"{steps} full training steps of S={S} B={B} fp32 after 1 warm-up, CPU oracle "
                     f"(oracle/vaegan_ref.py, same ATen CPU kernels the reference executes)",
           "s_per_step": round(dt, 3)}
    if (S, B) != (64, 16):
        dt1, st1 = sample(64, 16, 5.0)
        out["c1"] = {"workload": "S=64 B=16 fp32 (BASELINE configs[0])", "value": round(16 / dt1, 2),
                     "s_per_step": round(dt1, 3), "steps": st1}
    return out


def ops_binding():
    from importlib import import_module
    return import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.ops").BINDING


def build_models(V, S, dtype, dev):
    V.configure_seed(42)
    e = V.Encoder([3, S, S], 100, dtype=dtype)
    g = V.Generator(nz=100, img_size=S, dtype=dtype)
    d = V.Discriminator(img_size=S, dtype=dtype)
    g.apply(V.weights_init), d.apply(V.weights_init)
    e.to(dev), g.to(dev), d.to(dev)
    return e, g, d


def time_steps(fn, warm, steps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def parity_path(V, S, B, dev, inputs, ops, steps=10):
    """The fp32 engine (exact-f32 MFMA, v_mfma_f32_16x16x4_f32: the reference's own arithmetic; the path whose first-step
    losses are held to 1e-4 against the reference, tests/test_gpu_parity.py) on the same workload, hipGraph replay -- with
    its OWN roofline against the 157.3 TFLOP/s fp32-MFMA peak, timed like the bf16 one (HIP start/stop events around every
    launch in an eager pass of the same steps right after the replayed ones)."""
    e, g, d = build_models(V, S, "fp32", dev)
    tr = V.VAEGANTrainer(e, g, d, *(V.Adam(m.parameters(), lr=2e-4) for m in (e, g, d)))
    tr.train()
    dt = time_steps(lambda: tr.train_step_graphed(inputs[0], 60, *inputs[1:]), 3, steps)
    out = {"dtype": "f32", "value": round(B / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3),
           "steps": steps, "mode": "VAEGANTrainer.train_step_graphed, injected noise"}
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    n0 = ops.launch_count()
    for _ in range(steps):
        tr.train_step(inputs[0], 60, *inputs[1:])
    torch.cuda.synchronize()
    ops.set_timer(None)
    out["launches_per_step"] = (ops.launch_count() - n0) // steps
    fam = timer.summary()
    gg, wg = fam.get("gather_gemm"), fam.get("wgrad")
    if gg and gg["launches"]:
        ach = gg["flops"] / (gg["ms"] * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "gg_kernel<f32> (gather-GEMM on v_mfma_f32_16x16x4_f32: conv/convT/linear fprop + dgrad)",
                           "achieved": round(ach, 2), "peak": PEAK["fp32"], "unit": "TFLOP/s", "frac": round(ach / PEAK["fp32"], 4),
                           "traffic": None, "launches_per_step": gg["launches"] // steps,
                           "avg_launch_us": round(gg["ms"] * 1e3 / gg["launches"], 2),
                           "share_of_step": round(gg["ms"] / steps / (dt * 1e3), 3),
                           "timed_in": "eager pass after the replayed steps: HIP start/stop events around every launch on its stream",
                           "step_alg_tflops": round(ALG_GFLOP_PER_IMAGE.get(S, 0) * B / (dt * 1e3), 2)}
        if wg and wg["launches"]:
            awg = wg["flops"] / (wg["ms"] * 1e-3) / 1e12
            out["roofline"]["wgrad"] = {"achieved": round(awg, 2), "frac": round(awg / PEAK["fp32"], 4),
                                        "launches_per_step": wg["launches"] // steps,
                                        "share_of_step": round(wg["ms"] / steps / (dt * 1e3), 3)}
    return out


def elided_path(V, S, B, dev, dtype, inputs, steps=20):
    """Same engine and workload with the one piece of dead work of the reference iteration removed: the Discriminator
    weight gradients of the generator-loss pass, which vaegan_code.py:133 computes and the next opt_Dis.zero_grad()
    (:103) discards unread.  Observable results are identical; reported beside the
    headline number, which executes them as the reference does."""
    e, g, d = build_models(V, S, dtype, dev)
    tr = V.VAEGANTrainer(e, g, d, *(V.Adam(m.parameters(), lr=2e-4) for m in (e, g, d)), elide_dead_grads=True)
    tr.train()
    dt = time_steps(lambda: tr.train_step_graphed(inputs[0], 60), 4, steps)
    return {"value": round(B / dt, 1), "unit": "images/sec", "ms_per_step": round(dt * 1e3, 3), "steps": steps,
            "mode": "train_step_graphed(elide_dead_grads=True)"}


def denoise_path(V, S, B, dev, dtype, inputs, steps=20):
    """BASELINE configs[3]: the denoising evaluation forward of vaegan_code.py:147-171 at the bench size --
    eval-mode Encoder -> reparameterise -> Generator on clamp(img + 0.2 * eps, -1, 1), MSE + KL, PSNR and SSIM, noise
    drawn on the device; the two host reads of denoise_eval (losses, SSIM) are inside the timed region, as the
    reference's .item() calls are."""
    e, g, _ = build_models(V, S, "bf16" if dtype == "fp8" else dtype, dev)
    e.eval(), g.eval()
    res = {}

    def step():
        res.update(V.denoise_eval(e, g, inputs[0], sigma=0.2))

    dt = time_steps(step, 3, steps)
    return {"dtype": {"fp32": "f32", "bf16": "bf16", "fp8": "bf16"}[dtype], "value": round(B / dt, 1), "unit": "images/sec",
            "ms_per_batch": round(dt * 1e3, 3), "steps": steps, "sigma": 0.2,
            "psnr_db": round(res["psnr"], 3), "ssim": round(res["ssim"], 4),
            "mode": "denoise_eval (eval-mode E -> reparam -> G, MSE, KL, PSNR, SSIM; eager launches, 2 host reads per batch)"}


def dropin_path(V, S, B, dev, dtype, inputs, steps=10, graph=True):
    """the reference trainer's own code shape (vaegan_code.py:65-135 -- module calls, torch
    ops between them, nn.BCELoss / nn.MSELoss, .backward(), optimizer.step()) on the engine's nn.Modules + Adam:
    autograd drives the HIP kernel chains, every launch marshalled through ctypes, NCHW<->NHWC at every module edge."""
    encoder, decoder, discriminator = build_models(V, S, dtype, dev)
    opt_E, opt_Dec, opt_Dis = (V.Adam(m.parameters(), lr=2e-4) for m in (encoder, decoder, discriminator))
    bce, mse = torch.nn.BCELoss(), torch.nn.MSELoss(reduction="mean")
    encoder.train(), decoder.train(), discriminator.train()
    real_images = inputs[0]
    epoch, alpha_kl, alpha_adv = 60, 0.1, 0.1

    def step():
        batch_size = real_images.size(0)
        mu, logvar = encoder(real_images)
        logvar = torch.clamp(logvar, min=-10, max=10)
        std = torch.exp(0.5 * logvar)
        z = (mu + std * torch.randn_like(std)).unsqueeze(-1).unsqueeze(-1)
        recon_images = decoder(z)
        real_labels = torch.full((batch_size,), 0.9, device=dev)
        fake_labels = torch.full((batch_size,), 0.1, device=dev)
        real_images_noisy = real_images + 0.05 * torch.randn_like(real_images)
        recon_images_noisy = recon_images + 0.05 * torch.randn_like(recon_images)
        for _ in range(2):
            d_loss = bce(discriminator(real_images_noisy), real_labels) + \
                bce(discriminator(recon_images_noisy.detach()), fake_labels)
            opt_Dis.zero_grad()
            d_loss.backward()
            opt_Dis.step()
        fake_output = discriminator(recon_images_noisy)
        recon_loss = mse(recon_images, real_images)
        kl_loss = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / batch_size
        g_loss_adv = bce(fake_output, real_labels)
        total = recon_loss + alpha_kl * min(1.0, epoch / 50) * kl_loss + alpha_adv * g_loss_adv
        opt_E.zero_grad()
        opt_Dec.zero_grad()
        total.backward()
        opt_E.step()
        opt_Dec.step()
        return total

    dt = time_steps(step, 3, steps)
    out = {"dtype": {"fp32": "f32", "bf16": "bf16"}[dtype], "value": round(B / dt, 1), "unit": "images/sec",
           "ms_per_step": round(dt * 1e3, 3), "steps": steps,
           "mode": "reference loop on vaegan_amd nn.Modules + Adam (autograd, eager)", "binding": ops_binding()}
    if graph:
        # the same function behind vaegan_amd.graphed(): captured once (forward, autograd backward, zero_grad / step, the
        # randn_like draws), then replayed -- INTEGRATION.md section 1's fourth line
        gstep = V.graphed(step, modules=(encoder, decoder, discriminator), optimizers=(opt_E, opt_Dec, opt_Dis))
        dtg = time_steps(gstep, 4, 2 * steps)
        out["graphed"] = {"value": round(B / dtg, 1), "ms_per_step": round(dtg * 1e3, 3), "steps": 2 * steps,
                          "mode": "the same step function wrapped in vaegan_amd.graphed(...): hipGraph replay"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # 50 x 3.2 ms: long enough to average over clock / neighbour transients
    ap.add_argument("--inject-noise", type=int, default=0,
                    help="1: feed the three randn draws of the iteration as resident tensors (parity-test mode); "
                         "0 (default): the iteration draws them on the device every step, as the reference does")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--dtype", default=os.environ.get("VAEGAN_BENCH_DTYPE", "bf16"), choices=["fp32", "bf16", "fp8"],
                    help="bf16 = BASELINE configs[1] (bf16 storage, f32 accumulate, fp32 master weights); fp32 = parity path; "
                         "fp8 = BASELINE configs[4] (e4m3 operands for the forward GEMMs of the wide conv layers on the "
                         "block-scaled MFMA, everything else as bf16; quoted at --size 256 --batch 32; roofline run, no parity claim)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-paths", action="store_true",
                    help="skip the secondary legs (fp32 parity path, drop-in autograd path) reported beside the main number")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("VAEGAN_BENCH_GRAPH", "1")),
                    help="1: replay the iteration from captured hipGraph(s); with N > 1 the graph is cut at the all-reduces")
    ap.add_argument("--elide-dead-grads", action="store_true",
                    help="skip the D weight gradients of the generator-loss pass that the reference computes and discards")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: keep a private handle on it and point fd 1 at stderr, so that
    # nothing a library prints (RCCL writes its version banner to stdout at communicator creation) can get in front
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run for --gpus > 1 (see module docstring)")
    ndev = torch.cuda.device_count()
    local = local % max(ndev, 1)            # rehearsals put several ranks on one card (gloo); the driver uses 1 rank/GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # VAEGAN_FORCE_DIST=1: take the N > 1 code path (process group, reducer, segmented graphs) with a single rank --
    # the only way to run the RCCL calls themselves on a one-GPU box
    multi = world > 1 or os.environ.get("VAEGAN_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("VAEGAN_DIST_BACKEND", "nccl")      # "nccl" == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import vaegan_amd as V
    from importlib import import_module
    ops = import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.ops")
    ddp = import_module("vae-gan-based-model-for-image-generation-and-denoising_amd.ddp")

    S, B = args.size, args.batch
    e, g, d = build_models(V, S, args.dtype, dev)
    oE, oG, oD = (V.Adam(m.parameters(), lr=2e-4) for m in (e, g, d))
    reducer = None
    if multi:
        reducer = ddp.GradReducer()
        reducer.attach(oE, oG, oD)
        reducer.broadcast_parameters(oE, oG, oD)
    tr = V.VAEGANTrainer(e, g, d, oE, oG, oD, elide_dead_grads=args.elide_dead_grads, reducer=reducer)
    tr.train()
    real, ez, er, ec = (t.to(dev) for t in make_inputs(B, S, 1234 + rank))
    resident = (real, ez, er, ec)
    if not args.inject_noise:
        ez = er = ec = None             # vaegan_code.py:77,91,92: randn_like on the device, inside the timed step
        torch.cuda.manual_seed(4242 + rank)     # identical weights on every rank (seed 42), different noise streams
    epoch = 60

    use_graph = bool(args.graph)            # N > 1: segmented graphs, collectives launched between the segments
    step_fn = tr.train_step_graphed if use_graph else tr.train_step
    n_warm = max(args.warmup, 2 if use_graph else 0)
    done = 0
    try:
        for done in range(n_warm):
            step_fn(real, epoch, ez, er, ec)
        done = n_warm
    except Exception as ex:                 # noqa: BLE001 -- any capture failure: keep the measurement alive
        if not use_graph:
            raise
        # A failed capture has executed nothing (VAEGANTrainer restores its state); carry on with eager launches.
        # The collectives per iteration are the same in both modes, so ranks may even differ in mode.
        print(f"[bench] rank {rank}: hipGraph capture failed ({ex!r}); falling back to eager launches", file=sys.stderr)
        use_graph, step_fn = False, tr.train_step
        for _ in range(done, n_warm):
            step_fn(real, epoch, ez, er, ec)
    if use_graph and tr.graph_input() is not None:
        # zero-copy hand-off, as data.DeviceLoader.bind_output does it: the synthetic batch lives in the buffer the
        # captured iteration reads, so the replay has no input copy (inputs resident in HBM, bench contract)
        gi = tr.graph_input()
        gi.copy_(real)
        real = gi
    timer = ops.KernelTimer() if (rank == 0 and not use_graph) else None
    ops.set_timer(timer)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    n_launch_timed = ops.launch_count()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses = step_fn(real, epoch, ez, er, ec)
    torch.cuda.synchronize()
    n_launch_timed = ops.launch_count() - n_launch_timed
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.set_timer(None)
    if multi:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if args.graph:
        # graph replay: per-launch HIP events cannot be recorded inside captured graphs, so the kernel families are
        # timed in an extra EAGER pass of the same steps after the timed region (every rank runs it: it contains
        # the collectives; only rank 0 records).  Keyed on the REQUESTED mode so that a rank that fell back to
        # eager launches still takes part in these collectives (its timed-loop recording is simply replaced).
        timer = ops.KernelTimer() if rank == 0 else None
        ops.set_timer(timer)
        n_launch0 = ops.launch_count()
        for _ in range(args.steps):
            tr.train_step(real, epoch, ez, er, ec)
        torch.cuda.synchronize()
        ops.set_timer(None)
        launches_per_step = (ops.launch_count() - n_launch0) / args.steps
    else:
        launches_per_step = n_launch_timed / args.steps      # eager launches: the timed region itself
    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    ms = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    ld = tr.loss_dict(losses, epoch)
    assert all(v == v and abs(v) < 1e6 for v in ld.values()), f"non-finite losses: {ld}"

    # ---- roofline of the dominant kernel family (implicit-GEMM conv fprop+dgrad), measured live ----
    fam = timer.summary()
    gg = fam.get("gather_gemm", dict(launches=0, ms=1e-9, flops=0, bytes=0))
    wg = fam.get("wgrad", dict(launches=0, ms=1e-9, flops=0, bytes=0))
    ed = fam.get("edge", dict(launches=0, ms=0.0, flops=0, bytes=0))
    bn = fam.get("bn", dict(launches=0, ms=0.0, flops=0, bytes=0))
    ach = gg["flops"] / (gg["ms"] * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic(args.dtype)
    roofline = {"bound": "mfma", "kernel": "gg_kernel (gather-GEMM: conv/convT/linear fprop + dgrad)",
                "achieved": round(ach, 2), "peak": PEAK["bf16" if args.dtype == "fp8" else args.dtype], "unit": "TFLOP/s",
                "frac": round(ach / PEAK["bf16" if args.dtype == "fp8" else args.dtype], 4),
                # HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/), not re-measured here
                "traffic": traffic if (S, B) == (64, 128) else None,
                "traffic_source": (f"{traffic_src}: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this "
                                   f"command (FETCH_SIZE doubled, MI355X_MICROARCH.md section HBM), measured offline"
                                   if (S, B) == (64, 128) and traffic_src else None),
                "timed_in": ("eager pass after the timed region: HIP start/stop events around every launch on its stream "
                             "(a replayed hipGraph cannot carry per-launch events)" if args.graph else
                             "the timed region itself: HIP start/stop events around every launch on its stream"),
                "launches_per_step": gg["launches"] // args.steps,
                "avg_launch_us": round(gg["ms"] * 1e3 / max(gg["launches"], 1), 2),
                "alg_gflop_per_launch": round(gg["flops"] / max(gg["launches"], 1) / 1e9, 3),
                "share_of_step": round(gg["ms"] / args.steps / ms, 3),
                "alg_bytes_gbs": round(gg["bytes"] / (gg["ms"] * 1e-3) / 1e9, 1),
                "wgrad": {"achieved": round(wg["flops"] / (wg["ms"] * 1e-3) / 1e12, 2),
                          "launches_per_step": wg["launches"] // args.steps,
                          "share_of_step": round(wg["ms"] / args.steps / ms, 3)},
                "step_alg_tflops": round(ALG_GFLOP_PER_IMAGE.get(S, 0) * B / ms, 2)}
    busy, busy_all, busy_src = pmc_mfma_busy(args.dtype)
    if busy is not None and (S, B) == (64, 128):
        # north_star's own wording of the target ("MFMA utilisation ... from rocprof"): matrix-pipe busy cycles over
        # SIMD cycles, from a separate rocprofv3 PMC pass of this command (offline)
        roofline["mfma_busy_frac"] = round(busy, 4)
        roofline["mfma_busy_frac_whole_step"] = round(busy_all, 4)
        roofline["mfma_busy_source"] = f"{busy_src}: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass of bench.py --graph 0, measured offline"
    # the 3-channel image-side layers (<= 20 FLOP/B, priced against HBM): own kernels, own roofline
    roofline_edge = None
    if ed["launches"]:
        gbs = ed["bytes"] / (ed["ms"] * 1e-3) / 1e9
        etraffic, esrc = pmc_traffic(args.dtype, "edge")
        roofline_edge = {"bound": "hbm", "kernel": "edge layers: tnconv_kernel (narrow-N transposed conv, GEMM + col2im), "
                                                   "ggn_kernel (narrow-K direct conv)",
                         "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "traffic": etraffic if (S, B) == (64, 128) else None,
                         "traffic_source": (f"{esrc}: separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (FETCH_SIZE "
                                            f"doubled), measured offline" if (S, B) == (64, 128) and esrc else None),
                         "launches_per_step": ed["launches"] // args.steps,
                         "avg_launch_us": round(ed["ms"] * 1e3 / ed["launches"], 2),
                         "alg_mbytes_per_launch": round(ed["bytes"] / ed["launches"] / 1e6, 2),
                         "share_of_step": round(ed["ms"] / args.steps / ms, 3)}
    out = {"metric": "images/sec/GPU VAE-GAN train step", "value": round(value, 1), "unit": "images/sec",
           "per_gpu": round(value / world, 1), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": {"fp32": "f32", "bf16": "bf16", "fp8": "fp8(e4m3 fprop)+bf16"}[args.dtype], "data": "synthetic",
           "config": {"workload": f"CelebA-shaped {S}x{S} VAE-GAN full train step (E+G+5xD fwd, all bwd, 4 Adam), "
                                  f"batch {B}/GPU, global batch {B * world}", "img_size": S, "per_gpu_batch": B,
                      "global_batch": B * world, "parallelism": f"dp{world}", "epoch_kl_weight": 0.1,
                      "elide_dead_grads": bool(args.elide_dead_grads), "hip_graph": use_graph,
                      # graphs one iteration is replayed from: 1 = the whole iteration incl. its RCCL collectives (N > 1:
                      # captured inside the graph); > 1 = cut at the hand-offs to the reducer (gloo rehearsals, fallback)
                      "hip_graph_segments": (len(tr._graph[1]) if (use_graph and tr._graph is not None) else 0)},
           # kernel launches of libvaegan_hip.so per iteration (vg_launch_count over the eager pass that follows the timed
           # region; the replayed graph holds the same launches).  Round 3: 193 by rocprofv3.
           "kernel_launches_per_step": launches_per_step,
           "losses": {k: round(v, 5) for k, v in ld.items()},
           "roofline": roofline}
    if roofline_edge is not None:
        out["roofline_edge"] = roofline_edge
    if bn["launches"]:
        # the second-largest consumer of the step: every launch of csrc/bn_act.hip (BatchNorm finalize, normalise +
        # activation, backward reduce + apply, activation backward, bias-gradient column sums) against HBM.  Algorithmic
        # bytes: 2 |Y| per forward pass of a BatchNorm layer, 5 |Y| per backward pass (ops._bn_bytes)
        gbs = bn["bytes"] / (bn["ms"] * 1e-3) / 1e9
        out["roofline_bn"] = {"bound": "hbm", "kernel": "bn_act.hip: BatchNorm finalize / normalise + activation / backward reduce + apply, "
                                                        "activation backward, bias-gradient sums",
                              "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                              "traffic": None, "launches_per_step": bn["launches"] // args.steps,
                              "avg_launch_us": round(bn["ms"] * 1e3 / bn["launches"], 2),
                              "alg_mbytes_per_step": round(bn["bytes"] / args.steps / 1e6, 1),
                              "us_per_step": round(bn["ms"] * 1e3 / args.steps, 1),
                              "share_of_step": round(bn["ms"] / args.steps / ms, 3)}
    f8 = fam.get("gather_gemm_fp8", dict(launches=0, ms=0.0, flops=0, bytes=0))
    if f8["launches"]:
        a8 = f8["flops"] / (f8["ms"] * 1e-3) / 1e12
        out["roofline_fp8"] = {"bound": "mfma", "kernel": "gg_kernel<fp8> (conv / convT fprop on v_mfma_scale_f32_16x16x128_f8f6f4, e4m3)",
                               "achieved": round(a8, 2), "peak": PEAK["fp8"], "unit": "TFLOP/s", "frac": round(a8 / PEAK["fp8"], 4),
                               "traffic": None, "launches_per_step": f8["launches"] // args.steps,
                               "avg_launch_us": round(f8["ms"] * 1e3 / f8["launches"], 2),
                               "share_of_step": round(f8["ms"] / args.steps / ms, 3)}
    if world == 1 and not multi and not args.no_extra_paths:
        del tr, e, g, d, oE, oG, oD
        out["denoise_path"] = denoise_path(V, S, B, dev, args.dtype, resident)
        out["parity_path"] = parity_path(V, S, B, dev, resident, ops)
        out["elided_path"] = elided_path(V, S, B, dev, args.dtype, resident)
        out["dropin_path"] = dropin_path(V, S, B, dev, args.dtype, resident)
        try:                                    # the same loop with the hot ops bound through torch.ops.vaegan.*
            ops.set_binding("torchops")
            alt = dropin_path(V, S, B, dev, args.dtype, resident, graph=False)
            out["dropin_path"]["torchops_binding"] = {"value": alt["value"], "ms_per_step": alt["ms_per_step"]}
        except RuntimeError as ex:
            out["dropin_path"]["torchops_binding"] = {"error": str(ex)[:200]}
        finally:
            ops.set_binding("ctypes")
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(S, B)
    print(json.dumps(out), file=result_out, flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
