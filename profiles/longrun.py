"""bf16 vs f32 engine over a few hundred iterations on the same synthetic image set (smooth blobs), same seeds."""
import sys, os, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import torch
import vaegan_amd as V

def make_set(n, S, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, S), torch.linspace(-1, 1, S), indexing="ij")
    imgs = []
    for i in range(n):
        c = torch.rand(3, 2, generator=g) * 1.4 - 0.7
        r = torch.rand(3, generator=g) * 0.5 + 0.2
        col = torch.rand(3, 3, generator=g) * 2 - 1
        img = torch.zeros(3, S, S)
        for k in range(3):
            blob = torch.exp(-((xx - c[k, 0]) ** 2 + (yy - c[k, 1]) ** 2) / (2 * r[k] ** 2))
            img += col[k].view(3, 1, 1) * blob
        imgs.append(img.clamp(-1, 1))
    return torch.stack(imgs)

def run(dtype, steps, S=64, B=64):
    V.configure_seed(7)
    dev = torch.device("cuda", 0)
    e = V.Encoder([3, S, S], 100, dtype=dtype); g = V.Generator(nz=100, img_size=S, dtype=dtype); d = V.Discriminator(img_size=S, dtype=dtype)
    g.apply(V.weights_init), d.apply(V.weights_init)
    e.to(dev), g.to(dev), d.to(dev)
    oE, oG, oD = (V.Adam(m.parameters(), lr=2e-4) for m in (e, g, d))
    tr = V.VAEGANTrainer(e, g, d, oE, oG, oD); tr.train()
    data = make_set(512, S, 99).to(dev)
    torch.cuda.manual_seed(1234)
    log = []
    acc = torch.zeros(5, device=dev); n = 0
    for it in range(steps):
        idx = torch.arange(it * B, (it + 1) * B, device=dev) % data.shape[0]
        losses = tr.train_step_graphed(data[idx].contiguous(), 1 + it // 8)
        acc += losses[:5]; n += 1
        if (it + 1) % 50 == 0:
            v = (acc / n).tolist(); acc.zero_(); n = 0
            log.append([it + 1] + [round(x, 4) for x in v])
            if not all(math.isfinite(x) for x in v):
                break
    return log

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    out = {"columns": ["iter", "recon", "kl", "g_adv", "d1", "d2"], "bf16": run("bf16", steps), "f32": run("fp32", steps)}
    print(json.dumps(out))
