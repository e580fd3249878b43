"""Layer-by-layer: the bf16 engine's stored forward tensors against the bf16-storage emulation (oracle/vaegan_ref_bf16.py).
GPU.  python tools/bf16_layer_diff.py [S=64] [B=128]   -> per stored tensor: fraction of elements that differ, worst
difference in units of the bf16 spacing at that magnitude, relative Frobenius distance."""
import os, sys, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import torch
import torch.nn.functional as F
import vaegan_ref as R, vaegan_ref_bf16 as RB
from _inputs import make_inputs
import vaegan_amd as V
PKG = "vae-gan-based-model-for-image-generation-and-denoising_amd"
engine = importlib.import_module(PKG + ".engine")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
torch.set_num_threads(16)
V.configure_seed(42)
e = V.Encoder([3, S, S], 100, dtype="bf16"); g = V.Generator(nz=100, img_size=S, dtype="bf16"); d = V.Discriminator(img_size=S, dtype="bf16")
g.apply(V.weights_init); d.apply(V.weights_init)
for m in (e, g, d): m.to("cuda")
tr = V.VAEGANTrainer(e, g, d, *(V.Adam(m.parameters(), lr=0.0) for m in (e, g, d))); tr.train()
tr.group_d_passes = False            # separate real / fake passes: ctx of the first pass = the real batch
inp = make_inputs(B, S, 1234)
dev = [t.to("cuda") for t in inp]
stash = {}
orig = engine.StackEngine.forward
def fwd(self, x, B_, train, keep=True, groups=1, tail=None):
    out, ctxpack = orig(self, x, B_, train, keep, groups, tail)
    name = {id(e._engine): "E", id(g._engine): "G", id(d._engine): "D"}[id(self)]
    if name not in stash and keep:
        stash[name] = ([{k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in c.items()} for c in ctxpack[0]], out.clone() if isinstance(out, torch.Tensor) else out)
    return out, ctxpack
engine.StackEngine.forward = fwd
tr.train_step(dev[0], 60, *dev[1:]); torch.cuda.synchronize()
engine.StackEngine.forward = orig

# emulation with taps
em = RB.RefVAEGANbf16(img_size=S, seed=42, lr=0.0)
taps = {}
orig_bn = RB._BNAct.forward
cnt = {"n": 0}
def nchw(t): return t.permute(0, 2, 3, 1).contiguous()
class Tap:
    pass
orig_bn_act = em._bn_act
def bn_act(st, prefix, Y, slope, net, i, stats_rounded=False):
    out = orig_bn_act(st, prefix, Y, slope, net, i, stats_rounded)
    key = f"{net}.{i}"
    if key not in taps:
        taps[key] = (RB.bf16_round(Y.detach()), out.detach())
    return out
em._bn_act = bn_act
em.train_step(*inp, 60)

def cmp(name, hip, ref):
    hip = hip.double().cpu().reshape(-1); ref = ref.double().reshape(-1)
    n = min(hip.numel(), ref.numel())
    assert hip.numel() == ref.numel(), (name, hip.shape, ref.shape)
    dlt = (hip - ref).abs()
    ulp = torch.clamp(ref.abs(), min=1e-30) * 2.0 ** -8
    print(f"{name:16s} differ {float((dlt > 0).double().mean()):8.2e}  worst {float((dlt / ulp).max()):8.2f} bf16-ulps  frobenius {float(dlt.norm() / ref.norm()):.2e}")

for net, eng in (("E", e._engine), ("G", g._engine), ("D", d._engine)):
    ctx, out = stash[net]
    for i, c in enumerate(ctx):
        key = f"{net}.{i}"
        if key in taps and c.get("Y") is not None:
            Yq, A = taps[key]
            OCp = c["Y"].shape[-1]
            cmp(key + " Y", c["Y"][..., :Yq.shape[1]], nchw(Yq))
        if key in taps and i + 1 < len(ctx) and ctx[i + 1].get("x") is not None:
            Yq, A = taps[key]
            xin = ctx[i + 1]["x"]
            if xin.dim() == 4 and xin.shape[-1] >= A.shape[1]:
                cmp(key + " A", xin.reshape(A.shape[0], A.shape[2], A.shape[3], -1)[..., :A.shape[1]], nchw(A))
