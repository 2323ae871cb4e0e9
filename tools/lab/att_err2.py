import os, sys
sys.path.insert(0, os.getcwd())
import torch
from avcer_amd.engine import Engine
from avcer_amd.sp32 import from_sp32
eng = Engine(0); dev = eng.device
g = torch.Generator().manual_seed(0)
n, s, heads, d = 2, 99, 16, 64
e = heads * d
def run(qkv, label):
    q, k, v = (qkv[..., i * e:(i + 1) * e].double().view(n, s, heads, d).transpose(1, 2) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(n, s, e)
    qd = qkv.to(dev)
    o32 = torch.empty(n, s, e, device=dev); eng.attention(qd, o32, n, s, heads, d, 0.125, 0, 0)
    osp = torch.empty(n, s, 2 * e, dtype=torch.int16, device=dev); eng.attention(qd, osp, n, s, heads, d, 0.125, 0, 2)
    torch.cuda.synchronize()
    r = ref.pow(2).mean().sqrt()
    d32 = (o32.cpu().double() - ref); dx = (from_sp32(osp.cpu()).double() - ref)
    print(f"{label:34s} f32 VALU rms {d32.pow(2).mean().sqrt()/r:.2e} max {d32.abs().max():.2e}   x3 MFMA rms {dx.pow(2).mean().sqrt()/r:.2e} max {dx.abs().max():.2e}  (ref rms {r:.3f})")
    return dx
base = torch.randn(n, s, 3 * e, generator=g)
run(base, "random q k v")
t = base.clone(); t[..., :e] = 0
run(t, "q = 0 (uniform p = 1)")
t = base.clone(); t[..., 2 * e:] = 1.0
run(t, "v = 1 (out must be exactly 1)")
t = base.clone(); t[..., 2 * e:] = torch.round(t[..., 2 * e:] * 4) / 4
run(t, "v on a 1/4 grid (exact in f16)")
t = base.clone(); t[..., :2 * e] = torch.round(t[..., :2 * e] * 8) / 8
dx = run(t, "q, k on a 1/8 grid (scores exact)")
t = base.clone(); t[..., :2 * e] = torch.round(t[..., :2 * e] * 8) / 8; t[..., 2 * e:] = torch.round(t[..., 2 * e:] * 4) / 4
run(t, "both on grids")
