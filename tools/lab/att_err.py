import os, sys
sys.path.insert(0, os.getcwd())
import torch
from avcer_amd.engine import Engine
from avcer_amd.sp32 import from_sp32
eng = Engine(0); dev = eng.device
g = torch.Generator().manual_seed(0)
n, s, heads, d = 4, 99, 16, 64
e = heads * d
for qs in (1.0, 3.0, 8.0):
    qkv = torch.randn(n, s, 3 * e, generator=g)
    qkv[..., :e] *= qs
    q, k, v = (qkv[..., i * e:(i + 1) * e].double().view(n, s, heads, d).transpose(1, 2) for i in range(3))
    sc = q @ k.transpose(-1, -2) * 0.125
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(n, s, e)
    qd = qkv.to(dev)
    o32 = torch.empty(n, s, e, device=dev); eng.attention(qd, o32, n, s, heads, d, 0.125, 0, 0)
    osp = torch.empty(n, s, 2 * e, dtype=torch.int16, device=dev); eng.attention(qd, osp, n, s, heads, d, 0.125, 0, 2)
    torch.cuda.synchronize()
    r = ref.pow(2).mean().sqrt()
    e32 = ((o32.cpu().double() - ref).pow(2).mean().sqrt() / r).item()
    ex3 = ((from_sp32(osp.cpu()).double() - ref).pow(2).mean().sqrt() / r).item()
    print(f"q scale {qs}: |score| max {sc.abs().max():.1f}  rel rms err  f32 VALU {e32:.2e}   x3 MFMA {ex3:.2e}")
