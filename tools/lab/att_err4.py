import os, sys
sys.path.insert(0, os.getcwd())
import torch
from avcer_amd.engine import Engine
from avcer_amd.sp32 import from_sp32
eng = Engine(0); dev = eng.device
g = torch.Generator().manual_seed(0)
n, s, heads, d = 1, 99, 16, 64
e = heads * d
qkv = torch.randn(n, s, 3 * e, generator=g)
qkv[..., :e] = 0
osp = torch.empty(n, s, 2 * e, dtype=torch.int16, device=dev)
eng.attention(qkv.to(dev), osp, n, s, heads, d, 0.125, 0, 2)
torch.cuda.synchronize()
raw = osp.cpu()
got = from_sp32(raw).double()
v = qkv[0, :, 2 * e:]
for h, c in ((15, 33), (0, 23), (3, 5)):
    col = v[:, h * d + c].double()
    hi = v[:, h * d + c].half().double(); lo = (v[:, h * d + c] - v[:, h * d + c].half().float()).half().double()
    gi = h * d + c
    grp, w = gi // 32, gi % 32
    rh = raw[0, 0, grp * 64 + w].view(torch.float16) if False else raw[0, 0, grp * 64 + w: grp * 64 + w + 1].view(torch.float16).double().item()
    rl = raw[0, 0, grp * 64 + 32 + w: grp * 64 + 32 + w + 1].view(torch.float16).double().item()
    print(f"head {h} ch {c}: ref {col.mean():.9f}  got {got[0,0,gi]:.9f} (hi {rh:.9f} lo {rl:.3e})  sum(hi)/99 {hi.mean():.9f}  (hi+lo)/99 {(hi+lo).mean():.9f}  f16(ref) {float(torch.tensor(col.mean()).half()):.9f}")
