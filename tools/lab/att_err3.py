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
v = qkv[..., 2 * e:].double()
ref = v.mean(1, keepdim=True).expand(n, s, e)
osp = torch.empty(n, s, 2 * e, dtype=torch.int16, device=dev)
eng.attention(qkv.to(dev), osp, n, s, heads, d, 0.125, 0, 2)
torch.cuda.synchronize()
got = from_sp32(osp.cpu()).double()
err = (got - ref).abs()[0]           # [s, e]
print("per-query max err (first 20):", [f"{x:.1e}" for x in err.max(1).values[:20].tolist()])
print("per-channel-in-head max err:", [f"{x:.1e}" for x in err.view(s, heads, d).amax((0, 1)).tolist()])
print("per-head max err:", [f"{x:.1e}" for x in err.view(s, heads, d).amax((0, 2)).tolist()])
# is the result what you get if V lo is dropped for keys >= some index?
vh = v.float().half().double()
for lim in (99, 96, 64, 32, 0):
    alt = (torch.cat([v[:, :lim], vh[:, lim:]], 1)).mean(1, keepdim=True)
    print(f"keys >= {lim} without lo: max|got - alt| {(got - alt).abs().max():.2e}")
