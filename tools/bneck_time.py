import sys, os, torch
sys.path.insert(0, os.getcwd())
from avcer_amd.engine import Engine
if len(sys.argv) > 1:  # a one-off experiment build of the library
    from avcer_amd import _lib  # AVCER_LIB_OVERRIDE (argument, not environment)
    _lib.LIB = os.path.abspath(sys.argv[1])
eng = Engine(0)
dev = eng.device
def run(planes, nb, hw, nxt, iters=5, frags=False):
    p4 = 4 * planes
    M = nb * hw * hw
    t1 = torch.randint(-3000, 3000, (M, 2 * planes), dtype=torch.int16, device=dev)
    x = torch.randint(-3000, 3000, (M, 2 * p4), dtype=torch.int16, device=dev)
    out = torch.empty((M, 2 * p4), dtype=torch.int16, device=dev)
    t1n = torch.empty((M, 2 * planes), dtype=torch.int16, device=dev) if nxt else None
    w2f32 = torch.randn(planes, 9 * planes, device=dev) * 0.05
    w2 = eng.split_weight_rows(w2f32)
    kw = {"w2_frags": eng.weight_frags(w2f32)} if frags else {}
    w3 = eng.split_weight_rows(torch.randn(p4, planes, device=dev) * 0.1)
    w1 = eng.split_weight_rows(torch.randn(planes, p4, device=dev) * 0.05) if nxt else None
    b2, b3, b1 = torch.zeros(planes, device=dev), torch.zeros(p4, device=dev), (torch.zeros(planes, device=dev) if nxt else None)
    for _ in range(2):
        eng.bneck_chain(planes, nb, hw, hw, t1, x, out, t1n, w2, b2, w3, b3, w1, b1, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        eng.bneck_chain(planes, nb, hw, hw, t1, x, out, t1n, w2, b2, w3, b3, w1, b1, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"bneck planes={planes} next={nxt}{' spatial-tile form' if frags else ''}: {e0.elapsed_time(e1)/iters*1e3:8.1f} us")
run(64, 1024, 55, True); run(64, 1024, 55, False); run(128, 1024, 28, True); run(128, 1024, 28, False)
if "w2_frags" in eng.bneck_chain.__code__.co_varnames:
    for _ in range(2):
        run(64, 1024, 55, True, frags=True); run(64, 1024, 55, True)


def run_stem(n=1024, iters=5):
    planes = torch.randint(-3000, 3000, (2, n, 230, 230, 4), dtype=torch.int16, device=dev)
    w = eng.split_weight_rows(torch.randn(64, 224, device=dev) * 0.05)
    sc, bi = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    for _ in range(2):
        eng.stem_pool(planes, w, sc, bi, n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        eng.stem_pool(planes, w, sc, bi, n)
    e1.record(); torch.cuda.synchronize()
    print(f"stem_pool {n} frames: {e0.elapsed_time(e1)/iters*1e3:8.1f} us")


run_stem()
