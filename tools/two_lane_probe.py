#!/usr/bin/env python3
"""Would two half-batches on two HIP streams fill the partial rounds of config 2 (256 frames)?  (experiment, GPU)

The static CNN at 256 frames runs its stage-3 / stage-4 grids at 0.44-0.88 of one round of block slots; the same kernels on
two independent halves of the batch, queued on two streams, let the tail of one launch overlap the head of another.  This probe
measures that with what exists: two contexts (each its own workspace), 2 x n/2 frames on two streams, against one call of n.

    python tools/two_lane_probe.py [n ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, Engine  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 2048]
    sd = synth.static_state_dict(42)
    lanes = 4
    engs = [Engine(0) for _ in range(lanes)]
    for e in engs:
        e.load_static(sd)
    streams = [torch.cuda.Stream() for _ in range(lanes)]
    for n in sizes:
        frames = torch.from_numpy(synth.face_frames(1, n)).cuda()
        ref = engs[0].static_forward(frames, MODE_F16X3)
        torch.cuda.synchronize()

        def one():
            engs[0].static_forward(frames, MODE_F16X3)

        def split(k):
            def run():
                ev = torch.cuda.Event()
                ev.record()
                per = n // k
                outs = []
                for i in range(k):
                    with torch.cuda.stream(streams[i]):
                        streams[i].wait_event(ev)
                        outs.append(engs[i].static_forward(frames[i * per:(i + 1) * per], MODE_F16X3))
                for s in streams[:k]:
                    torch.cuda.current_stream().wait_stream(s)
                return outs
            return run

        def timeit(fn, reps=20):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3

        outs = split(2)()
        torch.cuda.synchronize()
        same = all(torch.equal(torch.cat([o[j] for o in outs]), ref[j]) for j in range(3))
        print(f"n={n:5d}  one call {timeit(one):7.3f} ms   2 lanes {timeit(split(2)):7.3f} ms   4 lanes {timeit(split(4)):7.3f} ms   "
              f"bit-identical {same}", flush=True)


if __name__ == "__main__":
    main()
