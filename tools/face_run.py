#!/usr/bin/env python3
"""Stage 0 of the reference's real run (get_face_images.py:38-63): RetinaFace-R50 over the 750 frames of a 30 s 640 x 360
video -- network, box decoding, device NMS, rows back on the host -- as `avcer_amd.face_tiles.RetinaFacePredictor.batch` runs it.

    python3 tools/face_run.py [passes=3] [fam] [serial]   wall time per pass; `fam`: per kernel family from HIP events (serial by
                                                            itself); `serial`: avcer_set_static_lanes(1) -- one stream, for traces
    rocprofv3 --kernel-trace --stats --output-format csv -d out -o face -- python3 tools/face_run.py 3 serial
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/face_run.py 1 serial   (then WRITE_SIZE; tools/pmc_table.py)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, Engine  # noqa: E402
from avcer_amd.face_tiles import RetinaFacePredictor  # noqa: E402

GFLOP_FRAME = 51.5  # conv_gemm FLOPs of one 640 x 360 frame (avcer_gemm_stats)

if __name__ == "__main__":
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    fam = "fam" in sys.argv
    n, h, w = 750, 360, 640
    eng = Engine(0)
    if "serial" in sys.argv:
        eng.set_static_lanes(1)  # the detector's batches on one stream: per-kernel durations of a trace mean something
    det = RetinaFacePredictor(eng, synth.to_torch(synth.retina_state_dict(42)), mode=MODE_F16X3)
    frames = torch.from_numpy(synth.video_frames(77, n, h, w)).cuda()
    det.batch(frames)  # warm-up: weight split, workspace
    torch.cuda.synchronize()
    if fam:
        eng.gemm_stats(reset=True)
        eng.profile_enable(True)
    dts = []
    for _ in range(passes):
        t0 = time.perf_counter()
        found = det.batch(frames)
        torch.cuda.synchronize()
        dts.append((time.perf_counter() - t0) * 1e3)
    how = "one lane" if ("serial" in sys.argv or fam) else "two lanes (the default)"
    print(f"detector over {n} frames of {w}x{h}, {how}: " + ", ".join(f"{d:.1f}" for d in dts) + f" ms per pass ({sum(len(d) for d in found)} boxes kept)")
    if fam:
        fams = eng.profile_read_families()
        eng.profile_enable(False)
        launches, flops = eng.gemm_stats(reset=True)
        tot = 0.0
        print(f"{'family':28s} {'ms/pass':>9s} {'launches':>9s} {'TFLOP/s':>9s} {'of 2500':>8s} {'GB/s compulsory':>16s}")
        for k, (ms, la, fl, by) in fams.items():
            if not la:
                continue
            tot += ms
            print(f"{k:28s} {ms / passes:9.2f} {la // passes:9d} {fl / ms / 1e9:9.1f} {fl / ms / 1e9 / 2500:8.3f} {by / ms / 1e6:16.0f}")
        wall = sorted(dts)[len(dts) // 2]
        print(f"MFMA kernels {tot / passes:.1f} ms of {wall:.1f} ms per pass; {flops / passes / n / 1e9:.1f} GFLOP per frame; "
              f"{flops / passes / (tot / passes) / 1e9:.0f} TFLOP/s inside them = {flops / tot / 1e9 / 2500:.3f} of the 16-bit MFMA peak "
              f"({flops / tot / 1e9 * 3 / 2500:.3f} executed); everything else (pre, max-pool, upsample, heads, decode, NMS, D2H) "
              f"{wall - tot / passes:.1f} ms")
