#!/usr/bin/env python3
"""`configs.run_inference` (one 30 s video through avcer_amd/run.py) under `rocprofv3 --kernel-trace`: where the GPU is busy
and where it waits for the host -- per HIP queue the span and the summed kernel time of the LAST call, the gaps longer than
100 us, and the kernels that own the time.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ri -o t -- python3 tools/trace_run_inference.py run [det]
    python3 tools/trace_run_inference.py show gpurun_out/ri
(`det`: with the RetinaFace-R50 detector in front -- `with_detector` of the bench line -- instead of scripted detections)
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MARK = "avcer_mark"  # a tiny torch kernel before the last call marks its start in the trace


def run():
    import time

    import torch
    from avcer_amd import run as arun
    from avcer_amd import synth
    from avcer_amd.engine import MODE_F16X3
    from avcer_amd.pipeline import AVPipeline

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    seconds, fps, h, w = 30, 25, 360, 640
    n = seconds * fps
    pipe = AVPipeline(device=0, seed=42, mode=MODE_F16X3)
    frames = torch.from_numpy(synth.video_frames(77, n, h, w)).cuda()
    dets = bench.scripted_detections(n, h, w)
    wav = torch.from_numpy(synth.waveforms(78, 1, seconds * 16000)[0]).cuda()
    kw = {"detections": dets}
    if "det" in sys.argv:
        from avcer_amd.face_tiles import RetinaFacePredictor

        real = RetinaFacePredictor(pipe.engine, synth.to_torch(synth.retina_state_dict(42)), mode=MODE_F16X3)

        class DetectorThenScript:  # the network, decode and NMS run; the scripted track is used behind them (synthetic weights find no faces)
            def batch(self, fr, rgb=False):
                real.batch(fr, rgb=rgb)
                return dets

        kw = {"detector": DetectorThenScript()}
    for i in range(4):
        torch.cuda.synchronize()
        if i == 3:
            torch.zeros(7, device="cuda").cumsum(0)  # marker launch: the last call starts behind it
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        arun.run_inference(pipe.engine, frames, wav, fps, mode=MODE_F16X3, **kw)
        torch.cuda.synchronize()
        print(f"call {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)


def show(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "cumsum" in r["Kernel_Name"].lower() or "scan" in r["Kernel_Name"].lower()]
    first = marks[-1] + 1 if marks else 0
    rows = rows[first:]
    t0 = int(rows[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in rows)
    print(f"last call: {len(rows)} launches, first start -> last end {(t1 - t0) / 1e6:.2f} ms")
    by_q = defaultdict(list)
    for r in rows:
        by_q[(r["Queue_Id"], r.get("Stream_Id", "?"))].append(r)
    for q, rs in sorted(by_q.items()):
        s, e = int(rs[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rs)
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
        print(f"  queue {q[0]} stream {q[1]}: {len(rs):4d} launches, {(s - t0) / 1e6:7.2f} -> {(e - t0) / 1e6:7.2f} ms, kernel time {busy / 1e6:6.2f} ms")
    # moments when NO kernel of any queue runs
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    idle, cur_end, gaps = 0, ev[0][1], []
    for s, e in ev[1:]:
        if s > cur_end:
            idle += s - cur_end
            if s - cur_end > 100_000:
                gaps.append((cur_end - t0, s - cur_end))
        cur_end = max(cur_end, e)
    print(f"  GPU idle inside the call (no kernel on any queue): {idle / 1e6:.2f} ms; gaps > 100 us:")
    for at, g in gaps:
        print(f"    at {at / 1e6:7.2f} ms: {g / 1e3:8.1f} us")
    tot = defaultdict(lambda: [0, 0])
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        tot[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        tot[n][1] += 1
    print("  kernel time by name:")
    for n, (ns, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"    {ns / 1e6:7.2f} ms  {c:4d} x  {n[:110]}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
