#!/usr/bin/env python3
"""One pipeline step under `rocprofv3 --kernel-trace`: prints every kernel launch of the LAST step in order with its duration.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 tools/trace_step.py run
    python3 tools/trace_step.py show gpurun_out/trace
"""
import csv
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from avcer_amd import synth
    from avcer_amd.engine import MODE_F16X3
    from avcer_amd.pipeline import AVPipeline

    pipe = AVPipeline(device=0, seed=42, mode=MODE_F16X3)
    pipe.overlap_branches = False  # one stream: the trace lists the launches of a step in order
    pipe.engine.set_static_lanes(1)  # ... and the static CNN's 2048 frames on one lane
    frames = torch.from_numpy(synth.face_frames(1234, 128 * 16)).reshape(128, 16, 224, 224, 3).cuda()
    wav = torch.from_numpy(synth.waveforms(5678, 128, 32000)).cuda()
    for _ in range(3):
        pipe.run_clips(frames, wav, 25)
        torch.cuda.synchronize()


def show(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # the last step starts at the last stem launch but one (two front passes per step)
    starts = [i for i, n in enumerate(names) if "stem_pool" in n]
    first = starts[-2]
    t0 = int(rows[first]["Start_Timestamp"])
    total = 0.0
    for r in rows[first:]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        total += dur
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} us  {dur:9.1f} us  grid {r.get('Grid_Size', '?'):>9s}  {n[:100]}")
    print(f"sum of kernel durations in the step: {total / 1e3:.2f} ms")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
