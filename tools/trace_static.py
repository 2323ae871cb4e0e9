#!/usr/bin/env python3
"""The static CNN at a given batch under `rocprofv3 --kernel-trace`: every launch of the LAST forward pass with its duration
and grid -- where a small batch (BASELINE configs[1]: 256 frames) loses against the 2048-frame pass.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ts -o t -- python3 tools/trace_static.py run 256
    python3 tools/trace_static.py show gpurun_out/ts
    ... run 1 audio | run 1 lstm; show <dir> wav_normalize | show <dir> gather_windows  (first kernel of the pass as the marker)
"""
import csv
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(batch, what="static"):
    import torch
    from avcer_amd import synth
    from avcer_amd.engine import MODE_F16X3, Engine

    eng = Engine(0)
    eng.set_static_lanes(1)  # one stream: the trace lists the launches of a call in order
    if what == "audio":   # `batch` windows of 4 s
        eng.load_audio(synth.audio_state_dict(42))
        wav = torch.from_numpy(synth.waveforms(5678, batch, 64000)).cuda()
        call = lambda: eng.audio_forward(wav, True, MODE_F16X3)
    elif what == "lstm":  # `batch` windows of 10 feature rows
        eng.load_dynamic(synth.dynamic_state_dict(42))
        win = torch.randn(batch, 10, 512, device="cuda")
        call = lambda: eng.dynamic_forward(win, MODE_F16X3)
    else:
        eng.load_static(synth.static_state_dict(42))
        frames = torch.from_numpy(synth.face_frames(1, batch)).cuda()
        call = lambda: eng.static_forward(frames, MODE_F16X3)
    for _ in range(4):
        call()
        torch.cuda.synchronize()


def show(d, marker="stem_pool"):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if not starts:  # no marker kernel: the passes are equally long, show the last quarter (run() makes four)
        starts = [len(rows) - len(rows) // 4]
    first = starts[-1]
    t0, total = int(rows[first]["Start_Timestamp"]), 0.0
    for r in rows[first:]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
        n = re.sub(r"\(.*", "", n)
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        total += dur
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "256")) or 256))
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {dur:8.1f} us  blocks {grid:6d}  {n[:90]}")
    print(f"sum of kernel durations: {total / 1e3:.3f} ms; first launch to last end: {(int(rows[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 256, sys.argv[3] if len(sys.argv) > 3 else "static")
    else:
        show(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "stem_pool")
