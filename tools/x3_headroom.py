#!/usr/bin/env python3
"""Range-contract headroom of AVCER_MODE_F16X3, tensor by tensor (round-5 review item 3).

Every sp32 activation tensor of the three models is read back from the GPU through the library's debug taps (the tensors the
kernels themselves wrote: hi / lo fp16 pairs) on the synthetic checkpoints, and two numbers are printed per tensor:
  max|x| / 65504      how close the tensor comes to fp16's range (the contract: < 1; an overflow is NaN + a counted event)
  below 2^-3          the share of its NON-ZERO elements whose lo half is an fp16 subnormal: these carry an absolute error of
                      <= 2^-25 instead of the 2^-22 relative one (exact zeros -- ReLU -- carry none and are listed apart)
The logit scale of tests/test_gpu_parity_breadth.py (1, 4, 8) multiplies the LAST Linear of each model only: its output is the
f32 logits, no sp32 tensor changes with it -- run with --scale 8 to see the identical table.  What does move the headroom is
the checkpoint's activation gain; the last column says how many times larger every activation of the tensor could be.

    python tools/x3_headroom.py [--scale S] [--seed N]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from avcer_amd import synth  # noqa: E402
from avcer_amd.engine import MODE_F16X3, Engine  # noqa: E402
from avcer_amd.sp32 import from_sp32  # noqa: E402


def stats(raw_i16):
    x = from_sp32(raw_i16.reshape(-1, 64)).reshape(-1).double()
    a = x.abs()
    nz = a > 0
    n, nnz = x.numel(), int(nz.sum())
    small = int(((a < 0.125) & nz).sum())
    tiny = int(((a < 2.0 ** -14) & nz).sum())
    return dict(n=n, amax=float(a.max()), zero=1 - nnz / n, small=small / max(nnz, 1), tiny=tiny / max(nnz, 1),
                rms=float((x[nz] ** 2).mean().sqrt()) if nnz else 0.0, nonfinite=int((~torch.isfinite(x)).sum()))


def walk(eng, call, names, cap_bytes, title):
    print(f"\n{title}")
    print(f"{'tensor (debug tap)':26s} {'elements':>10s} {'max|x|':>10s} {'/65504':>9s} {'x to limit':>10s} {'rms(nz)':>9s} {'zeros':>7s} {'nz<2^-3':>8s} {'nz<2^-14':>9s}")
    worst = (0.0, None)
    for name in names:
        dst = eng.debug_tap(name, cap_bytes // 2, dtype=torch.int16)
        call()
        torch.cuda.synchronize()
        nb = eng.debug_tap_copied()
        if nb <= 0:
            continue
        s = stats(dst[: nb // 2].cpu())
        if s["amax"] / 65504 > worst[0]:
            worst = (s["amax"] / 65504, name)
        print(f"{name:26s} {s['n']:10d} {s['amax']:10.3f} {s['amax'] / 65504:9.2e} {65504 / max(s['amax'], 1e-30):10.0f} "
              f"{s['rms']:9.3f} {s['zero']:7.1%} {s['small']:8.1%} {s['tiny']:9.2%}" + ("  NON-FINITE" if s["nonfinite"] else ""))
    print(f"closest to the limit: {worst[1]} at {worst[0]:.2e} of 65504 ({1 / worst[0]:.0f} x headroom)")
    return worst


if __name__ == "__main__":
    scale = float(sys.argv[sys.argv.index("--scale") + 1]) if "--scale" in sys.argv else 1.0
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 42
    eng = Engine(0)

    def scaled(sd, keys):
        sd = dict(sd)
        for k in keys:
            sd[k] = sd[k] * scale
        return sd

    eng.load_static(scaled(synth.static_state_dict(seed), ("fc2.weight", "fc2.bias")))
    eng.load_audio(scaled(synth.audio_state_dict(seed), ("feature_downsample.weight", "feature_downsample.bias")))
    print(f"synthetic checkpoints seed {seed}, logit scale {scale:g}; x3 mode; 8 frames (static CNN), 4 windows of 2 s (audio model)")
    frames = torch.from_numpy(synth.face_frames(2468, 8)).cuda()
    names = ["stem", "out:l1.0.c1.w"]
    blocks = {1: 3, 2: 4, 3: 6, 4: 3}
    for li in (1, 2, 3, 4):
        for b in range(blocks[li]):
            p = f"l{li}.{b}."
            names += [f"out:{p}c1.w", f"out:{p}c1.wf", f"out:{p}c2.w", f"chain_t1n:{p}", f"chain_out:{p}", f"out:{p}c3d.w", f"out:{p}c3.w",
                      f"out:{p}c3.wf", f"tail_out:{p}", f"tail_t1n:{p}"]
    walk(eng, lambda: eng.static_forward(frames, MODE_F16X3), list(dict.fromkeys(names)), 8 * 112 * 112 * 64 * 4,
         "static CNN (ResNet-50): every sp32 tensor a kernel writes")
    wav = torch.from_numpy(synth.waveforms(1357, 4, 32000)).cuda()
    names = ["conv0"] + [f"ln:fe{i}.ln" for i in range(1, 7)] + ["ln:fp.ln", "pos_in"]
    for l in range(12):
        names += [f"ln:enc{l}.ln1", f"att:enc{l}", f"ln:enc{l}.ln2", f"out:enc{l}.ff1.w"]
    for l in (1, 2):
        names += [f"pe:tl{l}", f"att:tl{l}", f"ln:tl{l}.ln1", f"out:tl{l}.ff1.w", f"ln:tl{l}.ln2"]
    walk(eng, lambda: eng.audio_forward(wav, True, MODE_F16X3), names, 4 * 6399 * 512 * 4,
         "audio model (wav2vec2 + 2 TransformerLayers): every sp32 tensor (residual streams, q/k/v and the heads stay f32)")
    print("\nLSTM: its sp32 tensors are the hidden states h = o * tanh(c), |h| < 1 by construction (65504 x headroom); the windows "
          "(relu(fc1 features)) are split on the fly by the contraction that reads them and are counted by the overflow counter.")
    print(f"range-contract counter after these passes: {eng.x3_overflow_count(reset=True)}")
