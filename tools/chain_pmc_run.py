#!/usr/bin/env python3
"""A few launches of the stage-1 / stage-2 bottleneck chains for a counter pass:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY ... --output-format csv -d out -- python3 tools/chain_pmc_run.py
(python3 tools/chain_pmc_run.py show out  prints the per-kernel sums)"""
import csv
import glob
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from avcer_amd.engine import Engine

    eng = Engine(0)
    dev = eng.device
    for planes, hw, frags in ((64, 55, True), (64, 55, False), (128, 28, False)):
        nb, p4 = 512, 4 * planes
        M = nb * hw * hw
        t1 = eng.split_weights(torch.relu(torch.randn(M * planes, device=dev))).view(M, -1)
        x = eng.split_weights(torch.relu(torch.randn(M * p4, device=dev))).view(M, -1)
        out, t1n = torch.empty_like(x), torch.empty_like(t1)
        w2f = torch.randn(planes, 9 * planes, device=dev) * 0.05
        w2 = eng.split_weight_rows(w2f)
        kw = {"w2_frags": eng.weight_frags(w2f)} if frags else {}
        w3 = eng.split_weight_rows(torch.randn(p4, planes, device=dev) * 0.1)
        w1 = eng.split_weight_rows(torch.randn(planes, p4, device=dev) * 0.05)
        b2, b3, b1 = torch.zeros(planes, device=dev), torch.zeros(p4, device=dev), torch.zeros(planes, device=dev)
        for _ in range(4):
            eng.bneck_chain(planes, nb, hw, hw, t1, x, out, t1n, w2, b2, w3, b3, w1, b1, **kw)
        torch.cuda.synchronize()
        del t1, x, out, t1n


def show(d):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = {}
    for r in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
        if "bneck" not in n:
            continue
        agg.setdefault(n, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for n, cs in agg.items():
        print(n)
        wc = sum(cs.get("SQ_WAVE_CYCLES", [0])) or 1.0
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} {sum(v) / len(v):16.0f} per launch   {sum(v) / wc * (len(cs.get('SQ_WAVE_CYCLES', [0])) / len(v)):8.3f} of SQ_WAVE_CYCLES")


if __name__ == "__main__":
    show(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] == "show" else run()
