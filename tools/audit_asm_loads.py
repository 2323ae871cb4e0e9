#!/usr/bin/env python3
"""Register audit of bneck_tail2_kernel (avcer_amd/csrc/fused.hip).

Its residual loads are inline asm (hipcc must not wait for them itself), so nothing tells the compiler that their
destination registers are not valid until the counted `s_waitcnt vmcnt(4)` two groups later.  This script checks the
generated ISA: between each residual load inside the loop and the second `s_waitcnt vmcnt(4)` after it (the one that
names the registers), no instruction may read or write the destination registers; and between the loop's last wait and
the inline-asm `s_waitcnt vmcnt(0)` behind the loop (which covers the last trip's never-consumed loads) no instruction
may touch any ring register.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/fused.s avcer_amd/csrc/fused.hip
    python tools/audit_asm_loads.py /tmp/fused.s
"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(path, kernel="bneck_tail2_kernel"):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^_ZN\S*" + kernel + r"\S*:", l)][0]
    body = []
    for l in lines[start:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l.strip())
    # the loop: from the first in-loop barrier to the last s_waitcnt vmcnt(4)
    waits = [i for i, t in enumerate(body) if t.startswith("s_waitcnt vmcnt(4)")]
    loads = [i for i, t in enumerate(body) if t.startswith("global_load_dwordx4") and waits and i > waits[0] - 400]
    if len(waits) != 2:
        return [f"expected the two unrolled group bodies (2 x s_waitcnt vmcnt(4)), found {len(waits)}"]
    w0, w1 = waits
    # loop body spans [top, w1]; find top = label after which the first mfma of group 0 starts: take the barrier before w0's group
    bars = [i for i, t in enumerate(body) if t == "s_barrier"]
    top = max(b for b in bars if b < w0 - 50)  # the barrier that ends the prologue / previous iteration
    in_loop = [i for i in loads if top < i < w1]
    problems = []
    if len(in_loop) != 4:
        problems.append(f"expected 4 residual loads in the loop, found {len(in_loop)}")
    for li in in_loop:
        dst = regs(body[li].split()[1].rstrip(","))
        # loads before w0 (group G) are named by the wait that ends group G+1 = w1; loads between w0 and w1 by w0 of the next trip
        if li < w0:
            span = list(range(li + 1, w1))
        else:
            span = list(range(li + 1, w1 + 1)) + list(range(top, w0))
        for i in span:
            t = body[i]
            if not t or t.startswith(";") or t.startswith(".") or i in in_loop:
                continue
            used = set()
            for k in re.findall(r"v\[\d+:\d+\]|v\d+", t):
                used |= regs(k)
            if used & dst:
                problems.append(f"line {i}: '{t[:70]}' touches {sorted(used & dst)} loaded at line {li} before their wait")
    # After the loop: the last trip's loads are never consumed.  They must be covered by the asm `s_waitcnt vmcnt(0)` that
    # follows the loop (it names the ring registers), and nothing between the loop's last wait and that one may touch them.
    post = [i for i in range(w1 + 1, len(body)) if body[i].startswith("s_waitcnt vmcnt(0)") and body[i - 1].startswith(";;#ASMSTART")]
    if not post:
        problems.append("no inline-asm s_waitcnt vmcnt(0) between the loop and s_endpgm: the last residual loads are never waited for")
    else:
        ring = set()
        for li in in_loop:
            ring |= regs(body[li].split()[1].rstrip(","))
        for i in range(w1 + 1, post[0]):
            t = body[i]
            if not t or t.startswith(";") or t.startswith("."):
                continue
            used = set()
            for k in re.findall(r"v\[\d+:\d+\]|v\d+", t):
                used |= regs(k)
            if used & ring:
                problems.append(f"line {i}: '{t[:70]}' touches ring registers {sorted(used & ring)} between the loop and the final wait")
    return problems


if __name__ == "__main__":
    p = audit(sys.argv[1])
    for x in p:
        print("AUDIT:", x)
    print("audit clean" if not p else f"{len(p)} problem(s)")
    sys.exit(1 if p else 0)
