#!/usr/bin/env python3
"""Register audit of every kernel that loads into VGPRs by inline asm (conv_gemm_wd_kernel in gemm.hip: the weight
fragments; bneck_tail2_kernel in fused.hip: the residual ring).

hipcc does not know that an asm `buffer_load` / `global_load` is asynchronous: nothing but the kernel's own counted
`s_waitcnt vmcnt(N)` (an asm statement that names the registers) keeps a later instruction from reading or overwriting the
destination before the data has landed.  This script proves the property on the generated ISA instead of trusting the source:

  * the kernel's control-flow graph is walked from its entry with the queue of outstanding vector-memory operations as the
    state (gfx9: loads, stores and LDS-DMA all count in vmcnt and retire in issue order; `s_waitcnt vmcnt(N)` leaves the N
    youngest outstanding); every (block, queue) pair is visited once, so loops are followed until the queue state repeats;
  * while an asm load is outstanding, NO instruction may read or write any of its destination VGPRs -- not an MFMA, not a
    copy the register allocator inserted, not a second load;
  * at `s_endpgm` no asm load may be outstanding (the drain wait behind the loop is there and names the registers).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/gemm.s avcer_amd/csrc/gemm.hip
    python tools/audit_asm_loads.py /tmp/gemm.s conv_gemm_wd_kernel
"""
import re
import sys

VM_OP = re.compile(r"^(buffer_(load|store|atomic)|global_(load|store|atomic)|flat_(load|store|atomic)|scratch_(load|store))")
LOAD_TO_VGPR = re.compile(r"^(buffer_load|global_load|flat_load)\S*\s+(v\[\d+:\d+\]|v\d+),")
BRANCH = re.compile(r"^(s_branch|s_cbranch_\w+)\s+(\.L\w+)")
LABEL = re.compile(r"^(\.L\w+):")
MAX_STATES = 200000


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return frozenset({int(m.group(1))}) if m else frozenset()


def vgprs_of(text):
    used = set()
    for k in re.findall(r"\bv\[\d+:\d+\]|\bv\d+\b", text):
        used |= regs(k)
    return used


def kernels(path, name):
    """[(mangled name, [instruction lines with asm markers])] for every kernel whose name contains `name`."""
    lines = open(path).read().split("\n")
    out = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w*" + re.escape(name) + r"\w*):", l)
        if not m:
            continue
        body = []
        for t in lines[i + 1:]:
            if t.startswith(".Lfunc_end"):
                break
            body.append(t.strip())
        out.append((m.group(1), body))
    return out


def blocks_of(body):
    """Basic blocks: {label: [(text, in_asm)]}, plus the fall-through order."""
    order, blocks, cur, in_asm = ["<entry>"], {"<entry>": []}, "<entry>", False
    for t in body:
        if not t:
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = LABEL.match(t)
        if m:
            cur = m.group(1)
            order.append(cur)
            blocks[cur] = []
            continue
        if t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        if t:
            blocks[cur].append((t, in_asm))
    return order, blocks


def audit_kernel(name, body):
    order, blocks = blocks_of(body)
    nxt = {order[i]: (order[i + 1] if i + 1 < len(order) else None) for i in range(len(order))}
    problems, seen, work = [], set(), [("<entry>", ())]
    n_asm_loads = sum(1 for b in blocks.values() for t, a in b if a and LOAD_TO_VGPR.match(t))
    while work:
        lab, q = work.pop()
        if (lab, q) in seen:
            continue
        seen.add((lab, q))
        if len(seen) > MAX_STATES:
            return [f"{name}: state space exceeds {MAX_STATES} (queue states do not reconverge)"], n_asm_loads
        q = list(q)  # oldest ... youngest; entries: frozenset of destination VGPRs (asm load) or None (any other vm op)
        ended = False
        for t, in_asm in blocks[lab]:
            busy = frozenset().union(*[e for e in q if e]) if any(q) else frozenset()
            m = LOAD_TO_VGPR.match(t)
            if busy:
                used = vgprs_of(t)
                if used & busy:
                    problems.append(f"{name} {lab}: '{t[:80]}' touches v{sorted(used & busy)} while an asm load into them is outstanding")
            if t.startswith("s_waitcnt"):
                c = re.search(r"vmcnt\((\d+)\)", t)
                if c:
                    n = int(c.group(1))
                    while len(q) > n:
                        q.pop(0)
            elif VM_OP.match(t):
                q.append(regs(m.group(2)) if (m and in_asm) else None)
            while q and q[0] is None:  # operations older than every outstanding asm load no longer matter
                q.pop(0)
            if len(q) > 63:  # the hardware counter has 6 bits: a path that never waits for its asm loads
                return sorted(set(problems + [f"{name} {lab}: more than 63 vector-memory operations behind an asm load that is never waited for"])), n_asm_loads
            if t.startswith("s_endpgm"):
                if any(q):
                    problems.append(f"{name} {lab}: s_endpgm with an asm load still outstanding")
                ended = True
                break
            b = BRANCH.match(t)
            if b:
                work.append((b.group(2), tuple(q)))
                if b.group(1) == "s_branch":
                    ended = True
                    break
        if not ended and nxt.get(lab):
            work.append((nxt[lab], tuple(q)))
    return sorted(set(problems)), n_asm_loads


def audit(path, kernel="bneck_tail2_kernel"):
    """Problems found in every instantiation of `kernel` in the assembly file (empty list = clean)."""
    ks = kernels(path, kernel)
    if not ks:
        return [f"no kernel matching {kernel} in {path}"]
    problems = []
    for name, body in ks:
        p, n = audit_kernel(name, body)
        if n == 0:
            p = p + [f"{name}: no inline-asm load found (the audit would be vacuous)"]
        problems += p
    return problems


if __name__ == "__main__":
    kern = sys.argv[2] if len(sys.argv) > 2 else "bneck_tail2_kernel"
    found = kernels(sys.argv[1], kern)
    p = audit(sys.argv[1], kern)
    for x in p[:40]:
        print("AUDIT:", x)
    print(f"{len(found)} instantiation(s) of {kern}: " + ("audit clean" if not p else f"{len(p)} problem(s)"))
    sys.exit(1 if p else 0)
