"""Build hygiene of the shipped library (runs on CPU: hipcc cross-compiles gfx950 without a GPU).

* no MFMA kernel instantiation spills (round-1 finding: spilling variants compiled into the product);
* the library reads no environment variable (round-1 finding: test knobs steering the product at run time);
* every symbol of include/avcer_hip.h is exported (also checked by __graft_entry__.build)."""
import os
import re
import shutil
import subprocess

import pytest

from avcer_amd import _lib, build

CSRC = build.CSRC


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        pytest.skip("hipcc not available")
    return exe


@pytest.mark.parametrize("src", ["gemm.hip", "fused.hip"])
def test_mfma_kernels_do_not_spill(src, tmp_path):
    r = subprocess.run([_hipcc()] + build.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o",
                                                   str(tmp_path / "x.o")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", r.stderr)]
    assert len(names) == len(scratch) == len(spills) and len(names) >= 10
    bad = [(n, s, v) for n, s, v in zip(names, scratch, spills) if s or v]
    assert not bad, bad


def test_library_reads_no_environment_and_exports_the_header():
    lib = build.build()
    und = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "getenv" not in und
    defined = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    header = open(os.path.join(CSRC, "..", "..", "include", "avcer_hip.h")).read()
    declared = set(re.findall(r"\b(avcer_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert re.search(rf"\b{name}\b", defined), name


def test_tail_kernel_asm_loaded_registers_are_untouched_until_their_wait(tmp_path):
    """bneck_tail2_kernel loads its residual ring by inline asm, so only the counted `s_waitcnt vmcnt(4)` that names the
    registers protects them; tools/audit_asm_loads.py checks the generated ISA for any earlier read or write."""
    import importlib.util

    asm = tmp_path / "fused.s"
    flags = [f for f in build.FLAGS if f not in ("-fPIC", "-shared")]
    r = subprocess.run([_hipcc()] + flags + ["-S", "--cuda-device-only", "-o", str(asm), os.path.join(CSRC, "fused.hip")],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    spec = importlib.util.spec_from_file_location("audit_asm_loads", os.path.join(os.path.dirname(CSRC), "..", "tools", "audit_asm_loads.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.audit(str(asm)) == []
