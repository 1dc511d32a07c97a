"""Build-time ISA check of the inline-assembly LDS-DMA launches (tools/check_dma_isa.py): the DMA blocks are intact,
nothing else writes M0, no spills -- the properties the hand-counted `s_waitcnt vmcnt(N)` of mvsn_conv_wino.hip rest on
that the compiler could silently break.  (hipcc cross-compiles gfx950 assembly on the CPU box: ~15 s.)"""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.skipif(not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")), reason="no hipcc")
def test_inline_asm_dma_kernels_keep_their_shape():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_dma_isa.py")], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("conv_wino_kernel<")]
    assert len(lines) >= 4 and all(ln.endswith("OK") for ln in lines), p.stdout
    s2 = [ln for ln in p.stdout.splitlines() if ln.startswith("conv_wino_s2_kernel:")]
    assert len(s2) == 1 and s2[0].endswith("OK") and " 1 counted wait" in s2[0], p.stdout
