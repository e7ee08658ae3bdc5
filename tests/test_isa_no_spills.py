"""No kernel of the matrix-core Lanczos translation unit may spill a VGPR or use scratch (`LzMfma4k8` sat at 256 VGPRs + one
spilled until round 5 moved the staging loads' row step into the buffer load's scalar offset).  The kernels run two waves per SIMD on a
256-register budget: a spill is silent (8 bytes of scratch, a reload in the inner loop) and only the code object's metadata shows it.
Compiles the TU to assembly for gfx950 (no GPU; ~30 s)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.timeout(900)
def test_no_kernel_of_k_lanczos_mfma_spills():
    import isa_stats

    rows = isa_stats.spills(os.path.join(ROOT, "videoprocessingframework_amd", "csrc", "k_lanczos_mfma.hip"))
    main = [r for r in rows if "k_lanczos_mfma<" in r[0]]
    assert len(main) >= 10, [r[0] for r in rows]                      # every instantiation of the main kernel is there
    for name, vgpr, vspill, sspill, scratch in rows:
        assert vspill == 0 and scratch == 0, (name, vgpr, vspill, scratch)
        assert vgpr <= 256, (name, vgpr)
