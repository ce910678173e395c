"""packed-fp32 results of the rasteriser went wrong while another HIP queue ran a matrix + vector kernel; raster.hip is therefore compiled
without packed-fp32 instructions (its `// GS_CXXFLAGS:` line, read by csrc/Makefile).  This test disassembles the SHIPPED library and fails if a kernel of
that file contains one again (a changed flag, a Makefile that stopped reading the line)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import packed_audit  # noqa: E402

LIB = os.path.join(ROOT, "gshell_amd", "lib", "libgshell_hip.so")


@pytest.mark.skipif(not (os.path.isfile(packed_audit.OBJDUMP) and os.path.isfile(LIB)), reason="llvm-objdump or the built library missing")
def test_the_rasterisers_kernels_contain_no_packed_fp32_instruction():
    counts = packed_audit.packed_counts(LIB)
    raster = {k: v for k, v in counts.items() if any(t in k for t in ("k_rast_", "k_xfm_", "k_interp_", "k_face_normal_"))}
    assert len(raster) >= 8, sorted(raster)                      # the kernels of csrc/raster.hip were found in the disassembly
    assert not {k: v for k, v in raster.items() if v}, {k: v for k, v in raster.items() if v}
    # the audit sees packed instructions where they are known to be (the SDF network's epilogues): it is not blind
    assert any(v > 100 for k, v in counts.items() if "k_h2_fwd" in k)


def test_raster_source_carries_its_per_file_flag_and_the_makefile_reads_it():
    src = open(os.path.join(ROOT, "gshell_amd", "csrc", "raster.hip")).read()
    assert "\n// GS_CXXFLAGS: -fno-slp-vectorize\n" in src
    mk = open(os.path.join(ROOT, "gshell_amd", "csrc", "Makefile")).read()
    assert "GS_CXXFLAGS" in mk and "$(call fileflags,$<)" in mk
