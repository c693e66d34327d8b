"""Instruction-level rules on the library as BUILT (the gfx950 code objects inside cloudaae_amd/libcloudaae_hip.so are
disassembled; no GPU needed)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import isa_scan  # noqa: E402

LIB = os.path.join(ROOT, "cloudaae_amd", "libcloudaae_hip.so")


@pytest.fixture(scope="module")
def instructions():
    if not isa_scan.tools_present():
        pytest.skip("no ROCm LLVM tools on this machine")
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    return isa_scan.disassemble(LIB)


def test_no_packed_fp32_instruction_takes_its_low_half_from_a_high_register(instructions):
    """The cause of the wrong neighbour lists under two processes on one GPU (profiles/notes_two_processes_one_gpu.md, round 6):
    `v_pk_add_f32 ... op_sel:[0,1]`, formed by the compiler for the candidate norms x*x + y*y + z*z of the layer-1 kNN kernels,
    returned `src0 + 0` in lanes 48-63 about once in 6000 executions when a wave of another process shared the SIMD.  csrc/Makefile
    compiles the files where the compiler forms such instructions without the packed-fp32 feature; this test keeps every file
    honest, whatever a later compiler or a later edit does."""
    assert len(instructions) > 100000, "disassembly looks empty"
    kernels = set(fn for fn, _ in instructions)
    assert any("knn3_wide_kernel" in k for k in kernels) and any("adam_tf_kernel" in k for k in kernels)
    bad = isa_scan.packed_f32_low_from_high(instructions)
    assert not bad, "packed-fp32 instructions with op_sel on a low half:\n" + "\n".join("%s | %s" % b for b in bad[:20])


def test_chamfer_digest_reads_no_xdl_result_too_early(instructions):
    """csrc/nn_distance.hip's digest (`v_min3_f32` in inline assembly) reads accumulators of
    v_mfma_f32_32x32x16_bf16, an XDL operation whose result needs 11 wait states before a vector instruction may read it; the
    compiler's hazard recogniser does not see inline assembly (a stale read already happened once: a few wrong neighbours per
    thousand).  The source pins the order with scheduling barriers; this checks the built code."""
    reads = [i for fn, i in instructions if "nn_distance_filter_kernel" in fn and i.startswith("v_min3_f32")]
    assert len(reads) >= 16, "the digest's v_min3_f32 were not found in the disassembly"
    early = isa_scan.xdl_results_read_too_early(instructions, "nn_distance_filter_kernel")
    assert not early, early[:5]
