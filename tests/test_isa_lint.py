"""CPU: ISA lint of the built library (tools/isa_lint.py).  The weight-gradient contraction reads its MFMA fragments with inline-asm
`ds_read_b64_tr_b16` and hand-placed `s_waitcnt`; in rounds 2-5 the register allocator copied loop-carried fragment registers AHEAD of
the wait (stale fragments whenever the LDS answered late: the round-5 mismatches under device sharing).  No instruction of ANY kernel may
touch the destination of an LDS read that the LGKM counter still covers -- checked on the code objects inside libfabind_hip.so, i.e. on
what actually runs."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

needs_llvm = pytest.mark.skipif(not os.path.exists(os.path.join(isa_lint.LLVM, "llvm-objdump")), reason="llvm-objdump not found")


@pytest.fixture(scope="module")
def report():
    if not os.path.exists(isa_lint.DEFAULT_LIB):
        from fabind_amd import build
        build.build(verbose=False)
    return isa_lint.run(verbose=True)


@needs_llvm
def test_no_instruction_touches_an_in_flight_lds_read(report):
    assert report["kernels"] > 200                         # the whole library was disassembled
    assert report["tn_kernels"] == 5 and report["tn_tr_reads"] >= 5 * 40      # the five instantiations and their transpose reads were seen
    assert not report["async_lds"], {k: v[:3] for k, v in report["async_lds"].items()}


@needs_llvm
def test_las_step_accumulates_with_scalar_fmas(report):
    assert len(report["las_kernels"]) == 2
    assert not report["pk_fma_acc"], report["pk_fma_acc"]


def test_lint_flags_the_round5_pattern():
    """The hazard of rounds 2-5 in miniature: reads issued, a copy of a destination ahead of the wait, then the wait."""
    ins = [(0, "ds_read_b64_tr_b16", "v[204:205], v130"), (8, "v_mfma_f32_16x16x32_bf16", "a[0:3], v[10:13], v[20:23], a[0:3]"),
           (16, "v_mov_b64_e32", "v[162:163], v[204:205]"), (20, "s_branch", "65530"), (24, "s_endpgm", "")]
    ins_loop = [(a + 100, m, o) for a, m, o in ins]
    f = isa_lint.lint_async_lds(ins)
    assert len(f) == 1 and f[0][0] == 16
    ok = [(0, "ds_read_b64_tr_b16", "v[204:205], v130"), (8, "s_waitcnt", "lgkmcnt(0)"), (12, "v_mov_b64_e32", "v[162:163], v[204:205]"), (16, "s_endpgm", "")]
    assert not isa_lint.lint_async_lds(ok)
    # across a back-edge: the read at the bottom of the loop, the copy at its top
    loop = [(0, "v_mov_b64_e32", "v[2:3], v[8:9]"), (4, "s_waitcnt", "lgkmcnt(0)"), (8, "ds_read_b64", "v[8:9], v1"), (16, "s_cbranch_scc1", "65531"),
            (20, "s_waitcnt", "lgkmcnt(0)"), (24, "s_endpgm", "")]
    f = isa_lint.lint_async_lds(loop)
    assert [x[0] for x in f] == [0]
    assert ins_loop
