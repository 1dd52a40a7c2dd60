"""Static check of the BUILT libraries' device code (no GPU): no AccVGPR access may sit closer than 19 wait states behind an
MFMA that wrote the register (tools/check_mfma_hazards.py).  The fp32 MFMA of the GEMM / GRU kernels is inline asm, which the
compiler's own hazard recognizer cannot see, so a register-allocator copy behind a K loop can read a tile that is still being
written; the scanner found that pattern in the unaligned-operand GEMM instantiations in round 3."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_mfma_hazards as H  # noqa: E402

PKG = os.path.join(ROOT, "neuraldialog-laed_amd", "laed_amd")


def _ins(lines):
    """[(mnemonic, operands, branch offset or None)] -> the scanner's instruction list (4 bytes per instruction)."""
    out = [("base", 0x1000, "", None)]
    for i, (op, args, off) in enumerate(lines):
        out.append((0x1000 + 4 * i, op, args, off))
    return out


def test_scanner_flags_a_copy_behind_an_mfma_and_accepts_a_drained_one():
    mfma = ("v_mfma_f32_32x32x2_f32", "a[0:15], v1, v2, a[0:15]", None)
    filler = ("v_add_u32_e32", "v3, v4, v5", None)
    racy = _ins([mfma] + [filler] * 5 + [("v_accvgpr_read_b32", "v9, a15", None), ("s_endpgm", "", None)])
    assert list(H.check_kernel("racy", racy).values()) == [("v_accvgpr_read_b32", "v9, a15", 5, 0x1000)]
    other_tile = _ins([mfma, ("v_accvgpr_mov_b32", "a16, a17", None), ("s_endpgm", "", None)])
    assert not H.check_kernel("other", other_tile)
    drained = _ins([mfma, ("s_nop", "15", None), ("s_nop", "3", None), ("v_accvgpr_read_b32", "v9, a15", None), ("s_endpgm", "", None)])
    assert not H.check_kernel("drained", drained)
    chained = _ins([mfma, mfma, ("s_endpgm", "", None)])                  # MFMA -> MFMA on one tile is interlocked
    assert not H.check_kernel("chained", chained)
    # a loop whose exit edge carries the copy: MFMA; branch back (offset 0) or fall through to the read
    loop = _ins([mfma, filler, ("s_cbranch_scc1", "65534", 0), ("v_accvgpr_mov_b32", "a63, a15", None), ("s_endpgm", "", None)])
    assert [v[:3] for v in H.check_kernel("loop", loop).values()] == [("v_accvgpr_mov_b32", "a63, a15", 2)]
    # the same with the wait states inside the loop body
    safe = _ins([mfma, ("s_nop", "15", None), ("s_nop", "3", None), ("s_cbranch_scc1", "65534", 0),
                 ("v_accvgpr_mov_b32", "a63, a15", None), ("s_endpgm", "", None)])
    assert not H.check_kernel("safe", safe)


@pytest.mark.skipif(not os.path.exists(os.path.join(H.LLVM, "llvm-objdump")), reason="no llvm-objdump in this image")
@pytest.mark.parametrize("lib", ["liblaed_hip.so", "liblaed_hip_hooks.so"])
def test_built_library_has_no_mfma_accvgpr_hazard(lib, capsys):
    path = os.path.join(PKG, lib)
    if not os.path.exists(path):
        if lib == "liblaed_hip.so":
            pytest.fail("the product library is not built: run __graft_entry__.build()")
        pytest.skip("experiment build not present")
    bad = H.check_library(path)
    out = capsys.readouterr().out
    assert bad == 0, out[-4000:]
    assert " 0 finding(s)" in out and "MFMA instructions" in out
