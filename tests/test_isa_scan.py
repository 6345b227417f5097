"""Static ordering checks on the gfx950 code object of liblamslide_hip.so (tools/isa_scan.py): runs without a GPU.

Guards what no functional test sees reliably: loads that land in an MFMA's accumulator registers a few instructions behind it (the
k_resident nondeterminism of round 2 came and went with that distance), and the inline-asm vector-memory instructions of the
token-stationary linear1 kernel, around which hipcc pads no wait states."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_scan  # noqa: E402

from lam_slide_amd import _lib  # noqa: E402


@pytest.fixture(scope="module")
def reports():
    if not os.path.exists(isa_scan.LLVM + "/llvm-objdump"):
        pytest.skip("no ROCm llvm-objdump in this environment")
    path = _lib.build()  # (no-op when the in-tree library is newer than its sources)
    return {r.name: r for r in isa_scan.scan_library(path)}


MFMA_FAMILIES = ("k_resident", "k_gemm_glds", "k_linear1_ts", "k_attention", "k_dense_mfma", "k_head_step_mfma", "k_embed_mfma")


def test_every_mfma_kernel_family_is_scanned(reports):
    for fam in MFMA_FAMILIES:
        hits = [r for n, r in reports.items() if fam in n and r.n_mfma > 0]
        assert hits, f"no kernel of family {fam} with MFMAs found in the code object"


def test_no_vector_memory_load_into_recent_mfma_accumulators(reports):
    bad = {n: r.r1 for n, r in reports.items() if r.r1}
    assert not bad, "vector-memory loads fewer than 12 instructions behind an MFMA whose C/D registers they overwrite:\n" + "\n".join(
        f"{n}: {v[0]} (+{len(v) - 1} more)" for n, v in bad.items())


def test_inline_asm_vector_memory_wait_states(reports):
    bad = {n: (r.r2 + r.r3) for n, r in reports.items() if r.r2 or r.r3}
    assert not bad, "missing wait states in front of a vector-memory instruction:\n" + "\n".join(f"{n}: {v[0]}" for n, v in bad.items())


def test_linear1_ts_issues_its_dma_and_stores(reports):
    # the structure the kernel's counted s_waitcnt relies on: LDS-DMA instructions and streaming stores are present in every instance
    for n, r in reports.items():
        if "k_linear1_ts" in n:
            assert r.n_mfma >= 160, n


# ---- the scanner itself: synthetic listings in llvm-objdump's format (a rule that cannot fail guards nothing) ----
def _listing(body: str) -> str:
    return "0000000000001000 <k_demo>:\n" + "".join("\t" + line.strip() + "  // 000000001000: 00000000\n" for line in body.strip().splitlines())


def _scan(body: str):
    kernels = isa_scan.parse(_listing(body))
    assert list(kernels) == ["k_demo"]
    return isa_scan.scan("k_demo", kernels["k_demo"], 12)


def test_scanner_flags_a_load_into_fresh_mfma_accumulators_and_respects_distance_and_branches():
    near = """
        v_mfma_f32_32x32x16_bf16 v[0:15], v[40:43], v[44:47], v[0:15]
        v_add_u32_e32 v20, v21, v22
        global_load_dwordx4 v[4:7], v[30:31], off
    """
    r = _scan(near)
    assert len(r.r1) == 1 and r.min_mfma_load_gap == 2
    far = near.replace("v_add_u32_e32 v20, v21, v22", "\n".join(["v_add_u32_e32 v20, v21, v22"] * 12))
    assert not _scan(far).r1
    other_regs = near.replace("v[4:7]", "v[64:67]")
    assert not _scan(other_regs).r1
    behind_branch = near.replace("v_add_u32_e32 v20, v21, v22", "s_cbranch_vccnz 12")  # layout order is not execution order
    assert not _scan(behind_branch).r1
    lds = near.replace("global_load_dwordx4 v[4:7], v[30:31], off", "ds_read_b128 v[4:7], v30")  # counted, not failed (hipcc's own pattern)
    r = _scan(lds)
    assert not r.r1 and r.r1_lds == 1


def test_scanner_flags_missing_wait_states_in_front_of_vector_memory():
    r2 = _scan("""
        v_readfirstlane_b32 s4, v1
        v_readfirstlane_b32 s5, v2
        s_nop 1
        global_store_dwordx4 v3, v[8:11], s[4:5] nt
    """)
    assert len(r2.r2) == 1
    ok = _scan("""
        v_readfirstlane_b32 s4, v1
        v_readfirstlane_b32 s5, v2
        s_nop 4
        global_store_dwordx4 v3, v[8:11], s[4:5] nt
    """)
    assert not ok.r2
    salu = _scan("""
        s_add_u32 s4, s6, s7
        s_addc_u32 s5, s8, 0
        global_load_dwordx4 v[8:11], v3, s[4:5]
    """)
    assert not salu.r2  # scalar-ALU writes are interlocked (hipcc itself emits this)
    r3 = _scan("""
        s_mov_b32 m0, s9
        global_load_lds_dwordx4 v3, s[4:5] offset:1024
    """)
    assert len(r3.r3) == 1
    ok3 = _scan("""
        s_add_u32 m0, s9, 16
        s_nop 0
        global_load_lds_dwordx4 v3, s[4:5] offset:1024
    """)
    assert not ok3.r3
