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
