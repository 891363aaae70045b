"""The register allocation of the kernels every measurement in DESIGN.md / MEASUREMENTS.md depends on, read from the
built library's gfx950 code objects (tools/kernel_regs.py; no GPU needed).  In round 5 a change to the shape of one
wait loop took k_grid_hb<4,24,4> from 124 to 145 vector registers and cost 10 % for weeks of box-time before it was
found by bisecting builds (MEASUREMENTS.md section R5.5): such a change now fails here, on the CPU."""
import importlib.util
import os
import shutil

import pytest

from libstb_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("kernel_regs", os.path.join(ROOT, "tools", "kernel_regs.py"))
kernel_regs = importlib.util.module_from_spec(spec)
spec.loader.exec_module(kernel_regs)

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(kernel_regs.LLVM, "llvm-readelf")) or shutil.which("c++filt") is None,
                                reason="llvm-readelf / c++filt not on this machine")

# kernel -> (most vector registers, LDS bytes at most): the allocation class each was measured in
PINNED = {
    "k_fill_hb<2, 0, 0>(fill_args, hb_args)": (96, 48 * 1024),    # one table: 5 waves a SIMD
    "k_fill_hb<4, 0, 0>(fill_args, hb_args)": (112, 80 * 1024),   # 8 tables
    "k_fill_hb<2, 1, 0>(fill_args, hb_args)": (72, 48 * 1024),    # samplea's fused evaluation
    # (round 6: the hand-over rings are one flat array with a spare ring for the lanes that hand nothing over: + 3.4 KB)
    "k_fill_hb<4, 1, 0>(fill_args, hb_args)": (80, 44 * 1024),    # 9-28 discounts (14 waves a workgroup: + 112 KB of staging rows, 160 KB in all at most)
    "k_fill_hb<3, 1, 0>(fill_args, hb_args)": (80, 44 * 1024),    # 4-8 discounts
    "k_grid_hb<4, 24, 4>(gh_args)": (124, 20 * 1024),             # 64-discount evaluation: 145 registers cost 10 %
    "k_grid_hb<2, 24, 2>(gh_args)": (96, 20 * 1024),
}


@pytest.fixture(scope="module")
def kernels():
    return kernel_regs.kernels(capi.LIB_PATH)


@pytest.mark.parametrize("name", sorted(PINNED))
def test_hot_kernel_keeps_its_allocation(kernels, name):
    assert name in kernels, [k for k in kernels if k.startswith(name.split("<")[0])][:8]
    k, (vmax, ldsmax) = kernels[name], PINNED[name]
    assert k["vgpr"] <= vmax, (name, k)
    assert k["spill"] == 0 and k["scratch"] == 0, (name, k)
    assert k["lds"] <= ldsmax, (name, k)


def test_no_hot_kernel_spills(kernels):
    hot = {n: k for n, k in kernels.items() if n.startswith(("k_fill_hb<", "k_grid_hb<4", "k_grid_hb<2", "k_count_cells",
                                                            "k_emit_", "k_scan_lists", "k_jobs_build", "k_bterms_one", "k_eval_tail"))}
    assert len(hot) >= 20   # (k_fill_chain, the form for dense pair sets, is compiled to 80 registers with 1-8 spilled: not a headline kernel)
    bad = {n: k for n, k in hot.items() if k["spill"] or k["scratch"]}
    assert not bad, bad
