"""The library's gfx950 assembly must not contain the two hazards neither the hardware nor LLVM covers for this code:

* the VMEM store-data hazard (a > 64-bit buffer store with an SGPR soffset whose data registers are written within the next two
  issue slots): the root cause of round 4's nondeterministic conv_first results (tools/isa_store_hazard.py);
* a VGPR touched while an (inline-asm) VMEM load into it is still outstanding — VGPR reads are not interlocked with VMEM returns
  (the loop-carried f0 fragments of corr_fwd_ring_kernel; tools/isa_load_hazard.py).

Compiles every csrc/*.hip to assembly ONCE with the library's own flags (hipcc cross-compiles without a GPU; skipped where there
is no hipcc) and runs both scanners over it; also pins each scanner on the recorded listing's essential lines."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "isa_store_hazard.py")
TOOL_LD = os.path.join(ROOT, "tools", "isa_load_hazard.py")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_scanner_flags_the_recorded_site_and_accepts_the_padded_form(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("_Zkernel:\n\tbuffer_store_dwordx4 v[96:99], v118, s[28:31], s43 offen\n\tv_mov_b32_e32 v96, 2.0\n\ts_endpgm\n")
    ok = tmp_path / "ok.s"
    ok.write_text("_Zkernel:\n\tbuffer_store_dwordx4 v[96:99], v118, s[28:31], s43 offen\n\t;;#ASMSTART\n\ts_nop 1\n\t;;#ASMEND\n"
                  "\tv_mov_b32_e32 v96, 2.0\n"
                  "\tbuffer_store_dwordx4 v[0:3], v8, s[28:31], 0 offen\n\tv_mov_b32_e32 v0, 2.0\n"        # no SGPR soffset: LLVM pads this form itself
                  "\tbuffer_store_dwordx4 v[4:7], v8, s[28:31], s9 offen\n\tv_mov_b32_e32 v8, 2.0\n\ts_endpgm\n")   # address register: no hazard
    r = subprocess.run([sys.executable, TOOL, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "1 hazard site(s)" in r.stdout and "v96" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, TOOL, str(ok)], capture_output=True, text=True)
    assert r.returncode == 0 and "0 hazard site(s)" in r.stdout, r.stdout


def test_load_scanner_flags_the_loop_carried_copy_and_accepts_the_covered_form(tmp_path):
    # the recorded shape (corr_fwd_ring_kernel, round 5 build): 6 LDS-DMA loads, the fragment load, 6 more DMA loads, then the
    # register allocator's phi copy at the end of the loop body with only vmcnt(12) in between
    dma = "\tbuffer_load_dwordx4 v140, s[24:27], 0 offen lds\n"
    body = "_Zkernel:\n.LBB0_1:\n" + dma * 6 + "\tbuffer_load_dwordx4 v[132:135], v140, s[0:3], 0 offen\n" + dma * 6
    bad = tmp_path / "bad.s"
    bad.write_text(body + "\ts_waitcnt vmcnt(12)\n\tv_mov_b64_e32 v[108:109], v[132:133]\n\ts_cbranch_scc0 .LBB0_1\n\ts_endpgm\n")
    ok = tmp_path / "ok.s"
    ok.write_text(body + "\ts_waitcnt vmcnt(6)\n\tv_mov_b64_e32 v[108:109], v[132:133]\n\ts_cbranch_scc0 .LBB0_1\n"
                  "\tbuffer_load_dword v34, v36, s[36:39], 0 offen\n\tbuffer_load_dwordx4 v[34:37], v0, s[36:39], 0 offen\n"   # in-order overwrite: fine
                  "\ts_waitcnt vmcnt(0)\n\tv_add_f32_e32 v1, v34, v35\n\ts_endpgm\n")
    r = subprocess.run([sys.executable, TOOL_LD, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "1 in-flight-load hazard site(s)" in r.stdout and "132" in r.stdout, r.stdout
    r = subprocess.run([sys.executable, TOOL_LD, str(ok)], capture_output=True, text=True)
    assert r.returncode == 0 and "0 in-flight-load hazard site(s)" in r.stdout, r.stdout


@pytest.fixture(scope="module")
def library_assembly():
    from unflow_amd import build as B
    if not os.path.exists(B.HIPCC):
        pytest.skip("no hipcc on this machine (%s): the assembly scans need the ROCm compiler" % B.HIPCC)
    import isa_store_hazard
    return isa_store_hazard.build_all()


@pytest.mark.slow
@pytest.mark.timeout(1500)
def test_library_assembly_has_no_uncovered_store_data_hazard(library_assembly):
    import isa_store_hazard
    sites = [s for f in library_assembly for s in isa_store_hazard.scan(f)]
    assert not sites, sites[:5]


@pytest.mark.slow
@pytest.mark.timeout(1500)
def test_library_assembly_touches_no_register_with_a_load_in_flight(library_assembly):
    import isa_load_hazard
    sites = [s for f in library_assembly for s in isa_load_hazard.scan(f)]
    assert not sites, sites[:5]
