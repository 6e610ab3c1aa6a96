"""The library's gfx950 assembly must not contain the VMEM store-data hazard LLVM leaves uncovered (a > 64-bit buffer store
with an SGPR soffset whose data registers are written within the next two issue slots): the root cause of round 4's
nondeterministic conv_first results (tools/isa_store_hazard.py).  Compiles every csrc/*.hip to assembly with the
library's own flags (hipcc cross-compiles without a GPU) and scans it; also pins the scanner on the two recorded listings'
essential lines."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "isa_store_hazard.py")


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


def test_library_assembly_has_no_uncovered_store_data_hazard():
    r = subprocess.run([sys.executable, TOOL, "--build"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 hazard site(s) in" in r.stdout
