"""The sliding evaluation (csrc/evalslide.hip) without a GPU: its plan builder (csrc/slideplan.hpp) and its band routine
(csrc/slidecore.hpp) are plain C++ shared with tools/slide_emul.cpp, which runs the routine one "lane" at a time on random alignments
and nested chains and compares every candidate's three counters with brute force — several events in one step, steps without
events, members that drop the column's reference base, most-degenerate members that do not accept it, strict positions anywhere,
gaps between windows, every k in 2..31, v in 0..3, 1 / 2 / 4 row words per lane, with and without exclusion words.  The GPU run
of the same code is compared with the oracle in test_hip_parity.py / test_scale_parity.py (MP_EVAL_SLIDE=1)."""
import os
import subprocess

from conftest import REPO


def test_band_routine_and_plan_builder_equal_brute_force(tmp_path):
    exe = str(tmp_path / "slide_emul")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", os.path.join(REPO, "tools", "slide_emul.cpp"), "-o", exe])
    for seed in ("12345", "7"):
        out = subprocess.run([exe, "400", seed], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        assert b"equal to brute force" in out.stdout
