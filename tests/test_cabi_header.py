"""include/pcreg.h consumed from C and from C++: tests/cabi/check_header.c is compiled by gcc (-std=c99) and
by g++ (-x c++) with -Wall -Wextra -Werror, linked against libpcreg_hip.so and run.  On a box without a GPU
it must exit 77 (every call returned PCREG_E_NODEVICE); on the MI355X it must pass its known-answer checks
(testTransformEstimation.m:2-14 through pcreg_estimate_transform / pcreg_calc_dists / pcreg_ransac)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cabi", "check_header.c")


def _build(lang):
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "pcreg_amd", "libpcreg_hip.so")):
        g.build()
    out = os.path.join(ROOT, "tests", "cabi", f"check_header_{lang}")
    cc = ["gcc", "-std=c99"] if lang == "c" else ["g++", "-x", "c++", "-std=c++17"]
    subprocess.check_call(cc + ["-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", out,
                                "-L" + os.path.join(ROOT, "pcreg_amd"), "-lpcreg_hip", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "pcreg_amd")])
    return out


@pytest.mark.parametrize("lang", ["c", "cpp"])
def test_header_compiles_links_and_reports_nodevice(lang):
    import torch
    exe = _build(lang)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 77, r.stdout + r.stderr
        assert "PCREG_E_NODEVICE" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("lang", ["c", "cpp"])
def test_header_known_answers_on_the_gpu(lang):
    r = subprocess.run([_build(lang)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
