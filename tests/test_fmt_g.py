"""The driver's number formatting (src/fmt_g.h) prints what printf("%g") prints -- the reference's
`stream << value` (io.cpp:219, :245, :271) -- for millions of floats and doubles: random bit patterns
of every exponent, values of a running simulation, ties and near-ties at six digits, powers of ten
and their neighbours, denormals, infinities and NaNs.  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_format_g6_equals_printf(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not found")
    exe = str(tmp_path / "fmt_g_check")
    subprocess.check_call([gxx, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "src"),
                           os.path.join(ROOT, "tests", "native", "fmt_g_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert " 0 mismatches" in out.stdout
