"""The C++ a maintainer is told to add to the reference (the ```cpp blocks of the integration guide) compiles
against include/lbm_hip.h: argument counts and types of every documented call match the header.
Stand-ins replace what the reference's own headers declare (Dst_t, FLT, the option globals)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PRELUDE = r'''
#include <algorithm>
#include <iostream>
#include <utility>
typedef float FLT;
typedef unsigned int UINT;
typedef const UINT DUI;
typedef const FLT DFL;
struct Dst_t { FLT *p; FLT *data() const { return p; } };      // Kokkos::View<FLT***>::data()
enum Scene { SHEAR_WAVE, LID_DRIVEN };
static UINT NX = 8, NY = 8, STEPS = 1, OUTPUT_EVERY = 0, Y_BELOW = 0, Y_GLOB = 8;
static FLT OMEGA = 1, RHO_INIT = 1, U_INIT = 0.1f;
static Scene SCENE = SHEAR_WAVE;
static bool PUSH = false;
static std::ostream *OUT_STREAM = &std::cout;
#define USE_MPI false
#define COALESCING false
static void output(Dst_t &, Dst_t &, UINT, UINT, UINT, UINT, int) {}
'''


def cpp_blocks():
    with open(os.path.join(ROOT, "INTEGRATION.md")) as fh:
        return re.findall(r"```cpp\n(.*?)```", fh.read(), flags=re.S)


def test_documented_shim_compiles_against_the_header(tmp_path):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    blocks = cpp_blocks()
    assert len(blocks) >= 4
    parts = [PRELUDE]
    for i, b in enumerate(blocks):
        if re.search(r"^void \w+\(", b, flags=re.M):
            parts.append(b)                      # definitions at file scope (the operator shim)
        else:
            # statements: the body of a function that has the names the text uses
            parts.append(f"void block_{i}(Dst_t f_view, Dst_t buf_view, Dst_t vel, void *stream, int k) {{\n"
                         "    static const int PREC = sizeof(FLT);\n    (void)PREC; (void)stream; (void)k; (void)vel;\n"
                         + ("    Dst_t f = f_view, buf = buf_view;\n" if ".data()" in b else
                            "    FLT *f = f_view.data(), *buf = buf_view.data();\n")
                         + b + "\n    (void)f; (void)buf;\n}\n")
    src = tmp_path / "shim.cpp"
    src.write_text("\n".join(parts))
    cmd = [cxx, "-std=c++17", "-fsyntax-only", "-Wall", "-Werror=return-type", "-I", os.path.join(ROOT, "include"),
           "-include", "lbm_hip.h", str(src)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-4000:]


def test_header_is_plain_c(tmp_path):
    cc = shutil.which("gcc")
    if cc is None:
        pytest.skip("no gcc")
    src = tmp_path / "use.c"
    src.write_text('#include "lbm_hip.h"\nint main(void) { return lbm_get_layout() < 0; }\n')
    cmd = [cc, "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src)]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-4000:]
