"""The C-ABI shared library loads and exports exactly what include/lbm_hip.h declares (no GPU)."""
import ctypes as C
import os
import re

from lbm_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "lbm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lbm_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(capi.SIGNATURES) == names


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert lib.lbm_abi_version() == 1
    assert lib.lbm_status_string(0) == b"ok"


def test_argument_validation_needs_no_gpu():
    lib = capi.load()
    # bad precision, null pointer, misaligned pointer, empty lattice: rejected before any HIP call
    assert lib.lbm_init_rest(C.c_void_p(4096), 3, 8, 8, 1.0, None) == 1
    assert b"precision" in lib.lbm_last_error()
    assert lib.lbm_init_rest(None, 4, 8, 8, 1.0, None) == 1
    assert lib.lbm_init_rest(C.c_void_p(4100), 4, 8, 8, 1.0, None) == 1
    assert lib.lbm_init_rest(C.c_void_p(4096), 4, 0, 8, 1.0, None) == 1
    assert lib.lbm_pull_periodic_no_mpi(C.c_void_p(4096), C.c_void_p(4096), 4, 8, 8, 0.5, None) == 1
    assert b"overlap" in lib.lbm_last_error()
    assert lib.lbm_load_kernels(3) == 1          # bad precision tag: refused before any HIP call
    assert lib.lbm_set_launch_shape(100, 1) == 1
    assert lib.lbm_set_launch_shape(0, 0) == 0


def test_product_package_does_not_use_the_oracle():
    """The oracle is test infrastructure: nothing under lbm_amd/, src/ or include/ may import, link
    or load it."""
    for top in ("lbm_amd", "src", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for fn in files:
                if not fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".txt", ".cmake")):
                    continue
                text = open(os.path.join(dirpath, fn)).read()
                for needle in ("import oracle", "from oracle", "liblbm_oracle", "lbm_oracle", "oracle/"):
                    assert needle not in text, f"{top}/{fn} refers to the oracle ({needle!r})"


def test_no_kernel_of_the_library_uses_scratch_memory(tmp_path):
    """Every kernel keeps its rings and temporaries in registers and LDS: a spill to scratch memory
    (private segment) costs a multi-step sweep half its rate, so the code object of the built
    library is read back and every kernel's private segment must be empty and its register count
    must allow two wavefronts per SIMD (256 VGPRs)."""
    import shutil
    import subprocess

    import pytest

    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) or shutil.which(os.path.basename(t)) for t in tools):
        pytest.skip("LLVM binary tools of the ROCm installation not found")
    objcopy, bundler, readelf = tools
    fat = str(tmp_path / "fat.bin")
    subprocess.check_call([objcopy, f"--dump-section=.hip_fatbin={fat}", capi.LIB_PATH, str(tmp_path / "rest.so")])
    # one offload bundle per source file of the library, back to back in the section
    blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert starts, "no offload bundle in the library"
    notes = ""
    for n, (lo, hi) in enumerate(zip(starts, starts[1:] + [len(blob)])):
        part, dev = str(tmp_path / f"bundle{n}.bin"), str(tmp_path / f"gfx950_{n}.co")
        open(part, "wb").write(blob[lo:hi])
        subprocess.check_call([bundler, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               f"--input={part}", f"--output={dev}"])
        notes += subprocess.check_output([readelf, "--notes", dev], text=True)
    kernels = re.findall(r"\.name:\s+(_Z\S+|\w+)\s*\n(?:.*\n)*?\s*\.private_segment_fixed_size:\s+(\d+)", notes)
    scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
    vgprs = [int(x) for x in re.findall(r"\.vgpr_count:\s+(\d+)", notes)]
    assert len(scratch) > 300 and len(vgprs) == len(scratch), "kernel metadata not found in the code object"
    assert max(scratch) == 0, f"{sum(1 for x in scratch if x)} kernel(s) spill to scratch memory"
    assert max(vgprs) <= 256
    assert any("k_fused" in name for name, _ in kernels)


def test_default_arithmetic_follows_the_environment():
    """The suite starts the library with LBM_ARITHMETIC=strict (tests/conftest.py); a process that says
    nothing gets the contracted arithmetic, and ./main / bench.py / Simulation inherit that."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from lbm_amd import capi; "
            "print(capi.load().lbm_get_arithmetic(), capi.load().lbm_get_unroll_loops())" % ROOT)
    for value, want in ((None, "2 1"), ("strict", "0 1"), ("unrolled", "0 1"), ("contracted", "2 1"), ("rolled", "1 0")):
        env = {k: v for k, v in os.environ.items() if k != "LBM_ARITHMETIC"}
        if value is not None:
            env["LBM_ARITHMETIC"] = value
        out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True).strip()
        assert out == want, (value, out)
    lib = capi.load()
    assert lib.lbm_get_arithmetic() == 0          # this process: strict
    # UNROLL_LOOPS=false selects the rolled loops, =true returns to the arithmetic the process started with
    assert lib.lbm_set_unroll_loops(0) == 0 and lib.lbm_get_arithmetic() == 1
    assert lib.lbm_set_unroll_loops(1) == 0 and lib.lbm_get_arithmetic() == 0
    assert lib.lbm_set_arithmetic(2) == 0 and lib.lbm_set_unroll_loops(1) == 0 and lib.lbm_get_arithmetic() == 2
    assert lib.lbm_set_arithmetic(0) == 0


def test_contexts_carry_their_own_settings():
    """lbm_ctx_*: settings start as those in force where the context is made, change independently of
    the process-wide ones, and are current per host thread."""
    import threading
    lib = capi.load()
    assert lib.lbm_ctx_get_current() is None
    a, b = C.c_void_p(), C.c_void_p()
    assert lib.lbm_ctx_create(C.byref(a)) == 0 and lib.lbm_ctx_create(C.byref(b)) == 0
    assert lib.lbm_ctx_create(None) == 1
    assert (lib.lbm_ctx_get_layout(a), lib.lbm_ctx_get_arithmetic(a)) == (lib.lbm_get_layout(), lib.lbm_get_arithmetic())
    assert lib.lbm_ctx_set_layout(a, 1) == 0 and lib.lbm_ctx_set_arithmetic(a, 2) == 0
    assert lib.lbm_ctx_set_layout(a, 7) == 1 and lib.lbm_ctx_set_arithmetic(a, 9) == 1
    assert lib.lbm_ctx_set_launch_shape(a, 100, 1) == 1 and lib.lbm_ctx_set_launch_shape(a, 64, 6) == 0
    assert lib.lbm_ctx_set_layout(None, 0) == 1 and lib.lbm_ctx_get_layout(None) == -1
    # the process-wide settings and the other context have not moved
    assert (lib.lbm_get_layout(), lib.lbm_get_arithmetic()) == (0, 0)
    assert (lib.lbm_ctx_get_layout(b), lib.lbm_ctx_get_arithmetic(b)) == (0, 0)
    # a context made while another is current copies that one's settings
    assert lib.lbm_ctx_make_current(a) == 0 and lib.lbm_ctx_get_current() == a.value
    c = C.c_void_p()
    assert lib.lbm_ctx_create(C.byref(c)) == 0
    assert (lib.lbm_ctx_get_layout(c), lib.lbm_ctx_get_arithmetic(c)) == (1, 2)
    # what the planner sees follows the current context: eight steps per launch are the contracted
    # arithmetic's, seven the strict one's (fp32, a large lattice)
    assert lib.lbm_plan_steps(4, 16384, 16384, 100) == 8
    seen = {}

    def other_thread():
        seen["current"] = lib.lbm_ctx_get_current()
        seen["plan"] = lib.lbm_plan_steps(4, 16384, 16384, 100)
        lib.lbm_ctx_make_current(b)
        seen["after"] = lib.lbm_ctx_get_current()

    t = threading.Thread(target=other_thread)
    t.start()
    t.join()
    assert seen == {"current": None, "plan": 7, "after": b.value}
    assert lib.lbm_ctx_get_current() == a.value      # this thread's choice is its own
    assert lib.lbm_ctx_destroy(a) == 0 and lib.lbm_ctx_get_current() is None   # destroying the current one un-currents it
    assert lib.lbm_plan_steps(4, 16384, 16384, 100) == 7
    assert lib.lbm_ctx_destroy(b) == 0 and lib.lbm_ctx_destroy(c) == 0 and lib.lbm_ctx_destroy(None) == 0


def test_tuning_switches_are_named():
    lib = capi.load()
    assert lib.lbm_set_tuning(b"fused_max_steps", 3) == 0
    assert lib.lbm_plan_steps(4, 16384, 16384, 100) == 3
    assert lib.lbm_set_tuning(b"fused_max_steps", 0) == 0
    assert lib.lbm_plan_steps(4, 16384, 16384, 100) == 7
    assert lib.lbm_set_tuning(b"no_such_switch", 1) == 1 and b"unknown tuning" in lib.lbm_last_error()
    assert lib.lbm_set_tuning(None, 1) == 1
