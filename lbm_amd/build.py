"""In-tree build of the HIP shared library (liblbm_hip.so) for gfx950.

hipcc cross-compiles without a GPU, so this runs in the authoring container; the built library
travels to the GPU box with the source tree.  FMA contraction is disabled on purpose: the kernels
must round exactly where the reference's expressions round (see csrc/lbm_device.h).
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "liblbm_hip.so")
# the C ABI, and the instantiations of the multi-step kernel in five sets that compile side by side
SOURCES = [os.path.join(HERE, "csrc", "lbm_capi.hip")] + \
          [os.path.join(HERE, "csrc", f"lbm_fused_inst_{i}.hip") for i in range(1, 6)]
DEPS = SOURCES + [os.path.join(HERE, "csrc", "lbm_device.h"), os.path.join(HERE, "csrc", "lbm_fused.h"),
                  os.path.join(HERE, "csrc", "lbm_fused_instances.h"), os.path.join(ROOT, "include", "lbm_hip.h")]
OBJ_DIR = os.path.join(HERE, "csrc", "obj")

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",       # no a*b+c fusion: bit-parity with a strict evaluation of the reference
    "-fno-fast-math",
    # instruction scheduling for latency rather than for the fewest registers: a dependent packed-fp32
    # operation issues ~2.2 x later than an independent one, and the default strategy leaves a third
    # of them back to back (16384^2 fp32, seven steps per launch: 380 -> 397 GLUPS)
    "-mllvm", "-amdgpu-sched-strategy=max-ilp",
    "-fPIC",
    "-Wall",
    "-Wextra",
]
LINK_FLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs"]   # a kernel no file defines fails the link


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP library cannot be built")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile liblbm_hip.so if missing or older than its sources; return its path."""
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(OBJ_DIR, exist_ok=True)
    objects = [os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o") for src in SOURCES]

    def compile_one(pair):
        src, obj = pair
        cmd = [hipcc()] + HIPCC_FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"),
                                         "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), max(1, (os.cpu_count() or 2) // 2))) as pool:
        list(pool.map(compile_one, zip(SOURCES, objects)))    # re-raises the first failure
    cmd = [hipcc()] + LINK_FLAGS + ["-o", LIB] + objects
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
