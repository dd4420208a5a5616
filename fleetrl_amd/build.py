"""In-tree build of libfleet_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ("fleet_kernels.hip", "fleet_capi.hip")
HEADERS = ("fleet_device.h", os.path.join("..", "..", "include", "fleet_hip.h"))
# -ffp-contract=off: no fused multiply-add contraction, so float64 results follow the reference's operation
# order bit for bit on the SOC path.  No -ffast-math for the same reason.
# -mllvm -disable-machine-licm: the machine-level loop-invariant code motion hoists every rare path's constant
# materialisation (polynomial coefficients of the exp / pow code, ...) out of the K-step loop and out of the lane loop of
# the N > 64 kernel -- 60 extra live vector registers, which halves the resident wavefronts of the multi-step kernel
# (185 -> 124 VGPRs, +31 % env-steps/s measured) and does nothing for the loop-free single-step kernel.
# -mllvm -amdgpu-kernarg-preload-count=12: the step kernel's first twelve argument dwords (env record / state record /
# action pointers, E, N) arrive in scalar registers with the wave instead of through an argument fetch (-2.5 % per step).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
         "-mllvm", "-amdgpu-kernarg-preload-count=12", "-shared"]


def lib_path() -> str:
    return os.path.join(_HERE, "libfleet_hip.so")


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def needs_build() -> bool:
    out = lib_path()
    if not os.path.isfile(out):
        return True
    deps = [os.path.join(_HERE, "csrc", f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return lib_path()
    csrc = os.path.join(_HERE, "csrc")
    extra = os.environ.get("FLEET_EXTRA_HIPCC_FLAGS", "").split()  # diagnostics only (ablation builds)
    cmd = [hipcc(), *FLAGS, *extra, *[os.path.join(csrc, s) for s in SOURCES], "-o", lib_path()]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stderr[-4000:])
    if verbose:
        print(" ".join(cmd))
    return lib_path()


if __name__ == "__main__":
    print(build(force=True, verbose=True))
