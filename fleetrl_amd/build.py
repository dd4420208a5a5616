"""In-tree build of libfleet_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ("fleet_kernels.hip", "fleet_capi.hip", "fleet_direct.hip")
HEADERS = ("fleet_device.h", "fleet_direct.h", os.path.join("..", "..", "include", "fleet_hip.h"))  # relative to csrc/
# -ffp-contract=off: no fused multiply-add contraction, so float64 results follow the reference's operation
# order bit for bit on the SOC path.  No -ffast-math for the same reason.
# -mllvm -disable-machine-licm: the machine-level loop-invariant code motion hoists every rare path's constant
# materialisation (polynomial coefficients of the exp / pow code, ...) out of the K-step loop and out of the lane loop of
# the N > 64 kernel -- 60 extra live vector registers, which halves the resident wavefronts of the multi-step kernel
# (185 -> 124 VGPRs, +31 % env-steps/s measured) and does nothing for the loop-free single-step kernel.
# -mllvm -amdgpu-kernarg-preload-count=12: the step kernel's first twelve argument dwords (env record / state record /
# action pointers, E, N) arrive in scalar registers with the wave instead of through an argument fetch (-2.5 % per step).
# -mllvm -amdgpu-sched-strategy=max-memory-clause: the machine scheduler groups the loads of a burst into clauses instead of
# interleaving them with arithmetic (round 6: -0.7 % per launch at 4096 x 50, -1 % at 2048 x 50, +3 % on the K-step entry, +-0 on the
# large shapes; max-ilp, wave-priority insertion, early if-conversion, no post-RA scheduler, 16 preloaded dwords: all equal or worse,
# profiles/r06_experiments/compiler_scheduling_flags.log).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
         "-mllvm", "-amdgpu-kernarg-preload-count=12", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause", "-shared"]


# the library also talks to the HSA runtime directly (fleet_direct.hip: AQL packets of its own for runs of steps)
LINK = ["-L/opt/rocm/lib", "-lhsa-runtime64"]
# ... and for that the step kernels once more as a plain code object, loaded through HSA: same source, same device flags
GENCO = ["--genco", "--no-gpu-bundle-output"] + [f for f in FLAGS if f not in ("-fPIC", "-shared")]


def lib_path() -> str:
    return os.path.join(_HERE, "libfleet_hip.so")


def code_object_path(lib: str | None = None) -> str:
    """The step kernels' code object that belongs to a library: <library without .so>.gfx950.hsaco (fleet_direct.hip looks there)."""
    lib = lib or lib_path()
    return (lib[:-3] if lib.endswith(".so") else lib) + ".gfx950.hsaco"


def source_sha(extra_flags=()) -> str:
    """Identifies what a library and the code object beside it were compiled from: hash of every source and header plus the flags.
    Both artefacts carry it (`FLEET_SRC_SHA`: a host-side string in the library, a `__device__` symbol in the code object) and
    fleet_direct_open refuses a code object that is not the library's twin."""
    import hashlib

    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(_HERE, "csrc", f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join([*FLAGS, *extra_flags]).encode())
    return h.hexdigest()[:24]


def _compile(out_so: str, out_co: str, extra_flags=()) -> list:
    """Both artefacts, side by side (two hipcc processes); returns the command lines.  Raises with the compiler's message."""
    csrc = os.path.join(_HERE, "csrc")
    sha = f'-DFLEET_SRC_SHA="{source_sha(extra_flags)}"'
    cmds = [[hipcc(), *FLAGS, *extra_flags, sha, *[os.path.join(csrc, s) for s in SOURCES], *LINK, "-o", out_so],
            [hipcc(), *GENCO, *extra_flags, sha, os.path.join(csrc, "fleet_kernels.hip"), "-o", out_co]]
    procs = [subprocess.Popen(c, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for c in cmds]
    outs = [p.communicate() for p in procs]
    for p, (_, err) in zip(procs, outs):
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + err[-4000:])
    return cmds


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required)")


def needs_build() -> bool:
    deps = [os.path.join(_HERE, "csrc", f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    for out in (lib_path(), code_object_path()):
        if not os.path.isfile(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            return True
    return False


def build_variant(out: str, extra_flags) -> str:
    """Diagnostic builds (in-kernel stamps, experiments): the same sources with extra hipcc flags, written to `out` -- which
    must not be the product library's path: a build with non-standard flags never carries the product's name."""
    out = os.path.abspath(out)
    if out == os.path.abspath(lib_path()):
        raise ValueError("a variant build must not overwrite the product library " + lib_path())
    _compile(out, code_object_path(out), list(extra_flags))
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile libfleet_hip.so if it is missing or older than its sources.  Safe under `torch.distributed.run`: the ranks
    serialise on a lock file, the first one compiles into a temporary file in the same directory and renames it into place
    (a rank never maps a half-written library), the others find the fresh library when they get the lock.
    The product library is always built with FLAGS and nothing else (no environment variable can change them; diagnostic
    builds go through build_variant to a different file)."""
    import fcntl
    import tempfile

    if os.environ.get("FLEET_EXTRA_HIPCC_FLAGS"):
        raise RuntimeError("FLEET_EXTRA_HIPCC_FLAGS is no longer honoured: the product library is built with fleetrl_amd.build.FLAGS "
                           "only; use fleetrl_amd.build.build_variant(out, flags) for a diagnostic build under another file name")
    if not force and not needs_build():
        return lib_path()
    with open(os.path.join(_HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():  # another process built it while this one waited
                return lib_path()
            tmps = []
            for suffix in (".so.tmp", ".hsaco.tmp"):
                fd, tmp = tempfile.mkstemp(prefix=".libfleet_hip.", suffix=suffix, dir=_HERE)
                os.close(fd)
                tmps.append(tmp)
            try:
                cmds = _compile(tmps[0], tmps[1])
            except Exception:
                for t in tmps:
                    if os.path.exists(t):
                        os.unlink(t)
                raise
            os.chmod(tmps[0], 0o755)
            os.replace(tmps[1], code_object_path())  # the code object first: a library never meets an older one
            os.replace(tmps[0], lib_path())
            if verbose:
                for c in cmds:
                    print(" ".join(c))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib_path()


if __name__ == "__main__":
    print(build(force=True, verbose=True))
