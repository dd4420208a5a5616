"""Static instruction mix of a step-kernel instance from `hipcc -S` output (build container).
usage: python tools/isa_mix.py [G DEG MULTI WIDE] (default 64 2 0 0); writes the kernel's ISA to /tmp/kernel_<...>.s"""
import re, subprocess, sys, os
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
key = tuple(sys.argv[1:5]) if len(sys.argv) >= 5 else ("64", "2", "0", "0")
extra = sys.argv[5:]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
                "-mllvm", "-amdgpu-kernarg-preload-count=12", "-S", "--cuda-device-only", *extra,
                os.path.join(ROOT, "fleetrl_amd/csrc/fleet_kernels.hip"), "-o", "/tmp/k.s"], check=True, stderr=subprocess.DEVNULL)
lines = open("/tmp/k.s").read().split("\n")
start = None
for i, l in enumerate(lines):
    m = re.match(r"_ZN12_GLOBAL__N_117fleet_step_kernelILi(\d+)ELi(\d)ELb(\d)ELb(\d)E\w*:", l)
    if m and m.groups() == key:
        start = i
        break
assert start is not None, key
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
open("/tmp/kernel_%s.s" % "_".join(key), "w").write("\n".join(body))
ins = [l.strip().split()[0] for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = Counter()
for x in ins:
    if x.startswith("v_"):
        k = "v_*_f64" if "f64" in x else "v_cmp" if x.startswith("v_cmp") else "v_cndmask" if "cndmask" in x else "v_mov" if "mov" in x else "v_other"
    elif x.startswith("s_"):
        k = "s_waitcnt" if "waitcnt" in x else "s_branch" if ("branch" in x or "cbranch" in x) else "s_other"
    elif x.startswith(("global_", "buffer_", "flat_", "scratch_")):
        k = "vmem_load" if "load" in x else "vmem_store"
    else:
        k = "other"
    c[k] += 1
print(key, "static instructions:", len(ins), dict(c))
