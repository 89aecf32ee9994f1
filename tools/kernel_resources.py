#!/usr/bin/env python3
"""Resource usage of every ahead-of-time instantiation of the production QP kernel (and of upr_api.hip's kernels with `api`):
VGPRs, spilled VGPRs, scratch bytes, LDS, occupancy -- from the compiler's kernel-resource-usage remarks (no GPU needed).
    python tools/kernel_resources.py [part ...| api] [-DUPR_...]"""
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parents[1] / "upright_amd" / "csrc"


def usage(part, extra=()):
    src = "upr_api.hip" if part == "api" else "upr_qp3_inst.hip"
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", *extra,
           *([] if part == "api" else [f"-DUPR_QP3_PART={part}"]), "--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, check=True)
    out, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1); out[name] = {}
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and name:
                out[name][key] = int(m.group(1))
    return out


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    parts = [a for a in sys.argv[1:] if not a.startswith("-")] or [str(k) for k in range(6)]
    with ThreadPoolExecutor(4) as ex:
        res = list(ex.map(lambda p: usage(p, extra), parts))
    for p, r in zip(parts, res):
        for n, u in r.items():
            if "qp3" in n or p == "api":
                print(f"part {p}  {demangle(n)[:110]:110s}  " + "  ".join(f"{k} {v}" for k, v in u.items()))
