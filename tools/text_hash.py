#!/usr/bin/env python3
"""sha256 of the gfx950 .text section of every object under a directory (default upright_amd/csrc/build): what has to stay equal when
a source edit is meant to change no production code (comment / dead-branch removal).  No GPU needed.
usage: python tools/text_hash.py [dir]"""
import hashlib
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")


def text_hash(obj):
    with tempfile.TemporaryDirectory() as td:
        co, fat = Path(td) / "dev.co", Path(td) / "fat.bin"
        subprocess.check_call([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(obj)])
        subprocess.check_call([str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.DEVNULL)
        txt = Path(td) / "text.bin"
        subprocess.check_call([str(LLVM / "llvm-objcopy"), "-O", "binary", "--only-section=.text", str(co), str(txt)])
        b = txt.read_bytes()
        return hashlib.sha256(b).hexdigest()[:16], len(b)


if __name__ == "__main__":
    d = Path(sys.argv[1] if len(sys.argv) > 1 else "upright_amd/csrc/build")
    for o in sorted(d.glob("*.o")):
        h, n = text_hash(o)
        print(f"{o.name:24s} {n:9d} B  {h}")
