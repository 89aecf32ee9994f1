"""Where a kernel's scratch traffic sits: compiles one part of the production QP kernel (or upr_api.hip with 'api') for gfx950 with line
tables and counts scratch_load / scratch_store instructions per kernel and per 25-line bucket of the source.  A spill COUNT says
little; a register array that LIVES in scratch shows up here as stores right behind its loads and reloads at every use
(DESIGN.md "Registers that lived in scratch").   python tools/scratch_by_line.py <part 0..6 | api> [top N] [extra -D flags ...]"""
import collections, os, re, subprocess, sys, tempfile
part = sys.argv[1] if len(sys.argv) > 1 else "0"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
extra = sys.argv[3:]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "upright_amd", "csrc")
src = "upr_api.hip" if part == "api" else "upr_qp3_inst.hip"
out = os.path.join(tempfile.gettempdir(), "scratch_by_line_%s.s" % part)
cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed",
       "-gline-tables-only", "-S", "--cuda-device-only", src, "-o", out] + ([] if part == "api" else ["-DUPR_QP3_PART=" + part]) + extra
subprocess.run(cmd, cwd=root, check=True, stderr=subprocess.DEVNULL)
files, cnt, cur, fn = {}, collections.defaultdict(collections.Counter), None, None
for l in open(out):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[m.group(1)] = (m.group(3) or m.group(2)); continue
    m = re.match(r'^(_Z\w+):', l)
    if m: fn = m.group(1); continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(m.group(1), m.group(1)).split('/')[-1], int(m.group(2))); continue
    if cur and ('scratch_load' in l or 'scratch_store' in l):
        cnt[fn][(cur[0], cur[1] // 25 * 25, 'load' if 'load' in l else 'store')] += 1
for f, c in sorted(cnt.items(), key=lambda x: -sum(x[1].values())):
    print(subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip()[:110], "| scratch instructions:", sum(c.values()))
    for (file, line, kind), n in sorted(c.items(), key=lambda x: -x[1])[:top]: print("     %-18s lines %4d-%4d  %-5s %d" % (file, line, line + 24, kind, n))
