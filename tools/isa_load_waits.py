#!/usr/bin/env python
"""Which kernels complete their global loads one at a time?  Disassembles the gfx950 code object of librlhip.so and counts, per kernel, the
`s_waitcnt vmcnt(0)` instructions that are reached with exactly ONE load outstanding -- the signature of a load inside a per-lane condition
(its own exec region: the compiler finishes it before it issues the next) or of a chain of dependent accesses.  Round 4 found the sixteen
gathers per thread of k_part_scatter, the eight of k_chain_prefix and the per-record loads of the bookkeeping block this way.
usage: python tools/isa_load_waits.py [path/to/librlhip.so]        (no GPU needed; needs /opt/rocm/lib/llvm/bin)
columns: serial waits, global loads, all vmcnt(0) waits, kernel"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ranklib_amd", "lib", "librlhip.so")
with tempfile.TemporaryDirectory() as d:
    tmp = os.path.join(d, "lib.so")
    with open(lib, "rb") as f, open(tmp, "wb") as g:
        g.write(f.read())
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = [os.path.join(d, n) for n in os.listdir(d) if "amdgcn" in n]
    if not objs:
        sys.exit("no gfx950 code object found in " + lib)
    asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn"] + objs, capture_output=True, text=True).stdout
cur, stats, pend = None, {}, 0
for line in asm.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
    if m:
        cur, pend = m.group(1), 0
        stats[cur] = [0, 0, 0]
        continue
    if cur is None:
        continue
    t = line.strip()
    if t.startswith(("global_load", "flat_load", "buffer_load")):
        pend += 1
        stats[cur][1] += 1
    elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
        stats[cur][0] += pend == 1
        stats[cur][2] += 1
        pend = 0
names = subprocess.run(["c++filt"], input="\n".join(stats), capture_output=True, text=True).stdout.splitlines()
rows = sorted(((v[0], v[1], v[2], n) for (k, v), n in zip(stats.items(), names) if v[1]), reverse=True)
print("serial  loads  waits  kernel")
for r in rows:
    print("%6d %6d %6d  %s" % (r[0], r[1], r[2], r[3][:140]))
