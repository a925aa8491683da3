#!/usr/bin/env python3
"""Static instruction budget of the codec kernels: the gfx950 ISA of lz4amd_device.hip (compiled with -gline-tables-only
--save-temps) binned by the source function every instruction comes from.  Instructions of inlined platform / common
helpers are charged to the kernel-header function that was current when they were emitted.  Static counts (loop bodies
count once): the dynamic totals are in the SQ counter files next to this one.
usage: isa_budget.py <kernel mangled-name prefix> <asm file>"""
import re, sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KDIR = os.path.join(ROOT, "lz4_amd", "csrc", "kernels")
MAIN = {"lz4_decompress_kernel.h", "lz4_preparse_kernel.h", "lz4_compress_kernel.h", "lz4_hc_kernel.h"}

def func_ranges(path):
    out, cur = [], None
    for i, line in enumerate(open(path), 1):
        m = re.match(r"^(?:template\s*<[^>]*>\s*)?(?:__device__|__host__|static|inline|__forceinline__|\s)*[\w:<>\*&\s]+?\b(\w+)\s*\([^;]*$", line)
        if m and not line.startswith((" ", "\t", "//", "#", "}")) and "(" in line and m.group(1) not in ("if", "for", "while", "static_assert", "enum", "return"):
            cur = m.group(1); out.append((i, cur))
    return out

def main():
    kname, asm = sys.argv[1], sys.argv[2]
    files, ranges = {}, {}
    lines = open(asm).read().split("\n")
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3))
    for f in MAIN:
        ranges[f] = func_ranges(os.path.join(KDIR, f))
    def owner(f, ln):
        best = "?"
        for start, name in ranges.get(f, []):
            if start <= ln: best = name
            else: break
        return f.replace("lz4_", "").replace("_kernel.h", "") + ":" + best
    inside, cur = False, "?"
    cnt = collections.defaultdict(lambda: collections.Counter())
    for l in lines:
        if l.startswith(kname): inside = True; continue
        if not inside: continue
        t = l.strip()
        if t.startswith("s_endpgm"): break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            f = files.get(int(m.group(1)), "")
            if f in MAIN: cur = owner(f, int(m.group(2)))
            continue
        kind = None
        if t.startswith("v_"): kind = "VALU"
        elif t.startswith("s_waitcnt") or t.startswith("s_nop"): kind = "wait/nop"
        elif t.startswith("s_cbranch") or t.startswith("s_branch"): kind = "branch"
        elif t.startswith("s_"): kind = "SALU"
        elif t.startswith("ds_"): kind = "LDS"
        elif t.startswith(("global_", "flat_", "scratch_", "buffer_")): kind = "VMEM"
        if kind: cnt[cur][kind] += 1
    print("# static gfx950 instruction counts of %s by source function (tools/isa_budget.py)" % kname)
    print("# %-46s %6s %6s %6s %6s %6s %8s" % ("function", "VALU", "SALU", "branch", "LDS", "VMEM", "wait/nop"))
    tot = collections.Counter()
    for fn, c in sorted(cnt.items(), key=lambda kv: -sum(kv[1].values())):
        print("  %-46s %6d %6d %6d %6d %6d %8d" % (fn, c["VALU"], c["SALU"], c["branch"], c["LDS"], c["VMEM"], c["wait/nop"]))
        tot.update(c)
    print("  %-46s %6d %6d %6d %6d %6d %8d" % ("total", tot["VALU"], tot["SALU"], tot["branch"], tot["LDS"], tot["VMEM"], tot["wait/nop"]))

main()
