#!/usr/bin/env python3
"""Developer aid: the gfx950 ISA of one kernel, every instruction tagged with the source file:line it comes from.
usage: isa_annot.py <kernel mangled-name prefix> <asm file made with -gline-tables-only -S> > out.txt"""
import re, sys
kname, asm = sys.argv[1], sys.argv[2]
lines = open(asm).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m: files[int(m.group(1))] = m.group(3).split("/")[-1]
inside, cur = False, ("?", 0)
for l in lines:
    if l.startswith(kname) and l.rstrip().split(";")[0].rstrip().endswith(":"): inside = True; continue
    if not inside: continue
    t = l.strip()
    if t.startswith("s_endpgm"): break
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m: cur = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
    if re.match(r"^(v_|s_|ds_|global_|flat_|buffer_|scratch_)", t) or re.match(r"^\.LBB\S*:", t):
        print("%-26s %5d  %s" % (cur[0][:26], cur[1], t.split(";")[0].rstrip()))
