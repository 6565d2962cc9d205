#!/usr/bin/env python3
"""Audit of a gfx950 assembly listing (hipcc -S --cuda-device-only): every s_barrier whose LDS traffic the compiler may have left undrained.
Round 6 met one in k_mapgen — a ds_write at the end of a loop body, the loop head's s_barrier with no s_waitcnt lgkmcnt(0) on the back edge —
and one tile in 6 000 was counted twice.  For every s_barrier: walk back through its basic block; a `s_waitcnt .. lgkmcnt(0)` met first: fine;
a ds_write / ds atomic met first: BAD; the block's head reached: the same walk from the end of every predecessor block (labels that branch
here, and the block above when it falls through), to a depth of 12 blocks.  Prints the barriers that are BAD or undecided, by kernel."""
import re, sys
src = open(sys.argv[1]).read().split("\n")
# split into functions
funcs, cur, name = {}, [], None
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        if name: funcs[name] = cur
        name, cur = m.group(1), []
    elif name is not None:
        cur.append(ln)
if name: funcs[name] = cur
bad_total = 0
for name, lines in funcs.items():
    # basic blocks
    blocks, labels, order = {}, {}, []
    cur_lbl, cur = "<entry>", []
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            blocks[cur_lbl] = cur; order.append(cur_lbl)
            cur_lbl, cur = m.group(1), []
            continue
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.append(t.split(";")[0].strip())
    blocks[cur_lbl] = cur; order.append(cur_lbl)
    preds = {b: set() for b in blocks}
    for i, b in enumerate(order):
        ins = blocks[b]
        falls = True
        for t in ins:
            m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)", t) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", t)
            if m and m.group(1) in preds:
                preds[m.group(1)].add(b)
        if ins and (ins[-1].startswith("s_branch") or ins[-1].startswith("s_endpgm") or ins[-1].startswith("s_setpc")):
            falls = False
        if falls and i + 1 < len(order):
            preds[order[i + 1]].add(b)
    def walk(b, idx, depth, seen):
        ins = blocks[b]
        for k in range(idx - 1, -1, -1):
            t = ins[k]
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t: return "ok"
            if t.startswith("s_barrier"): return "ok"          # (an earlier barrier: what was before it is its business)
            if re.match(r"ds_(write|add|or|and|max|min|inc|dec|wrxchg|cmpst|append|consume)", t) or re.match(r"(buffer|global)_load\w*\s.*\blds\b", t): return "BAD " + t
        if depth == 0: return "deep"
        res = "ok"
        for p in preds[b]:
            if (p, len(blocks[p])) in seen: continue
            seen.add((p, len(blocks[p])))
            r = walk(p, len(blocks[p]), depth - 1, seen)
            if r.startswith("BAD"): return r + "  (via " + p + ")"
            if r == "deep": res = "deep"
        return res
    for b in order:
        for k, t in enumerate(blocks[b]):
            if t.startswith("s_barrier"):
                r = walk(b, k, int(sys.argv[2]) if len(sys.argv) > 2 else 12, set())
                if r != "ok":
                    bad_total += 1
                    print(name[:70], b, r)
print("barriers flagged:", bad_total)
