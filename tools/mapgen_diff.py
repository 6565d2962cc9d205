#!/usr/bin/env python3
"""where the memoryless kernel's output first differs from the count / emit pair's (two child processes), with the bytes around it"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import torch, trre_amd, corpora
dev = torch.device("cuda", 0)
inp = corpora.printable_lines(int(sys.argv[3]), corpora.SEED0 + 2, dev)
out = torch.empty(inp.numel() * 3 + 4096, dtype=torch.uint8, device=dev)
p = trre_amd.Program(sys.argv[1], sys.argv[2])
torch.cuda.synchronize()
for rep in range(int(sys.argv[5])):
    p.enqueue(inp, out); m = p.finish()
out[:m].cpu().numpy().tofile(sys.argv[4])
inp.cpu().numpy().tofile(sys.argv[4] + ".in")
print(m)
'''
pat, eng, size = sys.argv[1], sys.argv[2], sys.argv[3]
lens = eval(sys.argv[4]) if len(sys.argv) > 4 else None          # {byte: output length} of the bytes that do not print one byte: the tiles' totals and places are checked
for name, env in (("new", {"TRRE_MAPGEN": "1", "TRRE_MAPGEN_DBG": "/tmp/mg_dbg"} if lens is not None else {"TRRE_MAPGEN": "1"}), ("old", {"TRRE_MAPGEN": "0"})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tools")), pat, eng, size, "/tmp/mg_" + name, os.environ.get("REPS", "1")], env=e, stdout=subprocess.PIPE)
    print(name, r.stdout.decode().strip())
import numpy as np
a = np.fromfile("/tmp/mg_new", dtype=np.uint8); b = np.fromfile("/tmp/mg_old", dtype=np.uint8)
inp = np.fromfile("/tmp/mg_new.in", dtype=np.uint8)
n = min(len(a), len(b))
d = np.nonzero(a[:n] != b[:n])[0]
print("sizes", len(a), len(b), "first diff", d[0] if len(d) else None, "diffs", len(d))
if len(d):
    i = int(d[0])
    print("new:", bytes(a[max(0, i - 60):i + 60]))
    print("old:", bytes(b[max(0, i - 60):i + 60]))
# line lengths of the input
nl = np.nonzero(inp == 10)[0]
ll = np.diff(np.concatenate(([-1], nl)))
print("lines", len(nl), "longest", ll.max(), "last byte", inp[-1], "zeros", int((inp == 0).sum()), "high", int((inp >= 128).sum()))

if lens is not None:
    tab = np.ones(256, dtype=np.int64)
    for k, v in lens.items():
        tab[ord(k) if isinstance(k, str) else k] = v
    per = tab[inp]
    T = 16384          # (map_block.hpp: kMapGenTile)
    nt = (len(inp) + T - 1) // T
    pad = np.zeros(nt * T, dtype=np.int64); pad[:len(inp)] = per
    tot = pad.reshape(nt, T).sum(axis=1)
    bas = np.concatenate(([0], np.cumsum(tot)[:-1]))
    dbg = np.fromfile("/tmp/mg_dbg", dtype=np.uint64).astype(np.int64)[:16 * nt].reshape(nt, 16)
    bad_t = np.nonzero(dbg[:, 0] != tot)[0]; bad_b = np.nonzero(dbg[:, 1] != bas)[0]
    print("tiles", nt, "wrong totals", len(bad_t), bad_t[:8], "wrong places", len(bad_b), bad_b[:8])
    for i in list(bad_t[:6]):
        print(" tile", i, "total", dbg[i, 0], "want", tot[i], "block", dbg[i, 10], "clock", dbg[i, 11])
    for i in list(bad_b[:4]):
        print(" tile", i, "place", dbg[i, 1], "want", bas[i], "diff", dbg[i, 1] - bas[i])
    if dbg.shape[1] > 13:
        first = dbg[1:, 12]
        clk = dbg[1:, 13]
        print("look-back: first answers enough for %.1f%% of the tiles; clocks in the rest of it: mean %.0f, median %.0f, p90 %.0f, p99 %.0f, max %.0f; when the first answers were enough: mean %.0f; when not: mean %.0f"
              % (100.0 * first.mean(), clk.mean(), np.median(clk), np.percentile(clk, 90), np.percentile(clk, 99), clk.max(), clk[first == 1].mean() if (first == 1).any() else 0, clk[first == 0].mean() if (first == 0).any() else 0))
        # per round: the spread of the workgroups' clocks
        G = int(dbg[:, 10].max()) + 1
        t = dbg[:, 11]
        for k in (2, 5, 10):
            if (k + 1) * G <= nt:
                r = t[k * G:(k + 1) * G]
                print("  round %d: the workgroups' look-back ends spread over %.0f clocks (p10..p90 %.0f); a round takes %.0f" % (k, r.max() - r.min(), np.percentile(r, 90) - np.percentile(r, 10), np.median(t[(k + 1) * G:(k + 2) * G]) - np.median(r) if (k + 2) * G <= nt else 0))
        blk = dbg[1:, 10]
        for m in (8, 32):
            rates = [first[blk % m == x].mean() for x in range(m)]
            print("  first answers enough, by workgroup index mod %d: %s" % (m, " ".join("%.2f" % v for v in rates)))
        cu = (blk // 8) % 32
