#!/usr/bin/env python3
"""A length-preserving scan whose input holds NUL bytes: what the repair of runtime.cpp (repair_lp_nuls) costs against the
clean scan and against the whole buffer on the general family (TRRE_NO_NUL_REPAIR=1).  Checks the output against the oracle
on slices around the NULs.
    python tools/nul_bench.py [--bytes N] [--nuls K]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import trre_amd
import corpora
from oracle_lib import Oracle

ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=1 << 30)
ap.add_argument("--nuls", type=int, default=8)
ap.add_argument("--pattern", default="(cat:dog|dog:cat)")
ap.add_argument("--engine", default="nft")
ap.add_argument("--corpus", default="catdog")
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
inp = corpora.by_name(a.corpus, a.bytes, corpora.SEED0, dev)
out = torch.empty(a.bytes + 4096, dtype=torch.uint8, device=dev)
p = trre_amd.Program(a.pattern, a.engine)


def run(label):
    p.enqueue(inp, out); m = p.finish()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        p.enqueue(inp, out); m = p.finish()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    print("%-28s out=%d  %.3f ms/step  %.1f GB/s" % (label, m, dt * 1e3, a.bytes / dt / 1e9), flush=True)
    return m


run("clean")
at = [int((k + 0.37) * a.bytes / a.nuls) for k in range(a.nuls)]
inp[at] = 0
m = run("with %d NULs" % a.nuls)
# check: the head up to behind the first NUL's line, and the tail, against the oracle (the tail by its length from the end)
o = Oracle(a.pattern, a.engine)
head_in = inp[: at[0] + 4096].cpu().numpy().tobytes()
head_in = head_in[: head_in.rfind(b"\n") + 1]
want = o.scan(head_in)
assert out[: len(want)].cpu().numpy().tobytes() == want, "head differs"
tail_from = at[-1] + 1
t_in = inp[tail_from:].cpu().numpy().tobytes()
t_in = t_in[t_in.find(b"\n") + 1:]                 # from the first line start behind the last NUL
want = o.scan(t_in)
assert out[m - len(want): m].cpu().numpy().tobytes() == want, "tail differs"
print("verified: head through the first cut line and the tail behind the last one against the oracle")
