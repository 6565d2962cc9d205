#!/usr/bin/env python3
"""Run one kernel family on one corpus a few times (for rocprofv3 / quick A-B).
    python tools/kbench.py --pattern '(cat:dog|dog:cat)' --engine nft --corpus catdog --bytes $((1<<30))
    python tools/kbench.py --dict 1000 --engine dft             # cfg 5 shape
Several --case 'pattern;;engine;;corpus;;kernel' run back to back in one process."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import trre_amd
import corpora

FAMS = {v: k for k, v in trre_amd.KERNEL_NAMES.items()}

ap = argparse.ArgumentParser()
ap.add_argument("--pattern", default="[a:A-z:Z]")
ap.add_argument("--engine", default="dft")
ap.add_argument("--kernel", default="auto")
ap.add_argument("--corpus", default="printable", help="printable | catdog | dict<N>")
ap.add_argument("--bytes", type=int, default=1 << 30)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dict", type=int, default=0, help="use the seeded N-entry key:value dictionary pattern and corpus (config 5)")
ap.add_argument("--out-mis", type=int, default=0, help="misalign the output buffer by this many bytes")
ap.add_argument("--sum", action="store_true", help="print a position-weighted checksum of the output (A/B runs in separate processes compare it)")
ap.add_argument("--case", action="append", default=[], help="pattern;;engine;;corpus;;kernel (repeatable)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
cases = [c.split(";;") for c in a.case]
if not cases:
    if a.dict:
        cases = [["@dict%d" % a.dict, a.engine, "dict%d" % a.dict, a.kernel]]
    else:
        cases = [[a.pattern, a.engine, a.corpus, a.kernel]]
bufs = {}
out = torch.empty(a.bytes * 2 + 64, dtype=torch.uint8, device=dev)[a.out_mis:]
for pat, eng, corp, kern in cases:
    if pat.startswith("@dict"):
        import dictgen
        keys, vals = dictgen.make_dictionary(int(pat[5:]))
        pat = dictgen.pattern(keys, vals)
    if corp not in bufs:
        bufs[corp] = corpora.by_name(corp, a.bytes, corpora.SEED0, dev)
    inp = bufs[corp]
    p = trre_amd.Program(pat, eng)
    p.set_kernel(FAMS[kern])
    p.enqueue(inp, out); m = p.finish()
    p.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        p.enqueue(inp, out)
    m = p.finish()
    dt = (time.perf_counter() - t0) / a.steps
    chk = ""
    if a.sum:
        acc = 0
        step = 1 << 28
        for lo in range(0, m, step):
            x = out[lo:min(m, lo + step)].to(torch.int64)
            w = (torch.arange(lo, lo + x.numel(), device=dev, dtype=torch.int64) % 65521) + 1
            acc = (acc + int((x * w).sum())) % (1 << 61)
        chk = "  sum=%016x" % acc
    print("pattern=%.40s engine=%s corpus=%s kernel=%s bytes=%d out=%d  %.3f ms/step  %.1f GB/s  (events %.3f ms)%s" % (
        pat, eng, corp, trre_amd.KERNEL_NAMES[FAMS[kern] or p.info.kernel], a.bytes, m, dt * 1e3, a.bytes / dt / 1e9, p.last_kernel_ms(), chk), flush=True)
