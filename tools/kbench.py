#!/usr/bin/env python3
"""Run one kernel family on the bench workload a few times (for rocprofv3 / quick A-B)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trre_amd
from bench import synth_lines

ap = argparse.ArgumentParser()
ap.add_argument("--pattern", default="[a:A-z:Z]")
ap.add_argument("--engine", default="dft")
ap.add_argument("--kernel", default="auto")
ap.add_argument("--bytes", type=int, default=1 << 30)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dict", type=int, default=0, help="use the seeded N-entry key:value dictionary pattern and corpus (config 5)")
a = ap.parse_args()
fam = {"auto": 0, "bytemap": 1, "tile_lp": 2, "tile_gen": 3, "stream_lp": 4, "stream_gen": 5}[a.kernel]
dev = torch.device("cuda", 0)
if a.dict:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import dictgen
    keys, vals = dictgen.make_dictionary(a.dict)
    a.pattern = dictgen.pattern(keys, vals)
    base = torch.frombuffer(bytearray(dictgen.corpus_fast(keys, min(a.bytes, 64 << 20))), dtype=torch.uint8).to(dev)
    inp = base.repeat((a.bytes + base.numel() - 1) // base.numel())[:a.bytes].contiguous()
    inp[-1] = 10
else:
    inp = synth_lines(a.bytes, 0x7472726531, dev)
out = torch.empty(a.bytes * 2 + 64, dtype=torch.uint8, device=dev)
p = trre_amd.Program(a.pattern, a.engine)
p.set_kernel(fam)
p.enqueue(inp, out); m = p.finish()
p.set_profiling(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    p.enqueue(inp, out)
m = p.finish()
dt = (time.perf_counter() - t0) / a.steps
print("pattern=%.40s engine=%s kernel=%s bytes=%d out=%d  %.3f ms/step  %.1f GB/s  (events %.3f ms)" % (
    a.pattern, a.engine, trre_amd.KERNEL_NAMES[p.info.kernel], a.bytes, m, dt * 1e3, a.bytes / dt / 1e9, p.last_kernel_ms()))
