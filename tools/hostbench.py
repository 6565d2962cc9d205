#!/usr/bin/env python3
"""PCIe-inclusive rate of trre_scan_host / trre_scan_host_multi on pageable host buffers (best of a few calls)."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import trre_amd
import corpora

ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=1 << 30)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
L = trre_amd.api.lib()
for pat, eng, corp in (("[a:A-z:Z]", "dft", "printable"), ("(cat:dog|dog:cat)", "nft", "catdog"), ("a:xyz", "dft", "printable")):
    host = corpora.by_name(corp, a.bytes, corpora.SEED0, "cuda").cpu().numpy()
    out = np.zeros(a.bytes + a.bytes // 4 + 4096, dtype=np.uint8)
    p = trre_amd.Program(pat, eng)
    m = ctypes.c_size_t()
    for name, call in (("scan_host", lambda: L.trre_scan_host(p._h, host.ctypes.data_as(ctypes.c_char_p), a.bytes, out.ctypes.data_as(ctypes.c_char_p), out.size, ctypes.byref(m), 0)),
                       ("scan_host_multi", lambda: L.trre_scan_host_multi(p._h, host.ctypes.data_as(ctypes.c_char_p), a.bytes, out.ctypes.data_as(ctypes.c_char_p), out.size, ctypes.byref(m), 0))):
        best = 1e9
        for _ in range(a.reps):
            t0 = time.perf_counter(); rc = call(); dt = time.perf_counter() - t0
            assert rc == 0, rc
            best = min(best, dt)
        print("%-18s %-20s %s  %.2f GB/s  (%.1f ms, out %d)" % (name, pat, eng, a.bytes / best / 1e9, best * 1e3, m.value), flush=True)
