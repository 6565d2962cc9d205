#!/usr/bin/env python3
"""Randomized differential check of the kernel bodies on the HOST (tests/cpu_shim.cpp) against the oracle: random patterns
(tests/fuzz_oracle.py's generator), random inputs with NULs and lines of every length, every kernel family a pattern
admits, both shim geometries, finished the way runtime.cpp's finish() finishes a scan (tests/shim_lib.py).  No GPU needed.
    python tools/shim_fuzz.py SEED SECONDS
The oracle runs in a forked child with a time limit (the reference's search is exponential on some pattern/input pairs)."""
import os
import random
import select
import signal
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_oracle as F  # noqa: E402
import shim_lib  # noqa: E402
import test_front_shim as T  # noqa: E402
import trre_amd  # noqa: E402
from oracle_lib import Oracle, OracleError  # noqa: E402


def bounded(fn, data, seconds):
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            os.close(r)
            try:
                out = b"\x01" + fn(data)
            except OracleError:
                out = b"\x00"
            view = memoryview(out)
            while view:
                view = view[os.write(w, view):]
        finally:
            os._exit(0)
    os.close(w)
    chunks = []
    t_end = time.time() + seconds
    try:
        while True:
            left = t_end - time.time()
            if left <= 0 or not select.select([r], [], [], left)[0]:
                os.kill(pid, signal.SIGKILL)
                raise TimeoutError()
            b = os.read(r, 1 << 20)
            if not b:
                break
            chunks.append(b)
    finally:
        os.close(r)
        os.waitpid(pid, 0)
    out = b"".join(chunks)
    if not out or out[:1] == b"\x00":
        raise OracleError(1, "scan failed")
    return out[1:]


def main():
    seed, secs = int(sys.argv[1]), float(sys.argv[2])
    rng = random.Random(seed)
    t_end = time.time() + secs
    checked = bad = 0
    while time.time() < t_end:
        pat = F.gen_soup(rng) if rng.random() < 0.2 else F.gen_expr(rng)
        if b"\0" in pat or not pat:
            continue
        data = F.gen_input(rng) + F.gen_input(rng)
        for eng in ("nft", "dft"):
            try:
                want = bounded(Oracle(pat, eng).scan, data, 5)
            except (OracleError, TimeoutError):
                want = None
            try:
                p = trre_amd.Program(pat, eng)
            except trre_amd.TrreError:
                continue
            fams = T.shim_families(p)
            if eng == "nft" and trre_amd.KERNEL_BACKTRACK in p.allowed_kernels():
                fams = fams + [shim_lib.BACKTRACK]         # (round 4: the backtracking fallback runs any NFT pattern)
            if eng == "dft" and trre_amd.KERNEL_DFT_LAZY in p.allowed_kernels() and shim_lib.DFT_LAZY not in fams:
                fams = fams + [shim_lib.DFT_LAZY]          # (round 5: the lazily determinised family runs any DFT scan pattern)
            # (round 6: the general families in ONE walk, one_block.hpp — the production geometry, tiny tiles with look-backs of 4 bytes, tight regions)
            if p.info.stream_states and shim_lib.has_g16(p.export_stream_tables()) and trre_amd.KERNEL_STREAM_GEN in p.allowed_kernels():
                fams = fams + [shim_lib.STREAM_ONE, shim_lib.STREAM_ONE_MISS, shim_lib.STREAM_ONE_TIGHT]
            if p.info.guided_rev_states and p.info.guided_rev_states <= 256 and trre_amd.KERNEL_GUIDED_GEN in p.allowed_kernels() and shim_lib.has_g16(p.export_guided_tables()[1]):
                fams = fams + [shim_lib.GUIDED_ONE, shim_lib.GUIDED_ONE_MISS]
            # (round 6: memoryless programs in one pass, map_block.hpp — the production geometry, tiles of one lane with windows of 48 bytes)
            if p.info.stream_states and shim_lib.has_mapgen(p) and trre_amd.KERNEL_STREAM_GEN in p.allowed_kernels():
                fams = fams + [shim_lib.STREAM_MAPGEN, shim_lib.STREAM_MAPGEN_TINY]
            for fam in fams:
                for geo in (1, 0):
                    try:
                        got = shim_lib.scan_like_runtime(p, data, geo=geo, family=fam)
                    except RuntimeError as e:
                        if "limits" in str(e):
                            continue              # the fallback's step budget / stack depth: an error at run time, not an answer
                        got = None                # diverges
                    if want is None:
                        continue
                    if got != want:
                        bad += 1
                        print("MISMATCH", repr(pat), eng, fam, geo, repr(data[:120]), flush=True)
                    checked += 1
    print("shim fuzz: seed %d, %d checks, %d mismatches" % (seed, checked, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
