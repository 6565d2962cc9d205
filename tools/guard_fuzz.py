#!/usr/bin/env python3
"""Randomized check of the stack guard (guard_block.hpp, stack_guard.cpp) on the HOST against the oracle's model of the reference's
65 536-item stack: random looping patterns, lines with long runs of their loop bytes around the lengths at which the search
overflows.  The guard must stop a scan exactly where the oracle fails, with the same partial output, and never elsewhere.
    python tools/guard_fuzz.py SEED SECONDS
The oracle runs in a forked child with a time limit (a line on which every attempt fails at the end of a run is quadratic)."""
import os
import random
import select
import signal
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import shim_lib  # noqa: E402
import trre_amd  # noqa: E402
from oracle_lib import Oracle, OracleError  # noqa: E402


def bounded(pat, data, seconds):
    """(output, None) | (partial, code) | None on a timeout"""
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            os.close(r)
            try:
                out = b"\x01" + Oracle(pat, "nft").scan(data)
            except OracleError as e:
                out = b"\x00" + struct.pack("<i", e.code) + e.partial
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
                return None
            b = os.read(r, 1 << 20)
            if not b:
                break
            chunks.append(b)
    finally:
        os.close(r)
        os.waitpid(pid, 0)
    out = b"".join(chunks)
    if out[:1] == b"\x01":
        return out[1:], None
    return out[5:], struct.unpack("<i", out[1:5])[0]


ATOMS = ["a", "b", "c", "ab", "[a-c]", "[ab]", ".", "a:x", "b:", ":y", "(a|b)", "(ab|a)", "(a|b|c)", "a?", "(a:x|b)", "[a-b]:z"]


def gen_pattern(rng):
    def loop():
        body = rng.choice(ATOMS)
        if rng.random() < 0.4:
            body = "(" + body + rng.choice(ATOMS) + ")"
        if rng.random() < 0.25:
            body = "(" + body + rng.choice(["*", "+"]) + rng.choice(ATOMS) + ")"
        return "(" + body + ")" + rng.choice(["*", "+", "*?", "+?", "{2,}"])
    parts = [rng.choice(["", "x", "a", "q:"]), loop()]
    if rng.random() < 0.3:
        parts.append(rng.choice(ATOMS))
        parts.append(loop())
    parts.append(rng.choice(["", "c", "c:x", "d", ":!", "(c|d)"]))
    return "".join(parts)


def main():
    seed, secs = int(sys.argv[1]), float(sys.argv[2])
    rng = random.Random(seed)
    t_end = time.time() + secs
    n = n_fail = n_to = bad = 0
    while time.time() < t_end:
        pat = gen_pattern(rng)
        try:
            p = trre_amd.Program(pat, "nft")
        except trre_amd.TrreError:
            continue
        k = p.export_guard_tables()
        if not k:
            continue
        h = struct.unpack("10I", k[:40])
        d, l_min = h[3], h[4]
        letters = [c for c in b"abc" if chr(c) in pat or "." in pat or "[" in pat] or [ord("a")]
        for _ in range(3):
            length = rng.choice([l_min - 3, l_min - 1, l_min, l_min + 1, l_min + 2, l_min + 40, 65536 // max(d - 1, 1) + 1, 66000, 70000])
            if length <= 0 or length > 70000:
                continue
            if rng.random() < 0.5:
                body = bytes([rng.choice(letters)]) * length
            else:
                unit = bytes(rng.choice(letters) for _ in range(rng.randint(1, 5)))
                body = (unit * (length // len(unit) + 1))[:length]
            if rng.random() < 0.3:                                  # a byte of another kind somewhere in the run
                at = rng.randrange(length)
                body = body[:at] + b"-" + body[at + 1:]
            tail = rng.choice([b"", b"c", b"d", b"cd"])
            data = b"head ab\n" + rng.choice([b"", b"x", b"q"]) + body + tail + b"\nafter ab c\n"
            ref = bounded(pat, data, 8)
            if ref is None:
                n_to += 1
                continue
            want, code = ref
            if code is not None and code != -3:
                continue
            g = shim_lib.stack_guard(p, data, budget=1 << 26)
            if g[0] == 2:
                n_to += 1
                continue
            if g[0] == 1:
                try:
                    pre = shim_lib.scan_like_runtime(p, data[:g[1]], geo=0) if g[1] else b""
                except RuntimeError:
                    continue                                        # (an epsilon cycle in the lines before: the table kernels' own report)
                got, why = pre + g[2], "stack"
            else:
                try:
                    got, why = shim_lib.scan_like_runtime(p, data, geo=0), None
                except RuntimeError:
                    continue                                        # (an epsilon cycle: the table kernels' own report)
            n += 1
            n_fail += code is not None
            if got != want or (why is None) != (code is None):
                bad += 1
                print("MISMATCH", repr(pat), "d=%d l_min=%d len=%d" % (d, l_min, len(data)), why, code, len(got), len(want), repr(data[8:40]), flush=True)
    print("guard fuzz: seed %d, %d lines checked (%d on which the reference fails), %d not decided in time, %d mismatches" % (seed, n, n_fail, n_to, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
