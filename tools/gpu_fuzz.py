#!/usr/bin/env python3
"""Randomized differential check of the GPU path (C ABI, every kernel family a pattern admits)
against the oracle.  Run on the GPU box:   python tools/gpu_fuzz.py --seconds 120 --seed 1

Random patterns (the generator of tests/fuzz_oracle.py over the alphabet abcxy), random inputs over
the same alphabet with lines of very different lengths (empty lines, lines longer than a lane's
sub-range, NULs now and then, with and without a final newline), random buffer sizes and random
misalignment of the input and output tensors.  --dict P: that share of the patterns are key:value lists of 80..250 keys
(large stream tables: the comb-packed fallback form)."""
import argparse
import os
import random
import signal
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import trre_amd  # noqa: E402
from fuzz_oracle import gen_expr  # noqa: E402
from oracle_lib import Oracle, OracleError  # noqa: E402

ALPHA = b"abcxy"


DICT_ALPHA = b"abcdefgh"       # (more byte classes: with 8 the tables of a key list stay small enough for the 16-byte form)


def gen_input(rng, n, alpha=ALPHA):
    out = bytearray()
    while len(out) < n:
        r = rng.random()
        if r < 0.05:
            ln = 0
        elif r < 0.85:
            ln = rng.randint(1, 160)
        elif r < 0.97:
            ln = rng.randint(1000, 6000)          # longer than a lane's sub-range
        else:
            ln = rng.randint(20000, 70000)        # longer than an LDS tile
        line = bytes(rng.choice(alpha) for _ in range(min(ln, 64)))
        line = (line * (ln // max(len(line), 1) + 1))[:ln]
        if rng.random() < 0.01 and ln:
            k = rng.randrange(ln)
            line = line[:k] + b"\0" + line[k + 1:]
        out += line + b"\n"
    out = bytes(out[:n])
    if rng.random() < 0.5 and not out.endswith(b"\n"):
        out = out[:-1] + b"\n"
    return out


def gen_replacement_list(rng):
    """alternation of literal key:value pairs, values of 0..12 bytes (inline, split and pooled outputs)"""
    pairs = []
    same_len = rng.random() < 0.4                      # length-preserving lists: the window kernels (keys up to 8 bytes)
    for _ in range(rng.randint(1, 6)):
        k = bytes(rng.choice(ALPHA) for _ in range(rng.randint(1, 8 if same_len else 4)))
        n = len(k) if same_len else rng.choice([0, 1, 3, 4, 5, 6, 7, 8, 9, 12])
        v = bytes(rng.choice(b"xyzXYZ01") for _ in range(n))
        pairs.append(k + b":" + v)
    pat = b"|".join(pairs)
    return b"(" + pat + b")" if rng.random() < 0.5 else pat


def gen_dictionary(rng):
    """a key:value list large enough for the fallback form of its stream table (StreamTables::fb_*): 80..250 keys of 2..8
    bytes over DICT_ALPHA, as the input of these cases (keys inside keys, keys that are prefixes of keys), values of 0..12 bytes"""
    keys = set()
    n = rng.randint(80, 250)
    while len(keys) < n:
        keys.add(bytes(rng.choice(DICT_ALPHA) for _ in range(rng.randint(2, 8))))
    keys = sorted(keys)             # (never the set's own order: it depends on PYTHONHASHSEED — a seed must name ONE run)
    order = rng.random()
    if order < 0.4:
        keys.sort(key=lambda k: (-len(k), k))           # the longer key first: NFT priority waits for it
    elif order < 0.7:
        keys.sort()
    else:
        rng.shuffle(keys)
    top = rng.choice([3, 8, 8, 12])
    return b"|".join(k + b":" + bytes(rng.choice(b"xyzXYZ01") for _ in range(rng.randint(0, top))) for k in keys)


def bounded(fn, data, seconds=20):
    """fn(data) in a forked child, given up after `seconds` (TimeoutError): the reference's backtracking search — and so
    the oracle's — is exponential on some pattern/input pairs, and a C call cannot be interrupted by an alarm.  The child
    only runs the CPU oracle (it never touches the GPU)."""
    import select
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        code = 1
        try:
            os.close(r)
            try:
                out = b"\x01" + fn(data)
            except OracleError:
                out = b"\x00"
            view = memoryview(out)
            while view:
                view = view[os.write(w, view):]
            code = 0
        finally:
            os._exit(code)
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verbose", action="store_true", help="print every case before it runs (to find a crashing one)")
    ap.add_argument("--dict", type=float, default=0.03, help="share of patterns that are large key:value lists (fallback form)")
    a = ap.parse_args()
    rng = random.Random(a.seed)
    t_end = time.time() + a.seconds
    n_pat = n_run = n_skip = bad = 0
    fams = {}
    while time.time() < t_end:
        if n_pat % 100 == 99:
            print("... %d patterns, %d scans, %d mismatches" % (n_pat, n_run, bad), flush=True)
        r = rng.random()
        alpha = DICT_ALPHA if r < a.dict else ALPHA
        pat = (gen_dictionary(rng) if r < a.dict else gen_replacement_list(rng) if r < a.dict + 0.3 else gen_expr(rng)).decode("latin-1")
        eng = rng.choice(["dft", "nft"])
        if a.verbose:
            print("pattern %r eng=%s" % (pat, eng), flush=True)      # (before it is compiled: a compile that does not end shows here)
        try:
            o = Oracle(pat, eng)
            p = trre_amd.Program(pat, eng)
        except (OracleError, trre_amd.TrreError):
            n_skip += 1
            continue
        n_pat += 1
        for _ in range(2):
            n = rng.choice([1, 7, 100, 5000, 70000, 300000, 1500000])
            if alpha is DICT_ALPHA and eng == "nft":
                n = min(n, 70000)                  # (the NFT oracle walks a few hundred alternatives per position)
            data = gen_input(rng, n, alpha)
            try:
                want = bounded(o.scan, data)
            except (OracleError, TimeoutError):
                n_skip += 1
                continue
            mis_in, mis_out = rng.choice([0, 0, 1, 5, 16, 33]), rng.choice([0, 0, 1, 5, 16, 33])
            buf = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
            buf[mis_in:mis_in + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            tin = buf[mis_in:mis_in + len(data)]
            for fam in p.allowed_kernels():
                if fam == trre_amd.KERNEL_BACKTRACK and len(data) > 70000:
                    continue                       # (the fallback on megabytes: seconds per case where the search backtracks)
                p.set_kernel(fam)
                if a.verbose:
                    print("case pat=%r eng=%s fam=%d n=%d mis=(%d,%d)" % (pat, eng, fam, len(data), mis_in, mis_out), flush=True)
                cap = max(len(want), len(data)) + 64
                obuf = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
                tout = obuf[mis_out:mis_out + cap]
                try:
                    m = p.scan_tensor(tin, out=tout)
                    got = m.cpu().numpy().tobytes() if hasattr(m, "cpu") else bytes(tout[:m].cpu().numpy())
                except trre_amd.TrreError as e:
                    if fam == trre_amd.KERNEL_BACKTRACK and e.code == trre_amd.api.E_UNSUPPORTED:
                        n_skip += 1                # (the fallback's run-time limits: an attempt deeper than its stack, its step budget)
                        continue
                    got = ("ERR " + str(e)).encode()
                n_run += 1
                fams[fam] = fams.get(fam, 0) + 1
                if got != want:
                    bad += 1
                    print("MISMATCH pat=%r eng=%s fam=%d n=%d mis=(%d,%d) got=%d want=%d" % (pat, eng, fam, len(data), mis_in, mis_out, len(got), len(want)), flush=True)
                    if bad > 20:
                        return 1
        # match mode (`trre -m`, NFT engine): lines of at most 8 bytes — the oracle's whole-line search is exponential in the
        # line length on some patterns (and a C call cannot be interrupted by the alarm)
        if eng == "nft" and rng.random() < 0.5:
            try:
                pm = trre_amd.Program(pat, "nft", mode="match")
            except trre_amd.TrreError:
                continue
            lines = [bytes(rng.choice(ALPHA) for _ in range(rng.randint(0, 8))) for _ in range(rng.randint(1, 400))]
            data = b"\n".join(lines) + (b"\n" if rng.random() < 0.7 else b"")
            try:
                want = bounded(o.match, data, 5)
            except (OracleError, TimeoutError):
                n_skip += 1
                continue
            try:
                got = pm.scan_tensor(torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()).cpu().numpy().tobytes() if data else b""
            except trre_amd.TrreError as e:
                got = ("ERR " + str(e)).encode()
            n_run += 1
            fams["match"] = fams.get("match", 0) + 1
            if got != want:
                bad += 1
                print("MISMATCH (match mode) pat=%r n=%d got=%d want=%d" % (pat, len(data), len(got), len(want)), flush=True)
    print("gpu fuzz: %d patterns, %d scans (%s), %d skipped, %d mismatches" % (n_pat, n_run, fams, n_skip, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
