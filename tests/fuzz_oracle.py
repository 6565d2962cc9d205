#!/usr/bin/env python3
"""Randomized differential check: oracle/libtrre_oracle.so vs the compiled
reference binaries oracle/_ref/{trre,trre_dft} (built by oracle/Makefile from
the sources under /root/reference).

Not collected by pytest (no test_ prefix); run by hand or by
tests/make_golden.py while pinning the oracle:

    python tests/fuzz_oracle.py --n 3000 --seed 1

For every random (pattern, input) pair the reference binary is run in scan
mode (`trre PATTERN FILE`).  When it exits 0 inside the timeout its stdout
must equal the oracle's output byte for byte; when it fails (exit 1, crash,
hang — the reference has unbounded loops and undefined behaviour on some
patterns) the oracle must report an error instead of an output.
"""
import argparse
import os
import random
import select
import signal
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from oracle_lib import Oracle, OracleError, REF_DIR  # noqa: E402

ALPHA = b"abcxy"
OPS = b":|*+?()[]{}.\\-,0123"


def gen_atom(rng, depth):
    r = rng.random()
    if r < 0.45 or depth > 3:
        return bytes([rng.choice(ALPHA)])
    if r < 0.55:
        return b"."
    if r < 0.65:
        a, b = sorted(rng.sample(list(ALPHA), 2))
        k = rng.random()
        if k < 0.4:
            return b"[" + bytes([a]) + b"-" + bytes([b]) + b"]"
        if k < 0.7:
            return b"[" + bytes([a]) + b":" + bytes([rng.choice(ALPHA)]) + b"-" + bytes([b]) + b":" + bytes([rng.choice(ALPHA)]) + b"]"
        return b"[" + bytes(rng.choice(ALPHA) for _ in range(rng.randint(1, 3))) + b"]"
    if r < 0.72:
        return b"\\" + bytes([rng.choice(OPS + ALPHA)])
    return b"(" + gen_expr(rng, depth + 1) + b")"


def gen_piece(rng, depth):
    a = gen_atom(rng, depth)
    r = rng.random()
    if r < 0.12:
        a += rng.choice([b"*", b"+", b"?", b"*?", b"+?", b"??"])
    elif r < 0.18:
        a += rng.choice([b"{2}", b"{,2}", b"{1,2}", b"{1,}", b"{,2}?", b"{0}", b"{2,1}"])
    return a


def gen_term(rng, depth):
    n = rng.randint(1, 3)
    t = b"".join(gen_piece(rng, depth) for _ in range(n))
    r = rng.random()
    if r < 0.35:
        t += b":" + b"".join(gen_piece(rng, depth) for _ in range(rng.randint(0, 2)))
    elif r < 0.42:
        t = b":" + t
    return t


def gen_expr(rng, depth=0):
    n = 1 if rng.random() < 0.6 else rng.randint(2, 3)
    return b"|".join(gen_term(rng, depth) for _ in range(n))


def gen_soup(rng):
    """operator soup: exercises the parser's corner cases"""
    n = rng.randint(1, 8)
    return bytes(rng.choice(ALPHA + OPS) for _ in range(n))


def gen_input(rng):
    lines = []
    for _ in range(rng.randint(1, 4)):
        n = rng.randint(0, 10)
        pool = ALPHA + (b"z<> " if rng.random() < 0.3 else b"")
        line = bytes(rng.choice(pool) for _ in range(n))
        if rng.random() < 0.04 and n:
            k = rng.randrange(n)
            line = line[:k] + b"\0" + line[k + 1:]
        lines.append(line)
    data = b"\n".join(lines)
    if rng.random() < 0.85:
        data += b"\n"
    return data


def bounded_oracle(pat, engine, data, seconds):
    """the oracle's scan in a forked child with a time limit: the reference's algorithms are exponential (or do not end) on some
    pattern / input pairs and the restatement follows them.  Returns (output, None), (None, error text) or (None, "timeout")."""
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            os.close(r)
            try:
                out = b"\x01" + Oracle(pat, engine).scan(data)
            except OracleError as e:
                out = b"\x00" + str(e).encode()
            view = memoryview(out)
            while view:
                view = view[os.write(w, view):]
        finally:
            os._exit(0)
    os.close(w)
    chunks = []
    t_end = time.time() + seconds
    timed_out = False
    try:
        while True:
            left = t_end - time.time()
            if left <= 0 or not select.select([r], [], [], left)[0]:
                os.kill(pid, signal.SIGKILL)
                timed_out = True
                break
            b = os.read(r, 1 << 20)
            if not b:
                break
            chunks.append(b)
    finally:
        os.close(r)
        os.waitpid(pid, 0)
    if timed_out:
        return None, "timeout"
    out = b"".join(chunks)
    if not out:
        return None, "oracle died"
    if out[:1] == b"\x00":
        return None, out[1:].decode(errors="replace")
    return out[1:], None


def run_ref(binary, pattern, path, timeout):
    try:
        p = subprocess.run([os.path.join(REF_DIR, binary), pattern, path],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    except subprocess.TimeoutExpired:
        return None, "timeout"
    if p.returncode != 0:
        return None, "rc=%d %s" % (p.returncode, p.stderr[:80])
    return p.stdout, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--timeout", type=float, default=0.25)
    ap.add_argument("--oracle-timeout", type=float, default=20.0, help="seconds the oracle may take on one case")
    ap.add_argument("--slow-timeout", type=float, default=120.0, help="last try for a reference run that the oracle answers and 20 s did not finish")
    ap.add_argument("--soup", type=float, default=0.3, help="fraction of operator-soup patterns")
    ap.add_argument("-v", action="store_true")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    bad = 0
    stats = {"equal": 0, "both_fail": 0, "ref_unfinished": 0}
    with tempfile.NamedTemporaryFile(delete=False) as tf:
        path = tf.name
    try:
        for it in range(args.n):
            pat = gen_soup(rng) if rng.random() < args.soup else gen_expr(rng)
            if b"\0" in pat or not pat or pat.startswith(b"-"):
                continue
            data = gen_input(rng)
            with open(path, "wb") as f:
                f.write(data)
            for engine, binary in (("nft", "trre"), ("dft", "trre_dft")):
                want, why = run_ref(binary, pat, path, args.timeout)
                got, err = bounded_oracle(pat, engine, data, args.oracle_timeout)
                if err == "timeout":
                    # the oracle does not come back: fine when the reference does not either, a finding when it does
                    if want is None:
                        stats["both_fail"] += 1
                    else:
                        bad += 1
                        print("ORACLE-TIMEOUT-REF-OK", engine, pat, data, "want", want)
                    continue
                if want is None and got is not None and why == "timeout":
                    want, why = run_ref(binary, pat, path, 20.0)   # slow-but-finite?
                if want is None and got is not None and why == "timeout":
                    # (./trre_dft allocates a table per state it meets: half a minute of page faults on a pattern like 'b|:c(.x).')
                    want, why = run_ref(binary, pat, path, args.slow_timeout)
                if want is None:
                    # reference failed / crashed / hung: the oracle must not invent an answer,
                    # unless the reference merely hit undefined behaviour that happened to kill it
                    if got is None:
                        stats["both_fail"] += 1
                    elif why == "timeout":
                        # nothing to compare with: the reference binary has not come back after --slow-timeout seconds (./trre_dft keeps a
                        # table per state: tens of gigabytes on such patterns) and the oracle — same states, hashed and small — has
                        stats["ref_unfinished"] += 1
                        print("REF-UNFINISHED-ORACLE-OK", engine, pat, data, got)
                    else:
                        bad += 1
                        print("REF-FAILED-ORACLE-OK", engine, pat, data, why, got)
                elif got is None and "error -2" in err:
                    stats["both_fail"] += 1      # undefined behaviour in the reference: whatever it printed is accidental
                elif got != want:
                    bad += 1
                    print("MISMATCH", engine, pat, data, "want", want, "got", got, "err", err)
                else:
                    stats["equal"] += 1
            if args.v and it % 200 == 0:
                print(it, stats, "bad", bad, flush=True)
    finally:
        os.unlink(path)
    print("done", stats, "bad", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
