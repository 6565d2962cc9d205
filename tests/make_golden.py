#!/usr/bin/env python3
"""Generate tests/golden/golden.json from the COMPILED REFERENCE binaries
(oracle/_ref/trre, oracle/_ref/trre_dft — built by oracle/Makefile from the
sources under /root/reference).  Run in the build container only:

    python tests/make_golden.py

The file holds data only: named inputs and, per (pattern, input, engine), the
bytes the reference printed in scan mode (or "fail" when it exited non-zero /
did not terminate).  Byte strings are zlib+base64 encoded.
"""
import base64
import json
import os
import subprocess
import sys
import tempfile
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
import corpus  # noqa: E402
import dictgen  # noqa: E402
from oracle_lib import REF_DIR, build_oracle  # noqa: E402


def enc(b):
    return base64.b64encode(zlib.compress(b, 9)).decode("ascii")


def dec_(s):
    return zlib.decompress(base64.b64decode(s)) if s != "fail" else b""


def run_ref_full(engine, pattern, path, flags=()):
    """(exit status or None on a timeout, stdout, stderr)"""
    binary = os.path.join(REF_DIR, "trre" if engine == "nft" else "trre_dft")
    try:
        p = subprocess.run([binary] + list(flags) + [pattern.encode("latin-1"), path], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=20)
    except subprocess.TimeoutExpired:
        return None, b"", b""
    return p.returncode, p.stdout, p.stderr


def run_ref(engine, pattern, path, flags=()):
    rc, out, _ = run_ref_full(engine, pattern, path, flags)
    return out if rc == 0 else None


def main():
    build_oracle()
    inputs = dict(corpus.edge_inputs())
    cases = []
    # 1. the reference's own S rows and README examples: their tiny inputs, expected text kept too
    for k, (inp, pat, exp) in enumerate(corpus.REF_S_CASES + corpus.README_CASES):
        name = "ref_%02d" % k
        inputs[name] = inp.encode("latin-1") + b"\n"
        cases.append({"pattern": pat, "input": name, "expect_nft_text": exp})
    small = ["words", "embedded_nul", "no_trailing_newline", "many_short", "high_bytes", "empty"]
    for pat in corpus.CONFIG_PATTERNS:
        for name in corpus.edge_inputs():
            cases.append({"pattern": pat, "input": name})
    for pat in corpus.QUIRK_PATTERNS + sorted({c[1] for c in corpus.REF_S_CASES}):
        for name in small:
            cases.append({"pattern": pat, "input": name})
    # 2. BASELINE config 5: the seeded 1000-entry dictionary (prefix-free keys: both engines agree) on a slice of its
    #    corpus, and a small dictionary whose keys overlap as prefixes and suffixes (the engines differ: Q9)
    keys, vals = dictgen.make_dictionary(1000)
    inputs["dict1000_slice"] = dictgen.corpus(keys, 60000, seed=7)
    cases.append({"pattern": dictgen.pattern(keys, vals), "input": "dict1000_slice"})
    import random
    rng = random.Random(11)
    inputs["abcd_soup"] = b"".join(bytes(rng.choice(b"abcd ") for _ in range(rng.randint(0, 70))) + b"\n" for _ in range(600))
    cases.append({"pattern": "ab:1|abc:2|abcd:3|b:4|bc:5|cab:6|ca:7|dab:8", "input": "abcd_soup"})
    cases.append({"pattern": "(abcd:3|abc:2|ab:1|dab:8|cab:6|ca:7|bc:5|b:4)", "input": "abcd_soup"})
    # 2b. key lists large enough for the fallback form of the stream table (StreamTables::fb_*): 160 keys of 2..8 bytes over
    #     eight letters — keys inside keys, keys that are prefixes of keys (the engines differ), in three orders — with
    #     replacement texts of 0..3 / 0..8 / 0..12 bytes, on inputs with near-misses, key upon key, NULs and keys at line ends
    for it, (order, top) in enumerate([("shuffled", 8), ("longest_first", 8), ("sorted", 12), ("shuffled", 3), ("longest_first", 12), ("sorted", 5)]):
        rng = random.Random(500 + it)
        letters = "abcdefgh"
        ks = set()
        while len(ks) < 160:
            ks.add("".join(rng.choice(letters) for _ in range(rng.randint(2, 8))))
        ks = sorted(ks)
        if order == "shuffled":
            rng.shuffle(ks)
        elif order == "longest_first":
            ks.sort(key=lambda k: (-len(k), k))
        vs = ["".join(rng.choice("XYZxyz01") for _ in range(rng.randint(0, top))) for _ in ks]
        toks = []
        while sum(map(len, toks)) < 6000:
            r = rng.random()
            if r < 0.35:
                t = rng.choice(ks)
            elif r < 0.5:
                t = rng.choice(ks)[:rng.randint(1, 8)] + rng.choice(letters)
            elif r < 0.6:
                t = rng.choice(ks) + rng.choice(ks)
            else:
                t = "".join(rng.choice(letters + "xyz") for _ in range(rng.randint(1, 9)))
            toks.append(t + rng.choice([" ", " ", "\n", ",", "\x00", ""]))
        inputs["keys160_%d" % it] = "".join(toks).encode("latin-1") + b"\n"
        cases.append({"pattern": "|".join("%s:%s" % kv for kv in zip(ks, vs)), "input": "keys160_%d" % it})
    # 3. epsilon cycles: the NFT engine's search runs round them for ever on some inputs ("stack max capacity
    #    reached", exit 1 -> "fail") and never meets them on others
    inputs.update({"eps_no_a": b"b\nxx\nzz\n", "eps_with_a": b"b\nca\n", "eps_ad": b"ad\n", "eps_abc": b"xyz\nabc\n",
                   # the bad line in the middle, with matches before the attempt that does not return: what the reference
                   # had printed when it exits is part of the vector ("nft_fail_stdout")
                   "eps_mid": b"cat one\nxx cat yy\nzzz cat bcd a cat\nnever printed cat\n",
                   "eps_tail": b"first\ncat cat cat abc"})
    for pat in ("a:*", "a(:y)*", "a(b*)*c|ad"):
        for name in ("eps_no_a", "eps_with_a", "eps_ad", "eps_abc"):
            cases.append({"pattern": pat, "input": name})
    for pat in ("cat:dog|a:*", "(cat:dog|b)*|a(:y)*", "a(b*)*c|ad|cat:x"):
        for name in ("eps_mid", "eps_tail", "eps_no_a"):
            cases.append({"pattern": pat, "input": name})
    out_cases = []
    with tempfile.TemporaryDirectory() as td:
        paths = {}
        for name, data in inputs.items():
            paths[name] = os.path.join(td, name)
            with open(paths[name], "wb") as f:
                f.write(data)
        for c in cases:
            for engine in ("nft", "dft"):
                rc, got, err = run_ref_full(engine, c["pattern"], paths[c["input"]])
                c[engine] = enc(got) if rc == 0 else "fail"
                if engine == "nft" and rc == 1 and err.startswith(b"error: stack max capacity reached"):
                    c["nft_fail_stdout"] = enc(got)      # exit() flushed stdout: everything printed before the failing attempt
            if "expect_nft_text" in c:
                want = (c["expect_nft_text"] + "\n").encode("latin-1")
                assert zlib.decompress(base64.b64decode(c["nft"])) == want, c
            out_cases.append(c)
        # 4. match mode (`trre -m`, NFT engine): the reference's own match table (test.sh M rows: the input is one line;
        #    the rows were written for -ma, the FIRST of their outputs is what -m prints) and whole-line patterns on
        #    the multi-line inputs
        match_cases = []
        for k, (inp, pat, first) in enumerate(corpus.REF_M_CASES):
            name = "refm_%02d" % k
            inputs[name] = inp.encode("latin-1") + b"\n"
            paths[name] = os.path.join(td, name)
            with open(paths[name], "wb") as f:
                f.write(inputs[name])
            got = run_ref("nft", pat, paths[name], ["-m"])
            assert got == ((first + "\n").encode("latin-1") if first is not None else b""), (inp, pat, got)
            match_cases.append({"pattern": pat, "input": name, "nft_m": enc(got)})
        for pat in corpus.MATCH_PATTERNS:
            for name in ("words", "many_short", "embedded_nul", "no_trailing_newline", "only_newlines", "empty", "abcd_soup", "eps_with_a"):
                got = run_ref("nft", pat, paths[name], ["-m"])
                match_cases.append({"pattern": pat, "input": name, "nft_m": "fail" if got is None else enc(got)})
        # 5. generator mode (`-a`: every accepting path prints, trre_nft.c:640-641,647-648): the reference's own match
        #    table with the command line its test.sh uses (`./trre -ma`, test.sh:4) — ALL outputs of all 51 rows —, whole-line
        #    patterns with several parses on multi-line inputs, and scan mode with -a (all outputs of the attempt at every
        #    position, then that position's raw byte) on short inputs.  Outputs grow fast (every start position, every
        #    parse): inputs are kept small; runs the reference does not finish in 20 s or that exit non-zero are "fail"
        #    with what they had printed (exit() flushes).
        inputs.update({"gen_lines": b"a\nab\naab\n\nabab\ncat\nb\n", "gen_text": b"cat dog\nthe cat sat\n\naaa\nabcabc",
                       "gen_nul": b"ab\0ab\nba\n"})
        for name in ("gen_lines", "gen_text", "gen_nul"):
            paths[name] = os.path.join(td, name)
            with open(paths[name], "wb") as f:
                f.write(inputs[name])
        all_cases = []

        def add_all(pat, name, flags):
            rc, got, err = run_ref_full("nft", pat, paths[name], flags)
            rec = {"pattern": pat, "input": name, "flags": "".join(flags), "out": enc(got) if rc == 0 else "fail"}
            if rc == 1 and err.startswith(b"error: stack max capacity reached"):
                rec["fail_stdout"] = enc(got)
            all_cases.append(rec)

        for k, (inp, pat, first) in enumerate(corpus.REF_M_CASES):
            add_all(pat, "refm_%02d" % k, ["-ma"])
            got = dec_(all_cases[-1]["out"])
            assert (got.split(b"\n")[0] if got else None) == (first.encode("latin-1") if first is not None else None), (inp, pat, got)
        for pat in corpus.GEN_MATCH_PATTERNS:
            for name in ("gen_lines", "gen_text", "gen_nul", "many_short", "only_newlines", "empty", "eps_with_a"):
                add_all(pat, name, ["-ma"])
        for pat in corpus.GEN_SCAN_PATTERNS:
            for name in ("gen_lines", "gen_text", "gen_nul", "no_trailing_newline", "only_newlines", "empty", "eps_with_a"):
                add_all(pat, name, ["-a"])
        # 6. (round 5) DFT patterns beyond any eager determinisation — a state per run length (two loops that print different texts
        #    before the byte that decides between them: the residuals grow with the run), 2^19 states (which of the last 19 bytes was
        #    an 'a') — the reference builds only the states its input visits (trre_dft.c:1135-1175).  ./trre_dft only.
        rng = random.Random(77)

        def soup(alpha, n_lines, top):
            return b"".join(bytes(rng.choice(alpha) for _ in range(rng.randint(0, top))) + b"\n" for _ in range(n_lines))
        inputs.update({
            "lazy_runs": b"aaab\naaac\nb\nc\n\naaaa\nxaaabyaacz\nab ac aab aac\n" + b"a" * 700 + b"b\n" + b"a" * 900 + b"c tail\n" + b"a" * 1200 + b"\n" +
                         soup(b"aaaaabc x", 120, 60) + b"aab\0aac\naac",
            "lazy_ab": soup(b"ab", 60, 90) + soup(b"aabbc ", 60, 70) + b"a" * 300 + b"\n" + b"ab" * 200 + b"\n" + b"abba\0abab\nbaab",
            "lazy_abcde": soup(b"abcde", 80, 60) + soup(b"abcdexy ", 80, 60) + b"abcabcabcabcd\nabcabce\nabc",
        })
        for name in ("lazy_runs", "lazy_ab", "lazy_abcde"):
            paths[name] = os.path.join(td, name)
            with open(paths[name], "wb") as f:
                f.write(inputs[name])
        lazy_cases = []
        for pat, names in [("((a:x)*b)|((a:y)*c)", ("lazy_runs", "lazy_ab", "words")),
                           ("((a:x)*b)|((a:yy)*c)", ("lazy_runs",)),
                           ("(a:x)*b|(a:y)*c|(a:z)*", ("lazy_runs",)),
                           ("(a:x|b:y)*c|(a:p|b:q)*d", ("lazy_abcde", "lazy_ab")),
                           ("([a-c]:x)*d|([a-c]:y)*e", ("lazy_abcde",)),
                           ("(.:x)*d|(.:y)*e", ("lazy_abcde", "words")),
                           ("(a|b)*a(a|b){14}:x", ("lazy_ab", "lazy_runs")),
                           ("(a|b)*a(a|b){18}:x", ("lazy_ab", "lazy_runs")),
                           ("(a|b)*a(a|b){22}:x", ("lazy_ab",)),
                           ("(a|b)*a(a|b){18}:x|((a:x)*c)|((a:y)*d)", ("lazy_ab", "lazy_abcde"))]:
            for name in names:
                rc, got, err = run_ref_full("dft", pat, paths[name])
                assert rc == 0, (pat, name, rc, err)
                lazy_cases.append({"pattern": pat, "input": name, "dft": enc(got)})
    doc = {"about": "scan-mode (and `-m`) outputs of the compiled reference (c0stya/trre @ 2025-05-23), see make_golden.py",
           "inputs": {k: enc(v) for k, v in inputs.items()}, "cases": out_cases, "match_cases": match_cases, "all_cases": all_cases,
           "lazy_cases": lazy_cases}
    os.makedirs(os.path.join(HERE, "golden"), exist_ok=True)
    path = os.path.join(HERE, "golden", "golden.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
    nfail = sum(1 for c in out_cases for e in ("nft", "dft") if c[e] == "fail")
    print("wrote %s: %d inputs, %d cases (%d reference failures), %d match cases, %d generator-mode cases (%d failures), %d bytes"
          % (path, len(inputs), len(out_cases), nfail, len(match_cases), len(all_cases), sum(1 for c in all_cases if c["out"] == "fail"),
             os.path.getsize(path)))


if __name__ == "__main__":
    main()
