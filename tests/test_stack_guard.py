"""The stack guard (guard_block.hpp, stack_guard.cpp): the reference's search keeps at most 65 536 untried alternatives
(trre_nft.c:35-36,548-556) and exits 1 on the push after that — a greedy loop over a run of 65 536 bytes fails there although a
match exists.  The guard finds the lines long enough for that and runs the reference's search on them, state by state; here its
bodies run on the host (tests/cpu_shim.cpp) against the oracle, which models the limit (and is pinned to the compiled reference
on it: tests/test_oracle_golden.py)."""
import random
import struct

import pytest

import shim_lib
import trre_amd
from oracle_lib import Oracle, OracleError


def guard_info(p):
    k = p.export_guard_tables()
    if not k:
        return None
    h = struct.unpack("10I", k[:40])
    return {"states": h[1], "d": h[3], "l_min": h[4], "window": h[5]}


def expected(pat, data):
    """(the reference's output, None) or (what it had printed, the error)"""
    try:
        return Oracle(pat, "nft").scan(data), None
    except OracleError as e:
        return e.partial, e


def guarded_scan(p, pat, data, in_mis=0):
    """what the runtime answers: the scan's output, or (partial, 'stack') when the guard stops it"""
    r = shim_lib.stack_guard(p, data, in_mis=in_mis)
    if r is None or r[0] != 1:
        return shim_lib.scan_like_runtime(p, data, geo=0), None
    hit, ls, part = r
    pre = shim_lib.scan_like_runtime(p, data[:ls], geo=0) if ls else b""
    return pre + part, "stack"


def test_which_patterns_are_guarded():
    for pat, d in [(" +: ", 1), ("(a|b)*c:x", 2), ("a*b:x", 1), ("[a-z]+ing:X", 2), ("((a|b)*c)*d:x", 3)]:
        g = guard_info(trre_amd.Program(pat, "nft"))
        assert g and g["d"] == d and g["l_min"] == 65536 // d - 1 and g["window"] % 16 == 0 and 2 * g["window"] <= g["l_min"], (pat, g)
    # no loop: an attempt reads a bounded number of bytes, the stack cannot fill
    for pat in ["cat:dog", "(cat:dog|dog:cat)", "a:xyz", "[a:A-z:Z]", "(a|b|c)(d|e)?:x"]:
        assert guard_info(trre_amd.Program(pat, "nft")) is None, pat
    # the deterministic engine has no such stack
    assert trre_amd.Program(" +: ", "dft").export_guard_tables() == b""


def test_the_pinned_case_and_its_neighbours():
    pat = " +: "
    p = trre_amd.Program(pat, "nft")
    for k in (65534, 65535, 65536, 65537, 70000):
        for data in (b"x" + b" " * k + b"y\n", b"ab  cd\n" * 7 + b"x" + b" " * k + b"y\nzz  z\n", b"q q\na  b" + b" " * k):
            want, err = expected(pat, data)
            got, why = guarded_scan(p, pat, data)
            assert got == want and (why is None) == (err is None), (k, len(data), why, err)
    # the first line that fails is the one reported, whatever comes after it
    data = b"u  v\n" + b" " * 70000 + b"\nmiddle\n" + b" " * 80000 + b"\n"
    want, err = expected(pat, data)
    assert err is not None and guarded_scan(p, pat, data) == (want, "stack")
    # a NUL ends its record: what lies behind it does not count (Q2)
    data = b"a  b\0" + b" " * 70000 + b"\nnext  line\n"
    want, err = expected(pat, data)
    assert err is None and guarded_scan(p, pat, data) == (want, None)


def test_thresholds_of_nested_loops_against_the_oracle():
    """lines around 65 536 / D bytes made of what keeps the loops going, closed by what lets the first attempt accept (the search
    is linear then; a line on which every attempt fails at its end is quadratic for the reference too): the guard's bound must
    never miss a failure"""
    rng = random.Random(9)
    cases = [("(a|b)*c:x", b"ab", b"c"), ("a*b:x", b"a", b"b"), ("[a-z]+ing:X", b"abcxyz", b"ing"), ("((a|b)*c)*d:x", b"abc", b"cd"),
             ("(a:x|b)*", b"ab", b""), ("(ab|a)*c:y", b"a", b"c"), ("(.:x)*.*", b"qz", b""), ("((a|b)|(c|a))*d:x", b"abc", b"d"),
             ("(a?b)*c:x", b"b", b"c"), ("(a|b)+:<>", b"ab", b"")]
    n_fail = n_ok = 0
    for pat, alphabet, tail in cases:
        p = trre_amd.Program(pat, "nft")
        g = guard_info(p)
        assert g, pat
        for k in sorted({g["l_min"] - 40, g["l_min"] - 1, g["l_min"], g["l_min"] + 1, g["l_min"] + 2, 65536 // max(g["d"] - 1, 1) + 2, 66000, 70000}):
            if k <= 0 or k > 70000:
                continue
            body = bytes(rng.choice(alphabet) for _ in range(k)) if rng.random() < 0.5 else bytes([alphabet[0]]) * k
            data = b"head line\n" + body + tail + b"\nafter\n"
            want, err = expected(pat, data)
            if err is not None and err.code != -3:
                continue                                       # (an epsilon cycle entered: the table kernels' own business)
            got, why = guarded_scan(p, pat, data)
            assert got == want and (why is None) == (err is None), (pat, k, why, err, len(got), len(want))
            n_fail += err is not None
            n_ok += err is None
    assert n_fail > 15 and n_ok > 15, (n_fail, n_ok)


def test_ordinary_text_costs_a_probe_and_nothing_else():
    import corpus
    rng = random.Random(1)
    data = corpus.printable_lines(rng, 1 << 20)
    for pat in [" +: ", "(a|b)*c:x", "((a|b)*c)*d:x"]:
        assert shim_lib.stack_guard(trre_amd.Program(pat, "nft"), data) == (0, 0, b"")


def test_an_undecided_line_is_said_to_be_undecided():
    p = trre_amd.Program("(a|aa)*b:x", "nft")
    g = guard_info(p)
    data = b"a" * (g["l_min"] + 10) + b"\n"                      # exponential for the reference too
    assert shim_lib.stack_guard(p, data, budget=1000)[0] == 2


def test_match_mode_runs_the_same_search():
    pat = "(a|b)*c"
    p = trre_amd.Program(pat, "nft", mode="match")
    assert guard_info(p)["d"] == 2
    data = b"abc\nxx\n" * 50 + b"ab" * 35000 + b"c\nabc\n"
    with pytest.raises(OracleError) as e:
        Oracle(pat, "nft").match(data)
    hit, ls, part = shim_lib.stack_guard(p, data)
    assert (hit, ls, part) == (1, 7 * 50, b"") and e.value.partial == b"abc\n" * 50
    ok = b"abc\nxx\n" * 50 + b"ab" * 16000 + b"c\nabc\n"
    assert shim_lib.stack_guard(p, ok)[0] == 0 and Oracle(pat, "nft").match(ok).endswith(b"c\nabc\n")


def test_a_run_too_short_to_overflow_is_not_searched():
    """the probe's windows are half the run an overflow needs at least: a line may hold whole windows of loop bytes and still no
    stretch on which an attempt could hold 65 536 items — one pass over the line says so, nothing is searched (a search of such
    lines, quadratic where every attempt fails at the end of the run, was the guard's worst case: 14 s a line)"""
    pat = "(a|b)*c:x"
    p = trre_amd.Program(pat, "nft")
    g = guard_info(p)
    for tail in (b"c", b""):
        data = b"head\n" + b"x" + b"ab" * ((g["l_min"] - 100) // 2) + tail + b"\nafter\n"
        assert shim_lib.stack_guard(p, data, budget=1000) == (0, 0, b""), tail      # (budget 1000: a search would be "not decided")
    data = b"head\n" + b"x" + b"ab" * ((g["l_min"] - 100) // 2) + b"c\nafter\n"
    assert guarded_scan(p, pat, data) == (Oracle(pat, "nft").scan(data), None)
    # two such runs in one line, two bytes of another kind between them (one could be the byte of the one state outside the loop): none long enough
    data = b"ab" * 12000 + b"--" + b"ab" * 12000 + b"c\n"
    assert shim_lib.stack_guard(p, data, budget=1000) == (0, 0, b"")
    assert guarded_scan(p, pat, data) == (Oracle(pat, "nft").scan(data), None)


def test_random_looping_patterns_at_their_thresholds():
    """a short run of tools/guard_fuzz.py: random looping patterns, lines with runs of their loop bytes around the lengths at which
    the reference's search overflows — the guard stops a scan exactly where the oracle fails, with the same partial output"""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "guard_fuzz.py")
    r = subprocess.run([sys.executable, script, "20250928", "15"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=200)
    out = r.stdout.decode("latin-1")
    assert r.returncode == 0 and " 0 mismatches" in out, out[-2000:]
