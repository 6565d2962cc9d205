"""The backtracking fallback (family 9: gen_block.hpp bt_lane): the reference's depth-first search itself, a lane per
sub-range.  What a pattern beyond every table form runs on; here its lane body runs on the host (tests/cpu_shim.cpp) on
every golden vector of the NFT engine and against the oracle."""
import random

import pytest

import corpus
import golden_lib
import shim_lib
import trre_amd
from oracle_lib import Oracle, OracleError

_progs = {}


def prog(pat):
    if pat not in _progs:
        _progs[pat] = trre_amd.Program(pat, "nft")
    return _progs[pat]


def test_golden_vectors_through_the_backtracking_fallback():
    n = n_fail = 0
    for pat, name, data, engine, exp in golden_lib.cases():
        if engine != "nft" or len(data) > 20000:
            continue
        p = prog(pat)
        assert trre_amd.KERNEL_BACKTRACK in p.allowed_kernels()
        for geo in (1, 0):
            if exp is None:
                with pytest.raises(RuntimeError, match="diverges"):
                    shim_lib.scan_like_runtime(p, data, geo=geo, family=shim_lib.BACKTRACK)
                n_fail += 1
                continue
            assert shim_lib.scan_like_runtime(p, data, geo=geo, family=shim_lib.BACKTRACK) == exp, (pat, name, geo)
            n += 1
    assert n >= 880 and n_fail == 26, (n, n_fail)


def test_a_diverging_scan_keeps_what_the_reference_had_printed():
    """the NFT binary exits 1 with its stdout flushed: the lines before the bad one and the bad line's output up to the attempt
    that does not return"""
    for pat, data in [("a:*", b"xx\nbab\nzz\n"), ("a(:y)*", b"q\n" * 40 + b"cca\nzz\n"), ("(x:y)|a(b*)*c|ad", b"xxx\nxxad\nq\n")]:
        with pytest.raises(OracleError) as e:
            Oracle(pat, "nft").scan(data)
        for geo in (1, 0):
            out, st = shim_lib.scan_backtrack(prog(pat), data, geo=geo)
            assert st & shim_lib.ST_DIVERGE and out == e.value.partial, (pat, geo, out, e.value.partial)


def test_patterns_beyond_every_table_form():
    """more than 64 nodes, a backward automaton beyond the guided limits, no fold: TRRE_E_UNSUPPORTED until round 4"""
    rng = random.Random(11)
    for pat, k_mid in [("a(a|b|c|d|e|f|g|h){12}c:x", 12), ("(:<)a(a|b|c|d|e|f|g|h){14}c:>", 14)]:
        p = prog(pat)
        assert p.info.kernel == trre_amd.KERNEL_BACKTRACK and p.allowed_kernels() == [trre_amd.KERNEL_BACKTRACK]
        o = Oracle(pat, "nft")
        lines = []
        for _ in range(60):
            k = rng.randrange(0, 40)
            s = bytes(rng.choice(b"abcdefgh") for _ in range(k))
            if rng.random() < 0.5:
                s += b"a" + bytes(rng.choice(b"abcdefgh") for _ in range(k_mid)) + b"c" + bytes(rng.choice(b"abch") for _ in range(rng.randrange(3)))
            lines.append(s)
        data = b"\n".join(lines) + b"\n"
        want = o.scan(data)
        assert want != data
        for geo in (1, 0):
            assert shim_lib.scan_like_runtime(p, data, geo=geo) == want, (pat, geo)


def test_the_limits_are_an_error_not_a_hang():
    p = prog("(a|aa)*b:x")
    data = b"a" * 40 + b"\n"                                   # exponential for the reference too
    out, st = shim_lib.scan_backtrack(p, data, budget=100000)
    assert st & shim_lib.ST_EDIT_OVERFLOW
    out, st = shim_lib.scan_backtrack(p, b"a" * 12 + b"\n" + b"aab\n", budget=100000)
    assert not st & shim_lib.ST_EDIT_OVERFLOW and out == Oracle("(a|aa)*b:x", "nft").scan(b"a" * 12 + b"\n" + b"aab\n")
    # an attempt deeper than the stack, an output longer than the path buffer
    out, st = shim_lib.scan_backtrack(prog("a*b:x"), b"a" * 100 + b"b\n", frames=64)
    assert st & shim_lib.ST_EDIT_OVERFLOW
    out, st = shim_lib.scan_backtrack(prog("(a:xyzw)*b"), b"a" * 100 + b"b\n", path_cap=128)
    assert st & shim_lib.ST_EDIT_OVERFLOW


def test_random_patterns_on_the_backtracking_fallback():
    import fuzz_oracle as F
    import time
    rng = random.Random(77)
    n = n_div = 0
    t_end = time.time() + 40
    for _ in range(270):                     # (pattern 281 of this seed is exponential for the reference's search: minutes)
        if time.time() > t_end:
            break
        pat = F.gen_soup(rng) if rng.random() < 0.2 else F.gen_expr(rng)
        if b"\0" in pat or not pat:
            continue
        try:
            p = trre_amd.Program(pat, "nft")
        except trre_amd.TrreError:
            continue
        assert trre_amd.KERNEL_BACKTRACK in p.allowed_kernels()
        data = F.gen_input(rng) + F.gen_input(rng)
        try:
            want = Oracle(pat, "nft").scan(data)
        except OracleError as e:
            if e.code != 1:
                continue
            out, st = shim_lib.scan_backtrack(p, data)
            assert st & shim_lib.ST_DIVERGE and out == e.partial, (pat, data)
            n_div += 1
            continue
        out, st = shim_lib.scan_backtrack(p, data, geo=rng.choice((0, 1)))
        if st & shim_lib.ST_EDIT_OVERFLOW:
            continue
        assert not st & shim_lib.ST_DIVERGE and out == want, (pat, data)
        n += 1
    assert n > 80, (n, n_div)


def test_match_mode_through_the_backtracking_fallback():
    """round 5: `trre -m` on a pattern beyond the guided tables runs the search in match form (FINAL accepts at the end of the
    line only, trre_nft.c:635-642) — every match vector of the compiled reference through it, and a pattern nothing else runs"""
    n = 0
    for pat, name, data, exp in golden_lib.match_cases():
        p = trre_amd.Program(pat, "nft", mode="match")
        assert trre_amd.KERNEL_BACKTRACK in p.allowed_kernels()
        for geo in (0, 1):
            out, st = shim_lib.scan_backtrack(p, data, geo)
            if exp is None:
                assert st & shim_lib.ST_DIVERGE, (pat, name)
            else:
                assert not st and out == exp, (pat, name, geo)
        n += 1
    assert n >= 120
    # (read right to left, the automaton has to know which of the next 16 bytes is a 'c': 2^16 states, the guided tables stop at 16 384)
    pat = "(a|b|c){15}c(a|b|c)*:x"
    p = trre_amd.Program(pat, "nft", mode="match")
    assert p.info.kernel == trre_amd.KERNEL_BACKTRACK and p.info.guided_rev_states == 0
    rng = random.Random(5)
    lines = [bytes(rng.choice(b"abcc") for _ in range(rng.choice((3, 15, 16, 17, 20, 40)))) for _ in range(80)]
    data = b"\n".join(lines) + b"\n"
    want = Oracle(pat, "nft").match(data)
    assert 10 < want.count(b"\n") < 80
    for geo in (0, 1):
        out, st = shim_lib.scan_backtrack(p, data, geo)
        assert not st and out == want


@pytest.mark.gpu
def test_match_mode_backtracking_and_the_larger_stacks_on_gpu():
    import torch

    def dev(p, data, fam=trre_amd.KERNEL_AUTO):
        p.set_kernel(fam)
        try:
            return p.scan_tensor(torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()).cpu().numpy().tobytes()
        finally:
            p.set_kernel(trre_amd.KERNEL_AUTO)
    for pat, name, data, exp in list(golden_lib.match_cases())[::2]:
        if exp is None or not data:
            continue
        p = trre_amd.Program(pat, "nft", mode="match")
        assert dev(p, data, trre_amd.KERNEL_BACKTRACK) == exp, (pat, name)
    pat = "(a|b|c){15}c(a|b|c)*:x"
    p = trre_amd.Program(pat, "nft", mode="match")
    assert p.info.kernel == trre_amd.KERNEL_BACKTRACK
    rng = random.Random(6)
    lines = [bytes(rng.choice(b"abcc") for _ in range(rng.choice((3, 15, 16, 17, 20, 40)))) for _ in range(20000)]
    data = b"\n".join(lines) + b"\n"
    assert dev(p, data) == Oracle(pat, "nft").match(data) and p.scan(data) == Oracle(pat, "nft").match(data)
    # attempts deeper than the first tier's stacks (4 096 frames) and path buffers (4 KiB): round 4 gave TRRE_E_UNSUPPORTED
    q = trre_amd.Program("(a:xy)*b", "nft")
    data = b"cat\n" * 3000 + b"a" * 30000 + b"b tail\n" + b"dog\n" * 3000 + b"a" * 5000 + b"b\n"
    assert dev(q, data, trre_amd.KERNEL_BACKTRACK) == Oracle("(a:xy)*b", "nft").scan(data)
    # ... and than the second tier's (65 536 / 64 KiB): an attempt of 320 000 bytes whose search holds 40 000 items — the reference runs it
    q2 = trre_amd.Program("(aaaaaaaa:xy)*b", "nft")
    data = b"zz\n" + b"a" * 320000 + b"b\n"
    want = b"zz\n" + b"xy" * 40000 + b"b\n"
    assert Oracle("(aaaaaaaa:xy)*b", "nft").scan(data) == want
    assert dev(q2, data, trre_amd.KERNEL_BACKTRACK) == want
