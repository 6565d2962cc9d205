"""The deterministic engine on tables that are built as the input asks for them (round 5; trre_amd/csrc/lazy_block.hpp,
dft_build.cpp: LazyDft) — what the reference does (trre_dft.c:1135-1175) and what the patterns beyond any eager
determinisation run on.  CPU tier: the kernel's lane body on the host shim with the library's own explore() between the
rounds, against vectors of the compiled reference (tests/make_golden.py, section 6) and the oracle.  GPU tier: the same
through the C ABI."""
import os
import random

import pytest

import golden_lib
import shim_lib
import trre_amd
from oracle_lib import Oracle

OVER_CAP = ["((a:x)*b)|((a:y)*c)", "(a:x|b:y)*c|(a:p|b:q)*d", "(a|b)*a(a|b){18}:x", "(a|b)*a(a|b){22}:x"]


@pytest.fixture
def no_seed(monkeypatch):
    monkeypatch.setenv("TRRE_LAZY_SEED_STATES", "0")      # nothing built at compile time: every state comes from a miss


def test_patterns_beyond_the_eager_caps_compile_and_name_their_family():
    for pat in OVER_CAP:
        p = trre_amd.Program(pat, "dft")
        assert p.info.kernel == trre_amd.KERNEL_DFT_LAZY and p.allowed_kernels() == [trre_amd.KERNEL_DFT_LAZY], pat
    # what fits the eager construction keeps its families, and may be asked to run lazily (the parity tests do)
    p = trre_amd.Program("(a|b)*a(a|b){14}:x", "dft")
    assert p.info.kernel == trre_amd.KERNEL_GUIDED_GEN and trre_amd.KERNEL_DFT_LAZY in p.allowed_kernels()
    assert trre_amd.KERNEL_DFT_LAZY not in trre_amd.Program("a:b", "nft").allowed_kernels()


def test_oracle_on_the_over_cap_vectors():
    n = 0
    for pat, name, data, exp in golden_lib.lazy_cases():
        assert Oracle(pat, "dft").scan(data) == exp, (pat, name)
        n += 1
    assert n >= 17


def _lazy_vectors(spec):
    rounds_max = 0
    for pat, name, data, exp in golden_lib.lazy_cases():
        p = trre_amd.Program(pat, "dft")
        for geo in (0, 1):
            for mis in (0, 5):
                got, st, rounds = shim_lib.scan_lazy(p, data, geo, in_mis=mis, spec=spec)
                assert not st and got == exp, (pat, name, geo, mis)
                rounds_max = max(rounds_max, rounds)
    return rounds_max


def test_over_cap_vectors_on_the_shim():
    _lazy_vectors(64)


def test_marks_of_another_launch_do_not_starve_this_one():
    """the host path keeps several chunks in flight on ONE device table; an edge that a later chunk's launch had marked as listed
    (in ITS miss list) used to void the earlier chunk's lanes without being listed there — a round with a miss and an empty list, for ever.
    Marks carry their launch's id now: every unexplored edge starts each round with a foreign mark, the scan still ends with the reference's bytes."""
    n = 0
    for pat, name, data, exp in golden_lib.lazy_cases():
        p = trre_amd.Program(pat, "dft")
        got, st, rounds = shim_lib.scan_lazy(p, data, 1, spec=0, foreign_marks=True)      # (scan_lazy asserts that a void round lists a miss)
        assert not st and got == exp, (pat, name)
        n += rounds > 1
    assert n >= 5


def test_over_cap_vectors_from_nothing(no_seed):
    assert _lazy_vectors(0) > 10           # (rounds were needed: the tables really grew from the misses)


def test_every_dft_golden_vector_lazily(no_seed):
    """every DFT vector of the golden set — the configs, the quirks, the dictionaries, NULs, the diverging ones — through the lazy
    family, tables built from nothing"""
    n = n_fail = 0
    progs = {}
    for pat, name, data, engine, exp in golden_lib.cases():
        if engine != "dft":
            continue
        if pat not in progs:
            progs[pat] = trre_amd.Program(pat, "dft")
        p = progs[pat]
        geo = 1 if len(data) < 20000 else 0
        got, st, _ = shim_lib.scan_lazy(p, data, geo, spec=8)
        if exp is None:
            assert st & shim_lib.ST_DIVERGE, (pat, name)
            n_fail += 1
            continue
        assert not st and got == exp, (pat, name)
        n += 1
    assert n >= 440 and n_fail >= 10, (n, n_fail)


def test_a_miss_list_that_fills_up(no_seed):
    """more distinct misses in a round than the list holds: the rest wait for the next round"""
    pat, data = "(a|b)*a(a|b){18}:x", golden_lib.load()["inputs"]["lazy_ab"]
    want = Oracle(pat, "dft").scan(data)
    got, st, rounds = shim_lib.scan_lazy(trre_amd.Program(pat, "dft"), data, 1, miss_cap=3, spec=0)
    assert not st and got == want and rounds > 50


def test_random_over_cap_patterns_against_the_oracle():
    rng = random.Random(20250928)
    alpha = "abc"
    n_lazy = 0
    for it in range(60):
        x, y = (rng.choice(["x", "yy", "", "a", "xyz"]) for _ in range(2))
        l1 = rng.choice(["a", "[ab]", "(a|b)", "(a|bc)", "."])
        l2 = rng.choice([l1, "a", "[ab]"])
        t1, t2 = rng.sample(["b", "c", "cc", "d", ""], 2)
        pat = rng.choice(["((%s:%s)*%s)|((%s:%s)*%s)", "(%s:%s)*%s|(%s:%s)+%s", "((%s:%s)*%s|(%s:%s)*%s)e?"]) % (l1, x, t1, l2, y, t2)
        if rng.random() < 0.3:
            pat = "(a|b)*a(a|b){%d}:%s|%s" % (rng.randint(16, 24), x, pat)
        lines = []
        for _ in range(rng.randint(1, 40)):
            lines.append("".join(rng.choice(alpha + "abd e"[rng.randint(0, 4)]) for _ in range(rng.randint(0, 120))))
        data = ("\n".join(lines) + rng.choice(["\n", ""])).encode()
        try:
            o = Oracle(pat, "dft")
            want = o.scan(data)
        except Exception:
            continue
        p = trre_amd.Program(pat, "dft")
        if p.info.kernel == trre_amd.KERNEL_DFT_LAZY:
            n_lazy += 1
        got, st, _ = shim_lib.scan_lazy(p, data, rng.randint(0, 1), spec=rng.choice([0, 4, 64]))
        if want is None:
            assert st & shim_lib.ST_DIVERGE, pat
        else:
            assert not st and got == want, (pat, data)
    assert n_lazy >= 15, n_lazy


def test_memory_limit_of_the_lazy_tables(monkeypatch):
    """the tables an input needs do not fit: TRRE_E_TOO_BIG at run time, as the reference runs out of memory — not at compile time"""
    monkeypatch.setenv("TRRE_LAZY_MAX_BYTES", str(200000))
    monkeypatch.setenv("TRRE_LAZY_SEED_STATES", "0")
    p = trre_amd.Program("((a:x)*b)|((a:y)*c)", "dft")
    with pytest.raises(trre_amd.TrreError) as e:
        shim_lib.scan_lazy(p, b"a" * 3000 + b"b\n", 0, spec=0)
    assert e.value.code == trre_amd.api.E_TOO_BIG
    assert shim_lib.scan_lazy(trre_amd.Program("((a:x)*b)|((a:y)*c)", "dft"), b"aaab\n", 0)[0] == b"xxxb\n"


# ---------------------------------------------------------------------------------------------------------------------
# GPU tier
# ---------------------------------------------------------------------------------------------------------------------
def _gpu_scan(p, data, family=None):
    import torch
    p.set_kernel(family or trre_amd.KERNEL_AUTO)
    try:
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
        return p.scan_tensor(t).cpu().numpy().tobytes()
    finally:
        p.set_kernel(trre_amd.KERNEL_AUTO)


@pytest.mark.gpu
def test_over_cap_vectors_on_gpu():
    """device buffers, host buffers (chunks) and shards; tables seeded and from nothing"""
    for seed in (None, "0"):
        if seed is not None:
            os.environ["TRRE_LAZY_SEED_STATES"] = seed
        try:
            for pat, name, data, exp in golden_lib.lazy_cases():
                p = trre_amd.Program(pat, "dft")
                assert _gpu_scan(p, data) == exp, (pat, name, "device")
                q = trre_amd.Program(pat, "dft")
                assert q.scan(data) == exp, (pat, name, "host")
                assert q.scan(data, device_mask=0) == exp, (pat, name, "multi")
        finally:
            os.environ.pop("TRRE_LAZY_SEED_STATES", None)


@pytest.mark.gpu
def test_every_dft_golden_vector_lazily_on_gpu():
    n = n_fail = 0
    progs = {}
    for pat, name, data, engine, exp in golden_lib.cases():
        if engine != "dft" or not data:
            continue
        if pat not in progs:
            progs[pat] = trre_amd.Program(pat, "dft")
        p = progs[pat]
        if exp is None:
            with pytest.raises(trre_amd.TrreError) as e:
                _gpu_scan(p, data, trre_amd.KERNEL_DFT_LAZY)
            assert e.value.code == trre_amd.api.E_DIVERGES, (pat, name)
            n_fail += 1
            continue
        assert _gpu_scan(p, data, trre_amd.KERNEL_DFT_LAZY) == exp, (pat, name)
        n += 1
    assert n >= 380 and n_fail >= 10, (n, n_fail)


@pytest.mark.gpu
def test_large_buffers_lazily_on_gpu():
    """64 MiB: 2^19-state pattern on random a/b text (every state is visited: the tables grow to their full size over the
    rounds), a run-length pattern on text; slices against the oracle, the whole against two half scans"""
    import numpy as np
    rng = np.random.default_rng(5)
    n = 64 << 20
    ab = rng.choice(np.frombuffer(b"abab ab\n", dtype=np.uint8), size=n).tobytes()
    text = rng.choice(np.frombuffer(b"aaabc xyz\n", dtype=np.uint8), size=n).tobytes()
    for pat, data in [("(a|b)*a(a|b){18}:x", ab), ("((a:x)*b)|((a:y)*c)", text)]:
        p = trre_amd.Program(pat, "dft")
        got = _gpu_scan(p, data)
        cut = data.rfind(b"\n", 0, n // 2) + 1
        assert got == _gpu_scan(p, data[:cut]) + _gpu_scan(p, data[cut:]), pat
        o = Oracle(pat, "dft")
        head = data[:data.rfind(b"\n", 0, 1 << 20) + 1]
        assert got[:len(o.scan(head))] == o.scan(head), pat
        tail = data[data.find(b"\n", n - (1 << 20)) + 1:]
        want = o.scan(tail)
        assert got[len(got) - len(want):] == want, pat
