"""The oracle (oracle/trre_oracle.c) against the reference's own results.

Pins the CPU restatement to (a) the reference's scan-mode test rows and README
examples and (b) outputs of the compiled reference on the configuration
patterns and quirk probes (tests/golden/golden.json, made by make_golden.py)."""
import pytest

import corpus
import golden_lib
from oracle_lib import Oracle, OracleError, ref_available, ref_scan, scan_mt


def test_reference_scan_rows_and_readme():
    # test.sh `S` rows and README examples: NFT engine (./trre)
    for inp, pat, exp in corpus.REF_S_CASES + corpus.README_CASES:
        got = Oracle(pat, "nft").scan(inp.encode("latin-1") + b"\n")
        assert got == exp.encode("latin-1") + b"\n", (inp, pat)


def test_oracle_matches_every_golden_vector():
    n = 0
    for pat, name, data, engine, exp in golden_lib.cases():
        if exp is None:
            with pytest.raises(OracleError):
                Oracle(pat, engine).scan(data)
        else:
            assert Oracle(pat, engine).scan(data) == exp, (pat, name, engine)
        n += 1
    assert n > 700


def test_oracle_keeps_what_the_reference_had_printed_when_it_fails():
    """an NFT scan that enters an epsilon cycle: the reference exits 1 ("stack max capacity reached") through exit(),
    which flushes stdout — the lines before the bad one and the bad line up to the attempt that does not return are
    printed.  The vectors hold that stdout; the oracle's error carries the same bytes."""
    n = 0
    for pat, name, data, printed in golden_lib.fail_cases():
        with pytest.raises(OracleError) as e:
            Oracle(pat, "nft").scan(data)
        assert e.value.partial == printed, (pat, name)
        n += 1
    assert n >= 13


def test_oracle_state_counts():
    # NFT sizes of the configuration patterns
    assert Oracle("cat:dog", "nft").nft_states == 7
    assert Oracle("(cat:dog|dog:cat)", "nft").nft_states == 15
    assert Oracle("[a:A-z:Z]", "nft").nft_states == 80
    assert Oracle("[a:A-z:Z]", "dft").nft_states == 81
    o = Oracle("[a:A-z:Z]", "dft")
    o.scan(bytes(range(1, 256)).replace(b"\n", b"") + b"\n")
    assert o.dft_states == 27          # start + 26 one-byte finals


def test_oracle_rejects_what_the_reference_rejects():
    for pat in ["(a", "a)", "[a", "a{1,2,3}", "a{x}", "|a", "*a", "[a-c:z]"]:
        with pytest.raises(OracleError):
            Oracle(pat, "nft")


def test_line_sharded_threads_equal_single_thread():
    import random
    data = corpus.word_soup(random.Random(3), 200000)
    for pat, eng in [("(cat:dog|dog:cat)", "nft"), ("[a:A-z:Z]", "dft"), ("a:xyz", "dft")]:
        assert scan_mt(pat, eng, 4, data) == Oracle(pat, eng).scan(data)


@pytest.mark.skipif(not ref_available(), reason="compiled reference (oracle/_ref) not present")
def test_oracle_equals_compiled_reference_on_fresh_inputs():
    import random
    rng = random.Random(11)
    data = corpus.word_soup(rng, 30000) + corpus.printable_lines(rng, 30000)
    for pat in corpus.CONFIG_PATTERNS + ["a:xyz", "[aie]:", "abc:2|ab:1"]:
        for eng in ("nft", "dft"):
            assert Oracle(pat, eng).scan(data) == ref_scan(pat, eng, data), (pat, eng)


def test_inverted_bounds_are_the_lower_bound_only():
    """'{2,1}' (upper bound below the lower one): the reference unrolls the two mandatory copies and no optional one
    (trre_nft.c:458-485), i.e. 'x{2,1}' is 'xx'.  A fuzz run (tests/fuzz_oracle.py --seed 777) met '.{2,1}' inside a
    large DFT pattern on which the compiled reference needs minutes and gigabytes (256-way branches squared) — slow,
    not non-terminating — so the oracle and the product answer it; the error promise of DESIGN.md §2 is for real
    non-termination only (epsilon cycles), which the 'eps_*' golden vectors pin."""
    from oracle_lib import Oracle
    import trre_amd
    data = b"axyb ab axb axyzb\naxxb\n\naxyb"
    for eng in ("nft", "dft"):
        for a, b in (("a.{2,1}b:x", "a..b:x"), ("(c{2,1}:y|ab)", "(cc:y|ab)"), ("[ab]{3,2}:z", "[ab][ab][ab]:z")):
            want = Oracle(b, eng).scan(data)
            assert Oracle(a, eng).scan(data) == want, (a, eng)
            assert trre_amd.Program(a, eng).info.nft_states == trre_amd.Program(b, eng).info.nft_states + 2   # (the iteration's two JOINs)


def test_match_mode_against_the_reference():
    """`trre -m`: the oracle's match mode (trre_oracle_match) against the outputs of the compiled reference, incl.
    the reference's own match table (test.sh M rows)"""
    from oracle_lib import Oracle, OracleError
    import golden_lib
    n = 0
    for pat, name, data, exp in golden_lib.match_cases():
        o = Oracle(pat, "nft")
        if exp is None:
            with pytest.raises(OracleError):
                o.match(data)
        else:
            assert o.match(data) == exp, (pat, name)
        n += 1
    assert n > 100
