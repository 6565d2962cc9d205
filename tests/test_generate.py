"""Generator mode (`trre -a`, `trre -ma`): every accepting path prints, trre_nft.c:640-641,647-648.

CPU tier: the oracle's restatement against the outputs of the compiled reference (tests/golden/golden.json,
"all_cases": the reference's own match table run the way its test.sh runs it — `./trre -ma`, all 51 rows with ALL their
outputs —, ambiguous whole-line patterns on multi-line inputs, and scan mode with -a), and the product's path with the
device's part — the backward viability sweep — run thread by thread on the host (tests/cpu_shim.cpp) feeding the
library's own enumeration.  GPU tier: the same vectors through the C ABI and the CLI, and larger inputs against the oracle."""
import random

import pytest

import corpus
import golden_lib
import shim_lib
import trre_amd
from oracle_lib import Oracle, OracleError


def mode_of(flags):
    return "match_all" if "m" in flags else "scan_all"


def oracle_run(pat, flags, data):
    o = Oracle(pat, "nft", all_outputs=True)
    return o.match(data) if "m" in flags else o.scan(data)


def test_oracle_generator_mode_against_the_reference():
    n = n_fail = 0
    for pat, flags, name, data, exp, printed in golden_lib.all_cases():
        if exp is None:
            with pytest.raises(OracleError) as e:
                oracle_run(pat, flags, data)
            if printed is not None:
                assert e.value.partial == printed, (pat, flags, name)
            n_fail += 1
        else:
            assert oracle_run(pat, flags, data) == exp, (pat, flags, name)
        n += 1
    assert n >= 200 and n_fail >= 10


def test_reference_match_table_has_all_its_outputs():
    """test.sh's M rows are written for `./trre -ma`: every row's expected text is the FIRST line of what -ma prints (or
    nothing); rows with several parses print more lines — all of them are in the vectors"""
    rows = [c for c in golden_lib.all_cases() if c[2].startswith("refm_")]
    assert len(rows) == len(corpus.REF_M_CASES) == 51
    several = 0
    for (inp, pat, first), (gpat, flags, name, data, exp, _) in zip(corpus.REF_M_CASES, rows):
        assert gpat == pat and flags == "-ma" and data == inp.encode("latin-1") + b"\n"
        lines = exp.split(b"\n")[:-1] if exp else []
        assert (lines[0].decode("latin-1") if lines else None) == first, (inp, pat)
        several += len(lines) > 1
    assert several >= 3


def test_generator_mode_enumeration_with_the_backward_sweep_on_the_host():
    n = n_fail = 0
    for pat, flags, name, data, exp, printed in golden_lib.all_cases():
        p = trre_amd.Program(pat, "nft", mode=mode_of(flags))
        assert p.info.kernel == trre_amd.api.KERNEL_GENERATE
        for geo, mis in ((1, 0), (0, 5)):
            if exp is None:
                if printed is None:
                    continue                 # the reference ran out of time or memory: nothing to compare
                with pytest.raises(trre_amd.TrreError) as e:
                    shim_lib.generate_like_runtime(p, data, geo, mis)
                assert e.value.code == trre_amd.api.E_DIVERGES and e.value.partial == printed, (pat, flags, name)
                n_fail += 1
            else:
                assert shim_lib.generate_like_runtime(p, data, geo, mis) == exp, (pat, flags, name, geo)
        n += 1
    assert n >= 200 and n_fail >= 10


def test_generator_mode_enumeration_kernel_body_on_the_host():
    """round 4: the enumeration is a kernel (gen_block.hpp): its lane body — count, exclusive sum, emit — on the host behind
    the backward sweep, against every vector of the compiled reference; with a stack of a few frames too, so that the route
    back to the host enumeration (a search deeper than a lane's stack, a path that never returns) runs as well"""
    n = n_host = n_dev = 0
    for pat, flags, name, data, exp, printed in golden_lib.all_cases():
        p = trre_amd.Program(pat, "nft", mode=mode_of(flags))
        for geo, mis, frames, cap in ((1, 0, 512, 2048), (0, 5, 512, 2048), (1, 3, 6, 24)):
            if exp is None:
                if printed is None:
                    continue
                with pytest.raises(trre_amd.TrreError) as e:
                    shim_lib.generate_on_device_like_runtime(p, data, geo, mis, frames, cap)
                assert e.value.code == trre_amd.api.E_DIVERGES and e.value.partial == printed, (pat, flags, name)
            else:
                out, host = shim_lib.generate_on_device_like_runtime(p, data, geo, mis, frames, cap)
                assert out == exp, (pat, flags, name, geo, frames)
                n_host += host
                n_dev += not host
        n += 1
    assert n >= 200 and n_dev > 350 and n_host >= 3, (n, n_dev, n_host)
    rng = random.Random(19)
    data = corpus.word_soup(rng, 40000, max_len=40) + b"nul\0cat\n" + b"cat cat"
    for pat, flags in [("(cat:dog|cat:cow|ca:C)", "-a"), ("a*", "-a"), (":=", "-a"), ("(cat:dog|cat:cow|.)*", "-ma"), ("[a-z ]*|.*", "-ma")]:
        p = trre_amd.Program(pat, "nft", mode=mode_of(flags))
        out, host = shim_lib.generate_on_device_like_runtime(p, data, 0)
        assert out == oracle_run(pat, flags, data) and not host, (pat, flags)


def test_generator_mode_without_a_viability_filter(monkeypatch):
    """round 5: a viability automaton of more than 256 states (symbols are bytes) no longer refuses the pattern: the filter lets
    every node through and the enumeration walks the failing branches too, as the reference does.  Every vector again with the
    limit at 3 states, and a pattern that is over the real one."""
    monkeypatch.setenv("TRRE_GEN_MAX_REV", "3")
    n = 0
    for pat, flags, name, data, exp, printed in golden_lib.all_cases():
        if len(data) > 4000:
            continue
        p = trre_amd.Program(pat, "nft", mode=mode_of(flags))
        assert p.info.guided_rev_states == 3
        if exp is None:
            if printed is None:
                continue
            with pytest.raises(trre_amd.TrreError) as e:
                shim_lib.generate_on_device_like_runtime(p, data, 1)
            assert e.value.code == trre_amd.api.E_DIVERGES and e.value.partial == printed, (pat, flags, name)
        else:
            out, _ = shim_lib.generate_on_device_like_runtime(p, data, 1)
            assert out == exp, (pat, flags, name)
            assert shim_lib.generate_like_runtime(p, data, 0, 3) == exp, (pat, flags, name, "host enumeration")
        n += 1
    assert n >= 190
    monkeypatch.delenv("TRRE_GEN_MAX_REV")
    # (which of the next ten / sixteen bytes is a 'c': thousands of viability states)
    data = b"abcabcabcabcabcabcabcabcabc\nacccccccccc acbacbacbacbacbacb\nabcabcabcabcabcc\nabcabcabcabcabccabc\n"
    for pat, flags in (("a(a|b|c){9}c:x", "-a"), ("(a|b|c){15}c(a|b|c)*:x", "-ma")):
        p = trre_amd.Program(pat, "nft", mode=mode_of(flags))
        assert p.info.guided_rev_states == 3
        out, _ = shim_lib.generate_on_device_like_runtime(p, data, 1)
        assert out == oracle_run(pat, flags, data), flags


def test_generator_mode_on_larger_inputs_against_the_oracle():
    rng = random.Random(19)
    data = corpus.word_soup(rng, 40000, max_len=40) + b"nul\0cat\n" + b"cat cat"
    for pat, flags in [("(cat:dog|cat:cow|ca:C)", "-a"), ("a*", "-a"), (":=", "-a"), ("(cat:dog|cat:cow|.)*", "-ma"), ("[a-z ]*|.*", "-ma"),
                       ("(a|a:x|.)*", "-a")]:
        small = data[:3000] if pat == "(a|a:x|.)*" else data
        p = trre_amd.Program(pat, "nft", mode=mode_of(flags))
        assert shim_lib.generate_like_runtime(p, small, 0) == oracle_run(pat, flags, small), (pat, flags)


def test_generator_mode_is_an_nft_feature():
    with pytest.raises(trre_amd.TrreError) as e:
        trre_amd.Program("a", "dft", mode="scan_all")
    assert e.value.code == trre_amd.api.E_UNSUPPORTED and e.value.message == "Not supported yet"      # trre_dft.c:1227-1229


@pytest.mark.gpu
def test_generator_mode_on_gpu():
    import torch
    n = 0
    for pat, flags, name, data, exp, printed in golden_lib.all_cases():
        p = trre_amd.Program(pat, "nft", mode=mode_of(flags))
        if exp is None:
            if printed is None:
                continue
            with pytest.raises(trre_amd.TrreError) as e:
                p.scan(data)
            assert e.value.code == trre_amd.api.E_DIVERGES and e.value.partial == printed, (pat, flags, name)
        else:
            assert p.scan(data) == exp, (pat, flags, name)
            if data and n % 5 == 0:
                t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
                assert p.scan_tensor(t, out=torch.empty(len(exp) + 64, dtype=torch.uint8, device="cuda")).cpu().numpy().tobytes() == exp
        n += 1
    assert n >= 200
    rng = random.Random(23)
    data = corpus.word_soup(rng, 3 << 20, max_len=60) + b"tail cat"
    for pat, flags in [("(cat:dog|cat:cow|ca:C)", "-a"), ("(cat:dog|cat:cow|.)*", "-ma"), ("a*", "-a")]:
        p = trre_amd.Program(pat, "nft", mode=mode_of(flags))
        want = oracle_run(pat, flags, data)
        assert p.scan(data) == want, (pat, flags)
        assert p.scan(data, device_mask=0) == want, (pat, flags, "multi")


@pytest.mark.gpu
def test_cli_generator_mode():
    from test_cli import BIN, REF, run
    from oracle_lib import ref_available
    n = 0
    for pat, flags, name, data, exp, printed in list(golden_lib.all_cases())[::3]:
        if b"\0" in pat.encode("latin-1"):
            continue
        rc, out, err = run(BIN["nft"], [flags, pat.encode("latin-1")], data)
        if exp is None:
            if printed is None:
                continue
            assert rc == 1 and out == printed and err.startswith(b"error: stack max capacity reached"), (pat, flags, name)
        else:
            assert (rc, out, err) == (0, exp, b""), (pat, flags, name, err)
            if ref_available() and n % 4 == 0:
                assert run(REF["nft"], [flags, pat.encode("latin-1")], data) == (0, exp, b"")
        n += 1
    assert n > 50
    rc, out, err = run(BIN["dft"], ["-a", "a"], b"a\n")
    assert (rc, out, err) == (1, b"", b"Not supported yet\n")
