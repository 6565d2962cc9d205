"""The command-line work-alikes trre_amd/bin/trre and trre_amd/bin/trre_dft (trre_amd/csrc/cli.cpp): same
command line, stderr texts and exit status as the reference binaries (trre_nft.c:728-773,
trre_dft.c:1217-1270), scan mode on the GPU through the C ABI.

The argument / pattern / file errors are decided before any device is touched and run in the CPU tier
(side by side with the compiled reference when oracle/_ref travelled along); the scans themselves are
`-m gpu`: golden vectors through both binaries, from stdin and from a FILE argument."""
import os
import subprocess
import tempfile

import pytest

import golden_lib
from oracle_lib import REF_DIR, ref_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = {"nft": os.path.join(ROOT, "trre_amd", "bin", "trre"), "dft": os.path.join(ROOT, "trre_amd", "bin", "trre_dft")}
REF = {"nft": os.path.join(REF_DIR, "trre"), "dft": os.path.join(REF_DIR, "trre_dft")}


def run(binary, args, data=b"", env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([binary] + list(args), input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=e)
    return r.returncode, r.stdout, r.stderr.replace(os.fsencode(binary), b"PROG")


@pytest.mark.parametrize("eng", ["nft", "dft"])
def test_cli_errors_match_the_reference(eng):
    usage = b"Usage: PROG [-d] [-m] expr [file]\n" if eng == "nft" else b"Usage: PROG [-dma] expr [file]\n"
    cases = [
        ([], b"error: missing trre expression\n"),                                  # trre_nft.c:746-749
        (["-x", "a"], b"PROG: invalid option -- 'x'\n" + usage),                   # trre_nft.c:739-743
        (["a", "/nonexistent/file"], b"error: can not open file /nonexistent/file\n"),   # trre_nft.c:766-771
        (["(a"], b"error: unmached parenthesis\n"),                                 # trre_nft.c:105-106
        (["a{1,2,3}"], b"error: more then one comma in curly brackets\n"),
        (["[a-c:z]"], b"error: unexpected range syntax\n"),
    ]
    for args, err in cases:
        rc, out, got = run(BIN[eng], args)
        assert (rc, out, got) == (1, b"", err), (eng, args, got)
        if ref_available():
            assert run(REF[eng], args) == (rc, out, got), (eng, args)


@pytest.mark.parametrize("eng", ["nft", "dft"])
def test_cli_refuses_the_cpu_only_modes(eng):
    """-d (Graphviz dumps) and trre_dft's -m (which only prints empty lines in the reference) are CPU features of the
    reference, not GPU paths: refused, status 1, nothing on stdout.  trre_dft -a answers like the reference."""
    for flag in ("-d",) + (("-m",) if eng == "dft" else ()):
        rc, out, err = run(BIN[eng], [flag, "a"])
        assert rc == 1 and out == b"" and err.startswith(b"error: " + flag.encode()), (eng, flag, err)
    if eng == "dft":
        assert run(BIN[eng], ["-a", "a"]) == (1, b"", b"Not supported yet\n")        # trre_dft.c:1227-1229
        if ref_available():
            assert run(REF[eng], ["-a", "a"]) == (1, b"", b"Not supported yet\n")


@pytest.mark.gpu
def test_cli_match_mode():
    """`trre -m PATTERN`: whole-line matches (the reference's test.sh M rows, first output)"""
    n = 0
    for pat, name, data, exp in list(golden_lib.match_cases())[::4]:
        if exp is None or b"\0" in pat.encode("latin-1"):
            continue
        rc, out, err = run(BIN["nft"], ["-m", pat.encode("latin-1")], data)
        assert (rc, out, err) == (0, exp, b""), (pat, name, err)
        n += 1
    assert n > 20


def _sample():
    """a spread of golden cases: every 11th, plus the reference's own scan rows and the config patterns"""
    picked = []
    for i, case in enumerate(golden_lib.cases()):
        pat, name, data, engine, exp = case
        if b"\0" in pat.encode("latin-1") or exp is None:
            continue
        if i % 11 == 0 or pat in ("(.:x)*.*", "<(.:)*?>", ":=", "(cat:dog|dog:cat)", "[a:A-z:Z]", "a:xyz"):
            picked.append(case)
    return picked


@pytest.mark.gpu
def test_cli_scans_golden_vectors_from_stdin_and_file():
    n = 0
    for pat, name, data, engine, exp in _sample()[:90]:
        arg = pat.encode("latin-1")
        rc, out, err = run(BIN[engine], [arg], data)
        assert (rc, out, err) == (0, exp, b""), (pat, name, engine, err)
        if n % 3 == 0:
            with tempfile.NamedTemporaryFile() as tf:
                tf.write(data)
                tf.flush()
                assert run(BIN[engine], [arg, tf.name]) == (0, exp, b""), (pat, name, engine, "FILE")
        n += 1
    assert n > 60


@pytest.mark.gpu
def test_cli_streams_blocks_and_shards():
    """input larger than the read block (TRRE_CLI_BLOCK) with lines that straddle block ends, several shards
    per device (TRRE_SHARDS_PER_DEVICE): the output must not depend on either"""
    import random
    import corpus
    from oracle_lib import Oracle
    rng = random.Random(3)
    data = corpus.word_soup(rng, 600000) + b"x" * 70000 + b" cat\n" + corpus.word_soup(rng, 100000) + b"last line without newline cat"
    for pat, eng in [("(cat:dog|dog:cat)", "nft"), ("a:xyz", "dft"), ("[aie]:", "nft"), ("[a:A-z:Z]", "dft")]:
        want = Oracle(pat, eng).scan(data)
        for env in ({}, {"TRRE_CLI_BLOCK": "65536"}, {"TRRE_CLI_BLOCK": "100000", "TRRE_SHARDS_PER_DEVICE": "3"}):
            rc, out, err = run(BIN[eng], [pat], data, env)
            assert (rc, err) == (0, b"") and out == want, (pat, eng, env, len(out), len(want))


@pytest.mark.gpu
def test_cli_reports_divergence_like_the_reference():
    """an epsilon cycle entered: the reference prints 'error: stack max capacity reached' and exits 1 — with the lines
    before the bad one and the bad line up to the failing attempt on stdout (exit() flushes).  Same status, same
    stderr, same stdout, whatever the read block size; compared with the compiled reference when it travelled along."""
    for pat, name, data, printed in golden_lib.fail_cases():
        for env in ({}, {"TRRE_CLI_BLOCK": "16"}):
            rc, out, err = run(BIN["nft"], [pat], data, env)
            assert rc == 1 and err.startswith(b"error: stack max capacity reached"), (pat, name, err)
            assert out == printed, (pat, name, env)
        if ref_available():
            rrc, rout, rerr = run(REF["nft"], [pat], data)
            assert (rrc, rout) == (1, printed) and rerr == b"error: stack max capacity reached\n"
    # the search's 65 536-item stack (round 4: the stack guard): a line that exhausts it, in a stream of ordinary lines
    data = b"a  b c\n" * 3000 + b"x" + b" " * 70000 + b"y\n" + b"d  e\n" * 10
    for env in ({}, {"TRRE_CLI_BLOCK": "65536"}):
        rc, out, err = run(BIN["nft"], [" +: "], data, env)
        assert rc == 1 and err.startswith(b"error: stack max capacity reached") and out == b"a b c\n" * 3000 + b"x", (env, rc, err, len(out))
    if ref_available():
        assert run(REF["nft"], [" +: "], data) == (1, b"a b c\n" * 3000 + b"x", b"error: stack max capacity reached\n")
