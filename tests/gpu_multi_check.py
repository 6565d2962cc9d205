#!/usr/bin/env python3
"""Run by tests/test_gpu_multi.py in a subprocess with TRRE_SHARDS_PER_DEVICE set (the library reads it once
per process): trre_scan_host_multi cuts the input into several shards per device, scans each from its own host
thread and reassembles — checked against the oracle for length-preserving and general programs, inputs with NUL
bytes, inputs smaller than the shard count, and the capacity protocol."""
import ctypes
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import corpus  # noqa: E402
import trre_amd  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def main():
    rng = random.Random(41)
    big = corpus.word_soup(rng, 3 << 20) + b"nul\0cat dog\n" + corpus.printable_lines(rng, 1 << 20) + b"tail cat without newline"
    inputs = [big, big.replace(b"\0", b" "), b"cat\n", b"cat", b"a\nb\n", b"\n\n\n", b""]
    bad = 0
    for pat, eng in [("[a:A-z:Z]", "dft"), ("(cat:dog|dog:cat)", "nft"), ("a:xyz", "dft"), ("[aie]:", "nft"), ("(a|b)*c:x", "nft"),
                     ("(.:x)*.*", "nft")]:
        p = trre_amd.Program(pat, eng)
        o = Oracle(pat, eng)
        for data in inputs:
            want = o.scan(data)
            got = p.scan(data, device_mask=0)
            if got != want:
                print("MISMATCH", pat, eng, len(data), len(got), len(want))
                bad += 1
    # capacity protocol: a buffer that is too small reports the size the whole output needs
    p = trre_amd.Program("a:xyz", "dft")
    want = Oracle("a:xyz", "dft").scan(big)
    small = ctypes.create_string_buffer(1 << 16)
    m = ctypes.c_size_t()
    rc = trre_amd.api.lib().trre_scan_host_multi(p._h, big, len(big), small, len(small), ctypes.byref(m), 0)
    if rc != trre_amd.api.E_CAPACITY or m.value != len(want):
        print("CAPACITY", rc, m.value, len(want))
        bad += 1
    rc = trre_amd.api.lib().trre_scan_host_multi(p._h, big, len(big), None, 0, ctypes.byref(m), 0)      # size query
    if rc != trre_amd.api.E_CAPACITY or m.value != len(want):
        print("SIZE QUERY", rc, m.value, len(want))
        bad += 1
    # a length-preserving program whose output NUL bytes shorten: a buffer of exactly the reported size must do
    # (shards whose place lay beyond `cap` ran as size queries and the call failed for ever)
    q = trre_amd.Program("[a:A-z:Z]", "dft")
    nul = (b"abc\0defghijklmnopqrstuvwxyz0123456789\n" * 40000) + big.replace(b"\0", b" ")[: 1 << 20]
    want_q = Oracle("[a:A-z:Z]", "dft").scan(nul)
    rc = trre_amd.api.lib().trre_scan_host_multi(q._h, nul, len(nul), None, 0, ctypes.byref(m), 0)
    exact = ctypes.create_string_buffer(m.value)
    m2 = ctypes.c_size_t()
    rc2 = trre_amd.api.lib().trre_scan_host_multi(q._h, nul, len(nul), exact, m.value, ctypes.byref(m2), 0)
    if rc != trre_amd.api.E_CAPACITY or m.value != len(want_q) or rc2 != 0 or exact.raw[:m2.value] != want_q:
        print("EXACT CAPACITY", rc, m.value, len(want_q), rc2, m2.value)
        bad += 1
    rc = trre_amd.api.lib().trre_scan_host_multi(p._h, big, len(big), small, len(small), ctypes.byref(m), 1 << 30)
    if rc != trre_amd.api.E_ARG:
        print("MASK", rc)
        bad += 1
    print("multi check: %s" % ("ok" if not bad else "%d failures" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
