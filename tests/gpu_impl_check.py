#!/usr/bin/env python3
"""Run by tests/test_gpu_parity.py in a subprocess with
TRRE_LANE_BYTES / TRRE_NO_G16 / TRRE_NO_FB / TRRE_FB_EMIT / TRRE_NO_FB_COPY set: checks the alternative implementations of the stream kernel families
against the oracle (the environment is read once per process by the library)."""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import torch  # noqa: E402

import corpus  # noqa: E402
import trre_amd  # noqa: E402
from oracle_lib import Oracle  # noqa: E402


def main():
    rng = random.Random(31)
    long_line = b"cat dog ca do there hello world abcdefgh abcdefg " * 300
    data = (corpus.word_soup(rng, 400000) + b"short cat\n" + long_line + b"\n" + corpus.printable_lines(rng, 300000)
            + b"nul\0cat dog\n" + b"tail cat without newline")
    clean = data.replace(b"\0", b" ")
    bad = 0
    for pat, eng in [("[a:A-z:Z]", "dft"), ("(cat:dog|dog:cat)", "nft"), ("(cat:dog|dog:cat)", "dft"), ("cat:dog", "nft"),
                     ("a:xyz", "dft"), ("[aie]:", "nft"), ("abc:2|ab:1", "nft"), ("abc:2|ab:1", "dft"),
                     ("there:THERE|cat:dog", "dft"), ("(hello:world|world:hello)", "nft"), ("abcdefgh:ABCDEFGH|dog:cat", "dft"),
                     ("(cat:elephant|dog:a-replacement-text-of-more-than-forty-bytes-0123456789|do:12345)", "dft"),
                     ("(cat:elephant|dog:a-replacement-text-of-more-than-forty-bytes-0123456789|do:12345)", "nft")]:
        p = trre_amd.Program(pat, eng)
        for fam in [f for f in p.allowed_kernels() if f in (trre_amd.KERNEL_STREAM_LP, trre_amd.KERNEL_STREAM_GEN)]:
            p.set_kernel(fam)
            for buf in (clean, data, b"cat\n", b"cat"):
                t = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
                got = p.scan_tensor(t).cpu().numpy().tobytes()
                if got != Oracle(pat, eng).scan(buf):
                    print("MISMATCH", pat, eng, fam, len(buf))
                    bad += 1
    # a table large enough for the fallback form (TRRE_FB_EMIT / TRRE_NO_FB choose who walks it): keys inside keys, NULs
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import dictgen
    keys, vals = dictgen.make_dictionary(300)
    keys += [keys[0][1:] + "q", keys[1] + "zz", keys[2][:2]]
    vals += ["inner", "", "a-long-replacement-text"]
    pat = dictgen.pattern(keys, vals)
    text = dictgen.corpus(keys, 150000) + b"nul\0" + keys[3].encode() + b" rest\n" + (keys[0] + keys[1] + keys[300]).encode()
    for eng in ("dft", "nft"):
        p = trre_amd.Program(pat, eng)
        t = torch.frombuffer(bytearray(text), dtype=torch.uint8).cuda()
        if p.scan_tensor(t).cpu().numpy().tobytes() != Oracle(pat, eng).scan(text):
            print("MISMATCH dictionary", eng)
            bad += 1
    # the same table without the empty text has a copy form (TRRE_NO_FB_COPY=1: the count / emit pair): clean input (the copy
    # form runs), misaligned, a NUL (void launch, rerun), texts every three bytes (more events than a lane's list holds: void
    # launch, rerun, and the program stops trying)
    vals[301] = "zz-inner"
    pat = dictgen.pattern(keys, vals)
    clean_text = dictgen.corpus(keys, 400000) + (keys[0] + keys[1] + keys[300]).encode()
    dense = (" ".join(keys[302] for _ in range(4000)) + "\n").encode() * 3
    for eng in ("dft", "nft"):
        p = trre_amd.Program(pat, eng)
        o = Oracle(pat, eng)
        for buf, skip in ((clean_text, 0), (clean_text, 5), (text, 0), (clean_text[:70000] + dense + clean_text[:5000], 0), (clean_text, 3)):
            if eng == "nft":
                buf = buf[: 1 << 17]
            t = torch.frombuffer(bytearray(b"x" * skip + buf), dtype=torch.uint8).cuda()[skip:]
            if p.scan_tensor(t).cpu().numpy().tobytes() != o.scan(buf):
                print("MISMATCH dictionary (copy form)", eng, len(buf), skip)
                bad += 1
    # the general families whatever implements them : small tables, guided tables, edits in
    # every byte of a piece (more than 7 per 64 bytes: overflow records), texts longer than a piece, a diverging line
    for pat, eng in [("a:xyz", "dft"), ("a:", "nft"), (" +: ", "nft"), ("(a|b)*c:x", "nft"), ("[0-9]+:N", "nft"), ("[a-z]:xy", "dft"),
                     ("e:a-replacement-text-of-more-than-sixty-four-bytes-0123456789-0123456789-0123456789-0123456789", "dft"), (":=", "nft")]:
        p = trre_amd.Program(pat, eng)
        for buf in (clean, data, b"aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa 123 b\n" * 3000, b"a"):
            t = torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()
            got = p.scan_tensor(t).cpu().numpy().tobytes()
            if got != Oracle(pat, eng).scan(buf):
                print("MISMATCH general", pat, eng, len(buf))
                bad += 1
    try:
        p = trre_amd.Program("cat:dog|a:*", "nft")
        t = torch.frombuffer(bytearray(b"cat b\ncat xa cat\n"), dtype=torch.uint8).cuda()
        p.scan_tensor(t)
        print("MISMATCH: no divergence")
        bad += 1
    except trre_amd.TrreError as e:
        if e.code != trre_amd.api.E_DIVERGES or e.partial.cpu().numpy().tobytes() != b"dog b\ndog x":
            print("MISMATCH diverge partial", e.code)
            bad += 1
    print("impl check: %s" % ("ok" if not bad else "%d mismatches" % bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
