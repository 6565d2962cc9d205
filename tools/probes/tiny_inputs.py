#!/usr/bin/env python3
"""Every kernel family of a pattern over many tiny random inputs at odd buffer alignments, against the oracle
(a hang shows as the last line printed):  python tools/probes/tiny_inputs.py PATTERN ENGINE [ROUNDS]"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import trre_amd
from oracle_lib import Oracle, OracleError
pat, eng = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 400
p = trre_amd.Program(pat, eng)
o = Oracle(pat, eng)
rng = random.Random(7)
bad = 0
for it in range(rounds):
    ln = rng.choice([1, 2, 5, 7, 7, 7, 9, 20, 70])
    data = bytes(rng.choice(b"abcxy\n\n") for _ in range(ln))
    try:
        want = o.scan(data)
    except OracleError:
        continue
    for fam in p.allowed_kernels():
        for mi, mo in ((1, 16), (0, 0), (5, 33)):
            print("it", it, "fam", fam, "mis", mi, mo, data, flush=True)
            buf = torch.zeros(len(data) + 64, dtype=torch.uint8, device="cuda")
            buf[mi:mi + len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            obuf = torch.empty(len(want) + len(data) + 128, dtype=torch.uint8, device="cuda")
            p.set_kernel(fam)
            got = p.scan_tensor(buf[mi:mi + len(data)], out=obuf[mo:]).cpu().numpy().tobytes()
            if got != want:
                bad += 1
                print("MISMATCH", fam, mi, mo, data, got, want, flush=True)
print("done, mismatches:", bad)
