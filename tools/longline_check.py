#!/usr/bin/env python3
"""Long lines (round 5): the general families with exact sub-ranges against the same scan the old way (TRRE_EXACT=0 in a child
process), and against the oracle on a head; rates.  python tools/longline_check.py [bytes] [line_len ...]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import trre_amd
import corpora

CASES = [(" +: ", "nft"), ("(a|b)*c:x", "nft"), ("a:xyz", "dft"), ("(a|b)*c:x", "dft"), ("(cat:dog|dog:cat)", "nft"), ("[0-9]+:N", "nft")]


def run(n, line_len, tag):
    dev = torch.device("cuda", 0)
    name = "long%d" % line_len if line_len else "printable"
    inp = corpora.by_name(name, n, corpora.SEED0 + 2, dev)
    out = torch.empty(2 * n + 4096, dtype=torch.uint8, device=dev)
    res = []
    for pat, eng in CASES:
        p = trre_amd.Program(pat, eng)
        p.enqueue(inp, out); m = p.finish()
        torch.cuda.synchronize()
        steps = 5
        t0 = time.perf_counter()
        for _ in range(steps):
            p.enqueue(inp, out)
        m = p.finish()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        import hashlib
        h = hashlib.md5(out[:m].cpu().numpy().tobytes()).hexdigest() if n <= (1 << 30) else "-"
        res.append((pat, eng, m, h))
        print("%-8s %-10s %-20s %s  family %-11s %8.3f ms  %7.1f GB/s  out %d" % (tag, name, pat, eng, trre_amd.KERNEL_NAMES[p.info.kernel], dt * 1e3, n / dt / 1e9, m), flush=True)
    return res


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 30
    lens = [int(x) for x in sys.argv[2:]] or [400000, 40000, 0]
    if os.environ.get("LL_CHILD"):
        for L in lens:
            print("RES", L, repr(run(n, L, os.environ["LL_CHILD"])), flush=True)
        sys.exit(0)
    outs = {}
    for tag, env in (("old", {"TRRE_EXACT": "0"}), ("auto", {}), ("always", {"TRRE_EXACT": "1"})):
        e = dict(os.environ, LL_CHILD=tag, **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        txt = r.stdout.decode()
        print("\n".join(l for l in txt.splitlines() if not l.startswith("RES")), flush=True)
        outs[tag] = {l.split(" ", 2)[1]: l.split(" ", 2)[2] for l in txt.splitlines() if l.startswith("RES")}
        if r.returncode:
            print("child", tag, "failed", r.returncode)
    ok = outs["old"] == outs["auto"] == outs["always"] and len(outs["old"]) == len(lens)
    print("OUTPUTS IDENTICAL ACROSS old / auto / always:", ok)
    sys.exit(0 if ok else 1)
