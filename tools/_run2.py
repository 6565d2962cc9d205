import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, trre_amd, corpora
dev = torch.device("cuda", 0)
n = 1 << 30
base = corpora.by_name("printable", n, corpora.SEED0, dev)
nl = (base == 10).nonzero().flatten()
print("lines", nl.numel())
for keep_every, label in [(1, "ordinary lines"), (400, "~40 KB lines"), (1000, "~100 KB lines"), (4000, "~400 KB lines")]:
    inp = base.clone()
    if keep_every > 1:
        drop = nl[torch.arange(nl.numel(), device=dev) % keep_every != 0]
        inp[drop] = 32
    out = torch.empty(2 * n + 64, dtype=torch.uint8, device=dev)
    for pat in ["(a|b)*c:x", " +: "]:
        p = trre_amd.Program(pat, "nft")
        try:
            p.enqueue(inp, out); m = p.finish()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            p.enqueue(inp, out); m = p.finish()
            dt = time.perf_counter() - t0
            print("%-16s %-10s %8.2f ms  %.1f GB/s out=%d" % (label, pat, dt * 1e3, n / dt / 1e9, m), flush=True)
        except trre_amd.TrreError as e:
            print(label, pat, "error", e.code, str(e)[:80], flush=True)
