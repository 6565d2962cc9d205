"""2000- and 4000-key dictionaries (beyond the comb-packed form: both passes on the 8-byte rows) on the GPU against the
oracle, both engines.  Run on the GPU box from the repo root."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, dictgen, trre_amd
from oracle_lib import Oracle
for n in (2000, 4000):
    keys, vals = dictgen.make_dictionary(n)
    pat = dictgen.pattern(keys, vals)
    data = dictgen.corpus_fast(keys, 2 << 20)
    for eng in ("dft", "nft"):
        part = data if eng == "dft" else data[:1 << 17]          # (the NFT oracle walks every key at every position)
        p = trre_amd.Program(pat, eng)
        t = torch.frombuffer(bytearray(part), dtype=torch.uint8).cuda()
        got = p.scan_tensor(t).cpu().numpy().tobytes()
        want = Oracle(pat, eng).scan(part)
        print(n, eng, p.info.stream_states, p.info.kernel, got == want, len(got))
