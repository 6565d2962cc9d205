import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, dictgen, trre_amd
from oracle_lib import Oracle
for n in (2000, 4000):
    keys, vals = dictgen.make_dictionary(n)
    pat = dictgen.pattern(keys, vals)
    data = dictgen.corpus_fast(keys, 2 << 20)
    p = trre_amd.Program(pat, "dft")
    t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
    got = p.scan_tensor(t).cpu().numpy().tobytes()
    want = Oracle(pat, "dft").scan(data)
    print(n, p.info.stream_states, p.info.kernel, got == want, len(got))
