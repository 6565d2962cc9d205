python -m pytest tests/test_generate.py -x -q -m gpu 2>&1 | tail -3
python - <<'PY'
import sys, time, random
sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import os
import trre_amd, corpus
rng = random.Random(23)
data = corpus.word_soup(rng, 32 << 20, max_len=60)
for pat, mode in [("(cat:dog|cat:cow|ca:C)", "scan_all"), ("(cat:dog|cat:cow|.)*", "match_all")]:
    p = trre_amd.Program(pat, "nft", mode=mode)
    p.scan(data[:1 << 20])
    t0 = time.perf_counter(); out = p.scan(data); dt = time.perf_counter() - t0
    print("device enumeration  %-26s %s  in %d out %d  %.3f s  %.1f MB/s" % (pat, mode, len(data), len(out), dt, len(data) / dt / 1e6), flush=True)
PY
TRRE_GEN_HOST=1 python - <<'PY'
import sys, time, random
sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import trre_amd, corpus
rng = random.Random(23)
data = corpus.word_soup(rng, 32 << 20, max_len=60)
for pat, mode in [("(cat:dog|cat:cow|ca:C)", "scan_all"), ("(cat:dog|cat:cow|.)*", "match_all")]:
    p = trre_amd.Program(pat, "nft", mode=mode)
    p.scan(data[:1 << 20])
    t0 = time.perf_counter(); out = p.scan(data); dt = time.perf_counter() - t0
    print("host enumeration    %-26s %s  in %d out %d  %.3f s  %.1f MB/s" % (pat, mode, len(data), len(out), dt, len(data) / dt / 1e6), flush=True)
PY
