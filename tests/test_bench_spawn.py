"""bench.py's N-rank path without a GPU: `--gpus N` starts N ranks by itself, the ranks rendezvous
(gloo here, RCCL on the GPU box), the timed region is bracketed by barriers, the elapsed time is the MAX over
ranks and `n_gpus` is the number of ranks that really ran.  TRRE_BENCH_STUB=1 replaces the scan by a stub
(nothing is measured, the line says "data": "stub"): the launch / timing / reduction plumbing is what runs.
Without the stub and without GPUs the same command must fail loudly instead of reporting one GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_gpus_flag_starts_that_many_ranks():
    r = run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--bytes", str(1 << 20)], {"TRRE_BENCH_STUB": "1"})
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()                  # rank 0 alone prints
    assert r.stdout.decode().rstrip("\n").splitlines()[-1] == lines[0]      # ... and it is the LAST line of stdout
    assert len(lines[0]) < 4096                                # the driver parses the last line out of an 8 KB tail (23 KB, parsed null)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["data"] == "stub"
    assert line["config"]["bytes_per_gpu"] == 1 << 20
    # whole-job value: both ranks' bytes over the slowest rank's time
    assert abs(line["value"] - 2 * (1 << 20) / (line["ms_per_step"] * 1e-3) / 1e9) < 0.02 * line["value"] + 0.01
    # the configurations that shard ride along: config 4 strong-scaled (one corpus over the ranks), config 5 weak
    by = {c["name"]: c for c in line["configs"]}
    assert by["cfg4_strong"]["scaling"] == "strong" and by["cfg4_strong"]["n_gpus"] == 2 and by["cfg4_strong"]["verified"] is True
    assert by["cfg5_weak"]["scaling"] == "weak"
    assert line["configs_verified"] is True
    # the prose and the per-config detail live in the file the line names
    full = json.load(open(os.path.join(ROOT, line["details"])))
    fby = {c["name"]: c for c in full["configs"]}
    assert fby["cfg4_strong"]["bytes_per_gpu"] == (1 << 20) // 2 and fby["cfg4_strong"]["bytes_total"] == 1 << 20
    assert fby["cfg5_weak"]["bytes_total"] == 2 << 20


def test_the_line_stays_parseable_whatever_the_records_hold():
    """compact_line() on the largest record a round has produced (round 5's 23 KB line, with its 26 config records and their
    paragraphs of prose) and on a record blown up further: one line, under 4 KB, with the contract's keys, `roofline` and
    `cpu_baseline` intact."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    assert len(json.dumps(full)) > 20000
    for blow in (1, 4):
        rec = dict(full)
        rec["configs"] = [dict(c, name="%s_%d" % (c.get("name"), k)) for k in range(blow) for c in full["configs"]]
        text = bench.compact_line(rec)
        assert "\n" not in text and len(text) < 4096, len(text)
        line = json.loads(text)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
            assert line[k] == full[k], k
        assert line["config"]["workload"].startswith("'[a:A-z:Z]' DFT scan")
        assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["roofline"]["bound"] == "hbm" and line["roofline"]["traffic"] == full["roofline"]["traffic"]
        assert line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"] and line["cpu_baseline"]["kind"] == "reference" and line["cpu_baseline"]["cores"] == 1
        if blow == 1:
            got = {c["name"][:-2]: c for c in line["configs"]}
            assert got["cfg4"]["GBps"] == [c for c in full["configs"] if c["name"] == "cfg4"][0]["input_GBps"] and got["cfg4"]["verified"] is True


def test_gpus_flag_fails_loudly_without_the_devices():
    """no GPU in this tier: `bench.py --gpus 2` must exit non-zero with a clear message, never print an n_gpus=1 line"""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--bytes", str(1 << 20)])
    assert r.returncode != 0
    assert b"GPU(s) are visible" in r.stderr and b"{" not in r.stdout


def test_world_size_must_match_the_flag():
    r = run(["--gpus", "2", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "TRRE_BENCH_STUB": "1"})
    assert r.returncode != 0 and b"WORLD_SIZE=1" in r.stderr
