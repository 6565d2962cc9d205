#!/usr/bin/env python3
"""bench.py — scan-mode throughput of the MI355X transducer scan engine.

One "step" = one pass of the hot path (the scan of a whole '\\n'-delimited
buffer, i.e. the scan branch of the reference's main(): trre_dft.c:1272-1286 /
trre_nft.c:775-790) over one batch of synthetic input that is already resident
in HBM.

Headline workload (`value`): BASELINE.json configs[1] — '[a:A-z:Z]' uppercase DFT
scan over synthetic ASCII lines — at the size north_star asks for, 8 GiB per GPU
(`--bytes` changes it).  N GPUs = N line shards of that size, one rank per GPU,
no data-path collective (`"scaling": "weak"`); `--scaling strong` splits ONE
config-4 corpus of `--bytes` over the ranks instead.

`--gpus N` with N > 1 and no launcher around it starts the N ranks itself (the
driver's own command line: torch.distributed.run on 127.0.0.1) and exits
non-zero if fewer than N GPUs are visible; under a launcher WORLD_SIZE must
equal --gpus.  `n_gpus` in the line is the number of ranks that really ran.

On one GPU the same JSON line carries a `configs` array: the other BASELINE
configurations (Caesar, the NFT cat/dog scan on its own corpus, the 1000-entry
dictionary) and the general kernel families, each with the kernels' HIP-event
time, its fraction of the HBM roofline, a `verified` flag and the reference CPU
binary timed on a bounded sample of the same workload.  On N > 1 GPUs it carries
the two configurations that BASELINE.json shards: configs[3] strong-scaled (one
corpus of --bytes cut into N line shards) and configs[4] weak-scaled (--bytes
per GPU: 64 GiB at 8 x 8 GiB), each verified on every rank.

    python bench.py                       # 1 GPU
    python bench.py --gpus 8              # starts 8 ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))   # corpora (synthetic inputs), dictgen
sys.path.insert(0, os.path.join(ROOT, "tests"))   # oracle bindings: verification and cpu_baseline legs only

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable

KERNEL_OF = {"generate": "k_rev_sweep (viability symbols) + k_gen<count> + k_gen<emit>", "bytemap": "k_bytemap", "tile_lp": "k_scan_lp", "tile_gen": "k_scan_count + k_scan_emit",
             "stream_lp": "k_stream_lpw (window form) / k_stream_g16<emit> alone",
             "stream_gen": "k_stream_g16<count> + <emit> (small tables) / k_fb_mark + k_fb_splice (large tables: the copy form)",
             "guided_lp": "k_rev_sweep + k_stream_g16<emit, sym>", "guided_gen": "k_rev_sweep + k_stream_g16<count, sym> + <emit, sym>",
             "backtrack": "k_bt<count> + k_chunk_scan + k_bt<emit>", "dft_lazy": "k_lazy<count> + k_chunk_scan + k_lazy<emit> (+ host exploration between rounds)"}


def synth_lines(n, seed, device):
    """cfg 2 / cfg 3 input (kept under this name for tools/ and tests/): corpora.printable_lines"""
    import corpora
    return corpora.printable_lines(n, seed, device)


def pmc_traffic(kernel_name, nbytes):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
    (profiles/*_bench_pmc_{FETCH,WRITE}_SIZE.txt, newest round) — measured under rocprofv3 in separate
    passes, NOT in this run; returned with its source, or (None, None) when no committed pass matches this
    kernel and size.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of a
    wide coalesced read (MI355X_MICROARCH.md, HBM section), so it is doubled."""
    import glob
    import re
    vals = {}
    src = []
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_pmc_%s.txt" % c)))
        if not files:
            return None, None
        text = open(files[-1]).read()
        if kernel_name not in text:
            return None, None
        m = re.search(r"%s\s+([0-9.]+)" % c, text)
        if not m:
            return None, None
        vals[c] = float(m.group(1))
        src.append(os.path.relpath(files[-1], ROOT))
    total = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
    if not 1.6 * nbytes <= total <= 2.6 * nbytes:          # a pass taken at another input size
        return None, None
    return total, "rocprofv3 --pmc passes of this command, committed: " + ", ".join(src)


# HBM traffic per GiB of input of the other configurations, from the committed PMC passes of tools/profile_round.sh /
# tools/pmc_dict4.sh (1 GiB runs; FETCH_SIZE doubled as for the headline — round 5's calibration, tools/probes/fetch_calib.hip, found the
# factor 2 right for every access pattern the kernels use): summed over the kernels of one scan.  Measured
# under rocprofv3 in separate passes, NOT in this run — labelled as such in the record.
CONFIG_PMC = {"expand_one": "expand_one", "expand_map": "expand_map", "delete_map": "delete_map", "cfg4": "cfg4_nft", "cfg4_guided": "cfg4_nft_guided", "cfg5_dft": "dict1000_dft", "cfg5_nft": "dict1000_dft", "expand": "expand_dft",
              "nft_loop": "nft_loop_guided", "dft_loop": "dft_loop_guided", "tile_fallback": "tile_dft"}


THIS_ROUND = "r06"          # PMC passes are collected per round (tools/profile_round.sh); a record that quotes another round's says so


def config_traffic(name):
    """HBM bytes per GiB of input from the committed PMC passes (separate rocprofv3 runs at 1 GiB, not this run).  This round's
    passes are preferred; an older round's are quoted only with "stale_round" in the record (same kernels or not, the reader is told)."""
    import glob
    import re
    tag = CONFIG_PMC.get(name)
    if not tag:
        return None
    total = 0.0
    src = []
    rounds = set()
    for c, mul in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc_%s.txt" % (tag, c))))
        if not files:
            return None
        vals = re.findall(r"^\s+%s\s+([0-9.]+)" % c, open(files[-1]).read(), re.M)
        if not vals:
            return None
        total += mul * sum(float(v) for v in vals) * 1024
        src.append(os.path.relpath(files[-1], ROOT))
        rounds.add(os.path.basename(files[-1]).split("_")[0])
    if len(rounds) != 1:
        return None                                   # FETCH and WRITE passes of different rounds do not add up to one scan
    rec = {"hbm_bytes_per_GiB_of_input": int(total),
           "source": "rocprofv3 --pmc passes at 1 GiB, committed (not this run): " + ", ".join(src) + "; FETCH_SIZE x 2, WRITE_SIZE x 1 (profiles/r05_fetch_calibration.txt)"}
    rnd = rounds.pop()
    if rnd != THIS_ROUND:
        rec["stale_round"] = rnd
    return rec


def cpu_baseline(pattern, engine, sample, cores_mt=0):
    """Time the reference's CPU path on the host cores over a bounded sample.  Prefers the compiled reference
    binary (kind "reference"); falls back to the oracle's C restatement (kind "port")."""
    from oracle_lib import Oracle, REF_DIR, ref_available, scan_mt
    nbytes = len(sample)
    if ref_available():
        binary = os.path.join(REF_DIR, "trre" if engine == "nft" else "trre_dft")
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
        with tempfile.NamedTemporaryFile(dir=tmpdir) as tf:
            tf.write(sample)
            tf.flush()
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as devnull:
                subprocess.run([binary, pattern, tf.name], stdout=devnull, check=True)
            dt = time.perf_counter() - t0
        res = {"value": round(nbytes / dt / 1e9, 6), "unit": "GB/s", "cores": 1, "kind": "reference"}
    else:
        o = Oracle(pattern, engine)
        t0 = time.perf_counter()
        o.scan(sample)
        dt = time.perf_counter() - t0
        res = {"value": round(nbytes / dt / 1e9, 6), "unit": "GB/s", "cores": 1, "kind": "port"}
    res["sample"] = "%.2f MiB of the same synthetic workload, file -> /dev/null, %.1f s" % (nbytes / 2**20, dt)
    if cores_mt:
        t0 = time.perf_counter()
        scan_mt(pattern, engine, cores_mt, sample)
        dt = time.perf_counter() - t0
        res["port_all_cores"] = {"value": round(nbytes / dt / 1e9, 4), "unit": "GB/s", "cores": cores_mt, "kind": "port",
                                 "sample": "same sample, line-sharded threads, %.1f s" % dt}
        # (the same figure as flat keys of cpu_baseline: a parser that keeps scalars only keeps it)
        res["value_all_cores"] = res["port_all_cores"]["value"]
        res["cores_all"] = cores_mt
        res["kind_all_cores"] = "port"
    return res


def host_sample(inp, nbytes):
    """the first nbytes of a device buffer as host bytes, cut after the last complete line"""
    s = inp[:min(inp.numel(), nbytes)].cpu().numpy().tobytes()
    return s[: s.rfind(b"\n") + 1]


def line_start_at_or_after(inp, pos):
    """first line start >= pos (device search in a 1 MiB window)"""
    if pos <= 0:
        return 0
    w = inp[pos - 1: pos - 1 + (1 << 20)]
    nl = (w == 10).nonzero()
    return pos + int(nl[0]) if nl.numel() else inp.numel()


def verify_scan(prog, oracle, inp, out, m, length_preserving, slice_bytes, tmp):
    """Checks one full-size scan against the oracle without a full-size CPU run.
    Length-preserving programs (output offset == input offset at every line start): oracle-checked slices at the
    head, right above the middle of the buffer (above the 4 GiB mark of an 8 GiB buffer) and at the tail.
    General programs: the buffer is scanned again in two halves cut at a line start above the middle; the full
    output must be the two half outputs back to back (offsets carry across the cut), and the head of each half
    output is oracle-checked.  Returns (ok, description)."""
    import torch
    n = inp.numel()
    cut = line_start_at_or_after(inp, n // 2 + (1 << 20) + 12345)
    if length_preserving:
        if m != n:
            return False, "output size %d != input size %d" % (m, n)
        starts = [0, cut, line_start_at_or_after(inp, max(n - slice_bytes, 0))]
        for s in starts:
            e = min(n, line_start_at_or_after(inp, min(n, s + slice_bytes)))
            if e <= s:
                continue
            want = oracle.scan(inp[s:e].cpu().numpy().tobytes())
            if out[s:e].cpu().numpy().tobytes() != want:
                return False, "slice at %d differs from the oracle" % s
        return True, "oracle-checked %d KiB slices at offsets %s" % (slice_bytes >> 10, starts)
    halves = []
    at = 0
    for lo, hi in ((0, cut), (cut, n)):
        if hi <= lo:
            continue
        part = prog.scan_tensor(inp[lo:hi], out=tmp)
        k = part.numel()
        if at + k > m or not torch.equal(out[at:at + k], part):
            return False, "full output differs from the half scans at output offset %d" % at
        e = min(hi, line_start_at_or_after(inp, min(hi, lo + slice_bytes)))
        want = oracle.scan(inp[lo:e].cpu().numpy().tobytes())
        if part[:len(want)].cpu().numpy().tobytes() != want:
            return False, "head of the half scan at input offset %d differs from the oracle" % lo
        halves.append(lo)
        at += k
    if at != m:
        return False, "half scans produce %d bytes, the full scan %d" % (at, m)
    return True, "full output == outputs of two half scans cut at input offset %d; %d KiB heads oracle-checked" % (cut, slice_bytes >> 10)


def verify_full(pattern, engine, inp, out, m, nbytes, threads):
    """The output against the oracle on ALL host cores (line-sharded threads: oracle_lib.scan_mt), slab by slab from the start of
    the buffer: the first `nbytes` of input (cut at a line end) and the output they become, every byte — the check VERDICT r3
    asked for in place of "slices and self-consistency" for variable-length outputs.  Returns (ok, description)."""
    from oracle_lib import scan_mt
    n = inp.numel()
    pos = opos = 0
    t0 = time.perf_counter()
    while pos < min(nbytes, n):
        end = min(n, pos + (1 << 30))
        slab = inp[pos:end].cpu().numpy().tobytes()
        if end < n:
            slab = slab[: slab.rfind(b"\n") + 1]
        want = scan_mt(pattern, engine, threads, slab)
        if opos + len(want) > m or out[opos:opos + len(want)].cpu().numpy().tobytes() != want:
            return False, "output differs from the all-cores oracle in the slab at input offset %d" % pos
        pos += len(slab)
        opos += len(want)
    if pos >= n and opos != m:
        return False, "the oracle's output ends at %d, the scan's at %d" % (opos, m)
    return True, "EVERY byte of the output of the first %.2f GiB of input against the oracle on %d host threads (%.0f s)" % (pos / 2**30, threads, time.perf_counter() - t0)


def run_config(trre_amd, spec, inp, out, tmp, want_cpu):
    """one record of the `configs` array"""
    import torch
    from oracle_lib import Oracle
    n = inp.numel()
    prog = trre_amd.Program(spec["pattern"], spec["engine"])
    if spec.get("force"):
        prog.set_kernel({v: k for k, v in trre_amd.KERNEL_NAMES.items()}[spec["force"]])
    info = prog.info
    rec = {"name": spec["name"], "workload": spec["workload"], "engine": spec["engine"], "bytes": n,
           "kernel_family": trre_amd.KERNEL_NAMES[info.kernel], "kernels": KERNEL_OF.get(trre_amd.KERNEL_NAMES[info.kernel], "?")}
    if len(spec["pattern"]) <= 64:
        rec["pattern"] = spec["pattern"]
    prog.enqueue(inp, out)
    m = prog.finish()                                          # warm-up: tables uploaded, workspaces sized
    prog.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(spec["steps"]):
        prog.enqueue(inp, out)
        if spec.get("finish_each"):
            m = prog.finish()            # (a scan that finish() has to run again on another family: every step pays for it)
    if not spec.get("finish_each"):
        m = prog.finish()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / spec["steps"]
    kernel_ms = prog.last_kernel_ms()
    if spec.get("finish_each"):
        kernel_ms = dt * 1e3             # (the events bracket the first launch only: a step that finish() completes with more launches is timed whole)
    rec.update({"output_bytes": m, "steps": spec["steps"], "ms_per_step": round(dt * 1e3, 4), "input_GBps": round(n / dt / 1e9, 1),
                "kernel_ms": round(kernel_ms, 4), "achieved_GBps": round(n / (kernel_ms * 1e-3) / 1e9, 1),
                "frac": round(n / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                "read_plus_write_GBps": round((n + m) / (kernel_ms * 1e-3) / 1e9, 1)})
    prog.set_profiling(False)
    if prog.info.kernel != info.kernel:          # (a bounded fold that overflowed: the scans after it run on the guided kernels)
        rec["kernel_family_after_first_scan"] = trre_amd.KERNEL_NAMES[prog.info.kernel]
    tr = config_traffic(spec["name"])
    if tr:
        # algorithmic bytes per GiB of input: the GiB itself and the output it becomes
        tr["x_algorithmic"] = round(tr["hbm_bytes_per_GiB_of_input"] / ((1 << 30) * (1.0 + m / float(n))), 2)
        rec["traffic"] = tr
    if spec.get("torch_check") is not None:
        ok = bool(m == n and spec["torch_check"](inp, out[:n]))
        rec["verified"], rec["verify"] = ok, "whole output against an independent torch byte map on the device"
    else:
        veng = spec.get("verify_engine", spec["engine"])
        oracle = Oracle(spec["pattern"], veng)
        lp = info.kernel in (trre_amd.KERNEL_BYTEMAP, trre_amd.KERNEL_TILE_LP, trre_amd.KERNEL_STREAM_LP, trre_amd.KERNEL_GUIDED_LP) and m == n
        rec["verified"], rec["verify"] = verify_scan(prog, oracle, inp, out, m, lp, spec.get("slice", 4 << 20), tmp)
        if veng != spec["engine"]:
            rec["verify"] += " (the %s oracle: the two engines agree on this prefix-free dictionary)" % veng.upper()
        if rec["verified"] and spec.get("own_head"):
            # and a head against the engine's own oracle (the NFT restatement does ~0.3 MB/s on this pattern)
            e = line_start_at_or_after(inp, spec["own_head"])
            want = Oracle(spec["pattern"], spec["engine"]).scan(inp[:e].cpu().numpy().tobytes())
            rec["verified"] = out[:len(want)].cpu().numpy().tobytes() == want
            rec["verify"] += "; %d KiB head against the %s oracle" % (spec["own_head"] >> 10, spec["engine"].upper())
        if rec["verified"] and spec.get("full_oracle"):
            ok, how = verify_full(spec["pattern"], veng, inp, out, m, spec["full_oracle"], os.cpu_count() or 1)
            rec["verified"] = ok
            rec["verify"] += "; " + how
    if want_cpu:
        rec["cpu_baseline"] = cpu_baseline(spec["pattern"], spec["engine"], host_sample(inp, spec["cpu_sample"]))
    prog.close()
    return rec


def cli_records_for(trre_amd, inp, out, n, want_cpu):
    """The binaries end to end, wall clock of the process (launch, HIP start-up, pattern compilation, file -> stdout):
      cfg1      BASELINE configs[0]: 'cat:dog' on 1 MB of ASCII through trre_amd/bin/trre, beside the reference binary on the same file
      cli_cfg2  the headline scan through trre_amd/bin/trre_dft: the corpus as a file in /dev/shm -> /dev/null
    """
    import random
    import shutil
    import torch
    import corpus as tcorpus
    from oracle_lib import REF_DIR, ref_available
    recs = []
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    bindir = os.path.join(ROOT, "trre_amd", "bin")

    def wall(cmd, stdout_path=None):
        t0 = time.perf_counter()
        with open(stdout_path or os.devnull, "wb") as so:
            r = subprocess.run(cmd, stdout=so, stderr=subprocess.PIPE)
        return time.perf_counter() - t0, r.returncode, r.stderr

    with tempfile.TemporaryDirectory(dir=shm) as td:
        # ---- cfg1 ----
        small = tcorpus.word_soup(random.Random(1), 1000000)
        f1, empty = os.path.join(td, "cfg1.txt"), os.path.join(td, "empty")
        open(f1, "wb").write(small)
        open(empty, "wb").close()
        o_cli, o_ref = os.path.join(td, "o_cli"), os.path.join(td, "o_ref")
        t_cli = min(wall([os.path.join(bindir, "trre"), "cat:dog", f1], o_cli)[0] for _ in range(3))
        t_empty = min(wall([os.path.join(bindir, "trre"), "cat:dog", empty])[0] for _ in range(3))
        rec = {"name": "cfg1", "workload": "BASELINE configs[0]: 'cat:dog' scan of %d bytes of ASCII words through the trre binary, file -> file (wall clock of the process)" % len(small),
               "pattern": "cat:dog", "engine": "nft", "bytes": len(small), "cli_wall_ms": round(t_cli * 1e3, 1), "cli_wall_ms_empty_input": round(t_empty * 1e3, 1),
               "input_GBps": round(len(small) / t_cli / 1e9, 5),
               "note": "launch + HIP start-up + pattern compilation dominate: the same binary on an EMPTY file takes cli_wall_ms_empty_input"}
        if ref_available():
            t_ref = min(wall([os.path.join(REF_DIR, "trre"), "cat:dog", f1], o_ref)[0] for _ in range(3))
            rec["reference_wall_ms"] = round(t_ref * 1e3, 1)
            rec["reference_GBps"] = round(len(small) / t_ref / 1e9, 5)
            rec["verified"] = open(o_cli, "rb").read() == open(o_ref, "rb").read()
            rec["verify"] = "stdout of the two binaries compared byte for byte"
        else:
            from oracle_lib import Oracle
            rec["verified"] = open(o_cli, "rb").read() == Oracle("cat:dog", "nft").scan(small)
            rec["verify"] = "stdout against the oracle"
        recs.append(rec)
        # ---- cli_cfg2 ----
        free = shutil.disk_usage(td).free
        nb = n
        while nb > (64 << 20) and nb + (2 << 30) > free:
            nb //= 2
        f2 = os.path.join(td, "cfg2.txt")
        with open(f2, "wb") as f:
            for lo in range(0, nb, 1 << 30):
                f.write(inp[lo:min(nb, lo + (1 << 30))].cpu().numpy().tobytes())
        dft = os.path.join(bindir, "trre_dft")
        wall([dft, "[a:A-z:Z]", f2])                               # (page cache and driver warm)
        t_full = min(wall([dft, "[a:A-z:Z]", f2])[0] for _ in range(2))
        t_empty = min(wall([dft, "[a:A-z:Z]", empty])[0] for _ in range(3))
        one = os.path.join(td, "one_line")
        open(one, "wb").write(b"a\n")
        t_one = min(wall([dft, "[a:A-z:Z]", one])[0] for _ in range(3))    # (an empty file starts no HIP runtime; one line does)
        # what the binary prints, checked on the first GiB (file -> file) against the device scan of the same bytes
        vb = min(nb, 1 << 30)
        if vb < nb:
            vb = line_start_at_or_after(inp, vb - (1 << 16))          # (whole records: the head file must not end inside one)
        f3, o3 = os.path.join(td, "head.txt"), os.path.join(td, "head.out")
        with open(f3, "wb") as f:
            f.write(inp[:vb].cpu().numpy().tobytes())
        _, rc3, err3 = wall([dft, "[a:A-z:Z]", f3], o3)
        import numpy as np
        got = torch.from_numpy(np.fromfile(o3, dtype=np.uint8))
        low = (inp[:vb] >= 97) & (inp[:vb] <= 122)
        want = torch.where(low, inp[:vb] - 32, inp[:vb])
        if vb == nb:
            want = want.clone()
            if int(want[-1]) != 10:
                want[-1] = 10                                      # (a last record without its newline loses its last byte: trre_dft.c:1274)
        ok = rc3 == 0 and got.numel() == vb and bool(torch.equal(got.to(inp.device), want))
        # the same through a pipe (cat file | trre_dft): read() instead of parallel pread()
        t0 = time.perf_counter()
        with open(os.devnull, "wb") as so:
            pc = subprocess.Popen(["cat", f2], stdout=subprocess.PIPE)
            subprocess.run([dft, "[a:A-z:Z]"], stdin=pc.stdout, stdout=so)
            pc.wait()
        t_pipe = time.perf_counter() - t0
        recs.append({"name": "cli_cfg2", "workload": "the headline scan through the trre_dft binary: %.2f GiB file in /dev/shm -> /dev/null, wall clock of the process (best of 2)" % (nb / 2**30),
                     "pattern": "[a:A-z:Z]", "engine": "dft", "bytes": nb, "cli_wall_ms": round(t_full * 1e3, 1), "cli_wall_ms_empty_input": round(t_empty * 1e3, 1),
                     "cli_wall_ms_one_line": round(t_one * 1e3, 1),
                     "input_GBps": round(nb / t_full / 1e9, 2), "streaming_GBps": round(nb / max(t_full - t_one, 1e-9) / 1e9, 2),
                     "pipe_input_GBps": round(nb / t_pipe / 1e9, 2),
                     "note": "input_GBps = bytes / wall; streaming_GBps = bytes / (wall - the wall of the same binary on a one-line file, i.e. the process, the HIP runtime's start-up and the first launch): what the reader / scan / writer "
                             "pipeline sustains once the process is up; pipe_input_GBps: `cat file | trre_dft` (one read() stream)",
                     "verified": bool(ok), "verify": "stdout for the first %.2f GiB (file -> file) against an independent torch byte map" % (vb / 2**30)})
    return recs


LINE_BUDGET = 3500          # bytes; the driver keeps an 8 KB tail of stdout and parses its last line (round 5's 23 KB line was cut: parsed null)
DETAILS_FILE = "bench_configs.json"


def _num(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(line):
    """The ONE line the driver parses: the contract's keys, `roofline` and `cpu_baseline` as flat scalars, and a numbers-only summary
    of the `configs` records.  The prose (workload / verify / traffic sources / notes) stays in DETAILS_FILE."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: line.get(k) for k in keep}
    cfg = line.get("config") or {}
    out["config"] = {k: cfg.get(k) for k in ("workload", "pattern", "engine", "corpus", "bytes_per_gpu", "output_bytes_per_gpu", "kernel", "parallelism")}
    rf = line.get("roofline") or {}
    out["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch",
                                              "achieved_read_plus_write")}
    cb = line.get("cpu_baseline")
    out["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "value_all_cores", "cores_all", "kind_all_cores") if k in cb} if cb else None
    out["verified"] = line.get("verified")
    for k in ("pcie_inclusive_GBps", "pcie_inclusive_pinned_GBps"):
        if k in line:
            out[k] = line[k]
    if "configs" in line:
        short = []
        for c in line["configs"]:
            r = {"name": c.get("name")}
            if "n_gpus" in c:
                r["n_gpus"], r["scaling"] = c["n_gpus"], c.get("scaling")
            if "input_GBps" in c:
                r["GBps"] = _num(c["input_GBps"])
            if "frac" in c or "frac_per_gpu_rank0" in c:
                r["frac"] = _num(c.get("frac", c.get("frac_per_gpu_rank0")))
            if isinstance(c.get("traffic"), dict):
                r["x_alg"] = c["traffic"].get("x_algorithmic")
                if c["traffic"].get("stale_round"):
                    r["x_alg_round"] = c["traffic"]["stale_round"]
            if "cli_wall_ms" in c:
                r["wall_ms"] = c["cli_wall_ms"]
            elif "input_GBps" not in c and "ms_per_step" in c:
                r["ms"] = c["ms_per_step"]
            r["verified"] = c.get("verified")
            short.append(r)
        out["configs"] = short
        out["configs_verified"] = line.get("configs_verified")
    out["details"] = DETAILS_FILE
    text = json.dumps(out, separators=(",", ":"))
    # never outgrow the budget: shed the optional parts, the contract's keys last
    for drop in ("configs", "pcie_inclusive_pinned_GBps", "pcie_inclusive_GBps"):
        if len(text) <= LINE_BUDGET:
            break
        if drop == "configs" and "configs" in out:
            out["configs"] = [{k: r[k] for k in ("name", "GBps", "frac", "verified") if k in r} for r in out["configs"]]
            text = json.dumps(out, separators=(",", ":"))
            if len(text) <= LINE_BUDGET:
                break
        out.pop(drop, None)
        text = json.dumps(out, separators=(",", ":"))
    if len(text) > LINE_BUDGET:
        out["config"]["workload"] = (out["config"].get("workload") or "")[:120]
        if out.get("cpu_baseline"):
            out["cpu_baseline"]["sample"] = (out["cpu_baseline"].get("sample") or "")[:80]
        text = json.dumps(out, separators=(",", ":"))
    assert len(text) <= LINE_BUDGET + 500, len(text)
    return text


def emit(line):
    """rank 0: the full record (every config with its prose) to DETAILS_FILE next to this script (and to gpurun_out/ when that exists,
    so it comes back from a GPU box); then, as the LAST line of stdout, the compact line."""
    full = json.dumps(line, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, DETAILS_FILE), "w") as f:
                    f.write(full + "\n")
            except OSError as e:
                print("bench.py: could not write %s in %s: %r" % (DETAILS_FILE, d, e), file=sys.stderr)
    sys.stderr.flush()
    print(compact_line(line), flush=True)


def switched_record(n, name, switch, pat, eng, kernels, what, void_mark, pmc_tag):
    """One pattern through an opt-in form (`switch`=1) in a child process: its rate (HIP events of the batch), its output's checksum against
    the count / emit pair's (a second child without the switch), the committed PMC passes of this round for its traffic."""
    import re

    def child(env):
        e = dict(os.environ)
        e.pop(switch, None)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "--bytes", str(n), "--steps", "10", "--sum", "--case",
                            "%s;;%s;;printable;;auto" % (pat, eng)], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        m = re.search(r"out=(\d+)\s+([0-9.]+) ms/step.*\(events ([0-9.]+) ms\)\s+sum=([0-9a-f]+)", r.stdout.decode())
        if r.returncode or not m:
            raise RuntimeError("kbench failed: " + r.stderr.decode()[-300:])
        return int(m.group(1)), float(m.group(2)), float(m.group(3)), m.group(4), r.stderr.decode()
    m1, ms1, ev1, sum1, err1 = child({switch: "1", "TRRE_TRACE": "1"})
    m0, ms0, ev0, sum0, _ = child({"TRRE_MAPGEN": "0"})          # (the count / emit pair: no one-pass form, not even the ones that are on by default)
    void = void_mark in err1 and "void" in err1
    rec = {"name": name, "pattern": pat, "engine": eng, "bytes": n, "output_bytes": m1, "kernel_family": "stream_gen", "kernels": kernels,
           "workload": "%s: '%s' %s, %.0f GiB, a child process with %s=1; beside it the count / emit pair in a child without the switch: %.3f ms (%.1f GB/s)"
                       % (what, pat, eng.upper(), n / 2**30, switch, ev0, n / (ev0 * 1e-3) / 1e9),
           "steps": 10, "ms_per_step": ms1, "kernel_ms": ev1, "input_GBps": round(n / (ms1 * 1e-3) / 1e9, 1), "achieved_GBps": round(n / (ev1 * 1e-3) / 1e9, 1),
           "frac": round(n / (ev1 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "pair_kernel_ms": ev0,
           "verified": bool(m1 == m0 and sum1 == sum0 and not void),
           "verify": "output size and a position-weighted checksum of every output byte equal the pair's; no void launch in the trace"}
    tr = config_traffic(pmc_tag)
    if tr:
        tr["x_algorithmic"] = round(tr["hbm_bytes_per_GiB_of_input"] / ((1 << 30) * (1.0 + m1 / float(n))), 2)
        rec["traffic"] = tr
    return rec


def one_walk_record(n):
    """'a:xyz' through the one-walk form of the general families (TRRE_ONE=1, one_block.hpp)"""
    return switched_record(n, "expand_one", "TRRE_ONE", "a:xyz", "dft", "k_stream_one (TRRE_ONE=1: one walk, the workgroup's output staged in LDS, look-back for its place)",
                           "general path in ONE walk (row f2, opt-in)", "one-pass launch", "expand_one")


def mapgen_records(n):
    """'a:xyz' and '[aie]:' through the memoryless one-pass kernel (TRRE_MAPGEN=1, map_block.hpp)"""
    k = "k_mapgen (TRRE_MAPGEN=1: no walk — lengths, DPP prefix sums, look-back, the texts at their places in an LDS window; one read of the input)"
    return [switched_record(n, "expand_map", "TRRE_MAPGEN", "a:xyz", "dft", k, "a memoryless program in ONE pass (row f2, opt-in)", "one-pass launch", "expand_map"),
            switched_record(n, "delete_map", "TRRE_MAPGEN", "[aie]:", "nft", k, "a memoryless program in ONE pass (row f2; the default for programs that print one byte or none per byte)", "one-pass launch", "delete_map")]


class StubProgram:
    """TRRE_BENCH_STUB=1 (tests/test_bench_spawn.py, no GPU): stands in for trre_amd.Program so that the launch, barrier,
    timing and reduction plumbing of the N-rank path runs on CPU over gloo.  It scans nothing; the line it yields says
    "data": "stub" and is not a measurement."""

    class _Info:
        kernel = 1
        table_rows = 0

    info = _Info()

    def __init__(self, *a, **k):
        self._n = 0

    def set_kernel(self, fam):
        pass

    def set_profiling(self, on=True):
        pass

    def enqueue(self, inp, out, stream=None):
        self._n = inp.numel()
        out[:self._n] = inp
        time.sleep(0.002)

    def finish(self):
        return self._n

    def last_kernel_ms(self):
        return 2.0

    def close(self):
        pass


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n_gpus, stub):
    """`python bench.py --gpus N` outside a launcher: start N ranks of this script, one per GPU (the same command line the
    driver uses: torch.distributed.run, rendezvous on 127.0.0.1).  Fails loudly when fewer than N devices are visible."""
    if not stub:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n_gpus and not (have and os.environ.get("TRRE_BENCH_SHARE_GPU") == "1"):
            raise SystemExit("bench.py: --gpus %d, but %d GPU(s) are visible on this node: nothing measured "
                             "(every rank needs a device of its own; the scan has no CPU path)" % (n_gpus, have))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bytes", type=int, default=8 << 30, help="input bytes per GPU (strong scaling: of the whole job)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--pattern", default=None, help="headline pattern (default: '[a:A-z:Z]', or config 4's with --scaling strong)")
    ap.add_argument("--engine", default=None, choices=["dft", "nft"])
    ap.add_argument("--corpus", default=None, help="printable | catdog | dict<N>")
    ap.add_argument("--kernel", default="auto", choices=["auto", "bytemap", "tile_lp", "tile_gen", "stream_lp", "stream_gen", "guided_lp", "guided_gen"])
    ap.add_argument("--cpu-sample-mib", type=int, default=256)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs array and the host-buffer rate")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    stub = os.environ.get("TRRE_BENCH_STUB") == "1"

    # ---- who runs: N ranks, one per GPU -------------------------------------------------------------------------------
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus, stub))               # N workers of this script; rank 0 of them prints the line
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but launched with WORLD_SIZE=%d: the line would misreport the GPUs used" % (args.gpus, world))

    import torch
    import corpora
    share = False
    if stub:
        import types
        dev = torch.device("cpu")
        trre_amd = types.SimpleNamespace(Program=StubProgram, KERNEL_NAMES={0: "auto", 1: "bytemap"}, KERNEL_BYTEMAP=1, KERNEL_TILE_LP=2,
                                         KERNEL_STREAM_LP=4, KERNEL_GUIDED_LP=6)
        sync = lambda: None                                   # noqa: E731
    else:
        import trre_amd
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the scan has no CPU path")
        # TRRE_BENCH_SHARE_GPU=1 (a debugging aid for 1-GPU boxes, marked in the line): ranks beyond the devices share
        # them and synchronise over gloo — the N-rank code path on real kernels, never a measurement of N GPUs
        share = os.environ.get("TRRE_BENCH_SHARE_GPU") == "1" and world > torch.cuda.device_count()
        if local >= torch.cuda.device_count() and not share:
            raise SystemExit("bench.py: rank %d has no device (%d visible)" % (rank, torch.cuda.device_count()))
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        sync = torch.cuda.synchronize
    dist = None
    if world > 1 or "RANK" in os.environ:          # launched by torch.distributed.run: one rank per GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo: the data path has no collective (north_star: "no RCCL collectives"); the only exchanges are the timing barrier and
        # scalar reductions of elapsed time / flags / byte counts, and they run on host tensors
        dist.init_process_group("gloo")

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    def reduce(x, op):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN, "sum": dist.ReduceOp.SUM}[op])
        return float(t.item())

    ranks_running = int(reduce(1.0, "sum"))                   # the ranks that really run (reported as n_gpus)

    def timed(p, inp, out, steps):
        """the contract's timed region: barrier + synchronize on both sides, EXACTLY `steps` scans, MAX over ranks"""
        p.set_profiling(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            p.enqueue(inp, out)
        m = p.finish()
        sync()
        elapsed = time.perf_counter() - t0
        barrier()
        kernel_ms = p.last_kernel_ms()                        # HIP events on the launch stream, avg per launch
        p.set_profiling(False)
        return reduce(elapsed, "max"), m, kernel_ms

    strong = args.scaling == "strong"
    pattern = args.pattern or ("(cat:dog|dog:cat)" if strong else "[a:A-z:Z]")
    engine = args.engine or ("nft" if strong else "dft")
    corpus_name = args.corpus or ("catdog" if strong else "printable")
    n = args.bytes // world if strong else args.bytes         # this rank's line shard
    cfg_index = {"printable": 2, "catdog": 4}.get(corpus_name, 5)
    fam = {v: k for k, v in trre_amd.KERNEL_NAMES.items()}[args.kernel] if not stub else 0
    prog = trre_amd.Program(pattern, engine)
    prog.set_kernel(fam)
    info = prog.info
    inp = corpora.by_name("printable" if stub else corpus_name, n, corpora.SEED0 + cfg_index + 1000 * rank, dev)
    out = torch.empty(n + n // 4 + 4096, dtype=torch.uint8, device=dev)

    def run_steps(p, k):
        for _ in range(k):
            p.enqueue(inp, out)
        return p.finish()

    # warmup + one verified pass
    m = run_steps(prog, max(args.warmup, 1))
    if stub:
        verified, verify_how = True, "stub: nothing scanned"
    elif pattern == "[a:A-z:Z]":
        low = (inp >= 97) & (inp <= 122)
        verified = bool(m == n and torch.equal(out[:n], torch.where(low, inp - 32, inp)))
        verify_how = "whole output against an independent torch byte map on the device"
        del low
    else:
        from oracle_lib import Oracle
        lp = info.kernel in (trre_amd.KERNEL_BYTEMAP, trre_amd.KERNEL_TILE_LP, trre_amd.KERNEL_STREAM_LP, trre_amd.KERNEL_GUIDED_LP)
        tmp = None if lp else torch.empty(out.numel() // 2 + (2 << 20), dtype=torch.uint8, device=dev)
        verified, verify_how = verify_scan(prog, Oracle(pattern, engine), inp, out, m, lp, 1 << 20, tmp)
        del tmp
    if reduce(1.0 if verified else 0.0, "min") < 1.0:
        raise SystemExit("bench: the output is wrong on some rank (rank %d: %s)" % (rank, verify_how))

    elapsed, m, kernel_ms = timed(prog, inp, out, args.steps)

    ms_per_step = elapsed / args.steps * 1e3
    value = world * n / (elapsed / args.steps) / 1e9          # whole-job input GB/s
    achieved = n / (kernel_ms * 1e-3) / 1e9                   # algorithmic: 1 byte read per input byte
    kname = trre_amd.KERNEL_NAMES[info.kernel]
    traffic, traffic_source = pmc_traffic("k_bytemap", n) if info.kernel == trre_amd.KERNEL_BYTEMAP and not stub else (None, None)
    cfg_note = ""
    if (pattern, engine, corpus_name) == ("[a:A-z:Z]", "dft", "printable"):
        cfg_note = " (BASELINE.json configs[1]" + (")" if n == 1 << 30 else " at north_star's size: >= 8 GiB)" if n >= 8 << 30 else " at another size)")
    line = {
        "metric": "input GB/s (scan mode)",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": ranks_running,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8",
        "data": "stub" if stub else ("synthetic; DEBUG: %d ranks share %d GPU(s), not a measurement of %d GPUs" % (world, torch.cuda.device_count(), world)
                                     if share else "synthetic"),
        "config": {
            "workload": "'%s' %s scan over %.3f GiB of synthetic %s lines per GPU%s"
                        % (pattern, engine.upper(), n / 2**30, corpus_name, cfg_note),
            "pattern": pattern, "engine": engine, "corpus": corpus_name, "bytes_per_gpu": n, "output_bytes_per_gpu": m,
            "kernel": kname, "table_rows": info.table_rows,
            "parallelism": "line-sharded x%d, one rank per GPU, no data-path collective" % world,
        },
        "roofline": {
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": KERNEL_OF.get(kname, kname),
            "kernel_ms": round(kernel_ms, 4),
            "algorithmic_bytes_per_launch": n,
            "achieved_read_plus_write": round((n + m) / (kernel_ms * 1e-3) / 1e9, 1),
            "note": "per GPU (rank 0)" if world > 1 else None,
        },
        "verified": verified,
        "verify": verify_how,
    }

    extras = rank == 0 and world == 1 and not args.no_extras and not stub
    want_cpu = rank == 0 and world == 1 and not args.no_cpu and not stub
    if want_cpu:
        line["cpu_baseline"] = cpu_baseline(pattern, engine, host_sample(inp, args.cpu_sample_mib << 20), os.cpu_count() or 1)
    elif rank == 0:
        line["cpu_baseline"] = None

    if world > 1 and not args.no_extras:
        # ---- the BASELINE configurations that shard, in the N-GPU line ---------------------------------------------------
        #   configs[3]  '(cat:dog|dog:cat)' NFT over ONE corpus of --bytes, line-sharded over the ranks (strong scaling)
        #   configs[4]  the 1000-entry dictionary, --bytes per GPU (weak scaling: 64 GiB at 8 x 8 GiB)
        # Same timed region as the headline (barrier, K scans, MAX over ranks); every rank verifies its own shard.
        import dictgen
        keys, vals = dictgen.make_dictionary(1000)
        multi = [
            {"name": "cfg4_strong", "pattern": "(cat:dog|dog:cat)", "engine": "nft", "corpus": "catdog", "seed": 4, "scaling": "strong",
             "bytes": args.bytes // world, "steps": 20,
             "workload": "BASELINE configs[3]: '(cat:dog|dog:cat)' NFT scan, one %.0f GiB word-soup corpus line-sharded over %d GPUs"
                         % (args.bytes / 2**30, world)},
            {"name": "cfg5_weak", "pattern": dictgen.pattern(keys, vals), "engine": "dft", "corpus": "dict1000", "seed": 5, "scaling": "weak",
             "bytes": args.bytes, "steps": 10,
             "workload": "BASELINE configs[4]: 1000-entry key:value dictionary, DFT engine, %.0f GiB per GPU = %.0f GiB over %d GPUs"
                         % (args.bytes / 2**30, args.bytes * world / 2**30, world)},
        ]
        del inp
        configs = []
        for spec in multi:
            nb = spec["bytes"]
            inp = corpora.by_name("printable" if stub else spec["corpus"], nb, corpora.SEED0 + spec["seed"] + 1000 * rank, dev)
            p = trre_amd.Program(spec["pattern"], spec["engine"])
            pinfo = p.info
            p.enqueue(inp, out)
            pm = p.finish()                                  # warm-up: tables uploaded, workspaces sized
            if stub:
                ok, how = True, "stub: nothing scanned"
            else:
                from oracle_lib import Oracle
                lp = pinfo.kernel in (trre_amd.KERNEL_BYTEMAP, trre_amd.KERNEL_TILE_LP, trre_amd.KERNEL_STREAM_LP, trre_amd.KERNEL_GUIDED_LP)
                tmp = None if lp else torch.empty(out.numel() // 2 + (2 << 20), dtype=torch.uint8, device=dev)
                ok, how = verify_scan(p, Oracle(spec["pattern"], spec["engine"]), inp, out, pm, lp, 2 << 20, tmp)
                del tmp
            all_ok = reduce(1.0 if ok else 0.0, "min") >= 1.0
            el, pm, kms = timed(p, inp, out, spec["steps"])
            total_in = reduce(float(nb), "sum")
            rec = {"name": spec["name"], "workload": spec["workload"], "engine": spec["engine"], "scaling": spec["scaling"],
                   "n_gpus": ranks_running, "bytes_per_gpu": nb, "bytes_total": int(total_in), "steps": spec["steps"],
                   "ms_per_step": round(el / spec["steps"] * 1e3, 4), "input_GBps": round(total_in / (el / spec["steps"]) / 1e9, 1),
                   "kernel_family": trre_amd.KERNEL_NAMES.get(pinfo.kernel, "?"), "kernel_ms_rank0": round(kms, 4),
                   "frac_per_gpu_rank0": round(nb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                   "verified": all_ok, "verify": "every rank, its own shard: " + how}
            if len(spec["pattern"]) <= 64:
                rec["pattern"] = spec["pattern"]
            configs.append(rec)
            p.close()
            del inp
        line["configs"] = configs
        line["configs_verified"] = all(c["verified"] for c in configs)
        inp = None

    if extras:
        # PCIe-inclusive rate through trre_scan_host on 1 GiB of pageable host memory, timed around the C call (never `value`)
        import ctypes
        import numpy as np
        hn = min(n, 1 << 30)
        if hn < n:
            hn = line_start_at_or_after(inp, hn - (1 << 16))            # (whole records)
        host = inp[:hn].cpu().numpy()
        hout = np.empty(hn + 4096, dtype=np.uint8)
        hm = ctypes.c_size_t()
        L = trre_amd.api.lib()
        best = None
        for _ in range(3):                                   # first call: staging buffers are allocated, pages touched
            t0 = time.perf_counter()
            rc = L.trre_scan_host(prog._h, host.ctypes.data_as(ctypes.c_char_p), hn, hout.ctypes.data_as(ctypes.c_char_p),
                                  hout.size, ctypes.byref(hm), local)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        line["pcie_inclusive_GBps"] = round(hn / best / 1e9, 2) if rc == 0 else None
        line["pcie_inclusive_note"] = "trre_scan_host, %.2f GiB pageable host in -> pageable host out, best of 3" % (hn / 2**30)
        del host, hout

        # the same with a pinned caller buffer (hipHostMalloc through torch): the library sends it over the link as it is, no staging copies
        try:
            pin_in = torch.empty(hn, dtype=torch.uint8).pin_memory()
            pin_out = torch.empty(hn + 4096, dtype=torch.uint8).pin_memory()
            pin_in.copy_(inp[:hn])
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                rc = L.trre_scan_host(prog._h, ctypes.c_char_p(pin_in.data_ptr()), hn, ctypes.c_char_p(pin_out.data_ptr()), hn + 4096, ctypes.byref(hm), local)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            line["pcie_inclusive_pinned_GBps"] = round(hn / best / 1e9, 2) if rc == 0 and hm.value == hn and bool(torch.equal(pin_out[:hn].to(dev), out[:hn])) else None
            del pin_in, pin_out
        except Exception as e:      # (the headline must not depend on it)
            line["pcie_inclusive_pinned_GBps"] = None
            line["pcie_inclusive_pinned_note"] = "failed: %r" % (e,)

        # ---- the command-line work-alikes end to end (BASELINE configs[0]; VERDICT r4: the binary, not the library call) ----
        cli_records = []
        try:
            cli_records = cli_records_for(trre_amd, inp, out, n, want_cpu)
        except Exception as e:
            cli_records = [{"name": "cli", "verified": False, "verify": "failed: %r" % (e,)}]

        # ---- the other configurations and kernel families, same JSON line --------------------------------------------
        import dictgen
        keys, vals = dictgen.make_dictionary(1000)
        dict_pat = dictgen.pattern(keys, vals)

        def caesar_check(x, y):
            up = torch.where((x >= 97) & (x <= 121), x + 1, torch.where(x == 122, torch.full_like(x, 97), x))
            return torch.equal(y, up)

        tmp = torch.empty(out.numel() // 2 + (2 << 20), dtype=torch.uint8, device=dev)
        configs = list(cli_records)
        printable = [
            {"name": "cfg3", "workload": "BASELINE configs[2]: Caesar '[a:b-y:zz:a]' DFT scan, %.0f GiB printable lines" % (n / 2**30),
             "pattern": "[a:b-y:zz:a]", "engine": "dft", "steps": 50, "torch_check": caesar_check, "cpu_sample": 128 << 20},
            {"name": "expand", "workload": "general path, expanding output: 'a:xyz' DFT", "pattern": "a:xyz", "engine": "dft", "steps": 20,
             "cpu_sample": 64 << 20, "full_oracle": 1 << 62},
            {"name": "nft_loop", "workload": "NFT pattern that does not fold (a loop before the decision): '(a|b)*c:x'",
             "pattern": "(a|b)*c:x", "engine": "nft", "steps": 20, "cpu_sample": 16 << 20, "full_oracle": 1 << 62},
            {"name": "dft_loop", "workload": "DFT pattern that does not fold (a loop before the decision; round 3: the tile kernels at 48 GB/s): '(a|b)*c:x'",
             "pattern": "(a|b)*c:x", "engine": "dft", "steps": 20, "cpu_sample": 16 << 20, "full_oracle": 1 << 62},
            {"name": "dft_suffix", "workload": "DFT, a range under a loop before a literal: '[a-z]+ing:X'", "pattern": "[a-z]+ing:X", "engine": "dft", "steps": 20,
             "cpu_sample": 16 << 20},
            {"name": "nft_range_loop", "workload": "NFT, a byte range under a loop: '[0-9]+:N'", "pattern": "[0-9]+:N", "engine": "nft", "steps": 20,
             "cpu_sample": 16 << 20},
            {"name": "nft_dot", "workload": "NFT, '.' rows of the reference's test.sh:114: '(.:x)*.*'", "pattern": "(.:x)*.*", "engine": "nft",
             "steps": 20, "cpu_sample": 4 << 20},
            {"name": "nft_greedy", "workload": "NFT, greedy loop ' +: ' (bounded fold, runs <= 64)", "pattern": " +: ", "engine": "nft", "steps": 20,
             "cpu_sample": 16 << 20},
        ]
        # BASELINE configs[1] at the size its text names (1 GiB; the headline above is the same scan at north_star's >= 8 GiB)
        def upper_check(x, y):
            return torch.equal(y, torch.where((x >= 97) & (x <= 122), x - 32, x))

        one_gib = corpora.printable_lines(1 << 30, corpora.SEED0 + 2, dev)
        configs.append(run_config(trre_amd, {"name": "cfg2_1gib", "workload": "BASELINE configs[1] as stated: '[a:A-z:Z]' DFT scan over 1 GiB of synthetic ASCII lines, 1 GPU",
                                             "pattern": "[a:A-z:Z]", "engine": "dft", "steps": 50, "torch_check": upper_check, "cpu_sample": 0}, one_gib, out, tmp, False))
        del one_gib
        for spec in printable:
            configs.append(run_config(trre_amd, spec, inp, out, tmp, want_cpu))
        # the general family in ONE walk (round 6, row f2: one_block.hpp — opt-in, TRRE_ONE=1, read once per process: a child of tools/kbench.py
        # with the switch, a second one without it for the checksum of the pair's output)
        try:
            configs.append(one_walk_record(n))
        except Exception as e:      # (the headline must not depend on it)
            configs.append({"name": "expand_one", "verified": False, "verify": "failed: %r" % (e,)})
        try:
            configs.extend(mapgen_records(n))
        except Exception as e:
            configs.append({"name": "expand_map", "verified": False, "verify": "failed: %r" % (e,)})
        nt = min(n, 1 << 30)
        # long lines (round 5: exact sub-ranges): the same text with one line end left per 400 KB — JSON lines, minified files
        try:
            ll = corpora.long_lines(nt, corpora.SEED0 + 2, dev, 400000)
            for lname, lpat, leng in (("long_lines_greedy", " +: ", "nft"), ("long_lines_loop", "(a|b)*c:x", "nft"), ("long_lines_cfg4", "(cat:dog|dog:cat)", "nft")):
                rec = run_config(trre_amd, {"name": lname, "pattern": lpat, "engine": leng, "steps": 10, "cpu_sample": 0,
                                            "workload": "'%s' on %.0f GiB of text with ONE line end per 400 KB (rounds 1-4: a lane walked every such line alone, "
                                                        "5.8-9.0 GB/s)" % (lpat, nt / 2**30)}, ll, out, tmp, False)
                rec["avg_line_bytes"] = round(nt / max(int((ll == 10).sum()), 1))
                configs.append(rec)
            del ll
        except Exception as e:      # (the headline must not depend on it)
            configs.append({"name": "long_lines", "verified": False, "verify": "failed: %r" % (e,)})
        # the fallbacks, measured rather than assumed: the headline scan on the same corpus with ONE NUL
        # byte per GiB — a NUL cuts its line short (C-string semantics, trre_dft.c:1277), so the positional launch is void and
        # the whole buffer runs again on the general family —, and the tile kernels (what a DFT pattern that does not fold runs on)
        nul_at = [int((k + 0.5) * (1 << 30)) for k in range(n >> 30)] or [n // 2]
        saved = inp[nul_at].clone()
        inp[nul_at] = 0
        configs.append(run_config(trre_amd, {"name": "cfg2_with_nuls", "pattern": "[a:A-z:Z]", "engine": "dft", "steps": 5, "cpu_sample": 0, "finish_each": True,
                                             "workload": "headline scan, %d NUL byte(s) in the %.0f GiB: bytemap launch void, the buffer runs again on the "
                                                         "general family (ms_per_step = both launches, host-timed; kernel_ms = the second alone)"
                                                         % (len(nul_at), n / 2**30)}, inp, out, tmp, False))
        inp[nul_at] = saved
        # a bounded fold that meets a run it was not built for (' +: ' folds runs of up to 64 spaces; one run of 300 per GiB): the count
        # pass finds out, the emit pass leaves at once, the guided kernels take the buffer — and the scans after it go there directly
        run_at = [int((k + 0.5) * (1 << 30)) for k in range(n >> 30)] or [n // 2]
        idx = torch.cat([torch.arange(p0, p0 + 300, device=dev) for p0 in run_at])
        saved = inp[idx].clone()
        inp[idx] = 32
        configs.append(run_config(trre_amd, {"name": "greedy_overflow", "pattern": " +: ", "engine": "nft", "steps": 5, "cpu_sample": 0, "finish_each": True,
                                             "workload": "' +: ' NFT with %d run(s) of 300 spaces in the %.0f GiB: the bounded stream table overflows on the first scan "
                                                         "(count pass void, emit pass skipped, guided kernels take over), later scans go to the guided "
                                                         "kernels directly (ms_per_step host-timed over all steps but the warm-up, which is the one that overflows)"
                                                         % (len(run_at), n / 2**30)}, inp, out, tmp, False))
        inp[idx] = saved
        del idx, saved
        nt = min(n, 1 << 30)
        configs.append(run_config(trre_amd, {"name": "tile_fallback", "pattern": "a:xyz", "engine": "dft", "steps": 3, "force": "tile_gen",
                                             "workload": "the LDS-tile kernels (fallback of last resort for DFT patterns that do not fold): 'a:xyz' forced "
                                                         "through tile_gen, %.0f GiB" % (nt / 2**30)}, inp[:nt], out, tmp, False))
        # the last resort of the DFT engine (round 5): patterns beyond any eager determinisation (rounds 1-4: TRRE_E_TOO_BIG at compile time) —
        # the tables grow with the input, between the rounds of the first scan; the timed scans find them complete
        for lname, lpat, lwhat in (("dft_lazy", "(a|b)*a(a|b){18}:x", "2^19 determinised states"), ("dft_lazy_runs", "((a:x)*b)|((a:y)*c)", "a state per run length")):
            configs.append(run_config(trre_amd, {"name": lname, "pattern": lpat, "engine": "dft", "steps": 3, "cpu_sample": 16 << 20,
                                                 "workload": "lazy determinisation (DFT patterns beyond the eager construction: %s): '%s', %.0f GiB"
                                                             % (lwhat, lpat, nt / 2**30)}, inp[:nt], out, tmp, want_cpu))
        # the fallback of the NFT engine (round 4): a pattern beyond every table form (98 nodes, a backward automaton of more than
        # 16 384 states, no fold; round 3: TRRE_E_UNSUPPORTED) — the reference's search on the device, a thread per KiB
        nb = min(n, 256 << 20)
        configs.append(run_config(trre_amd, {"name": "backtrack_fallback", "pattern": "a(a|b|c|d|e|f|g|h){12}c:x", "engine": "nft", "steps": 3, "full_oracle": 1 << 62,
                                             "workload": "the backtracking fallback (NFT patterns beyond every table form): 'a(a|...|h){12}c:x', %.2f GiB"
                                                         % (nb / 2**30)}, inp[:nb], out, tmp, False))
        # the reference's stack limit (round 4: the stack guard): one line of 70 000 spaces in the buffer — the scan runs, the guard's
        # probe finds the line, the reference's search on it overflows, the lines before it are scanned again: the whole error path
        if n > (64 << 20):
            p0 = n // 2
            saved = inp[p0:p0 + 70000].clone()
            inp[p0:p0 + 70000] = 32
            prog = trre_amd.Program(" +: ", "nft")
            t0 = time.perf_counter()
            err = None
            try:
                prog.scan_tensor(inp, out=out)
            except trre_amd.TrreError as e:
                err = e
            dt = time.perf_counter() - t0
            ok = err is not None and err.code == trre_amd.api.E_DIVERGES and "stack max capacity reached" in err.message
            part = int(err.partial.numel()) if ok else 0
            # what the reference had printed: the lines before that line, scanned, and the line up to the run (the run is where its search overflows)
            line0 = p0 - (1 << 20) + int((inp[p0 - (1 << 20):p0] == 10).nonzero()[-1]) + 1
            good = trre_amd.Program(" +: ", "nft").scan_tensor(inp[:line0])
            head = trre_amd.Program(" +: ", "nft").scan_tensor(torch.cat([inp[line0:p0], torch.tensor([10], dtype=torch.uint8, device=dev)]))[:-1]
            want = torch.cat([good, head])
            same = ok and part == int(want.numel()) and bool(torch.equal(err.partial, want))
            configs.append({"name": "stack_limit", "pattern": " +: ", "engine": "nft", "bytes": n, "verified": bool(same), "output_bytes": part, "ms_per_step": round(dt * 1e3, 2),
                            "workload": "' +: ' NFT with ONE run of 70 000 spaces in the %.0f GiB (trre_nft.c:548-556: the reference's search runs out of its 65 536-item stack "
                                        "there and exits 1 with what it had printed): the scan, the stack guard's probe, the reference's search on the one suspect line, the "
                                        "scan of the lines before it; TRRE_E_DIVERGES with the reference's partial output (checked here against a scan of exactly those "
                                        "lines); ms_per_step = the whole call, host-timed" % (n / 2**30)})
            inp[p0:p0 + 70000] = saved
            del saved, good, head, want
        del inp
        inp = corpora.cat_dog_soup(n, corpora.SEED0 + 4, dev)
        lines = int((inp == 10).sum())
        rec = run_config(trre_amd, {"name": "cfg4", "pattern": "(cat:dog|dog:cat)", "engine": "nft", "steps": 20, "cpu_sample": 128 << 20, "full_oracle": 1 << 62,
                                    "workload": "BASELINE configs[3]: '(cat:dog|dog:cat)' NFT scan, %.0f GiB word soup, ~10 %% cat/dog tokens "
                                                "+ near-misses, %d lines" % (n / 2**30, lines)}, inp, out, tmp, want_cpu)
        rec["lines"] = lines
        configs.append(rec)
        configs.append(run_config(trre_amd, {"name": "cfg4_guided", "pattern": "(cat:dog|dog:cat)", "engine": "nft", "steps": 20,
                                             "workload": "the same scan forced through the guided family (what an NFT pattern that does not "
                                                         "fold runs on)", "force": "guided_lp"}, inp, out, tmp, False))
        del inp
        inp = corpora.dictionary_soup(n, corpora.SEED0 + 5, dev, keys)
        # (the dictionary's keys are prefix-free, so the two engines print the same bytes — SURVEY Q9: the NFT run is checked
        # against the DFT oracle on 4 MiB slices, which the NFT oracle at 0.3 MB/s cannot cover, and against its own on a head)
        for eng, steps, sample, sl, veng in (("dft", 10, 16 << 20, 4 << 20, "dft"), ("nft", 10, 256 << 10, 4 << 20, "dft")):
            configs.append(run_config(trre_amd, {"name": "cfg5_" + eng, "pattern": dict_pat, "engine": eng, "steps": steps, "cpu_sample": sample,
                                                 "slice": sl, "verify_engine": veng, "own_head": 128 << 10 if eng == "nft" else 0,
                                                 # (the DFT oracle on this 24 kB pattern does 0.05 GB/s on all cores — every thread builds its own
                                                 # lazy tables —: the first GiB in full, the rest by the slices and the half scans)
                                                 "full_oracle": 1 << 30,
                                                 "workload": "BASELINE configs[4] shape: 1000-entry key:value dictionary (%d-byte pattern), %s engine, "
                                                             "%.0f GiB per GPU, 30 %% of the tokens are keys" % (len(dict_pat), eng.upper(), n / 2**30)},
                                      inp, out, tmp, want_cpu))
        # generator modes (row f4): the enumeration kernel behind the viability sweep, host buffer in -> host buffer out
        # (trre_scan_host: the output is unbounded in the input, the C ABI hands it over on the host)
        try:
            import random
            import corpus as tcorpus
            from oracle_lib import Oracle
            gdata = tcorpus.word_soup(random.Random(23), 32 << 20, max_len=60)
            for gname, gpat, gmode in (("gen_a", "(cat:dog|cat:cow|ca:C)", "scan_all"), ("gen_ma", "(cat:dog|cat:cow|.)*", "match_all")):
                gp = trre_amd.Program(gpat, "nft", mode=gmode)
                head = gdata[:1 << 20]
                head = head[:head.rfind(b"\n") + 1]
                o = Oracle(gpat, "nft", all_outputs=True)
                ok = gp.scan(head) == (o.match(head) if gmode == "match_all" else o.scan(head))
                t0 = time.perf_counter()
                gout = gp.scan(gdata)
                dt = time.perf_counter() - t0
                configs.append({"name": gname, "workload": "generator mode (trre %s): every accepting path prints; %d MiB of word soup, host buffer in, host buffer out"
                                                         % ("-ma" if gmode == "match_all" else "-a", len(gdata) >> 20),
                                "pattern": gpat, "engine": "nft", "kernel_family": "generate", "kernels": "k_rev_sweep (viability symbols) + k_gen<count> + k_chunk_scan + k_gen<emit>",
                                "bytes": len(gdata), "output_bytes": len(gout), "ms_per_step": round(dt * 1e3, 2), "input_GBps": round(len(gdata) / dt / 1e9, 4),
                                "verified": bool(ok), "verify": "a 1 MiB head against the oracle"})
                gp.close()
        except Exception as e:      # (the headline must not depend on it)
            configs.append({"name": "gen_ma", "verified": False, "verify": "failed: %r" % (e,)})
        line["configs"] = configs
        line["configs_verified"] = all(c.get("verified") for c in configs)

    if rank == 0:
        emit(line)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
