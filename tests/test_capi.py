"""The C ABI: the shared library loads, exports every symbol include/trre_mi355x.h
declares, compiles patterns on the host and reports reference-style errors.
No compute calls (no GPU in this tier)."""
import ctypes
import os
import re

import pytest

import trre_amd
from trre_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "trre_mi355x.h")).read()
    return sorted(set(re.findall(r"\b(trre_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(api.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), s


def test_compile_reports_reference_error_text():
    for pat, msg in [("(a", "error: unmached parenthesis"), ("a)", "error: unmached parenthesis"),
                     ("[a", "error: unmached square brackets"), ("a{1,2,3}", "error: more then one comma in curly brackets"),
                     ("|a", "error: unexpected symbol |"), ("[a-c:z]", "error: unexpected range syntax")]:
        with pytest.raises(trre_amd.TrreError) as e:
            trre_amd.Program(pat, "nft")
        assert e.value.code == api.E_SYNTAX and e.value.message == msg


def test_engine_limits_are_errors_not_fallbacks():
    big = "|".join("w%03d:x" % k for k in range(40))       # 160 CONS states: too many for the bitmask kernels,
    p = trre_amd.Program(big, "nft")                        # but the scan loop folds into a stream table
    assert p.info.kernel == trre_amd.KERNEL_STREAM_GEN and trre_amd.KERNEL_TILE_GEN not in p.allowed_kernels()
    assert trre_amd.Program(big, "dft").info.table_rows == 7   # start, w, w0, w00..w03 (trie)
    # unbounded look-ahead and > 64 CONS states: round 1 refused it, the guided families run it
    p = trre_amd.Program("(" + big.replace(":x", "") + ")*z:y", "nft")
    assert p.info.kernel == trre_amd.KERNEL_GUIDED_GEN and p.info.nft_nodes == 161
    # a backward DFA of more than 256 states (which of the next ten bytes is a 'c': 4 604) and more nodes than mask bits:
    # round 2 refused it, the wide guided tables (16-bit symbols) run it
    p = trre_amd.Program("a(a|b|c|d|e|f|g|h){9}c:x", "nft")
    assert p.info.kernel == trre_amd.KERNEL_GUIDED_GEN and p.info.guided_rev_states == 4604 and trre_amd.KERNEL_GUIDED_LP not in p.allowed_kernels()
    # more than 16 384 backward states with more than 64 nodes: round 3 refused it, the backtracking fallback runs it
    p = trre_amd.Program("a(a|b|c|d|e|f|g|h){12}c:x", "nft")
    assert p.info.kernel == trre_amd.KERNEL_BACKTRACK and p.allowed_kernels() == [trre_amd.KERNEL_BACKTRACK]
    # the same pattern on the DFT engine — 99 determinised states, no fold, beyond the guided tables: rounds 1-5 ran it on the tile kernels
    # (48 GB/s), round 6 on the lazily determinised family (twice that); the tile kernels stay selectable
    p = trre_amd.Program("a(a|b|c|d|e|f|g|h){12}c:x", "dft")
    assert p.info.kernel == trre_amd.KERNEL_DFT_LAZY and trre_amd.KERNEL_TILE_GEN in p.allowed_kernels()
    p = trre_amd.Program("a(a|b|c){9}c:x", "nft")            # 29 nodes: the bitmask tile kernels could run it, the wide tables are
    assert p.info.kernel == trre_amd.KERNEL_GUIDED_GEN and p.info.guided_rev_states > 256 and trre_amd.KERNEL_TILE_GEN in p.allowed_kernels()   # 2x faster
    # an epsilon cycle is not a compile error: like the reference's lazy tables, the scan fails (TRRE_E_DIVERGES) only
    # on an input that makes it explore the cycle (tests/test_front_shim.py, golden 'eps_*' inputs)
    assert trre_amd.Program("(a*)*", "dft").info.dft_states >= 1


def test_info_and_table_export():
    p = trre_amd.Program("(cat:dog|dog:cat)", "dft")
    blob = p.export_tables()
    sblob = p.export_stream_tables()
    rblob, gblob = p.export_guided_tables()      # (round 4: the deterministic engine has guided tables too)
    assert blob[:4] == b"TRD1" and sblob[:4] == b"TRS1" and len(blob) + len(sblob) + len(rblob) + len(gblob) == p.info.table_bytes
    p = trre_amd.Program("(cat:dog|dog:cat)", "nft")
    assert p.export_tables()[:4] == b"TRN1"
    with pytest.raises(trre_amd.TrreError):
        p.set_kernel(trre_amd.KERNEL_BYTEMAP)


def test_shard_bounds_cut_at_newlines():
    data = b"".join(b"line %d\n" % k for k in range(1000))
    b = trre_amd.shard_bounds(data, 8)
    assert b[0] == 0 and b[-1] == len(data) and b == sorted(b)
    for x in b[1:-1]:
        assert data[x - 1:x] == b"\n"
    assert trre_amd.shard_bounds(b"no newline at all", 4) == [0, 17, 17, 17, 17]


def test_scan_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(trre_amd.TrreError) as e:
        trre_amd.Program("cat:dog", "nft").scan(b"cat\n")
    assert e.value.code == api.E_DEVICE


def test_no_barrier_is_left_with_lds_stores_in_flight(tmp_path):
    """round 6: hipcc emitted a loop head's s_barrier without the s_waitcnt lgkmcnt(0) for a ds_write at the loop's end, and one
    tile in 6 000 was counted twice.  tools/barrier_audit.py walks a device listing for every s_barrier that an LDS store can reach without
    that wait; here over map_kernels.hip (seconds to compile; scan_kernels.hip takes two minutes: `hipcc -S --cuda-device-only`, by hand) —
    and over a listing with the fault, which it must flag."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    audit = os.path.join(ROOT, "tools", "barrier_audit.py")
    lst = str(tmp_path / "map_kernels.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(ROOT, "trre_amd", "csrc", "map_kernels.hip"), "-o", lst],
                   check=True, stderr=subprocess.DEVNULL)
    r = subprocess.run([sys.executable, audit, lst, "60"], stdout=subprocess.PIPE, check=True)
    assert r.stdout.decode().strip().splitlines()[-1] == "barriers flagged: 0", r.stdout.decode()[-2000:]
    bad = str(tmp_path / "bad.s")
    with open(bad, "w") as f:
        f.write("_Z3badv:\n\ts_load_dword s0, s[4:5], 0x0\n.LBB0_1:\n\ts_barrier\n\tds_read_b32 v1, v0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32_e32 v1, 1, v1\n"
                "\tds_write_b32 v0, v1\n\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    r = subprocess.run([sys.executable, audit, bad], stdout=subprocess.PIPE, check=True)
    assert "BAD ds_write_b32" in r.stdout.decode() and r.stdout.decode().strip().splitlines()[-1] == "barriers flagged: 1"
