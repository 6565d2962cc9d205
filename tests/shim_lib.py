"""ctypes binding of tests/cpu_shim.cpp — TEST INFRASTRUCTURE (see that file)."""
import ctypes
import os
import struct
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpu_shim.cpp")
SO = os.path.join(HERE, "_shim", "libcpu_shim.so")
CSRC = os.path.join(os.path.dirname(HERE), "trre_amd", "csrc")

ST_NUL, ST_DIVERGE, ST_CAPACITY, ST_LONGLINE, ST_NEEDSCRATCH, ST_OVERFLOW, ST_MISMATCH = 1, 2, 4, 8, 16, 32, 1 << 30


def build():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("scan_block.hpp", "scan_core.hpp", "device_blob.hpp", "splice_block.hpp", "gen_block.hpp", "lazy_block.hpp", "guard_block.hpp", "one_block.hpp", "map_block.hpp")]
    if os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return SO
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unknown-pragmas", "-Wno-stringop-overflow", SRC, "-o", SO], check=True)
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.shim_scan.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                ctypes.POINTER(ctypes.c_uint32)]
        L.shim_scan_guided.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p,
                                       ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                       ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint32)]
        L.shim_rev_sweep.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p]
        L.shim_generate.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p,
                                    ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint32)]
        L.shim_backtrack.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint32)]
        L.shim_guard.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int),
                                 ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.shim_lazy_round.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t, ctypes.c_int]
        _lib = L
    return _lib


def mask_bytes_of(blob, engine):
    if engine == 1:
        return 0
    n_cons = struct.unpack_from("<I", blob, 4)[0]
    return 1 if n_cons <= 8 else 2 if n_cons <= 16 else 4 if n_cons <= 32 else 8


def shim_scan(blob, engine, family, data, geo=1, in_mis=0, out_mis=0, scratch=True, cap=None):
    """Run one kernel family of the device code on the host.  Returns (output bytes, status)."""
    auto_cap = cap is None
    if cap is None:
        cap = len(data) * 8 + 64 if family in (3, 7, 9, 22, 23, 27, 29, 32, 33, 34, 35, 36, 37, 38) else len(data)     # (20, 21: length-preserving)
    for _ in range(2):
        out = ctypes.create_string_buffer(max(cap, 1))
        m = ctypes.c_size_t()
        st = ctypes.c_uint32()
        rc = lib().shim_scan(blob, engine, mask_bytes_of(blob, engine), family, geo, data, len(data), in_mis, out, cap,
                             out_mis, 1 if scratch else 0, ctypes.byref(m), ctypes.byref(st))
        if rc == -5:
            return None, 0                      # this table has no window form
        if rc:
            raise RuntimeError("shim rc %d" % rc)
        if auto_cap and st.value & ST_CAPACITY and m.value > cap:
            cap = m.value + 64                  # (a pattern that prints more than 8 bytes per input byte: the size the count pass asked for,
            continue                            #  as a caller of the C ABI does on TRRE_E_CAPACITY)
        break
    return out.raw[:m.value], st.value


def has_g16(stream_blob):
    """a stream-form blob carries the 16-byte entries (StreamBlobHeader::g16_bytes)"""
    return bool(stream_blob) and len(stream_blob) >= 144 and _g16_bytes(stream_blob) != 0


def _g16_bytes(blob):
    return struct.unpack_from("<I", blob, 60)[0]          # (device_blob.hpp: the 16th word of StreamBlobHeader)


def has_mapgen(prog):
    """the stream tables carry the memoryless form (StreamBlobHeader::mg_max: the longest text; 0: none)"""
    blob = prog.export_stream_tables()
    return bool(blob) and len(blob) >= 192 and struct.unpack_from("<48I", blob, 0)[46] != 0


def has_fallback_form(prog):
    """The stream tables carry the fallback form of a large table (StreamBlobHeader::fb_states)."""
    blob = prog.export_stream_tables()
    return bool(blob) and len(blob) >= 144 and struct.unpack_from("<36I", blob, 0)[20] != 0


def has_mark_form(prog):
    """... and the 32-bit mark form of the comb (StreamBlobHeader::fb4_slots)"""
    blob = prog.export_stream_tables()
    return has_copy_form(prog) and len(blob) >= 192 and struct.unpack_from("<48I", blob, 0)[36] != 0


def has_copy_form(prog):
    """... and the copy form on top of it (StreamBlobHeader::off_fb_lit_meta): every replacement text is non-empty and
    every cell of the fallback form is a copy, an owed text or an escape that says which input bytes it stands for."""
    blob = prog.export_stream_tables()
    return has_fallback_form(prog) and struct.unpack_from("<36I", blob, 0)[34] != 0


ST_EDIT_OVERFLOW = 64


# shim ids of the guided families (ABI ids 6, 7): LP by the emit pass alone on the 16-byte entries (as the runtime
# launches it), general on the 16-byte / 8-byte entries, LP by the older LDS-ring walker, LP emit on the 8-byte entries
GUIDED_LP, GUIDED_GEN, GUIDED_GEN8, GUIDED_LP_RING, GUIDED_LP8 = 10, 11, 12, 13, 14
GUIDED_GEN_EXACT, GUIDED_GEN_EXACT_MISS = 17, 18     # general guided family with exact sub-ranges (round 5); 18: a look-back of 4 bytes (repair rounds)
STREAM_G16_EXACT, STREAM_G16_EXACT_MISS = 32, 33     # stream general family, the same
# ONE walk (round 6, one_block.hpp): the production geometry (geo 0: 256 lanes of 128 bytes; geo 1: 5 lanes of 64); _MISS: look-backs of 4 bytes and tiles
# of 3 lanes (wrong guesses: repair rounds inside a tile, void launches across tiles); _TIGHT: regions of 76 bytes for 64 of input (outgrown regions)
STREAM_ONE, STREAM_ONE_MISS, STREAM_ONE_TIGHT = 34, 35, 36
GUIDED_ONE, GUIDED_ONE_MISS = 40, 41
# memoryless programs of any output length in one pass (round 6, map_block.hpp): the production geometry (geo 0: 8 waves of 64 lanes, a window of
# 40 KiB; geo 1: 2 waves of 3 lanes, 256 bytes); _TINY: tiles of one lane (64 bytes), windows of 48 bytes (several windows per tile)
STREAM_MAPGEN, STREAM_MAPGEN_TINY = 37, 38
ST_ONE_VOID = 256
one_stats = {"runs": 0, "void": 0}                  # how often the one-pass form answered / left the buffer to the pair
GUIDED_LP_ALL = (GUIDED_LP, GUIDED_LP_RING, GUIDED_LP8)
STREAM_LP_EMIT, STREAM_LP_EMIT8 = 20, 21            # stream LP family by the emit pass alone (no window form)
STREAM_LPW_PAIR = 26                                 # the window kernel on the pair form of its entries (what the runtime launches when the tables have one)
STREAM_FB_SPLICE8 = 29                               # ... with the mark pass on the 8-byte comb (tables without the mark form; round 3's first pass)
STREAM_FB_SPLICE = 27                                # ... the second pass as the wave-cooperative splice (what the runtime launches by default)
STREAM_FB_SPLICE8 = 29                               # ... with the mark pass on the 8-byte comb (tables without the mark form; round 3's first pass)
STREAM_FB_SPLICE = 27                                # ... the second pass as the wave-cooperative splice (what the runtime launches by default)
STREAM_FB, STREAM_FB_COUNT = 22, 23                  # stream general family on the fallback form of a large table: both passes /
                                                     # the count pass only, emit on the 8-byte rows (what the runtime launches)


def shim_scan_guided(prog, family, data, geo=1, in_mis=0, out_mis=0):
    rblob, gblob = prog.export_guided_tables()
    assert rblob and gblob
    cap = len(data) * 8 + 64 if family not in GUIDED_LP_ALL else len(data)
    for _ in range(2):
        out = ctypes.create_string_buffer(max(cap, 1))
        m = ctypes.c_size_t()
        st = ctypes.c_uint32()
        rc = lib().shim_scan_guided(rblob, gblob, family, geo, data, len(data), in_mis, out, cap, out_mis, ctypes.byref(m),
                                    ctypes.byref(st))
        if rc:
            raise RuntimeError("shim rc %d" % rc)
        if st.value & ST_CAPACITY and m.value > cap:
            cap = m.value + 64                  # (more than 8 bytes of output per input byte)
            continue
        break
    return out.raw[:m.value], st.value


def last_rounds():
    """repair rounds of the last exact-sub-range run on the shim"""
    return lib().shim_last_rounds()


def scan_guided_like_runtime(prog, data, geo=1, family=GUIDED_LP, in_mis=0, out_mis=0):
    if family in (GUIDED_ONE, GUIDED_ONE_MISS):
        # like finish(): a void one-pass launch leaves the buffer to the count / emit pair with exact sub-ranges
        try:
            out, st = shim_scan_guided(prog, family, data, geo, in_mis, out_mis)
        except RuntimeError as e:
            if "rc -5" not in str(e):
                raise
            out, st = None, ST_ONE_VOID
        one_stats["runs"] += 1
        assert not st & (1 << 29), "the backward pass's repair rounds did not converge"
        if not st & (ST_ONE_VOID | ST_DIVERGE | ST_OVERFLOW):
            return out
        one_stats["void"] += 1
        family = GUIDED_GEN_EXACT if family == GUIDED_ONE else GUIDED_GEN_EXACT_MISS
    if family in (GUIDED_GEN_EXACT, GUIDED_GEN_EXACT_MISS):
        lib()
        rblob, gblob = prog.export_guided_tables()
        cap = len(data) * 8 + 64
        o = ctypes.create_string_buffer(max(cap, 1))
        m = ctypes.c_size_t()
        stt = ctypes.c_uint32()
        rc = lib().shim_scan_guided(rblob, gblob, family, geo, data, len(data), in_mis, o, cap, out_mis, ctypes.byref(m), ctypes.byref(stt))
        if rc == -5:
            family = GUIDED_GEN
        elif rc:
            raise RuntimeError("shim rc %d" % rc)
        elif stt.value & (ST_DIVERGE | ST_OVERFLOW):
            family = GUIDED_GEN              # (an attempt that does not return: the old way names the first such lane)
        else:
            assert not stt.value & (1 << 29), "the repair rounds did not converge"
            return o.raw[:m.value]
    out, st = shim_scan_guided(prog, family, data, geo, in_mis, out_mis)
    assert not st & ST_MISMATCH, "count and emit passes disagree"
    if st & ST_DIVERGE and not (family in GUIDED_LP_ALL and st & ST_NUL):      # (void by a NUL: the general family decides)
        raise RuntimeError("diverges")
    if family in GUIDED_LP_ALL and st & ST_NUL:
        out, st = shim_scan_guided(prog, GUIDED_GEN, data, geo, in_mis, out_mis)
        assert not st & ST_MISMATCH
        if st & ST_DIVERGE:
            raise RuntimeError("diverges")
    return out


def scan_like_runtime(prog, data, geo=1, family=None, in_mis=0, out_mis=0):
    """Mirror of runtime.cpp's policy (finish()): auto family, NUL -> general family."""
    info = prog.info
    fam = family
    if not fam:                       # ABI ids -> shim ids (the shim's 6..9 are the direct walkers of the stream families)
        # (ABI 4 / 5, the stream families: the window kernel — the emit pass alone where the tables have no window form — and the direct
        # count / emit pair, as runtime.cpp launches them; the shim's own 4 / 5, the LDS-tile walkers, went in round 6)
        fam = {4: 8, 5: 7, 6: GUIDED_LP, 7: GUIDED_GEN}.get(info.kernel, info.kernel)
    if fam in (GUIDED_LP, GUIDED_GEN, GUIDED_GEN8, GUIDED_LP_RING, GUIDED_LP8, GUIDED_GEN_EXACT, GUIDED_GEN_EXACT_MISS, GUIDED_ONE, GUIDED_ONE_MISS):
        return scan_guided_like_runtime(prog, data, geo, fam, in_mis, out_mis)
    if family == DFT_LAZY or (not family and info.kernel == 10):
        out, st, _ = scan_lazy(prog, data, geo, in_mis)
        if st & ST_EDIT_OVERFLOW:
            raise RuntimeError("limits")
        if st & ST_DIVERGE:
            raise RuntimeError("diverges")
        return out
    if family == BACKTRACK or (not family and info.kernel == 9):       # (ABI family 9; the shim's own 9 is a direct walker of the stream family)
        out, st = scan_backtrack(prog, data, geo, in_mis)
        if st & ST_EDIT_OVERFLOW:
            raise RuntimeError("limits")
        if st & ST_DIVERGE:
            raise RuntimeError("diverges")
        return out
    blob = prog.export_stream_tables() if fam in (6, 7, 8, 9, 20, 21, 22, 23, 27, 29, 32, 33, 34, 35, 36, 37, 38, STREAM_LPW_PAIR) else prog.export_tables()
    if fam in (STREAM_MAPGEN, STREAM_MAPGEN_TINY):
        out, st = shim_scan(blob, info.engine, fam, data, geo, in_mis, out_mis)
        assert out is not None, "the tables have no memoryless form"
        assert not st & ST_MISMATCH, "count and expand disagree"
        if not st & ST_NUL:
            return out
        fam = 7                             # like finish(): a NUL voids the launch, the general small-table family takes the buffer
    if fam in (STREAM_ONE, STREAM_ONE_MISS, STREAM_ONE_TIGHT):
        out, st = shim_scan(blob, info.engine, fam, data, geo, in_mis, out_mis)
        one_stats["runs"] += 1
        if out is not None and not st & (ST_ONE_VOID | ST_DIVERGE | ST_OVERFLOW):
            return out
        one_stats["void"] += 1
        fam = STREAM_G16_EXACT_MISS if fam == STREAM_ONE_MISS else STREAM_G16_EXACT      # like finish(): the pair takes the buffer
    if fam in (STREAM_G16_EXACT, STREAM_G16_EXACT_MISS):
        out, st = shim_scan(blob, info.engine, fam, data, geo, in_mis, out_mis)
        if out is not None and not st & (ST_DIVERGE | ST_OVERFLOW):
            assert not st & (1 << 29), "the repair rounds did not converge"
            return out
        fam = 7                             # no 16-byte entries, a bounded fold that overflowed, an attempt that does not return: the old way
    out, st = shim_scan(blob, info.engine, fam, data, geo, in_mis, out_mis)
    if out is None:                         # family 8 / 26 without a window (pair) form: the emit pass alone, like the runtime
        fam = STREAM_LP_EMIT
        out, st = shim_scan(blob, info.engine, fam, data, geo, in_mis, out_mis)
    if fam in (7, 9) and st & ST_OVERFLOW:           # bounded stream table: the guided (or the tile) kernels take over
        if info.guided_rev_states:
            return scan_guided_like_runtime(prog, data, geo, GUIDED_GEN, in_mis, out_mis)
        fam = 3
        blob = prog.export_tables()
        out, st = shim_scan(blob, info.engine, fam, data, geo, in_mis, out_mis)
    assert not st & ST_MISMATCH, "count and emit passes disagree"
    lp = fam not in (3, 7, 9, 22, 23)
    if st & ST_DIVERGE and not (lp and st & ST_NUL):      # (a positional launch that met a NUL is void, whatever else it says)
        raise RuntimeError("diverges")
    if lp and st & ST_NUL:
        gen = 7 if info.stream_states else 3
        blob = prog.export_stream_tables() if gen == 7 else prog.export_tables()
        out, st = shim_scan(blob, info.engine, gen, data, geo, in_mis, out_mis)
        assert not st & ST_MISMATCH
        if st & ST_DIVERGE:
            raise RuntimeError("diverges")
    return out


def stack_guard(prog, data, in_mis=0, budget=1 << 27):
    """the stack guard's bodies on the host, as the runtime drives them: None when the pattern has no guard; else (hit, line_start,
    the part of that line's output the reference had printed) — hit 0: no line overflows, 1: one does, 2: a line was not decided"""
    kblob = prog.export_guard_tables()
    if not kblob:
        return None
    cap = 16 * len(data) + 65536
    out = ctypes.create_string_buffer(cap)
    hit, ls, part = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_size_t()
    rc = lib().shim_guard(kblob, data, len(data), in_mis, budget, ctypes.byref(hit), ctypes.byref(ls), out, cap, ctypes.byref(part))
    if rc:
        raise RuntimeError("shim rc %d" % rc)
    return hit.value, ls.value, out.raw[:part.value]


BACKTRACK = 30                                       # the backtracking fallback (ABI family 9)


def scan_backtrack(prog, data, geo=1, in_mis=0, frames=4096, path_cap=4096, budget=16 << 20):
    """the backtracking fallback's lane body on the host, as the runtime drives it: (output, status) — with ST_DIVERGE the output is
    what the reference had printed when it gave up"""
    nblob = prog.export_gen_tables()
    assert nblob
    cap = max(1 << 16, 4 * len(data))
    for _ in range(2):
        out = ctypes.create_string_buffer(cap)
        m = ctypes.c_size_t()
        st = ctypes.c_uint32()
        rc = lib().shim_backtrack(nblob, geo, data, len(data), in_mis, out, cap, frames, path_cap, budget, ctypes.byref(m), ctypes.byref(st))
        if rc:
            raise RuntimeError("shim rc %d" % rc)
        assert not st.value & ST_MISMATCH, "count and emit passes disagree"
        if st.value & ST_CAPACITY:
            cap = m.value + 64
            continue
        return out.raw[:m.value], st.value
    raise RuntimeError("capacity")


DFT_LAZY = 31                                        # the deterministic engine on tables still being built (ABI family 10)
ST_MISS = 128


def scan_lazy(prog, data, geo=1, in_mis=0, miss_cap=1 << 16, spec=64, budget=1 << 32, max_rounds=100000, foreign_marks=False):
    """the lazy family as the runtime runs it: rounds of (count pass for the lanes without a result; the library explores the edges
    the round listed) until a round lists none, then the emit pass.  Returns (output, status, rounds); ST_DIVERGE: the reference dies here"""
    import numpy as np
    lane_bytes = 1024 if geo == 0 else 64
    n_lanes = ((in_mis + 15) + len(data) + lane_bytes - 1) // lane_bytes + 2
    lane_counts = np.full(n_lanes, 0xffffffff, dtype=np.uint32)
    miss = np.zeros(2 + 16 * miss_cap, dtype=np.uint32)
    cap = max(1 << 16, 4 * len(data))
    rounds = 0
    while True:
        head, ent, pool = prog.lazy_tables()
        n_cls, cls = struct.unpack_from("<I", head, 0)[0], head[8:8 + 256]
        entb = np.frombuffer(bytearray(ent), dtype=np.uint64).copy()
        out = ctypes.create_string_buffer(cap)
        m = ctypes.c_size_t()
        st = ctypes.c_uint32()
        rc = lib().shim_lazy_round(cls, entb.ctypes.data, pool if pool else b"\0", n_cls, geo, data, len(data), in_mis, lane_counts.ctypes.data,
                                   miss.ctypes.data, miss_cap, budget, out, cap, ctypes.byref(m), ctypes.byref(st), entb.size, int(foreign_marks))
        if rc:
            raise RuntimeError("shim rc %d" % rc)
        rounds += 1
        assert not st.value & ST_MISMATCH, "count and emit passes disagree"
        if st.value & (ST_DIVERGE | ST_EDIT_OVERFLOW):
            return b"", st.value, rounds
        if st.value & ST_MISS:
            n = min(int(miss[0]), miss_cap)
            assert n > 0, "a void round listed no miss"
            assert rounds < max_rounds
            prog.lazy_explore(miss[2:2 + 16 * n], spec)
            continue
        if st.value & ST_CAPACITY:
            cap = m.value + 64
            continue
        return out.raw[:m.value], st.value, rounds


def rev_symbols(rblob, data, geo=1, in_mis=0):
    """the backward pass on the host: one symbol per byte of `data`"""
    out = ctypes.create_string_buffer(max(len(data), 1))
    rc = lib().shim_rev_sweep(rblob, geo, data, len(data), in_mis, out)
    if rc:
        raise RuntimeError("shim rc %d" % rc)
    return out.raw[:len(data)]


def generate_like_runtime(prog, data, geo=1, in_mis=0):
    """generator modes as the runtime runs them: viability symbols from the backward kernel's body (here on the host), then
    the library's own enumeration (generate.cpp)"""
    rblob, _ = prog.export_guided_tables()
    assert rblob
    return prog.generate_with_symbols(data, rev_symbols(rblob, data, geo, in_mis))


def generate_on_device_like_runtime(prog, data, geo=1, in_mis=0, frames=512, path_cap=2048):
    """generator modes as the runtime runs them since round 4: backward sweep, then the enumeration kernel's lane body (count,
    exclusive sum, emit) on the host; an input the kernel hands back (a path that never returns, a search deeper than a lane's
    stack) goes to the library's host enumeration, as in runtime.cpp.  Returns (output, went_to_the_host)."""
    rblob, _ = prog.export_guided_tables()
    nblob = prog.export_gen_tables()
    assert rblob and nblob
    cap = 1 << 16
    for _ in range(2):
        out = ctypes.create_string_buffer(cap)
        m = ctypes.c_size_t()
        st = ctypes.c_uint32()
        rc = lib().shim_generate(rblob, nblob, geo, data, len(data), in_mis, out, cap, frames, path_cap, ctypes.byref(m), ctypes.byref(st))
        if rc:
            raise RuntimeError("shim rc %d" % rc)
        assert not st.value & ST_MISMATCH, "count and emit passes disagree"
        if st.value & (ST_DIVERGE | ST_EDIT_OVERFLOW):
            return generate_like_runtime(prog, data, geo, in_mis), True
        if st.value & ST_CAPACITY:
            cap = m.value + 64
            continue
        return out.raw[:m.value], False
    raise RuntimeError("capacity")
