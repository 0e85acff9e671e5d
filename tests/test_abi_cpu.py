"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/psld_hip.h declares (no compute calls: there is no GPU here)."""
import os
import re

from psld_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "psld_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psld_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"libpsld_hip.so does not export {s}"
        assert s in _lib.SIGNATURES, f"{s} declared in the header but not bound in psld_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in Python but not declared in include/psld_hip.h"
    assert lib.psld_version() == _lib.ABI_VERSION == 12


def test_error_reporting_without_gpu():
    lib = _lib.load()
    # argument validation happens on the host before any launch
    st = lib.psld_axpby_f32(None, 1.0, None, 0.0, None, 4, 0, None)
    assert st != 0 and b"psld_axpby_f32" in lib.psld_last_error()


def test_shape_rules_are_host_side():
    """The *_supported queries are pure host logic (shape rules of the kernels behind them): they answer without a GPU,
    the way the host side decides which entry point to call."""
    lib = _lib.load()
    # GroupNorm backward in one pass (and with it the column sums of dx): an (image, 32-channel) slab in registers
    assert lib.psld_gn_bwd_colsum_supported(128, 32 * 32, 256, 32) == 1
    assert lib.psld_gn_bwd_colsum_supported(16, 16 * 16, 512, 32) == 1
    assert lib.psld_gn_bwd_colsum_supported(128, 64 * 64, 128, 32) == 0        # 64 pixels per thread: the three-pass form
    assert lib.psld_gn_bwd_colsum_supported(2, 8 * 8, 6, 1) == 0               # channels per group not a multiple of 4
    assert lib.psld_gn_bwd_colsum_supported(0, 64, 256, 32) == 0
    # fused attention forward: 16x16 and 8x8 maps, 128 or 256 channels
    assert lib.psld_attn_fwd_split_supported(256, 256) == 1 and lib.psld_attn_fwd_split_supported(64, 128) == 1
    assert lib.psld_attn_fwd_split_supported(1024, 256) == 0 and lib.psld_attn_fwd_split_supported(256, 64) == 0
    # the kernel selector is a process-wide setting with validated values
    assert lib.psld_get_gn_bwd_kernel() in (0, 1)
    before = lib.psld_get_gn_bwd_kernel()
    assert lib.psld_set_gn_bwd_kernel(7) != 0 and b"psld_set_gn_bwd_kernel" in lib.psld_last_error()
    assert lib.psld_set_gn_bwd_kernel(1) == 0 and lib.psld_get_gn_bwd_kernel() == 1
    assert lib.psld_set_gn_bwd_kernel(before) == 0


def test_graft_entry_build_runs():
    """The driver's build check: make (a no-op when the objects are current) + import + symbol binding."""
    import __graft_entry__ as g
    g.build()


def test_launch_tape_stubs_are_current_and_cover_the_launching_entry_points():
    """tape_stubs.inc is generated from _lib.SIGNATURES (tools/gen_tape_stubs.py): the committed file is current; the
    rule that picks the launching entry points (status-returning, stream last) agrees with the header's parameter names;
    the built library has a stub for each of them and for nothing else."""
    import subprocess
    import sys
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_tape_stubs.py"), "--check"]).returncode == 0, \
        "psld_amd/csrc/tape_stubs.inc is stale: run python tools/gen_tape_stubs.py and rebuild"
    text = open(os.path.join(ROOT, "include", "psld_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    lib = _lib.load()
    n = 0
    for m in re.finditer(r"\b(?:int|void|long long|const char\*|void\*)\s+(psld_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", text):
        name, params = m.group(1), m.group(2)
        takes_stream = bool(re.search(r"hipStream_t\s+stream\s*$", params.strip()))
        assert _lib.is_launch(name) == takes_stream, name
        assert (lib.psld_tape_fn_index(name.encode()) >= 0) == takes_stream, name
        n += takes_stream
    assert n >= 70


def test_launch_tape_packs_arguments_and_replay_validates_entries():
    """Host side of the tape without a GPU: argument words (negative ints, floats and doubles as bit patterns,
    structures copied), and psld_tape_replay rejecting a malformed entry / forwarding a launcher's own status."""
    import ctypes as C
    import struct
    import numpy as np
    from psld_amd import tape as T
    lib = _lib.load()
    tp = T.LaunchTape()
    epi = _lib.Epilogue()
    epi.alpha = 0.5
    name = "psld_gemm_f32"
    args = (0, 1, -3, 4, 5, 1 << 40, 7, 8, None, 9, 10, 12345, 11, -12, 1, C.byref(epi), None)
    words = tp._convert(name, args)
    assert words[2] == 0xFFFFFFFD and words[5] == 1 << 40 and words[8] == 0 and words[13] == (1 << 64) - 12
    epi.alpha = 2.0                                         # the tape holds a COPY of the structure
    assert C.cast(words[15], C.POINTER(_lib.Epilogue)).contents.alpha == 0.5
    w = tp._convert("psld_axpby_f32", (1, 1.5, 2, -0.25, 3, 4, 0, None))
    assert w[1] == struct.unpack("<I", struct.pack("<f", 1.5))[0] and w[3] == struct.unpack("<I", struct.pack("<f", -0.25))[0]
    w = tp._convert("psld_ema_f32", (1, 2, 3, 0.9999, None))
    assert w[3] == struct.unpack("<Q", struct.pack("<d", 0.9999))[0]
    # replay: a launcher's argument check fires from inside the tape (no launch happens: null pointers)
    tp.add_launch(lib.psld_tape_fn_index(b"psld_axpby_f32"), "psld_axpby_f32", (None, 1.0, None, 0.0, None, 4, 0, None))
    tp.finish()
    kind, arr, ptr, n = tp.segments[0]
    assert kind == "c" and n == 1 and arr.dtype.itemsize == 8 + 8 * T.MAX_ARGS
    failed = C.c_int(-1)
    assert lib.psld_tape_replay(ptr, n, C.byref(failed)) != 0 and failed.value == 0
    assert b"psld_axpby_f32" in lib.psld_last_error()
    # a recording tape refuses launches on a stream the step never forked to (another thread's launches)
    class _S:
        cuda_stream = 0x1000
    t2 = T.LaunchTape()
    t2.stream = _S()
    idx = lib.psld_tape_fn_index(b"psld_axpby_f32")
    t2.add_launch(idx, "psld_axpby_f32", (None, 1.0, None, 0.0, None, 4, 0, C.c_void_p(0x1000)))
    t2._streams.update((0x1000, 0x2000))
    t2.add_launch(idx, "psld_axpby_f32", (None, 1.0, None, 0.0, None, 4, 0, C.c_void_p(0x2000)))
    try:
        t2.add_launch(idx, "psld_axpby_f32", (None, 1.0, None, 0.0, None, 4, 0, C.c_void_p(0x3000)))
        raise AssertionError("a launch on a foreign stream was recorded")
    except RuntimeError as e:
        assert "neither the recorded compute stream" in str(e)
    bad = np.zeros(2, dtype=T.ENTRY)
    bad["fn"][:] = (lib.psld_tape_fn_index(b"psld_axpby_f32"), 10 ** 6)
    bad["nargs"][0] = 3                                     # wrong argument count for that function
    assert lib.psld_tape_replay(bad.ctypes.data, 1, C.byref(failed)) == 1 and b"arguments" in lib.psld_last_error()
    assert lib.psld_tape_replay(bad.ctypes.data + bad.dtype.itemsize, 1, C.byref(failed)) == 1


def test_product_library_has_no_ablation_modes():
    """Timing-only ablation variants (wrong results by construction) and the kernel families that lost their A/B are
    compiled only into libpsld_hip_abl.so (-DPSLD_ABLATIONS): the product library holds neither their switches nor
    their kernels, and the benchmark refuses to run under them."""
    import subprocess
    import sys
    data = open(_lib.LIB_PATH, "rb").read()
    for needle in (b"_ABL", b"PSLD_WINO_W4", b"PSLD_WINO_NMAJOR", b"PSLD_WINO_PERSIST", b"wino_conv4_kernel", b"wino_conv8p_kernel"):
        assert needle not in data, f"{needle!r} found in {_lib.LIB_PATH}"
    assert b"wino_conv8s_kernel" in data
    env = dict(os.environ, PSLD_WINO_ABL="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "PSLD_WINO_ABL" in r.stderr
