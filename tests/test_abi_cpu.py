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
    assert lib.psld_version() == _lib.ABI_VERSION == 14


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


def test_launching_entry_points_take_the_stream_last():
    """Every entry point that enqueues work returns a status and takes `hipStream_t stream` as its last parameter - the
    rule _lib.is_launch applies to the bound signatures agrees with the header's parameter names."""
    text = open(os.path.join(ROOT, "include", "psld_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    n = 0
    for m in re.finditer(r"\b(?:int|void|long long|const char\*|void\*)\s+(psld_[a-z0-9_]+)\s*\(([^;{]*)\)\s*;", text):
        name, params = m.group(1), m.group(2)
        takes_stream = bool(re.search(r"hipStream_t\s+stream\s*$", params.strip()))
        assert _lib.is_launch(name) == takes_stream, name
        n += takes_stream
    assert n >= 70


def test_product_switches_are_the_documented_ones():
    """the product reads a short, documented list of PSLD_* switches - every getenv of the kernel sources
    and every os.environ read of the package is in INTEGRATION.md's table, and nothing else is."""
    import glob
    found = set()
    for f in glob.glob(os.path.join(ROOT, "psld_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "psld_amd", "csrc", "*.h")):
        text = open(f).read()
        text = re.sub(r"#ifdef PSLD_ABLATIONS.*?#endif", "", text, flags=re.S)       # the ablation library's own switches
        found |= set(re.findall(r'getenv\("(PSLD_[A-Z0-9_]+)"\)', text))
    for f in glob.glob(os.path.join(ROOT, "psld_amd", "*.py")):
        found |= set(re.findall(r'environ(?:\.get)?[\[(]\s*"(PSLD_[A-Z0-9_]+)"', open(f).read()))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    listed = set(re.findall(r"^\| `(PSLD_[A-Z0-9_]+)`", doc, flags=re.M))
    assert found == listed, (sorted(found - listed), sorted(listed - found))
    kernel_or_policy = found - {"PSLD_HIP_LIB", "PSLD_DIST_TIMEOUT_S", "PSLD_PG_TIMEOUT_S", "PSLD_GRAPHS"}
    assert len(kernel_or_policy) <= 10, sorted(kernel_or_policy)
    n_getenv = sum(open(f).read().count("getenv(") for f in glob.glob(os.path.join(ROOT, "psld_amd", "csrc", "*.hip")))
    assert n_getenv <= 12, n_getenv


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


def test_hand_scheduled_groupnorm_backward_has_no_scratch():
    """gn_bwd_pipe_kernel orders its asm loads with hand-counted `s_waitcnt vmcnt(N)`; a compiler-inserted vector
    memory operation (a scratch spill or reload) would shift those counts.  The build must report zero scratch and zero
    spills for every instance of that kernel (and of the one-slab kernel with the early third operand, which sits at 238
    of 256 registers)."""
    import subprocess
    src = os.path.join(ROOT, "psld_amd", "csrc")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-I../../include", "-I.",
                        "-Rpass-analysis=kernel-resource-usage", "-c", "norm_act.hip", "-o", "/dev/null"],
                       cwd=src, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "gn_bwd_pipe_kernel" in name or "gn_bwd_fused_kernelILi16ELb1" in name or "gn_bwd_fused_kernelILi4ELb1" in name:
            seen += 1
            scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
            spills = [int(v) for v in re.findall(r"[SV]GPRs Spill: (\d+)", b)]
            assert scratch == 0 and not any(spills), (name, scratch, spills)
    assert seen >= 4
