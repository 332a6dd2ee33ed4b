"""The C-ABI library builds for gfx950 without a GPU, loads, and exports exactly the entry points
that include/ds_kernels.h declares (no compute calls here: there is no device in this container)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ds_kernels.h")


def _declared(tuning=False):
    """Entry points the header declares: the contract (default), or the ones inside its `#ifdef DS_TUNING` blocks."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    blocks = re.findall(r"#ifdef DS_TUNING(.*?)#endif", src, flags=re.S)
    src = " ".join(blocks) if tuning else re.sub(r"#ifdef DS_TUNING.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from tumblr_emotions_amd import _lib
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(_lib.TUNING_LIB_PATH)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tumblr_emotions_amd", "csrc"), "-j4"], check=True)
    return _lib


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 25
    dll = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), "libds_kernels.so lacks %s" % n
    assert sorted(lib.SIGNATURES) == names, "ctypes table and header disagree"
    assert lib.load().ds_version() >= 1


def test_shipped_library_exports_only_the_contract_and_reads_no_environment(lib):
    """SURVEY 8(b) / VERDICT r05 weak #11: the tuning rig is a SECOND library.  libds_kernels.so exports exactly the header's
    contract -- no ds_debug_* setter -- and does not import getenv (every DS_* knob of the selection rules takes its default);
    libds_kernels_tuning.so (-DDS_TUNING, what scripts/ and the tile-pinning kernel tests load) adds both."""
    def syms(path, flag):
        out = subprocess.run(["nm", "-D", flag, path], capture_output=True, text=True, check=True).stdout
        return {l.split()[-1].split("@")[0] for l in out.splitlines() if l.strip()}
    debug = _declared(tuning=True)
    assert len(debug) >= 8 and all(n.startswith("ds_debug_") for n in debug) and sorted(lib.DEBUG_SIGNATURES) == debug
    exported = {n for n in syms(lib.LIB_PATH, "--defined-only") if n.startswith("ds_")}
    assert exported == set(_declared()), sorted(exported ^ set(_declared()))
    assert "getenv" not in syms(lib.LIB_PATH, "--undefined-only")
    t_exported = {n for n in syms(lib.TUNING_LIB_PATH, "--defined-only") if n.startswith("ds_")}
    assert t_exported == set(_declared()) | set(debug), sorted(t_exported ^ (set(_declared()) | set(debug)))
    assert "getenv" in syms(lib.TUNING_LIB_PATH, "--undefined-only")
    t = lib.load_tuning()
    assert t.ds_debug_conv_set_tile(3, 1) == -1 and t.ds_debug_conv_set_tile(0, 0) == 0
    assert not hasattr(ctypes.CDLL(lib.LIB_PATH), "ds_debug_conv_set_tile")


def test_python_engine_switches_are_inert_beside_the_product_library(lib, monkeypatch):
    """The Python engine's A/B switches (DS_ZCAT, DS_FUSE_B3, DS_STEM_POOL, ...) go through _lib.tuning_env: honoured only when
    DS_LIB selects a tuning build, like the C library's knobs; no other DS_* variable is read under tumblr_emotions_amd/."""
    monkeypatch.delenv("DS_LIB", raising=False)
    monkeypatch.setenv("DS_ZCAT", "0")
    assert lib.tuning_env("DS_ZCAT", "1") == "1"
    monkeypatch.setenv("DS_LIB", lib.TUNING_LIB_PATH)
    assert lib.tuning_env("DS_ZCAT", "1") == "0"
    pkg = os.path.join(ROOT, "tumblr_emotions_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                for m in re.finditer(r'os\.environ\.get\("(DS_[A-Z0-9_]+)"', text):
                    assert m.group(1) == "DS_LIB", (os.path.join(dirpath, f), m.group(1))


def test_errors_are_reported_not_thrown(lib):
    l = lib.load()
    assert l.ds_gather_rows(None, None, None, 1, 1, 1, 1, 1, None) == -1          # DS_ERR_ARG
    assert b"ds_gather_rows" in l.ds_last_error()


def test_product_path_has_no_cpu_fallback(lib):
    import torch
    from tumblr_emotions_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.fill(torch.zeros(4), 4, 1.0)
    if not torch.cuda.is_available():
        from tumblr_emotions_amd.net import SentimentNet
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            SentimentNet(mode="text")


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tumblr_emotions_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") or f.endswith(".hip") or f.endswith(".h") or f.endswith(".cpp"):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(dirpath, f)


def test_product_path_never_calls_the_debug_switches():
    """the library keeps no mutable process-global state on the product path.  The ds_debug_* entry
    points (tile / kernel-family pins, LSTM phase stamps) are A-B aids for tests/ and scripts/ only."""
    pkg = os.path.join(ROOT, "tumblr_emotions_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py") and f != "_lib.py":
                text = open(os.path.join(dirpath, f)).read()
                assert "ds_debug_" not in text, os.path.join(dirpath, f)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "ds_debug_" not in bench


def test_lstm_seq_entry_points_take_rows_and_keep_no_global(lib):
    """`rows` is an argument of ds_lstm_seq_fwd / _bwd (round 2 kept it in a process global set before each launch)."""
    assert "ds_lstm_seq_set_rows" not in lib.SIGNATURES
    dll = ctypes.CDLL(lib.LIB_PATH)
    assert not hasattr(dll, "ds_lstm_seq_set_rows")
    l = lib.load()
    # workspace: forward + backward counters per 32-row group and two error words, 16-byte granules
    assert l.ds_lstm_seq_workspace(256, 512) == (2 * 16 * 65 + 2 + 63) // 64 * 256 + 2 * 256 * (2048 + 512) * 4      # control words (two blocks of 16 counters + 16 x 64 placement words, two error words) + the backward and the forward exchange ring
    assert l.ds_lstm_seq_supported(256, 512) == 1 and l.ds_lstm_seq_supported(256, 48) == 0
    # argument errors are reported before anything is launched (no device needed)
    assert l.ds_lstm_seq_fwd(None, None, 0, None, None, None, 1, 1, 32, 1.0, 1, None, 0, None) == -1


def test_no_undefined_names_in_python_sources():
    """scripts/lint_names.py over the package, the tests and bench.py (a NameError that only shows on the GPU box costs
    minutes of its budget: no pyflakes is installed here)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lint_names", os.path.join(ROOT, "scripts", "lint_names.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    paths = [os.path.join(ROOT, p) for p in ("tumblr_emotions_amd", "tests", "bench.py", "__graft_entry__.py")]
    files = []
    for a in paths:
        if os.path.isdir(a):
            for d, _, fs in os.walk(a):
                files += [os.path.join(d, f) for f in fs if f.endswith(".py")]
        else:
            files.append(a)
    bad = [b for f in sorted(files) for b in m.check(f)]
    assert not bad, bad


def test_winograd_f4_launch_time_model_answers_without_a_device(lib):
    """ds_conv_wino4_supported / _prefer / _partials are pure host functions (the launch-time model that picks
    F(4x4,3x3), with one or two channel blocks, or F(2x2,3x3) per shape): the shapes of the joint step at B = 256."""
    L = lib.load()
    assert L.ds_conv_wino4_supported(56, 56, 64, 192) == 1
    assert L.ds_conv_wino4_supported(14, 14, 24, 64) == 0            # Cin % 16 != 0: ds_conv_wino / implicit GEMM
    assert L.ds_conv_wino4_supported(14, 14, 64, 30) == 0            # Cout % 4 != 0 (16-byte output stores)
    assert L.ds_conv_wino4_supported(7, 13, 16, 16) == 1             # partial border tiles are fine
    for hw, ci, co in ((56, 64, 192), (56, 192, 64), (28, 96, 128), (28, 128, 192), (28, 16, 32), (14, 96, 208), (7, 160, 320)):
        assert L.ds_conv_wino4_prefer(256, hw, hw, ci, co) > 0, (hw, ci, co)
    assert L.ds_conv_wino4_prefer(256, 14, 14, 24, 64) == 0          # unsupported -> never preferred
    # (through round 4 two rounds of long F(2x2) workgroups beat three F(4x4) ones on this dgrad; with the round-5 kernel
    # F(4x4) wins it too -- 221.6 against 231.2 us, profiles/r05_notes.md -- and the refreshed launch model says so)
    assert L.ds_conv_wino4_prefer(256, 14, 14, 320, 160) > 0
    assert L.ds_conv_wino4_prefer(2, 14, 14, 16, 16) >= 0             # (tiny launches: either answer is a valid kernel)
    assert L.ds_conv_wino4_partials(256, 56, 56) == 256 * 14 * 14 // 32
    assert L.ds_conv_wino4_partials(2, 7, 7) == 1                    # 2 x 2 x 2 tiles -> one group of 32


def test_pool_gradient_sums_launch_shape_answers_without_a_device(lib):
    """ds_maxpool3_bwd_sums_partials is a pure host function: the number of unit blocks (= BatchNorm-sum partials per
    channel) of the chunked launch -- a workgroup owns 16..64 channel quads x 256 / quads unit groups, one image column per
    group, at most 2048 blocks.  The Branch_3 pools of the joint step at B = 256 and the degenerate widths."""
    L = lib.load()
    want = {(256, 28, 256): 1792,      # 64 quads x 4 groups: 7168 columns / 4
            (256, 14, 480): 896,       # 60 x 4
            (256, 14, 512): 896,       # 64 x 4
            (256, 14, 528): 326,       # 22 x 11 (132 quads fill no workgroup; 44 x 5 and 33 x 7 use fewer threads)
            (256, 7, 832): 112,        # 16 x 16
            (2, 5, 12): 1,             # 3 quads x 85 groups: ten columns, one block
            (3, 4, 68): 1}             # 17 quads (no divisor in 16..64): one chunk x 15 groups
    for (n, w, c), p in want.items():
        assert L.ds_maxpool3_bwd_sums_partials(n, w, c) == p, (n, w, c)
    for n, w, c in ((1, 1, 4), (4096, 28, 192), (100000, 7, 1024), (7, 3, 1020)):
        p = L.ds_maxpool3_bwd_sums_partials(n, w, c)
        assert 1 <= p <= 2048 and p <= n * w, (n, w, c, p)
    assert L.ds_maxpool3_bwd_sums_partials(0, 7, 64) == 0 and L.ds_maxpool3_bwd_sums_partials(4, 7, 0) == 0


def test_conv_plan_picks_the_kernel_family_without_a_device(lib):
    """SURVEY 8(b) / VERDICT r03 weak #11: kernel-family selection lives BEHIND the ABI.  ds_conv_plan is host-only, so
    its choices for the tower's shapes can be pinned here: Winograd F(4x4) / F(2x2) / implicit GEMM for the 3x3 layers,
    the stem kernel, the register-direct bf16 / fp8 / f32x3 kernels, the BatchNorm-sums epilogue only where a kernel
    carries it, and errors as return codes."""
    import ctypes as C
    L = lib
    l = lib.load()

    def plan(role, arith, opts, N, H, W, ci, co, k, s, flags=0, ldx=None, ldz=None):
        p = L.LayerPlanStruct()
        rc = l.ds_conv_plan(C.byref(p), role, arith, opts, N, H, W, ci, co, k, s, ci if ldx is None else ldx,
                            co if ldz is None else ldz, flags)
        return rc, p

    B = 256
    rc, p = plan(L.DS_CONV_FWD, L.DS_ARITH_F32, 0, B, 28, 28, 96, 128, 3, 1, L.DS_EPI_STATS)
    assert rc == 0 and p.family == L.DS_FAM_WINO4 and p.w_bytes == 4 * 36 * 96 * 128 and p.partials > 0
    assert p.alg_flops == 2.0 * B * 28 * 28 * 128 * 9 * 96
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_F32, L.DS_PLAN_NO_WINO4, B, 28, 28, 96, 128, 3, 1)[1].family == L.DS_FAM_WINO2
    rc, p = plan(L.DS_CONV_FWD, L.DS_ARITH_F32, L.DS_PLAN_NO_WINO, B, 28, 28, 96, 128, 3, 1)
    assert p.family == L.DS_FAM_IGEMM and p.w_bytes == 0 and p.d.w_k_stride == 128 and p.d.flip == 0
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_F32, 0, B, 7, 7, 48, 64, 3, 1)[1].family == L.DS_FAM_IGEMM   # 7x7 map, narrow
    # dgrad: channel roles swapped, flipped taps; the 1x1 dgrad can carry the BatchNorm-sums epilogue, a bf16 one cannot
    rc, p = plan(L.DS_CONV_DGRAD, L.DS_ARITH_F32, 0, B, 28, 28, 192, 176, 1, 1, 0, ldx=176, ldz=192)
    assert rc == 0 and p.family == L.DS_FAM_IGEMM and (p.d.Cin, p.d.Cout, p.d.flip) == (176, 192, 1) and p.partials == 0
    P = l.ds_conv_plan_enable_bnsums(C.byref(p), 192)
    assert P > 0 and p.partials == P and p.d.flags & L.DS_EPI_BNSUMS and p.d.ldmask == 192
    rc, p = plan(L.DS_CONV_DGRAD, L.DS_ARITH_F32, 0, B, 28, 28, 96, 128, 3, 1, 0, ldx=128, ldz=96)
    assert p.family in (L.DS_FAM_WINO2, L.DS_FAM_WINO4) and l.ds_conv_plan_enable_bnsums(C.byref(p), 96) > 0
    rc, p = plan(L.DS_CONV_DGRAD, L.DS_ARITH_BF16, 0, B, 28, 28, 192, 176, 1, 1, 0, ldx=176, ldz=192)
    assert p.family == L.DS_FAM_BF16D and l.ds_conv_plan_enable_bnsums(C.byref(p), 192) > 0 and p.w_bytes > 0
    # the 16-bit configurations' 3x3 input gradients: F(4x4) of the bf16-rounded operands on the bf16 matrix cores, with the
    # BatchNorm-sums epilogue; the two 14 x 14 layers with >= 288 reduction channels stay on the register-direct kernel;
    # DS_PLAN_NO_WINO4H restores the round-4 choice (LDS-staged bf16 kernel below 160 columns: no sums epilogue)
    rc, p = plan(L.DS_CONV_DGRAD, L.DS_ARITH_BF16, 0, B, 28, 28, 96, 128, 3, 1, 0, ldx=128, ldz=96)
    assert p.family == L.DS_FAM_WINO4H and p.w_bytes == 4 * 36 * 96 * 128 and l.ds_conv_plan_enable_bnsums(C.byref(p), 96) > 0
    assert plan(L.DS_CONV_DGRAD, L.DS_ARITH_FP8, 0, B, 56, 56, 64, 192, 3, 1, 0, ldx=192, ldz=64)[1].family == L.DS_FAM_WINO4H
    assert plan(L.DS_CONV_DGRAD, L.DS_ARITH_BF16, 0, B, 14, 14, 144, 288, 3, 1, 0, ldx=288, ldz=144)[1].family == L.DS_FAM_BF16D
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_BF16, 0, B, 28, 28, 96, 128, 3, 1)[1].family == L.DS_FAM_BF16D      # forward: unchanged
    rc, p = plan(L.DS_CONV_DGRAD, L.DS_ARITH_BF16, L.DS_PLAN_NO_WINO4H, B, 28, 28, 96, 128, 3, 1, 0, ldx=128, ldz=96)      # LDS-staged bf16 kernel
    assert p.family == L.DS_FAM_IGEMM and l.ds_conv_plan_enable_bnsums(C.byref(p), 96) == 0
    # stem: the packed-RGB kernel, or the generic kernel with KW folded into the channel axis
    rc, p = plan(L.DS_CONV_FWD, L.DS_ARITH_F32, L.DS_PLAN_PACKED_RGB, B, 224, 224, 4, 64, 7, 2, L.DS_EPI_STATS)
    assert rc == 0 and p.family == L.DS_FAM_STEM and (p.d.OH, p.d.pad_t) == (112, 2) and p.partials > 0
    assert p.alg_flops == 2.0 * B * 112 * 112 * 64 * 147
    rc, p = plan(L.DS_CONV_FWD, L.DS_ARITH_F32, L.DS_PLAN_PACKED_RGB | L.DS_PLAN_NO_STEM_DIRECT, B, 224, 224, 4, 64, 7, 2)
    assert p.family == L.DS_FAM_IGEMM and (p.d.fold_cin, p.d.Cin, p.d.KW) == (4, 28, 1)
    # the 16-bit configurations: the same packed-RGB kernel on the bf16 matrix cores (ds_conv_stem_bf16)
    rc, p = plan(L.DS_CONV_FWD, L.DS_ARITH_BF16, L.DS_PLAN_PACKED_RGB, B, 224, 224, 4, 64, 7, 2, L.DS_EPI_STATS)
    assert p.family == L.DS_FAM_STEM and p.d.dtype == L.DS_DTYPE_BF16 and p.partials > 0
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_FP8, L.DS_PLAN_PACKED_RGB | L.DS_PLAN_NO_STEM_DIRECT, B, 224, 224, 4, 64, 7, 2)[1].family == L.DS_FAM_IGEMM
    # arithmetic configurations
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_F32X3, 0, B, 28, 28, 192, 176, 1, 1)[1].family == L.DS_FAM_F32X3
    assert plan(L.DS_CONV_DGRAD, L.DS_ARITH_F32X3, 0, B, 28, 28, 192, 176, 1, 1, ldx=176, ldz=192)[1].family == L.DS_FAM_IGEMM
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_F32X3, 0, B, 28, 28, 96, 128, 3, 1)[1].family == L.DS_FAM_WINO4
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_BF16, 0, B, 28, 28, 192, 16, 1, 1)[1].family == L.DS_FAM_IGEMM      # narrow
    rc, p = plan(L.DS_CONV_FWD, L.DS_ARITH_BF16, L.DS_PLAN_ACT16, B, 28, 28, 192, 16, 1, 1)
    assert p.family == L.DS_FAM_BF16D and p.x16_ok == 1 and p.d.dtype == L.DS_DTYPE_BF16
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_BF16, L.DS_PLAN_NO_BF16_DIRECT, B, 28, 28, 192, 176, 1, 1)[1].family == L.DS_FAM_IGEMM
    rc, p = plan(L.DS_CONV_FWD, L.DS_ARITH_FP8, 0, B, 14, 14, 96, 208, 3, 1, L.DS_EPI_STATS)
    assert p.family == L.DS_FAM_FP8D and p.a_format == L.DS_FP8_E4M3 and p.wscale_floats == 4 + 512 and p.partials > 0
    rc, p = plan(L.DS_CONV_DGRAD, L.DS_ARITH_FP8, L.DS_PLAN_FP8_EVERYWHERE, B, 14, 14, 96, 208, 3, 1, ldx=208, ldz=96)
    assert p.family == L.DS_FAM_FP8D and p.a_format == L.DS_FP8_E5M2
    # ... only where fp8 beats the bf16 kernels (forward 3x3, >= 96 input channels, 14 x 14 and larger), unless forced
    assert plan(L.DS_CONV_DGRAD, L.DS_ARITH_FP8, 0, B, 14, 14, 96, 208, 3, 1, ldx=208, ldz=96)[1].family != L.DS_FAM_FP8D
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_FP8, 0, B, 7, 7, 192, 384, 3, 1)[1].family == L.DS_FAM_BF16D
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_FP8, 0, B, 28, 28, 256, 288, 1, 1)[1].family == L.DS_FAM_BF16D
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_FP8, L.DS_PLAN_ACT16, B, 14, 14, 16, 48, 3, 1)[1].family == L.DS_FAM_BF16D
    assert plan(L.DS_CONV_FWD, L.DS_ARITH_FP8, L.DS_PLAN_FP8_EVERYWHERE, B, 14, 14, 16, 48, 3, 1)[1].family == L.DS_FAM_FP8D
    assert plan(L.DS_CONV_DGRAD, L.DS_ARITH_FP8, L.DS_PLAN_NO_WINO4H, B, 56, 56, 64, 192, 3, 1, ldx=192, ldz=64)[1].family == L.DS_FAM_IGEMM
    # zcat probe: BatchNorm + ReLU on load is the wide 1x1 kernel's (and the f32x3 kernel's)
    rc, p = plan(L.DS_CONV_FWD, L.DS_ARITH_F32, 0, B, 28, 28, 256, 288, 1, 1, L.DS_EPI_STATS)
    assert l.ds_conv_plan_norm_supported(C.byref(p)) == 1
    assert l.ds_conv_plan_norm_supported(C.byref(plan(L.DS_CONV_FWD, L.DS_ARITH_F32, 0, B, 28, 28, 96, 128, 3, 1)[1])) == 0
    # errors are codes
    assert plan(2, L.DS_ARITH_F32, 0, B, 28, 28, 96, 128, 3, 1)[0] == -1 and b"role" in l.ds_last_error()
    assert plan(L.DS_CONV_DGRAD, L.DS_ARITH_F32, 0, B, 28, 28, 96, 128, 3, 2)[0] == -1
    assert l.ds_conv_run(None, None, None, None, None, None) == -1
    assert l.ds_conv_prepare_weights(None, None, None, None, None) == -1


def test_engine_reaches_the_conv_kernels_only_through_the_plan_interface():
    """The image engine names no kernel family: every conv of the tower goes ds_conv_plan -> ds_conv_prepare_weights ->
    ds_conv_run (ops.LayerPlan).  The family-level entry points stay for kernel tests and tuning scripts."""
    src = open(os.path.join(ROOT, "tumblr_emotions_amd", "engine_image.py")).read()
    for banned in ("WinoPlan", "Bf16Plan", "Fp8Plan", "F32x3Plan", "StemPlan", "ConvPlan(", "wino4_prefer", "wino_fwd",
                   "wino_dgrad", "wino_transform_weights", "weights_to_"):
        assert banned not in src, banned
    assert src.count("ops.LayerPlan(") >= 3
