"""CPU-side checks of the C-ABI shared library: it loads and exports every symbol that
include/text_alignment_amd.h declares (no compute calls without a GPU)."""
import os
import re

import pytest

from conftest import REPO


def _declared_symbols():
    hdr = open(os.path.join(REPO, "include", "text_alignment_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ta_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_exported(native):
    syms = _declared_symbols()
    assert "ta_nw_batch" in syms and "ta_version" in syms
    for s in syms:
        assert hasattr(native.lib, s), s
    assert sorted(native.EXPORTS) == syms


def test_version_and_sizes(native):
    lib = native.lib
    assert lib.ta_version() >= 100
    assert lib.ta_nw_workspace_bytes(0, 10) == 0
    b = lib.ta_nw_workspace_bytes(4096, 4096)
    # 1 byte per cell + the skew padding of the strip layout, sized for the finest strips a caller can ask
    # for (TA_NW_ROWS(1): 63 columns + a group's round-up per 64-row strip, 4.7 % at 4096 columns)
    assert b % 1024 == 0 and 4096 * 4096 <= b <= int(4096 * 4096 * 1.05)
    assert lib.ta_nw_max_m() >= 8192
    assert lib.ta_nw_general_ptr_bytes(3, 4) == 20


def test_argument_errors_without_gpu(native):
    lib = native.lib
    # validation happens before any HIP call, so this is safe on a CPU-only box
    rc = lib.ta_nw_batch(None, None, None, None, 1, None, 0, None, None, None, None, None,
                         10, 10, 100, 3, None)
    assert rc == native.TA_EINVAL
    assert b"null" in lib.ta_last_error()
    with pytest.raises(ValueError):
        native.check(rc, "ta_nw_batch")
    assert lib.ta_nw_batch(None, None, None, None, -1, None, 0, None, None, None, None, None,
                           0, 0, 0, 3, None) == native.TA_EINVAL
    # every other entry point refuses null pointers the same way, before touching the device
    calls = [
        lambda: lib.ta_nw2_batch(None, None, None, None, 1, None, 0, None, None, None, None, None, 10, 10, 100, 3, None),
        lambda: lib.ta_nw_general(None, 3, None, 3, None, None, 0, None, None, None, None, None),
        lambda: lib.ta_lstm_forward(None, None, None, None, 1, None, None, None, 0, None, None, None, None),
        lambda: lib.ta_nw_general_batch(None, None, None, None, 2, None, 0, None, None, None, None, None, None, None, None),
        lambda: lib.ta_lstm_output(None, 16, None, 96, None, None, None, None),
        lambda: lib.ta_decode_summary(None, None, None, 1, 0.7, None, None, None, None, None),
        lambda: lib.ta_decode(None, None, None, 1, 96, 0.7, None, None, None, None, None),
        lambda: lib.ta_linenorm_measure(None, None, None, None, 1, None, None, None, None, None, None, None,
                                        None, None, None, None, None),
        lambda: lib.ta_linenorm_resample(None, None, None, None, 1, None, None, None, None, None, None, None,
                                         None, None, None, None),
        lambda: lib.ta_pp_histogram(None, 10, None, None),
        lambda: lib.ta_pp_threshold(None, 10, 128, 0, None, None),
        lambda: lib.ta_pp_label(None, 4, 4, None, None, None, None),
        lambda: lib.ta_pp_components(None, None, 4, 4, None, 8, None, None),
        lambda: lib.ta_pp_filter_components(None, None, None, 4, 4, 1, 9, None),
        lambda: lib.ta_pp_invert(None, 10, None),
        lambda: lib.ta_pp_angle_histograms(None, 4, 4, 1, None, 2, None, None),
        lambda: lib.ta_pp_rotate(None, 4, 4, None, 4, 4, None, None),
        lambda: lib.ta_pp_open_runs(None, None, 4, 4, 2, 0, None),
        lambda: lib.ta_pp_row_sums(None, 4, 4, None, None),
        lambda: lib.ta_pp_clear_rows(None, 4, None, 1, None),
    ]
    for call in calls:
        assert call() == native.TA_EINVAL
        assert b"null" in lib.ta_last_error()
    assert lib.ta_lstm_packed_weight_floats(0) == 2 * 7 * 4 * 38 * 64


def test_phase1_plan_picks_the_measured_launch_shapes(native):
    """ta_nw2_phase1_plan is a pure host function: the shapes below are the ones whose workgroup widths
    were timed on MI355X (tools/p1_time.py) -- the model must keep choosing the fastest."""
    import ctypes
    from text_alignment_amd import _native
    out = (ctypes.c_int32 * 8)()
    same = _native.TA_NW_OPENS_SAME

    def plan(n, m, nprob, alphabet):
        assert native.lib.ta_nw2_phase1_plan_batch(n, m, nprob, (alphabet << _native.TA_NW_ALPHABET_SHIFT) | same, out) == 0
        return {"mode": out[0], "waves": out[1], "lds": out[2], "samego": out[3]}
    assert plan(4096, 4096, 4096, 31) == {"mode": 2, "waves": 4, "lds": 41312, "samego": 1}
    assert plan(8192, 8192, 512, 27)["waves"] == 8     # 32 strips, a batch that needs wide workgroups to fill the chip: 6.3 ms at 8, 7.2 at 4, 13.7 at 2
    assert plan(2048, 2048, 2048, 27)["waves"] == 2    # 1.78 ms at 2, 1.83 at 4, 2.07 at 8
    assert plan(2048, 2048, 1024, 27)["waves"] == 4    # half the batch: 0.99 ms at 4, 1.06 at 2, 1.10 at 8
    assert plan(1024, 1024, 2048, 27)["waves"] == 2    # 4 strips: 0.53 ms at 2, 0.59 at 4, 0.58 at 1
    assert plan(4096, 4096, 4096, 0)["mode"] == 1      # no alphabet hint: compare-select cell
    assert plan(4096, 4096, 4096, 0)["waves"] == 4     # 14.5 ms at 4, 14.9 at 8, 16.9 at 2
    assert plan(300, 60000, 64, 27)["mode"] == 1       # a profile that does not fit beside 60000 codes
    assert native.lib.ta_nw2_phase1_plan(4096, 4096, (31 << _native.TA_NW_ALPHABET_SHIFT) | same, out) == 0 and out[1] == 4
    assert native.lib.ta_nw2_phase1_plan_batch(-1, 5, 1, 0, out) != 0


def test_scoring_parse_and_errors():
    from text_alignment_amd import textSeqCompare as tsc
    assert tsc.default_sys == [8, -4, -7, -7, -3, 0] and tsc.gap_extend == -1
    assert tsc.parse_scoring_system(None)[0] == [8, -4, -7, -7, -3, 0]
    assert tsc.parse_scoring_system([10, -5, -7, -2])[0] == [10, -5, -7, -7, -2, -2]
    f = lambda a, b: 1
    p, fn = tsc.parse_scoring_system([f, -1, -2, -3, -4])
    assert fn is f and p[2:] == [-1, -2, -3, -4]
    for bad in ([1, 2, 3], [1, 2, 3, 4, 5], [1, 2, 3, 4, 5, 6, 7], []):
        with pytest.raises(ValueError) as ei:
            tsc.perform_alignment(list("ab"), list("ab"), bad)
        assert str(ei.value) == 'scoring_system {} invalid'.format(bad)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from text_alignment_amd import textSeqCompare as tsc
    with pytest.raises(RuntimeError):
        tsc.perform_alignment(list("abc"), list("abd"))


def test_product_never_imports_oracle():
    """No file of the package imports, loads or executes anything under oracle/: no import statement,
    no dlopen of its library, and the word itself only where a docstring names a checker FILE
    (`oracle/<name>`) the tests compare the kernels with."""
    import re
    pkg = os.path.join(REPO, "text_alignment_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(root, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert not re.search(r"import_module\([^)]*oracle|__import__\([^)]*oracle|libnw_oracle|CDLL\([^)]*oracle", src), f
                rest = re.sub(r"oracle/[A-Za-z0-9_]+(\.py|\.c)?", "", src)
                assert "oracle" not in rest.replace("the oracle", "").replace("CPU oracle", ""), f


# ---- the C ABI from plain C (no Python, no torch): tests/native/abi_c_example.c ---------------
_C_SRC = os.path.join(REPO, "tests", "native", "abi_c_example.c")
_C_BIN = os.path.join(REPO, "tests", "native", "build", "abi_c_example")


def _build_c_example():
    import subprocess
    os.makedirs(os.path.dirname(_C_BIN), exist_ok=True)
    libdir = os.path.join(REPO, "text_alignment_amd")
    subprocess.check_call(
        ["gcc", "-std=c99", "-Wall", "-I", os.path.join(REPO, "include"), "-I", "/opt/rocm/include",
         "-D__HIP_PLATFORM_AMD__", _C_SRC, "-L", libdir, "-lta_hip", "-L", "/opt/rocm/lib", "-lamdhip64",
         "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", _C_BIN])
    return _C_BIN


def test_c_example_builds_against_header(native):
    """the header is C (not C++), self-contained, and the library links from gcc"""
    assert os.path.exists(_build_c_example())


@pytest.mark.gpu
def test_c_example_runs_and_matches_oracle(native):
    import subprocess
    from oracle import nw_oracle
    exe = _build_c_example()
    for t, o in [("dominus deus meus", "domnus dcus  meuss"), ("abc", "xbcd"), ("aaaa", "a"),
                 ("quoniam confirmata est super nos misericordia eius", "quonia cofirmata est supcr nos miscricordia cius")]:
        out = subprocess.run([exe, t, o], check=True, capture_output=True, text=True, timeout=120).stdout.split("\n")
        tra, ocr = nw_oracle.perform_alignment(list(t), list(o))
        assert out[0] == "".join(tra) and out[1] == "".join(ocr), (t, o, out)
        # TA_NW_CHECK_IDS: an alphabet the ids do not fit is refused, a true one gives the same alignment
        assert out[2] == "check_ids: wrong alphabet rc=-1, true alphabet rc=0, two-phase equals one-pass=1", out[2]


def test_inline_asm_f64_mfmas_keep_their_wait_states():
    """csrc/ta_lstm_f64.hip issues v_mfma_f64_16x16x4_f64 as inline assembly (AGPR-resident weights named in the
    instruction), which the compiler's hazard recogniser does not see: the wait states around a chain are the kernel's
    own (mfma_begin / mfma_settle).  Checked on the disassembly of the built library (tools/check_mfma_hazard.py): no
    instruction but the next MFMA of the chain touches a result within 18 wait states, no VALU write of a source
    within 2 before."""
    from tools import check_mfma_hazard as chk
    from text_alignment_amd import _native
    found, nmfma, nco = chk.check(_native.LIB_PATH)
    assert nco >= 1 and nmfma >= 625             # the recurrence's 25 x 25 tiles are there
    assert found == []
    # the checker itself
    bad = """
0000000000001000 <kern>:
	v_mfma_f64_16x16x4_f64 v[42:49], a[0:1], v[102:103], v[42:49]  // 000000001000: D3EE002A 04AACD00
	v_mfma_f64_16x16x4_f64 v[42:49], a[2:3], v[104:105], v[42:49]  // 000000001008: D3EE002A 04AAD102
	s_nop 15                                                   // 000000001010: BF80000F
	v_max_f64 v[52:53], v[48:49], v[48:49]                     // 000000001014: D2680034 00026130
	v_mov_b32_e32 v60, v1                                      // 00000000101C: 7E780301
	v_mfma_f64_16x16x4_f64 v[70:77], v[60:61], v[104:105], v[70:77]  // 000000001020: D3EE0046 051AD13C
	s_nop 15                                                   // 000000001028: BF80000F
	s_nop 2                                                    // 00000000102C: BF800002
	v_max_f64 v[52:53], v[76:77], v[76:77]                     // 000000001030: D2680034 0002994C
"""
    got, n = chk.findings(bad)
    assert n == 3 and len(got) == 2, got
    assert "after 16 wait state(s)" in got[0] and "v_max_f64" in got[0]
    assert "v_mov_b32_e32" in got[1] and "0 wait state(s) before" in got[1]


def test_no_buffer_store_data_hazard_in_the_built_library():
    """16-byte buffer stores must not carry an SGPR soffset, and no VALU
    instruction may write a store's data VGPRs within two issue slots behind it.  Checked on the
    disassembly of the gfx950 code objects inside the built libta_hip.so (tools/check_store_hazard.py):
    a compiler update that re-introduces the form would otherwise only show as a numerically wrong batch."""
    from tools import check_store_hazard as chk
    from text_alignment_amd import _native
    found, nstores, nco = chk.check(_native.LIB_PATH)
    assert nco >= 1 and nstores >= 16            # the straight-line bottom-row stores of nw_score_kernel are there
    assert found == []
    # the checker itself, on the two forms it exists for
    bad = """
0000000000001000 <kern>:
	buffer_store_dwordx4 v[4:7], v30, s[16:19], s3 offen offset:16 // 000000001000: E07C1010 03040 41E
	v_add_u32_e32 v9, v1, v2                                   // 000000001008: 68120501
	buffer_store_dwordx4 v[8:11], v30, s[16:19], 0 offen       // 00000000100C: E07C1000 8004081E
	s_nop 0                                                    // 000000001014: BF800000
	v_max_i32_e32 v10, v1, v2                                  // 000000001018: 1A140501
	buffer_store_dwordx4 v[12:15], v30, s[16:19], 0 offen      // 00000000101C: E07C1000 80040C1E
	s_nop 1                                                    // 000000001024: BF800001
	v_max_i32_e32 v12, v1, v2                                  // 000000001028: 1A180501
"""
    got = chk.findings(bad)
    assert len(got) == 2 and "soffset" in got[0] and "v_max_i32_e32 v10" in got[1], got


def test_traceback_plan_is_a_pure_function_of_batch_shape(native):
    """ta_nw2_traceback_plan (host only): the phase-2 launch shape ta_nw2_batch picks -- several waves per problem for
    small batches, half-strip pairs (alone or with speculating waves) for larger ones under ONE scoring system, never
    a pair kernel with a scoring system per problem (their halves run in lockstep); TA_NW_TBWAVES forces a shape."""
    plan = native.lib.ta_nw2_traceback_plan
    assert [plan(n, 0, 0) for n in (1, 64, 832)] == [4, 4, 4]
    assert [plan(n, 0, 0) for n in (833, 1024, 1088)] == [6, 6, 6]
    assert [plan(n, 0, 0) for n in (1089, 2048, 2560)] == [5, 5, 5]
    assert [plan(n, 0, 0) for n in (2561, 4096, 100000)] == [3, 3, 3]
    assert all(plan(n, 6, 0) in (1, 2, 4) for n in (1, 500, 1000, 1500, 2187, 4096))
    for w in (1, 2, 3, 4, 5, 6):
        assert plan(4096, 0, w << native.TA_NW_TBWAVES_SHIFT) == w and plan(7, 6, w << native.TA_NW_TBWAVES_SHIFT) == w
