"""GPU parity tests of the NW hot path: HIP kernels (through the C ABI) vs the CPU oracle and
vs golden vectors captured from the reference (textSeqCompare.py:13-177).  Bit-exact."""
import hashlib

import numpy as np
import pytest

from conftest import kat_scoring, load_golden, unrle

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def tsc():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from text_alignment_amd import textSeqCompare
    return textSeqCompare


@pytest.fixture(params=[False, True], ids=["one-pass", "two-phase"])
def two_phase(request):
    return request.param


@pytest.fixture(params=[4, 2, 1], ids=["rows4", "rows2", "rows1"])
def rows(request):
    """rows per lane of the one-pass fill (256- / 128- / 64-row strips; the library picks 4 or 2, 1 is
    reachable through TA_NW_ROWS only); ignored by the two-phase aligner"""
    return request.param


def _sha16(tra, ocr):
    return hashlib.sha256(("".join(tra) + "|" + "".join(ocr)).encode()).hexdigest()[:16]


def _decode_ptr(ws, n, m, R=4):
    """Pointer bytes of one problem from the kernel's strip layout (nw_cell.h PtrLayout)."""
    SR, SPG = 64 * R, 16 // R
    ngroups = (m + 63 + SPG - 1) // SPG
    strip_bytes = ngroups * 1024
    i = np.arange(1, n + 1)[:, None]
    j = np.arange(1, m + 1)[None, :]
    i0 = i - 1
    strip, l, r = i0 // SR, (i0 % SR) // R, i0 % R
    k = (j - 1) + l
    addr = strip * strip_bytes + ((k // SPG) * 64 + l) * 16 + (k % SPG) * R + r
    b = ws[addr]
    return (2 - (b & 3)) | ((2 - ((b >> 2) & 3)) << 2) | ((2 - ((b >> 4) & 3)) << 4)


def test_golden_kat(tsc):
    g = load_golden("nw_kat.json")
    for c in g["cases"]:
        tra, ocr = tsc.perform_alignment(c["transcript"], c["ocr"], kat_scoring(c))
        assert tra == c["tra_align"], c["name"]
        assert ocr == c["ocr_align"], c["name"]


def test_golden_random_small_one_launch(tsc):
    g = load_golden("nw_random_small.json")
    pairs = [(list(c["t"]), list(c["o"])) for c in g["cases"]]
    systems = [c["scoring"] for c in g["cases"]]
    res = tsc.perform_alignment_batch(pairs, systems)
    for c, (tra, ocr) in zip(g["cases"], res):
        assert "".join(tra) == c["tra"] and "".join(ocr) == c["ocr"], c


def test_golden_synth(tsc):
    from tools.synth import synth_pair
    g = load_golden("nw_synth.json")
    for c in g["cases"]:
        t, o = synth_pair(c["n"], c["m"], c["seed"])
        sc = np.array(c["scoring"]) if (c["scoring"] is not None and c["seed"] == 99) else c["scoring"]
        tra, ocr = tsc.perform_alignment(t, o, sc)
        assert len(tra) == c["align_len"], (c["n"], c["m"])
        assert _sha16(tra, ocr) == c["sha16"], (c["n"], c["m"], c["seed"])


def test_inputs_not_mutated_and_types(tsc):
    t, o = list("abcd"), list("xbcy")
    tra, ocr = tsc.perform_alignment(t, o)
    assert t == list("abcd") and o == list("xbcy")
    assert isinstance(tra, list) and isinstance(ocr, list)
    assert ("".join(tra), "".join(ocr)) == ("a_bcd", "_xbcy")
    assert tsc.perform_alignment([], []) == ([], [])
    assert tsc.perform_alignment([], list("ab")) == (['_', '_'], ['a', 'b'])
    assert tsc.perform_alignment(list("ab"), []) == (['a', 'b'], ['_', '_'])


SYSTEMS = [[8, -4, -7, -7, -3, 0], [10, -5, -7, -7, -7, -7], [5, -10, -2, -7, 0, -5],
           [11, -4, -2, -2, 0, 0], [1, -1, -1, -1, -1, -1], [3, -3, 0, 0, 0, 0],
           [2, -1, 1, -3, -1, 1], [0, 0, 0, 0, 0, 0], [4, -6, -9, -1, -2, -4], [7, 7, 3, 2, 1, 1]]


def _random_problem(rng, n, m, asz, related):
    t = rng.integers(0, asz, size=n).astype(np.int32)
    o = rng.integers(0, asz, size=m).astype(np.int32)
    if related and n and m:
        k = min(n, m)
        o[:k] = np.where(rng.random(k) < 0.8, t[:k], o[:k])
    return t, o


def test_ragged_batch_vs_oracle_full_pointer_matrix(tsc, two_phase, rows):
    """Ragged sizes around every strip / group boundary, per-problem scoring systems; compares
    the whole pointer matrix (not only the traceback path) with the oracle."""
    from oracle import nw_oracle
    rng = np.random.default_rng(7)
    sizes = [(0, 0), (0, 5), (5, 0), (1, 1), (1, 70), (70, 1), (3, 64), (64, 3), (63, 63), (64, 64),
             (65, 65), (255, 300), (256, 61), (257, 130), (511, 17), (512, 512), (513, 200),
             (300, 1000), (1000, 300), (1024, 64), (1025, 1023), (700, 2049), (2049, 130),
             (100, 4500), (2100, 600)]
    t_list, o_list, prm = [], [], []
    for k, (n, m) in enumerate(sizes):
        t, o = _random_problem(rng, n, m, [2, 4, 27][k % 3], k % 2 == 0)
        t_list.append(t); o_list.append(o); prm.append(SYSTEMS[k % len(SYSTEMS)])
    if two_phase and rows != 4:
        pytest.skip("rows per lane is a one-pass launch shape")
    batch = tsc.NWBatch(t_list, o_list, prm, two_phase=two_phase)
    batch.rows = rows
    batch.run()
    torch.cuda.synchronize()
    res = batch.results()
    ws = batch.ws.cpu().numpy()
    ws_off = batch.ws_off.cpu().numpy()
    for k, (n, m) in enumerate(sizes):
        want_ops, want_ptr, _ = nw_oracle.align_ids(t_list[k], o_list[k], prm[k], want_ptr=True)
        assert res[k].tolist() == want_ops.tolist(), (k, n, m, prm[k])
        if n and m and not two_phase:
            got_ptr = _decode_ptr(ws[ws_off[k]:], n, m, R=rows)
            assert np.array_equal(got_ptr, want_ptr[1:, 1:]), (k, n, m, prm[k])


def test_two_phase_strip_borders_and_start_probe(tsc):
    """The two-phase traceback takes a step out of a strip with its next state pending and reads the
    state off the strip above; a walk that STARTS in a strip's first row (n = 256 s + 1) probes the
    strip above before its first step.  Every scoring system (gap-heavy ones make the path leave
    strips in states 1 and 2 as well), OCR strings of 1, 2 and a few tokens, and long ones."""
    from oracle import nw_oracle
    rng = np.random.default_rng(11)
    t_list, o_list, prm = [], [], []
    shapes = [(257, 300), (513, 1), (257, 1), (513, 2), (769, 640), (257, 64), (512, 300), (258, 257),
              (513, 513), (257, 5), (1025, 70), (1281, 1300), (2049, 2049), (256, 256), (1024, 3)]
    for k, (n, m) in enumerate(shapes):
        for sc in SYSTEMS:
            t, o = _random_problem(rng, n, m, [2, 3, 27][k % 3], k % 2 == 1)
            t_list.append(t); o_list.append(o); prm.append(sc)
    batch = tsc.NWBatch(t_list, o_list, prm, two_phase=True)
    batch.run()
    torch.cuda.synchronize()
    res = batch.results()
    for k in range(len(t_list)):
        want = nw_oracle.align_ids(t_list[k], o_list[k], prm[k])
        assert res[k].tolist() == want.tolist(), (len(t_list[k]), len(o_list[k]), prm[k])


@pytest.mark.parametrize("wide", [True, False], ids=["wide", "narrow"])
def test_one_pass_launch_shapes(tsc, wide, rows):
    """The one-pass fill spread over several workgroups per problem (hand-off rows in HBM, the
    default for batches smaller than the CU count) against one workgroup per problem: whole
    pointer matrices and alignments vs the oracle, ragged strip counts in one launch."""
    from oracle import nw_oracle
    rng = np.random.default_rng(23)
    sizes = [(1030, 70), (1300, 3000), (5000, 2000), (8192, 300), (2049, 64), (1024, 1024), (1025, 1),
             (4100, 4100), (300, 300), (0, 7), (2560, 129)]
    t_list, o_list, prm = [], [], []
    for k, (n, m) in enumerate(sizes):
        t, o = _random_problem(rng, n, m, [2, 4, 27][k % 3], k % 2 == 0)
        t_list.append(t); o_list.append(o); prm.append(SYSTEMS[k % len(SYSTEMS)])
    batch = tsc.NWBatch(t_list, o_list, prm, two_phase=False, wide=wide)
    batch.rows = rows
    for _ in range(2):                       # second run: progress words are re-armed per launch
        batch.run()
    torch.cuda.synchronize()
    res = batch.results()
    ws = batch.ws.cpu().numpy()
    ws_off = batch.ws_off.cpu().numpy()
    for k, (n, m) in enumerate(sizes):
        want_ops, want_ptr, _ = nw_oracle.align_ids(t_list[k], o_list[k], prm[k], want_ptr=True)
        assert res[k].tolist() == want_ops.tolist(), (k, n, m, prm[k])
        if n and m:
            got_ptr = _decode_ptr(ws[ws_off[k]:], n, m, R=rows)
            assert np.array_equal(got_ptr, want_ptr[1:, 1:]), (k, n, m, prm[k])


def test_waves_per_problem_variants(tsc, two_phase):
    """Same problems through batches whose largest problem selects W = 1, 2, 4, 8 waves."""
    from oracle import nw_oracle
    rng = np.random.default_rng(11)
    base = [_random_problem(rng, n, m, 27, True) for n, m in [(200, 333), (90, 1200), (256, 256)]]
    want = [nw_oracle.align_ids(t, o, SYSTEMS[0]).tolist() for t, o in base]
    for big_n in (250, 400, 1100, 2300, 4200):
        big = _random_problem(rng, big_n, 700, 27, True)
        probs = base + [big]
        batch = tsc.NWBatch([p[0] for p in probs], [p[1] for p in probs], SYSTEMS[0], two_phase=two_phase)
        batch.run()
        res = batch.results()
        for k in range(len(base)):
            assert res[k].tolist() == want[k], (big_n, k)
        assert res[-1].tolist() == nw_oracle.align_ids(big[0], big[1], SYSTEMS[0]).tolist(), big_n


def test_long_rows_and_long_columns(tsc, two_phase):
    from oracle import nw_oracle
    from tools.synth import synth_pair_ids
    for n, m, seed in [(8192, 8192, 5), (3000, 12000, 6), (12000, 900, 7)]:
        t, o = synth_pair_ids(n, m, seed)
        batch = tsc.NWBatch([t], [o], SYSTEMS[0], two_phase=two_phase)
        batch.run()
        got = batch.results()[0]
        want = nw_oracle.align_ids(t, o, SYSTEMS[0])
        assert got.tolist() == want.tolist(), (n, m)


def test_config2_batch_properties(tsc, two_phase):
    """BASELINE.json configs[1] (1024 problems of 2048 x 2048, default scoring): oracle comparison on
    a sample, size-independent properties on every problem of the batch."""
    from oracle import nw_oracle
    from tools.synth import synth_pair_ids
    nprob = 1024
    probs = [synth_pair_ids(2048, 2048, 1234 + k) for k in range(nprob)]
    batch = tsc.NWBatch([p[0] for p in probs], [p[1] for p in probs], SYSTEMS[0], two_phase=two_phase)
    batch.run()
    res = batch.results()
    for k, ops in enumerate(res):
        c = np.bincount(ops, minlength=3)
        assert c[0] + c[1] == 2048 and c[0] + c[2] == 2048, k      # every token exactly once
        assert len(c) == 3
    for k in (0, 1, 17, 100, 255, 600, 1023):
        want = nw_oracle.align_ids(probs[k][0], probs[k][1], SYSTEMS[0])
        assert res[k].tolist() == want.tolist(), k
    g = load_golden("nw_synth.json")
    c2048 = [c for c in g["cases"] if c["n"] == 2048][0]
    assert res[0].tolist() == unrle(c2048["ops_rle"])
    # idempotence: a second run over the same buffers gives the same bytes
    batch.run()
    res2 = batch.results()
    assert all(np.array_equal(a, b) for a, b in zip(res, res2))


def test_large_token_alphabets_and_hashable_tokens(tsc):
    """Tokens are arbitrary hashables compared with == (textSeqCompare.py:32): tuples, ints,
    strings; thousands of distinct ids go through the u16 code path."""
    from oracle import nw_oracle
    rng = np.random.default_rng(21)
    t = [(int(a), "x") for a in rng.integers(0, 3000, size=700)]
    o = [tok if rng.random() < 0.7 else (int(rng.integers(0, 3000)), "x") for tok in t[:650]]
    o += [int(v) for v in rng.integers(0, 50, size=40)]
    assert tsc.perform_alignment(t, o) == nw_oracle.perform_alignment(t, o)
    assert tsc.perform_alignment(t, o, [3, -2, -4, -1, -1, -2]) == nw_oracle.perform_alignment(t, o, [3, -2, -4, -1, -1, -2])


def test_random_scoring_systems_one_launch(tsc):
    """The parameter grid of evaluate_text_alignment.py:181-198 in one launch: 243 scoring systems
    drawn from the reference's 3^6 grid against one page-sized pair."""
    from oracle import nw_oracle
    from tools.synth import synth_pair
    grid = [(m1, m2, gx, gy, ex, ey) for m1 in (5, 8, 11) for m2 in (-10, -7, -4)
            for gx in (-7, -5, -2) for gy in (-7, -5, -2) for ex in (-5, -3, 0) for ey in (0,)]
    t, o = synth_pair(420, 390, 77)
    res = tsc.perform_alignment_batch([(t, o)] * len(grid), [list(g) for g in grid])
    for g, (tra, ocr) in list(zip(grid, res))[::9]:
        assert (tra, ocr) == nw_oracle.perform_alignment(t, o, list(g)), g
    assert len(res) == 243


def test_fuzz_2000_problems_one_launch(tsc, two_phase):
    """2000 random ragged problems (sizes 0..700, alphabets 2/4/27, 10 scoring systems incl.
    positive gap scores and all-zero) in one launch, every alignment against the oracle."""
    from oracle import nw_oracle
    rng = np.random.default_rng(2026)
    t_list, o_list, prm = [], [], []
    for k in range(2000):
        big = (k % 10 == 0)
        n = int(rng.integers(0, 700 if big else 120))
        m = int(rng.integers(0, 700 if big else 120))
        t, o = _random_problem(rng, n, m, [2, 4, 27][k % 3], k % 2 == 0)
        t_list.append(t); o_list.append(o); prm.append(SYSTEMS[int(rng.integers(0, len(SYSTEMS)))])
    batch = tsc.NWBatch(t_list, o_list, prm, two_phase=two_phase)
    batch.run()
    res = batch.results()
    bad = []
    for k in range(2000):
        want = nw_oracle.align_ids(t_list[k], o_list[k], prm[k])
        if res[k].tolist() != want.tolist():
            bad.append(k)
    assert not bad, bad[:10]


@pytest.mark.parametrize("wide", [None, True], ids=["auto", "wide"])
def test_fuzz_tall_problems_every_one_pass_launch_shape(tsc, rows, wide):
    """800 ragged problems up to 2600 x 2600 -- several strips per problem, strips handed from wave to wave
    (progress words in LDS) and, in the wide launch, from workgroup to workgroup (write-through rows in
    HBM) -- for both strip heights, twice over the same buffers.  This is the test that catches a missing
    progress wait: round 3's first block loop fetched a hand-off group without one, which only a fuzz of
    tall problems showed (1 problem in 800, not every run)."""
    from oracle import nw_oracle
    rng = np.random.default_rng(11)
    special = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049]
    t_list, o_list, prm = [], [], []
    for k in range(800):
        n = int(rng.choice(special)) if rng.random() < 0.4 else int(rng.integers(1, 2600))
        m = int(rng.choice(special)) if rng.random() < 0.4 else int(rng.integers(1, 2600))
        t, o = _random_problem(rng, n, m, [2, 3, 5, 27, 31][k % 5], rng.random() < 0.6)
        t_list.append(t); o_list.append(o); prm.append(SYSTEMS[int(rng.integers(0, len(SYSTEMS)))])
    batch = tsc.NWBatch(t_list, o_list, prm, two_phase=False, wide=wide)
    batch.rows = rows
    want = [nw_oracle.align_ids(t_list[k], o_list[k], prm[k]).tolist() for k in range(800)]
    for rep in range(2):
        batch.run()
        res = batch.results()
        bad = [k for k in range(800) if res[k].tolist() != want[k]]
        assert not bad, (rep, [(k, len(t_list[k]), len(o_list[k]), prm[k]) for k in bad[:5]])


def test_overflow_guard_routes_to_general_kernel(tsc):
    from oracle import nw_oracle
    t, o = list("abcabcabc"), list("abcbcaabc")
    big = [2 ** 21, -2 ** 20, -7, -7, -3, 0]
    assert tsc.perform_alignment(t, o, big) == nw_oracle.perform_alignment(t, o, big)


def test_ocr_longer_than_the_one_pass_lds_row_stays_on_the_integer_path(tsc):
    """m beyond ta_nw_max_m() (the one-pass kernel's LDS hand-off row): the two-phase aligner keeps
    its hand-off rows in the workspace and takes OCR strings up to ta_nw2_max_m() tokens -- the
    reference has no such limit (textSeqCompare.py:45-50).  Beyond that, the float64 kernel."""
    from oracle import nw_oracle
    from tools.synth import synth_pair, synth_pair_ids
    from text_alignment_amd import _native
    assert _native.lib.ta_nw2_max_m() >= 40000 > _native.lib.ta_nw_max_m()
    t, o = synth_pair_ids(700, 40000, 77)
    batch = tsc.NWBatch([t, t[:300]], [o, o[:17000]], SYSTEMS[0])
    assert batch.two_phase                                   # chosen automatically: too wide for one pass
    batch.run()
    res = batch.results()
    assert res[0].tolist() == nw_oracle.align_ids(t, o, SYSTEMS[0]).tolist()
    assert res[1].tolist() == nw_oracle.align_ids(t[:300], o[:17000], SYSTEMS[0]).tolist()
    with pytest.raises(OverflowError):
        tsc.NWBatch([t], [o], SYSTEMS[0], two_phase=False)
    tt, oo = synth_pair(700, 40000, 77)
    res = tsc.perform_alignment_batch([(tt, oo), (list("abc"), list("abd"))])
    assert res[0] == nw_oracle.perform_alignment(tt, oo)
    assert res[1] == nw_oracle.perform_alignment(list("abc"), list("abd"))
    m = _native.lib.ta_nw2_max_m() + 300                     # even wider: float64 kernel, same answer
    tt, oo = synth_pair(150, m, 78)
    assert tsc.perform_alignment(tt, oo) == nw_oracle.perform_alignment(tt, oo)


def test_float_scoring_grid_is_one_launch(tsc, monkeypatch):
    """A grid of NON-integral scoring systems (the float case of textSeqCompare.py:30-40 under the
    loop of evaluate_text_alignment.py:178-198) goes through ta_nw_general_batch: one launch, one
    workgroup per system, results as the float64 oracle's."""
    from oracle import nw_oracle
    from tools.synth import synth_pair
    from text_alignment_amd import _native, nw_general
    grid = [[m1 + 0.5, m2 - 0.25, gx, gy - 0.5, ex, 0.0] for m1 in (5, 8, 11) for m2 in (-10, -7, -4)
            for gx in (-7, -5, -2) for gy in (-7, -5, -2) for ex in (-5, -3, 0.5)]
    assert len(grid) == 243
    t, o = synth_pair(140, 130, 5)
    calls = []
    real = nw_general.align_batch
    monkeypatch.setattr(nw_general, "align_batch", lambda *a: (calls.append(len(a[0])), real(*a))[1])
    res = tsc.perform_alignment_batch([(t, o)] * len(grid), grid)
    assert calls == [243]
    for g, got in list(zip(grid, res))[::7]:
        assert got == nw_oracle.perform_alignment(t, o, g), g
    ragged = [(list("abcd"), list("xbcy")), ([], list("ab")), (t, o[:40]), (list("a"), [])]
    for (a, b), got in zip(ragged, tsc.perform_alignment_batch(ragged, [8.5, -4.25, -7, -7, -3, 0])):
        assert got == nw_oracle.perform_alignment(a, b, [8.5, -4.25, -7, -7, -3, 0])
    fn = lambda a, b: 3.5 if a == b else -2.0                                      # noqa: E731
    sys5 = [fn, -7, -7, -3, 0]                                  # the 5-element callable form, as ONE system
    got = tsc.perform_alignment_batch(ragged[:3], sys5)
    assert got == [nw_oracle.perform_alignment(a, b, sys5) for a, b in ragged[:3]]


def test_two_phase_many_problems_small_walk_window(tsc):
    """Batches of >= 2048 problems use the small LDS walk sub-window in phase 2 (several walkers
    per SIMD): ragged sizes, a few problems large enough for multi-strip, multi-window walks."""
    from oracle import nw_oracle
    rng = np.random.default_rng(31)
    sizes = [(int(rng.integers(1, 400)), int(rng.integers(1, 400))) for _ in range(2060)]
    sizes[7] = (2100, 1900); sizes[500] = (1500, 3000); sizes[2059] = (3000, 700); sizes[1000] = (0, 9)
    t_list, o_list = [], []
    for k, (n, m) in enumerate(sizes):
        t, o = _random_problem(rng, n, m, [2, 4, 27][k % 3], k % 2 == 0)
        t_list.append(t); o_list.append(o)
    batch = tsc.NWBatch(t_list, o_list, SYSTEMS[0], two_phase=True)
    batch.run()
    res = batch.results()
    for k in list(range(0, 2060, 41)) + [7, 500, 1000, 2059]:
        want = nw_oracle.align_ids(t_list[k], o_list[k], SYSTEMS[0])
        assert res[k].tolist() == want.tolist(), (k, sizes[k])


def test_bench_default_shape_properties(tsc):
    """The bench's own workload (4096 problems of 4096 x 4096, default scoring, two-phase) through
    size-independent properties: every token of both strings appears exactly once, no alignment
    column is a double gap, replicas of a problem give identical bytes, and the seed-1234 problem
    reproduces the alignment captured from the reference (tests/golden/nw_synth.json)."""
    from tools.synth import synth_pair_ids
    nprob, distinct = 4096, 16
    uniq = [synth_pair_ids(4096, 4096, 1234 + k) for k in range(distinct)]
    probs = [uniq[k % distinct] for k in range(nprob)]
    batch = tsc.NWBatch([p[0] for p in probs], [p[1] for p in probs], SYSTEMS[0])
    assert batch.two_phase
    batch.run()
    res = batch.results()
    for k, ops in enumerate(res):
        c = np.bincount(ops, minlength=3)
        assert len(c) == 3 and c[0] + c[1] == 4096 and c[0] + c[2] == 4096, k
    for k in range(distinct, nprob):
        assert np.array_equal(res[k], res[k % distinct]), k
    g = load_golden("nw_synth.json")
    c4096 = [c for c in g["cases"] if c["n"] == 4096][0]
    assert res[0].tolist() == unrle(c4096["ops_rle"])
    assert len(res[0]) == c4096["align_len"]


def test_wide_launch_oversubscribed(tsc):
    """255 problems of 4096 x 4096 take the wide one-pass launch (fewer problems than CUs) with
    more workgroups (255 x 4 chunks, 41 KB of LDS each) than the GPU holds at once: later chunks
    wait on earlier ones that must already be resident or done (in-order dispatch)."""
    from oracle import nw_oracle
    from tools.synth import synth_pair_ids
    distinct = 5
    uniq = [synth_pair_ids(4096, 4096, 4321 + k) for k in range(distinct)]
    probs = [uniq[k % distinct] for k in range(255)]
    batch = tsc.NWBatch([p[0] for p in probs], [p[1] for p in probs], SYSTEMS[0], two_phase=False)
    batch.run()
    res = batch.results()
    want = [nw_oracle.align_ids(t, o, SYSTEMS[0]) for t, o in uniq]
    for k, ops in enumerate(res):
        assert np.array_equal(ops, want[k % distinct]), k


def test_two_phase_with_16_bit_codes(tsc):
    """Alphabets beyond 254 symbols take the u16 code path of phase 1 (TA_NW_CODES8 unset)."""
    from oracle import nw_oracle
    rng = np.random.default_rng(77)
    probs = []
    for n, m in [(700, 650), (1300, 2100), (300, 5)]:
        t = rng.integers(0, 3000, size=n).astype(np.int32)
        o = rng.integers(0, 3000, size=m).astype(np.int32)
        k = min(n, m)
        o[:k] = np.where(rng.random(k) < 0.75, t[:k], o[:k])
        probs.append((t, o))
    batch = tsc.NWBatch([p[0] for p in probs], [p[1] for p in probs], SYSTEMS[1], two_phase=True)
    assert not batch.codes8
    batch.run()
    for (t, o), got in zip(probs, batch.results()):
        assert got.tolist() == nw_oracle.align_ids(t, o, SYSTEMS[1]).tolist()


def test_two_phase_walk_runs_off_the_lane_window(tsc):
    """Phase 2 keeps the pointer bytes of 32 lanes (128 rows) around the point where the walk enters a
    chunk.  Transcript-side gap runs of 100 .. 900 tokens at various row offsets (in strips, across
    strip borders, at the start / end of the transcript) make the walk climb out of that window --
    several times within one chunk for the long ones -- and the chunk is re-filled around the new
    position; scoring systems with free and with costly gap extension."""
    from oracle import nw_oracle
    rng = np.random.default_rng(2024)
    t_list, o_list, prm = [], [], []
    for runlen, at, system in [(100, 300, [8, -4, -7, -7, 0, 0]), (130, 250, SYSTEMS[0]), (300, 0, [8, -4, -7, -7, 0, 0]),
                               (517, 700, [8, -4, -7, -7, -1, -1]), (900, 130, [8, -4, -7, -7, 0, 0]),
                               (256, 1024, SYSTEMS[0]), (400, 1400, [11, -4, -2, -2, 0, 0]),
                               (129, 1, [5, -10, -2, -7, 0, -5]), (640, 896, [3, -3, 0, 0, 0, 0])]:
        base = rng.integers(0, 27, size=1500).astype(np.int32)
        junk = (27 + rng.integers(0, 4, size=runlen)).astype(np.int32)          # tokens the OCR string never has
        t = np.concatenate([base[:at], junk, base[at:]])
        o = base.copy()
        o[rng.random(o.size) < 0.05] = 26
        t_list.append(t); o_list.append(o); prm.append(system)
        t_list.append(np.concatenate([base, junk])); o_list.append(o); prm.append(system)      # the run ends the transcript
    batch = tsc.NWBatch(t_list, o_list, prm, two_phase=True)
    for _ in range(2):
        batch.run()
    for k, got in enumerate(batch.results()):
        want = nw_oracle.align_ids(t_list[k], o_list[k], prm[k])
        assert got.tolist() == want.tolist(), (k, len(t_list[k]), prm[k])
        assert (want == 1).sum() >= len(t_list[k]) - len(o_list[k])


@pytest.mark.parametrize("tb_waves", [1, 2, 3, 4, 5, 6])
def test_two_phase_traceback_waves_per_problem(tsc, tb_waves):
    """Phase 2's launch shapes: one wave per problem (large batches) and two / four waves that deal the chunks of
    the path among themselves and re-fill the chunk each EXPECTS ahead of the walk (nw_trace2w_kernel; the library
    picks by batch size, TA_NW_TBWAVES forces; 3 = the large-batch shape: two problems per wave, each walking back
    half-strips in 32 lanes, nw_trace2h_kernel; 5 / 6 = both at once: two / four waves per PAIR of problems on
    half-strips, nw_trace2hw_kernel).  Same alignments, bit for bit, on everything that stresses the
    speculation: walks that start in a strip's first row (probe) and leave strips in every state (pending states),
    long gap runs (the expected chunk is wrong many times in a row: the walk leaves the strip early, or stays in a
    chunk's lane range for hundreds of columns), table edges, all-tie tables, ragged sizes from 0, two runs of the
    same batch (nothing left over in the workspace or LDS)."""
    from oracle import nw_oracle
    rng = np.random.default_rng(4242)
    t_list, o_list, prm = [], [], []
    for k, (n, m) in enumerate([(257, 300), (513, 1), (257, 1), (513, 2), (769, 640), (257, 64), (512, 300), (258, 257),
                                (513, 513), (257, 5), (1025, 70), (1281, 1300), (2049, 2049), (256, 256), (1024, 3),
                                (0, 0), (0, 9), (9, 0), (1, 1), (63, 65), (64, 64), (300, 4500), (2100, 600),
                                (129, 200), (385, 1), (385, 2), (129, 64), (641, 700), (128, 128), (1153, 90)]):
        for sc in (SYSTEMS[0], SYSTEMS[k % len(SYSTEMS)], SYSTEMS[(3 * k + 1) % len(SYSTEMS)]):
            t, o = _random_problem(rng, n, m, [2, 3, 27][k % 3], k % 2 == 1)
            t_list.append(t); o_list.append(o); prm.append(sc)
    base = rng.integers(0, 27, size=1500).astype(np.int32)
    ins = rng.integers(0, 27, size=3000).astype(np.int32)
    for t, o in [(rng.integers(0, 10, size=2500).astype(np.int32), rng.integers(20, 30, size=2700).astype(np.int32)),
                 (base, np.concatenate([base[:700], ins, base[700:]])), (np.concatenate([base[:700], ins, base[700:]]), base),
                 (np.zeros(3000, np.int32), np.zeros(2600, np.int32)), (base, base[::-1].copy())]:
        for sc in (SYSTEMS[0], SYSTEMS[6], SYSTEMS[9], [3, -2, -1, -8, 0, -3]):
            t_list.append(t); o_list.append(o); prm.append(sc)
    for runlen, at, system in [(130, 250, SYSTEMS[0]), (517, 700, [8, -4, -7, -7, -1, -1]), (900, 130, [8, -4, -7, -7, 0, 0]),
                               (256, 1024, SYSTEMS[0]), (640, 896, [3, -3, 0, 0, 0, 0])]:
        junk = (27 + rng.integers(0, 4, size=runlen)).astype(np.int32)
        o = base.copy()
        o[rng.random(o.size) < 0.05] = 26
        t_list.append(np.concatenate([base[:at], junk, base[at:]])); o_list.append(o); prm.append(system)
        t_list.append(o); o_list.append(np.concatenate([base[:at], junk, base[at:]])); prm.append(system)
    for k in range(300):
        n, m = int(rng.integers(0, 900)), int(rng.integers(0, 900))
        t, o = _random_problem(rng, n, m, [2, 4, 27][k % 3], k % 2 == 0)
        t_list.append(t); o_list.append(o); prm.append(SYSTEMS[int(rng.integers(0, len(SYSTEMS)))])
    batch = tsc.NWBatch(t_list, o_list, prm, two_phase=True)
    batch.tb_waves = tb_waves
    for _ in range(2):
        batch.ops.fill_(7)
        batch.run()
    bad = [k for k, got in enumerate(batch.results())
           if got.tolist() != nw_oracle.align_ids(t_list[k], o_list[k], prm[k]).tolist()]
    assert not bad, (tb_waves, bad[:10], [(len(t_list[k]), len(o_list[k]), prm[k]) for k in bad[:5]])


@pytest.mark.parametrize("tb_waves", [3, 5, 6])
def test_fuzz_pair_kernels_ragged_sizes(tsc, tb_waves):
    """The traceback kernels that carry TWO problems per wave (half-strips of 128 rows; with and without speculating
    waves) on 1 200 random problems of very different sizes -- the two halves of a wave finish their problems rounds
    apart, cross half-strip and strip borders at different times, and one of them is often idle -- first under one
    scoring system (what the library pairs up by itself), then under a system per problem; odd batch size."""
    from oracle import nw_oracle
    rng = np.random.default_rng(777 + tb_waves)
    t_list, o_list = [], []
    for k in range(1201):
        n = int(rng.integers(0, 1400)) if k % 7 else int(rng.integers(0, 40))
        m = int(rng.integers(0, 1400)) if k % 5 else int(rng.integers(0, 40))
        t, o = _random_problem(rng, n, m, [2, 4, 27][k % 3], k % 2 == 0)
        t_list.append(t); o_list.append(o)
    for prm in (SYSTEMS[0], [SYSTEMS[int(v)] for v in rng.integers(0, len(SYSTEMS), size=len(t_list))]):
        batch = tsc.NWBatch(t_list, o_list, prm, two_phase=True)
        batch.tb_waves = tb_waves
        batch.run()
        res = batch.results()
        bad = []
        for k in range(len(t_list)):
            sysk = prm if isinstance(prm[0], int) else prm[k]
            if res[k].tolist() != nw_oracle.align_ids(t_list[k], o_list[k], sysk).tolist():
                bad.append(k)
        assert not bad, (tb_waves, bad[:10], [(len(t_list[k]), len(o_list[k])) for k in bad[:5]])


def test_adversarial_paths(tsc, two_phase):
    """Paths that stress the windowed traceback and the tie rules: disjoint alphabets (the path
    hugs the table edges), a 3000-token insertion in the middle (one horizontal / vertical run
    crossing many windows and strips), constant strings (every cell ties), scoring systems that
    reward gaps."""
    from oracle import nw_oracle
    rng = np.random.default_rng(123)
    base = rng.integers(0, 27, size=1500).astype(np.int32)
    ins = rng.integers(0, 27, size=3000).astype(np.int32)
    cases = [
        (rng.integers(0, 10, size=2500).astype(np.int32), rng.integers(20, 30, size=2700).astype(np.int32)),
        (base, np.concatenate([base[:700], ins, base[700:]])),
        (np.concatenate([base[:700], ins, base[700:]]), base),
        (np.zeros(3000, np.int32), np.zeros(2600, np.int32)),
        (np.tile(np.array([1, 2], np.int32), 1400), np.tile(np.array([2, 1], np.int32), 1300)),
        (base, base[::-1].copy()),
    ]
    systems = [SYSTEMS[0], SYSTEMS[6], SYSTEMS[7], SYSTEMS[9], [3, -2, -1, -8, 0, -3]]
    t_list, o_list, prm = [], [], []
    for t, o in cases:
        for s in systems:
            t_list.append(t); o_list.append(o); prm.append(s)
    batch = tsc.NWBatch(t_list, o_list, prm, two_phase=two_phase)
    batch.run()
    for k, got in enumerate(batch.results()):
        want = nw_oracle.align_ids(t_list[k], o_list[k], prm[k])
        assert got.tolist() == want.tolist(), (k // len(systems), prm[k])


def _phase1_plan(batch):
    import ctypes
    from text_alignment_amd import _native
    out = (ctypes.c_int32 * 4)()
    flags = (_native.TA_NW_CODES8 if batch.codes8 else 0) | batch.phase1_flags()
    assert _native.lib.ta_nw2_phase1_plan_batch(batch.max_n, batch.max_m, batch.nprob, flags, out) == 0
    return {"mode": out[0], "waves": out[1], "lds": out[2], "samego": out[3]}


@pytest.mark.parametrize("variant", ["profile", "compare"])
@pytest.mark.parametrize("waves", [None, 1, 2, 4, 8])
def test_phase1_variants_and_forced_hints(tsc, variant, waves):
    """Phase 1 of the two-phase aligner in both forms (score profile in LDS / compare-select) and
    with every workgroup width, on a launch whose hints are FORCED on although some problems do not
    meet them (positive gap opens, gap_open_x != gap_open_y, substitution scores beyond a byte):
    those problems must still come out right, through the kernel's general path."""
    from oracle import nw_oracle
    from text_alignment_amd import _native
    def shape(b):                                 # launch-shape overrides through the ABI's flag bits
        b.no_profile = (variant == "compare")
        b.waves = waves
    rng = np.random.default_rng(41)
    systems = SYSTEMS + [[120, -100, -7, -7, -3, 0], [8, -4, -7, -7, -3, 0], [6, -3, -5, -5, 0, -2],
                         [127, -128, -3, -3, 0, 0], [200, -4, -7, -7, -3, 0]]
    sizes = [(2300, 900), (1030, 2100), (300, 300), (64, 700), (513, 129), (1, 1), (0, 4), (2049, 257),
             (257, 4100), (700, 64), (1200, 1200), (90, 90), (2600, 70), (255, 256), (1024, 1024)]
    t_list, o_list, prm = [], [], []
    for k, (n, m) in enumerate(sizes):
        t, o = _random_problem(rng, n, m, [2, 4, 27, 31][k % 4], k % 2 == 0)
        t_list.append(t); o_list.append(o); prm.append(systems[k % len(systems)])
    batch = tsc.NWBatch(t_list, o_list, prm, two_phase=True)
    assert batch.hints == 0                       # mixed systems: nothing may be assumed
    batch.hints = (31 << _native.TA_NW_ALPHABET_SHIFT) | _native.TA_NW_OPENS_SAME
    shape(batch)
    plan = _phase1_plan(batch)
    assert plan["mode"] == (2 if variant == "profile" else 1)
    assert waves is None or plan["waves"] == waves
    for _ in range(2):
        batch.run()
    for k, got in enumerate(batch.results()):
        want = nw_oracle.align_ids(t_list[k], o_list[k], prm[k])
        assert got.tolist() == want.tolist(), (k, sizes[k], prm[k])
    # launches that qualify as a whole (what the hints are for): one gap open, and two different ones
    for system in ([8, -4, -7, -7, -3, 0], [9, -5, -6, -2, -1, -3]):
        same = tsc.NWBatch(t_list, o_list, system, two_phase=True)
        assert same.hints == (31 << _native.TA_NW_ALPHABET_SHIFT) | \
            (_native.TA_NW_OPENS_SAME if system[2] == system[3] else 0)
        shape(same)
        plan = _phase1_plan(same)
        assert plan["mode"] == (2 if variant == "profile" else 1) and plan["samego"] == int(system[2] == system[3])
        same.run()
        for k, got in enumerate(same.results()):
            want = nw_oracle.align_ids(t_list[k], o_list[k], system)
            assert got.tolist() == want.tolist(), (k, sizes[k], system)


@pytest.mark.gpu
def test_check_ids_flag_refuses_a_false_alphabet_assertion(tsc):
    """TA_NW_ALPHABET(a) / TA_NW_CODES8 are caller assertions the kernel does not check: an id >= a indexes LDS beyond
    the score profile and the alignment is silently wrong.  TA_NW_CHECK_IDS (debug guard, one stream synchronisation)
    verifies them on the device first: a false assertion fails the call with TA_EINVAL (NativeArgumentError here) and
    launches nothing; a true one changes nothing."""
    from oracle import nw_oracle
    from text_alignment_amd import _native
    rng = np.random.default_rng(5)
    probs = [_random_problem(rng, n, m, 27, True) for n, m in [(300, 280), (700, 64), (65, 513)]]
    t_list, o_list = [p[0] for p in probs], [p[1] for p in probs]
    batch = tsc.NWBatch(t_list, o_list, [8, -4, -7, -7, -3, 0], two_phase=True)
    amax = int(max(max(t.max(), o.max()) for t, o in probs)) + 1
    assert batch.hints >> _native.TA_NW_ALPHABET_SHIFT & 0xFF == amax
    batch.check_ids = True
    batch.run()                                               # the wrapper's own hints are true
    for (t, o), got in zip(probs, batch.results()):
        assert got.tolist() == nw_oracle.align_ids(t, o, [8, -4, -7, -7, -3, 0]).tolist()
    batch.ops_len.zero_()
    batch.hints = (batch.hints & ~(0xFF << _native.TA_NW_ALPHABET_SHIFT)) | ((amax - 3) << _native.TA_NW_ALPHABET_SHIFT)
    with pytest.raises(_native.NativeArgumentError, match="token id is outside"):
        batch.run()
    import torch
    torch.cuda.synchronize()
    assert int(batch.ops_len.abs().sum()) == 0                # refused before anything was launched
