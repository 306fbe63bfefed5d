"""Row N3: Gamera-free preprocessing.  The projection / peak numerics -- the product's copy
(text_alignment_amd/textAlignPreprocessing.py, host Python as in the reference) and the checker's
(oracle/preproc_ref.py) -- are pinned to golden vectors captured from the imported reference
(textAlignPreprocessing.py:38-157); the checker's scipy image operations that stand in for Gamera
are exercised on a synthetic page (parity unpinned); the product's image operations are HIP kernels,
compared with the checker in tests/test_preproc_gpu.py."""
import numpy as np
import pytest

from conftest import load_golden


def _numerics(which):
    if which == "product":
        from text_alignment_amd import textAlignPreprocessing as pp
    else:
        from oracle import preproc_ref as pp
    return pp


@pytest.mark.parametrize("which", ["product", "checker"])
def test_peak_numerics_golden(which):
    pp = _numerics(which)
    g = load_golden("preproc.json")
    for c in g["profiles"]:
        data = np.array(c["data"], dtype=float)
        if c["smoothed"] is None:
            assert pp.find_peak_locations(data) == c["peaks"]
            continue
        sm = pp.moving_avg_filter(data, c["filter_size"])
        assert sm.tolist() == c["smoothed"]
        assert pp.find_peak_locations(sm) == c["peaks"]
        assert pp.find_peak_locations(sm, tol=0.5) == c["peaks_tol05"]
        assert [[int(a), float(b)] for a, b in pp.find_peak_locations(sm, ranked=True)] == c["ranked"]
        assert [float(pp.calculate_peak_prominence(sm, i)) for i in range(0, len(sm), 7)] == c["prominence_every7"]
    for c in g["coincide"]:
        assert bool(pp.vertically_coincide(*c["args"])) == c["result"], c


def test_product_peak_finder_equals_the_reference_loop():
    """The product scores only the local maxima of a projection (everything else has prominence 0); the
    checker keeps the reference's loop over every row (textAlignPreprocessing.py:113-144).  Same peaks,
    same ranked prominences, on integer, smoothed, tie-rich and sub-unit (negative log) profiles, for
    tolerances on both sides of 0, and the same exception on empty data."""
    from oracle import preproc_ref as ref
    from text_alignment_amd import textAlignPreprocessing as pp
    rng = np.random.default_rng(1)
    for k in range(200):
        n = int(rng.integers(1, 300))
        kind = k % 5
        if kind == 0:
            d = rng.integers(0, 50, size=n).astype(float)
        elif kind == 1:
            d = ref.moving_avg_filter(rng.integers(0, 400, size=n), int(rng.integers(1, 20)))
        elif kind == 2:
            d = np.round(rng.random(n) * 5)
        elif kind == 3:
            d = rng.random(n) * 0.9
        else:
            d = rng.integers(0, 3, size=n)
        for tol in (0.7, 0.5, 0.0, -0.1):
            assert pp.find_peak_locations(d, tol=tol) == ref.find_peak_locations(d, tol=tol), (k, tol)
            a = [(int(i), float(v)) for i, v in pp.find_peak_locations(d, tol=tol, ranked=True)]
            b = [(int(i), float(v)) for i, v in ref.find_peak_locations(d, tol=tol, ranked=True)]
            assert a == b, (k, tol)
    for fn in (pp.find_peak_locations, ref.find_peak_locations):
        with pytest.raises(ValueError):
            fn(np.zeros(0))


def test_candidate_prominences_three_ways(native):
    """float64 projections take the library's host loops (ta_pp_peak_prominence_args), other dtypes the
    array form; both give, value for value, what calculate_peak_prominence gives row by row"""
    from text_alignment_amd import textAlignPreprocessing as pp
    rng = np.random.default_rng(8)
    for k in range(120):
        n = int(rng.integers(3, 500))
        d = [rng.integers(0, 9, n).astype(np.float64), rng.random(n) * 300,
             pp.moving_avg_filter(rng.integers(0, 1400, n + 40))][k % 3]
        top = d.max()
        cand = [i for i in range(1, n - 1)
                if not (d[i - 1] > d[i] or d[i + 1] > d[i] or (d[i - 1] == d[i] and d[i + 1] == d[i]))]
        want = [float(pp.calculate_peak_prominence(d, i, top)) for i in cand]
        assert [float(v) for v in pp._candidate_prominences_native(d, cand, top)] == want, k
        assert [float(v) for v in pp._candidate_prominences(d, cand, top)] == want, k
    assert pp._candidate_prominences_native(np.zeros(5), [], 0.0) == []
    with pytest.raises(ValueError):
        pp._candidate_prominences_native(np.zeros(5), [7], 0.0)


def _synthetic_page(nlines=6, angle=0.0, seed=0):
    """White page with `nlines` rows of word-like ink blobs (ink density bell-shaped across each
    line, like text); returns (uint8 image, line centres)."""
    from scipy import ndimage
    rng = np.random.default_rng(seed)
    h, w = 200 + 140 * nlines, 1400
    ink = np.zeros((h, w), dtype=bool)
    centres = []
    yy = np.arange(h)[:, None]
    for k in range(nlines):
        cy = 150 + 140 * k
        centres.append(cy)
        dens = 0.55 * np.exp(-0.5 * ((yy - cy) / 8.0) ** 2)
        x = 80
        while x < w - 160:
            ww = int(rng.integers(50, 120))
            ink[:, x:x + ww] |= rng.random((h, ww)) < dens
            x += ww + int(rng.integers(24, 40))
    ink = ndimage.binary_closing(ink, structure=np.ones((3, 3), bool), iterations=2)
    img = np.where(ink, 0, 255).astype(np.uint8)
    if angle:
        img = ndimage.rotate(img, angle, reshape=False, order=1, mode='constant', cval=255).astype(np.uint8)
    img[5:8, 5:8] = 0                                   # a speck the despeckler should remove
    return img, centres


def test_lines_found_on_synthetic_page():
    from oracle import preproc_ref as pp
    img, centres = _synthetic_page(6)
    image_bin, eroded, angle = pp.preprocess_images(img)
    assert abs(angle) <= 0.3
    assert image_bin.dim.ncols >= 1400 and not image_bin.ink[5:8, 5:8].any()
    strips, peaks, smoothed = pp.identify_text_lines(image_bin, eroded)
    # the moving average (61 rows) is wider than a line, so every peak is flat-topped; the
    # reference's duplicate removal skips the last pair (range(len - 2), reference :133-134), so the
    # last line may legitimately appear twice -- the restatement reproduces that
    assert len(peaks) in (6, 7) and len(strips) == len(peaks)
    for s, cy, pk in zip(strips, centres, peaks):
        assert abs(pk - cy) <= 16                 # flat-topped peaks report a corner, not the centre
        assert s.offset_y <= cy <= s.offset_y + s.height
        assert s.pixels.shape == (s.height, s.width) and s.pixels.dtype == np.uint8
        assert (s.pixels == 0).mean() > 0.3             # mostly ink inside a line strip


@pytest.mark.parametrize("skew", [2.0, -3.3, 4.0])
def test_deskew_sign_and_box_unrotation(skew):
    """A page skewed by a known signed angle: the reported angle has the sign that makes
    rotate_bbox(box, -angle, image.dim, raw.dim) -- what process() applies to every syllable box
    (reference alignToOCR.py:327-328) -- land boxes of the deskewed page back on the raw page's ink.
    (With the opposite sign a point 400 px from the centre misses by ~60 px.)  The checker's
    pipeline here; the device pipeline reports the same angles (tests/test_preproc_gpu.py)."""
    from oracle import preproc_ref as pp
    from text_alignment_amd import alignToOCR as atocr, textAlignPreprocessing as product
    from text_alignment_amd.page import Image
    assert product.reported_angle(1.25) == pp.reported_angle(1.25) == -1.25 and product.reported_angle(0) == 0.0
    img, _ = _synthetic_page(5, angle=skew, seed=3)
    image_bin, _, angle = pp.preprocess_images(img)
    assert abs(angle - skew) <= 0.4               # same sense as the scipy rotation that skewed the page
    raw_ink = pp.to_onebit(img)
    raw_dim = Image(img.shape[1], img.shape[0]).dim
    ys, xs = np.nonzero(image_bin.ink)
    pick = np.random.default_rng(1).choice(len(ys), size=300, replace=False)
    boxes = [atocr.CharBox('x', (int(xs[k]), int(ys[k])), (int(xs[k]), int(ys[k]))) for k in pick]
    back = atocr.rotate_bboxes(boxes, -1 * angle, image_bin.dim, raw_dim)
    hits = 0
    for b in back:
        x, y = int(b.ulx), int(b.uly)
        hits += bool(raw_ink[max(y - 2, 0):y + 3, max(x - 2, 0):x + 3].any())
    assert hits >= 0.95 * len(back), hits
    # the single-box form agrees with the vectorised one
    one = atocr.rotate_bbox(boxes[0], -1 * angle, image_bin.dim, raw_dim)
    assert (one.ulx, one.uly) == (back[0].ulx, back[0].uly)


def test_prepared_page_passes_through():
    from text_alignment_amd import textAlignPreprocessing as pp
    from text_alignment_amd.page import PreparedPage
    pg = PreparedPage((100, 80), (100, 80), 1.5, [], [10, 40])
    image, eroded, angle = pp.preprocess_images(pg)
    assert angle == 1.5 and pp.identify_text_lines(image, eroded) == ([], [10, 40], None)


@pytest.mark.gpu
def test_process_from_raw_image():
    """process() from a raw text-layer array: preprocessing -> line normaliser -> HIP recogniser
    -> HIP aligner -> JSON (shape checks only: there is no oracle for this path)."""
    from text_alignment_amd import alignToOCR as atocr, ocr
    from text_alignment_amd.page import Image

    class Raw(object):                                   # what callers hand to process(): pixels + dim
        def __init__(self, px):
            self.pixels = px
            self.dim = Image(px.shape[1], px.shape[0]).dim
    img, centres = _synthetic_page(4)
    model = ocr.LineModel.random(5, no=30)
    model.W2[0, 0] += 4.0
    res = atocr.process(Raw(img), "dominus dixit ad me filius meus es tu", model,
                        seq_align_params=[8, -1, -9, -9, -4, -4])
    assert res is not None
    syl_boxes, image, peaks, all_chars = res
    js = atocr.to_JSON_dict(syl_boxes, peaks)
    assert len(peaks) in (4, 5) and js["median_line_spacing"] > 100
    for b in js["syl_boxes"]:
        assert 0 <= b["ul"][0] <= b["lr"][0] <= image.dim.ncols


@pytest.mark.gpu
def test_process_batch_from_raw_images():
    """process_batch on raw page arrays (`parallel` is accepted and unused): per page the same boxes
    as process() on that page alone."""
    from text_alignment_amd import alignToOCR as atocr, ocr
    from text_alignment_amd.page import Image

    class Raw(object):
        def __init__(self, px):
            self.pixels = px
            self.dim = Image(px.shape[1], px.shape[0]).dim
    model = ocr.LineModel.random(5, no=30)
    model.W2[0, 0] += 4.0
    rec = ocr.LineRecognizer(model)
    params = [8, -1, -9, -9, -4, -4]
    pages = [Raw(_synthetic_page(4, seed=k)[0]) for k in range(3)]
    trs = ["dominus dixit ad me filius meus es tu", "ego hodie genui te alleluia", "quare fremuerunt gentes"]
    batch = atocr.process_batch(pages, trs, rec, params, parallel=2)
    for pg, tr, got in zip(pages, trs, batch):
        alone = atocr.process(pg, tr, rec, seq_align_params=params, verbose=False)
        assert atocr.to_JSON_dict(got[0], got[2]) == atocr.to_JSON_dict(alone[0], alone[2])


@pytest.mark.gpu
def test_bare_arrays_are_pages_too():
    """process / process_batch / sharding.process_pages on PLAIN numpy arrays (no .pixels, no .dim):
    the raw page's size is its shape.  A page of a type nobody can read fails before any GPU work."""
    from text_alignment_amd import alignToOCR as atocr, ocr, sharding
    model = ocr.LineModel.random(5, no=30)
    model.W2[0, 0] += 4.0
    rec = ocr.LineRecognizer(model)
    params = [8, -1, -9, -9, -4, -4]
    pages = [_synthetic_page(4, seed=k)[0] for k in range(2)]
    trs = ["dominus dixit ad me filius meus es tu", "ego hodie genui te alleluia"]
    alone = [atocr.process(pg, tr, rec, seq_align_params=params, verbose=False) for pg, tr in zip(pages, trs)]
    want = [atocr.to_JSON_dict(a[0], a[2]) for a in alone]
    assert all(len(w["syl_boxes"]) > 0 for w in want)
    batch = atocr.process_batch(pages, trs, rec, params)
    assert [atocr.to_JSON_dict(b[0], b[2]) for b in batch] == want
    out = sharding.process_pages(pages, trs, rec, seq_align_params=params)
    assert [out[k] for k in range(2)] == want
    with pytest.raises(TypeError):
        atocr.process_batch([np.zeros(7)], ["dominus"], rec, params)


def test_line_boxes_equal_the_per_line_loop():
    """textAlignPreprocessing.line_boxes (one peaks x components table) against the reference's loop (:253-276): for
    each peak the components that vertically_coincide with it, their union, lines without any left out -- on seeded
    component sets with components straddling strips, touching strip edges, far outside, and with odd / fractional
    median heights (the strip's half height is int(collision / 2))."""
    from text_alignment_amd import textAlignPreprocessing as pp
    rng = np.random.default_rng(77)
    assert pp.line_boxes([], np.zeros((0, 4)), 10.0) == []
    assert pp.line_boxes([5, 9], np.zeros((0, 4)), 10.0) == []
    for case in range(60):
        ncomp, npeaks = int(rng.integers(1, 120)), int(rng.integers(1, 40))
        uly = rng.integers(0, 1500, ncomp)
        ulx = rng.integers(0, 4000, ncomp)
        comps = np.stack([ulx, uly, ulx + rng.integers(0, 60, ncomp), uly + rng.integers(0, 50, ncomp)], axis=1)
        peaks = sorted(set(rng.integers(0, 1600, npeaks).tolist()))
        collision = [np.float64(np.median(comps[:, 3] - comps[:, 1] + 1)), 7.0, 1.0, 2.5, 33.5][case % 5]
        want = []
        for loc in peaks:
            hit = [c for c in comps.tolist() if pp.vertically_coincide(loc, c[1], c[3] - c[1] + 1, collision)]
            if hit:
                want.append([min(c[0] for c in hit), min(c[1] for c in hit), max(c[2] for c in hit), max(c[3] for c in hit)])
        assert pp.line_boxes(peaks, comps, collision) == want, case
        assert pp.line_boxes_numpy(peaks, comps, collision) == want, case


def test_host_arithmetic_of_the_stages_equals_numpy():
    """The library's host loops between the preprocessing stages against the numpy expressions they replace: Otsu
    thresholds (preproc_gpu.otsu_from_histogram) and the skew sweep's choice -- np.var per histogram row TO THE LAST
    BIT (numpy's pairwise summation restated in csrc/ta_common.cpp) and its argmax -- on flat, two-peaked, empty and
    single-bin histograms and on rows of 1 .. 1400 counts."""
    import ctypes
    from text_alignment_amd import _native, preproc_gpu as pg
    rng = np.random.default_rng(5)
    hists = []
    for trial in range(400):
        kind = trial % 5
        if kind == 0:
            h = rng.integers(0, 70000, 256)
        elif kind == 1:
            h = np.zeros(256, np.int64)
            h[rng.integers(0, 256, 3)] = rng.integers(1, 6000000, 3)
        elif kind == 2:
            x = np.clip(np.concatenate([rng.normal(60, 20, 3000), rng.normal(200, 15, 58000)]), 0, 255).astype(np.uint8)
            h = np.bincount(x, minlength=256) * 100
        elif kind == 3:
            h = np.zeros(256, np.int64)
        else:
            h = rng.integers(0, 3, 256) * rng.integers(0, 2000000, 256)
        hists.append(h.astype(np.int32))
    assert pg.otsu_thresholds(np.stack(hists)).tolist() == [pg.otsu_from_histogram(h) for h in hists]
    pages, nang, hs = [], [], []
    for trial in range(300):
        a_, h_ = int(rng.integers(1, 60)), int(rng.integers(1, 1400)) if trial % 7 else int(rng.integers(1, 140))
        kind = trial % 4
        if kind == 0:
            H = rng.integers(0, 1400, (a_, h_))
        elif kind == 1:
            H = rng.poisson(3.0, (a_, h_))
        elif kind == 2:
            base = rng.integers(0, 900, h_)
            H = np.stack([np.roll(base, int(sft)) for sft in rng.integers(0, 3, a_)])        # (ties between rows)
        else:
            H = np.zeros((a_, h_), np.int64)
            H[:, rng.integers(0, h_)] = rng.integers(0, 5)
        pages.append(H.astype(np.int32).ravel()); nang.append(a_); hs.append(h_)
    offs = np.concatenate(([0], np.cumsum([len(p) for p in pages])))[:-1]
    flat = np.concatenate(pages)
    best, some = pg.sharpest_rows(flat, offs, nang, hs)
    want_best, want_some = pg._sharpest_rows_numpy(flat, offs, nang, hs)
    assert best.tolist() == want_best.tolist() and some.tolist() == want_some.tolist()
    # the variances themselves, bit for bit
    var = np.zeros(sum(nang))
    b_, s_ = np.zeros(len(pages), np.int32), np.zeros(len(pages), np.uint8)
    o64, a32, h32 = offs.astype(np.int64), np.array(nang, np.int32), np.array(hs, np.int32)
    assert _native.lib.ta_host_sharpest_rows(flat.ctypes.data, o64.ctypes.data, a32.ctypes.data, h32.ctypes.data, len(pages),
                                             b_.ctypes.data, s_.ctypes.data, var.ctypes.data) == 0
    want = np.concatenate([np.var(p.reshape(a_, h_), axis=1) for p, a_, h_ in zip(pages, nang, hs)])
    assert var.view(np.uint64).tolist() == want.view(np.uint64).tolist()


def test_batched_peaks_equal_the_per_page_functions():
    """textAlignPreprocessing.peaks_of_projections (host loops of the library around one np.log) against
    moving_avg_filter + find_peak_locations + the white rows between peaks page by page: smoothed projections bit for
    bit, the same peaks, the same rows -- on the golden profiles' raw data, seeded text-like projections, flat-topped
    peaks (equal prominences), short pages (no room for the filter) and constant pages."""
    from text_alignment_amd import textAlignPreprocessing as pp
    rng = np.random.default_rng(11)
    pages = []
    for c in load_golden("preproc.json")["profiles"]:
        pages.append(np.array(c["data"], dtype=float).astype(np.int64))
    for trial in range(120):
        m = int(rng.integers(1, 5000)) if trial % 6 else int(rng.integers(1, 80))
        kind = trial % 4
        if kind == 0:                                     # text lines: humps on a noisy floor
            y = rng.integers(0, 8, m).astype(np.int64)
            for c0 in range(int(rng.integers(20, 90)), m, int(rng.integers(90, 160))):
                w = int(rng.integers(20, 60))
                y[c0:c0 + w] += int(rng.integers(200, 900))
        elif kind == 1:                                   # plateaus: flat-topped peaks, equal prominences
            y = np.repeat(rng.integers(0, 4, m // 50 + 1) * 300, 50)[:m].astype(np.int64)
        elif kind == 2:
            y = rng.integers(0, 1400, m).astype(np.int64)
        else:
            y = np.full(m, int(rng.integers(0, 3)), np.int64)
        pages.append(y)
    lens = [len(p) for p in pages]
    offs = (np.concatenate(([0], np.cumsum([(m + 63) // 64 * 64 for m in lens])))[:-1]).astype(np.int64)
    flat = np.zeros(int(offs[-1]) + lens[-1] + 64, np.int64)
    for o, p in zip(offs, pages):
        flat[o:o + len(p)] = p
    got = pp.peaks_of_projections(flat, offs, lens)
    want = pp.peaks_of_projections_numpy(flat, offs, lens)
    assert sum(len(w[1]) for w in want) > 500
    for k, (g, w) in enumerate(zip(got, want)):
        assert g[0].view(np.uint64).tolist() == w[0].view(np.uint64).tolist(), k
        assert g[1] == w[1], k
        assert g[2].tolist() == w[2].tolist(), k
    for tol in (0.0, 0.3, 0.95):
        got = pp.peaks_of_projections(flat, offs[:40], lens[:40], tol=tol)
        want = pp.peaks_of_projections_numpy(flat, offs[:40], lens[:40], tol=tol)
        assert [g[1] for g in got] == [w[1] for w in want] and [g[2].tolist() for g in got] == [w[2].tolist() for w in want]
