"""Device page preprocessing (csrc/ta_preproc.hip, preproc_gpu.py) against the checker
oracle/preproc_ref.py (numpy / scipy.ndimage): same components, same skew angle, same rotated and
filtered bits, same line strips.  (Gamera is absent: the checker is a parity-unpinned restatement of
reference textAlignPreprocessing.py:160-285.)"""
import numpy as np
import pytest

from test_preprocessing import _synthetic_page

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _noisy_page(seed, nlines=6, angle=0.0):
    img, centres = _synthetic_page(nlines, angle=angle, seed=seed)
    rng = np.random.default_rng(seed + 100)
    img = img.copy()
    ys, xs = rng.integers(0, img.shape[0], 400), rng.integers(0, img.shape[1], 400)
    img[ys, xs] = 0                                         # specks
    img[200:420, 30:36] = 0                                 # a tall stroke (taller than 150 rows)
    grey = np.where(img == 0, rng.integers(0, 80, img.shape), rng.integers(180, 256, img.shape)).astype(np.uint8)
    return grey


def test_labels_match_scipy():
    from scipy import ndimage
    from text_alignment_amd import preproc_gpu as G
    rng = np.random.default_rng(1)
    d = G._Dev()
    for shape, dens in [((64, 80), 0.5), ((200, 333), 0.3), ((37, 1000), 0.62), ((1, 50), 0.5), ((300, 300), 0.45)]:
        a = rng.random(shape) < dens
        if shape == (300, 300):                              # a spiral: many propagation rounds
            a[:] = False
            for k in range(0, 140, 4):
                a[k, k:300 - k] = True; a[k:300 - k, 299 - k] = True
                a[299 - k, k + 2:300 - k] = True; a[k + 4:300 - k, k + 2] = True
        ink = torch.from_numpy(a.astype(np.uint8)).cuda()
        lab, stats = d.label(ink)
        lab = lab.cpu().numpy()
        want, n = ndimage.label(a, structure=np.ones((3, 3), bool))
        assert ((lab >= 0) == a).all()
        # same partition: our label is the raster-first pixel of the component
        first = np.full(n + 1, -1, np.int64)
        flat = want.ravel()
        idx = np.arange(flat.size)
        for k in range(flat.size - 1, -1, -1):
            first[flat[k]] = idx[k]
        assert np.array_equal(lab.ravel()[flat > 0], first[flat[flat > 0]])
        recs = d.components(torch.from_numpy(lab).cuda(), stats)
        assert len(recs) == n
        area = np.bincount(flat, minlength=n + 1)[1:]
        assert sorted(recs[:, 1].tolist()) == sorted(area.tolist())
        objs = ndimage.find_objects(want)
        boxes = sorted((sl[1].start, sl[0].start, sl[1].stop - 1, sl[0].stop - 1) for sl in objs)
        assert sorted(map(tuple, recs[:, 2:6].tolist())) == boxes


def test_labels_across_tile_borders():
    """The stitching pass links a border pixel only where no other link implies the connection (straight
    neighbour first, diagonals only beside background, first pixel of a run only): patterns whose components
    touch ONLY through tile corners / diagonals / long runs along tile borders (tiles are 16 x 64), and pages
    that are nearly all ink (one component across hundreds of tiles), against scipy's partition."""
    from scipy import ndimage
    from text_alignment_amd import preproc_gpu as G
    rng = np.random.default_rng(12)
    d = G._Dev()
    yy, xx = np.mgrid[0:130, 0:260]
    pats = [(yy + xx) % 2 == 0,                                   # checkerboard: diagonal links only
            (yy - xx) % 7 == 0, (yy + xx) % 5 == 0,               # diagonal stripes, both directions
            (yy % 16 == 15) | (xx % 64 == 63),                    # the tile borders themselves (one side)
            (yy % 16 == 0) | (xx % 64 == 0),                      # ... the other side
            ((yy % 16 == 15) & (xx % 3 == 0)) | ((yy % 16 == 0) & (xx % 3 == 1)),   # only diagonal contacts across rows 15/16
            ((xx % 64 == 63) & (yy % 2 == 0)) | ((xx % 64 == 0) & (yy % 2 == 1)),   # ... across columns 63/64
            rng.random((130, 260)) < 0.92, rng.random((130, 260)) < 0.08]
    pats += [rng.random((530, 700)) < q for q in (0.97, 0.55, 0.4)]
    for a in pats:
        ink = torch.from_numpy(np.ascontiguousarray(a).astype(np.uint8)).cuda()
        lab = d.label(ink)[0].cpu().numpy()
        want, n = ndimage.label(a, structure=np.ones((3, 3), bool))
        assert ((lab >= 0) == a).all()
        flat = want.ravel()
        first = np.full(n + 1, flat.size, np.int64)
        np.minimum.at(first, flat, np.arange(flat.size))
        assert np.array_equal(lab.ravel()[flat > 0], first[flat[flat > 0]])


def test_angle_histograms_from_the_point_list_equal_those_from_the_page():
    """ta_pp_ink_points + ta_pp_angle_histograms_points against ta_pp_angle_histograms (which walks the whole
    decimated page per angle): same counts for every angle and row, decimated and not, and an empty page"""
    from text_alignment_amd import _native
    rng = np.random.default_rng(6)
    lib = _native.lib
    for (h, w), step, dens in [((700, 1100), 1, 0.1), ((2400, 1500), 2, 0.07), ((300, 200), 3, 0.5), ((64, 64), 1, 0.0)]:
        ink = torch.from_numpy((rng.random((h, w)) < dens).astype(np.uint8)).cuda()
        hs, ws = (h + step - 1) // step, (w + step - 1) // step
        ang = np.deg2rad(np.arange(-6, 6.01, 0.75))
        cs = np.empty(2 * len(ang)); cs[0::2], cs[1::2] = np.cos(ang), np.sin(ang)
        d_cs = torch.from_numpy(cs).cuda()
        a = torch.empty((len(ang), hs), dtype=torch.int32, device="cuda")
        b = torch.empty_like(a)
        pts = torch.empty(hs * ws, dtype=torch.int32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        _native.check(lib.ta_pp_angle_histograms(ink.data_ptr(), h, w, step, d_cs.data_ptr(), len(ang), a.data_ptr(), None), "a")
        _native.check(lib.ta_pp_ink_points(ink.data_ptr(), h, w, step, pts.data_ptr(), cnt.data_ptr(), None), "p")
        _native.check(lib.ta_pp_angle_histograms_points(pts.data_ptr(), cnt.data_ptr(), hs, ws, d_cs.data_ptr(), len(ang),
                                                        b.data_ptr(), None), "b")
        assert int(cnt.item()) == int(ink[::step, ::step].sum().item())
        assert torch.equal(a, b), ((h, w), step)
    # a decimated grid that does not fit 16-bit point coordinates is refused, not wrapped
    tall = torch.zeros((70000, 2), dtype=torch.uint8, device="cuda")
    rc = lib.ta_pp_ink_points(tall.data_ptr(), 70000, 2, 1, pts.data_ptr(), cnt.data_ptr(), None)
    assert rc == _native.TA_ELIMIT and b"16-bit" in lib.ta_last_error()


def test_strip_cutter():
    """ta_pp_cut_strips: boxes (inclusive corners, touching the page edges too) out of an ink plane into one
    packed buffer, ink 0 on 255"""
    import ctypes
    from text_alignment_amd import _native
    rng = np.random.default_rng(4)
    ink = (rng.random((300, 500)) < 0.4).astype(np.uint8)
    boxes, total = [], 0
    for ulx, uly, lrx, lry in [(0, 0, 499, 0), (0, 0, 0, 299), (10, 20, 400, 90), (499, 299, 499, 299), (3, 250, 77, 299)]:
        boxes.append((ulx, uly, lrx, lry, total))
        total += (lry - uly + 1) * (lrx - ulx + 1)
    d_ink = torch.from_numpy(ink).cuda()
    d_boxes = torch.tensor(boxes, dtype=torch.int64).cuda()
    out = torch.zeros(total, dtype=torch.uint8, device="cuda")
    _native.check(_native.lib.ta_pp_cut_strips(d_ink.data_ptr(), 300, 500, d_boxes.data_ptr(), len(boxes),
                                               out.data_ptr(), None), "ta_pp_cut_strips")
    got = out.cpu().numpy()
    for ulx, uly, lrx, lry, off in boxes:
        want = np.where(ink[uly:lry + 1, ulx:lrx + 1] > 0, 0, 255).astype(np.uint8)
        assert np.array_equal(got[off:off + want.size].reshape(want.shape), want)
    assert _native.lib.ta_pp_cut_strips(d_ink.data_ptr(), 300, 500, None, 0, None, None) == 0     # nothing to cut


def test_pages_without_lines_in_a_batch():
    """a blank page and a page of specks between two text pages: no strips, no peaks, and the text pages'
    results are what they are alone"""
    from text_alignment_amd import preproc_gpu as G
    a, b = _noisy_page(2), _noisy_page(9)
    blank = np.full_like(a, 255)
    specks = blank.copy()
    specks[::97, ::89] = 0
    got = G.find_lines_batch([a, blank, specks, b])
    assert [len(g[3]) for g in got[1:3]] == [0, 0]
    for alone, both in ((G.find_lines(a), got[0]), (G.find_lines(b), got[3])):
        assert alone[2] == both[2] and list(alone[4]) == list(both[4]) and len(alone[3]) == len(both[3]) >= 4
        assert all(np.array_equal(x.pixels, y.pixels) for x, y in zip(alone[3], both[3]))


@pytest.mark.parametrize("seed,angle", [(0, 0.0), (3, 2.0), (5, -3.3)])
def test_preprocess_and_lines_match_host(seed, angle):
    from oracle import preproc_ref as H
    from text_alignment_amd import preproc_gpu as G
    grey = _noisy_page(seed, angle=angle)
    b0, e0, a0, s0, p0 = H.find_lines(grey)
    b1, e1, a1, s1, p1 = G.find_lines(grey)
    assert a0 == a1
    assert np.array_equal(b0.ink, b1.ink)
    assert np.array_equal(e0.ink, e1.ink)
    assert list(p0) == list(p1)
    assert len(s0) == len(s1) and len(s0) >= 4
    for x, y in zip(s0, s1):
        assert (x.offset_x, x.offset_y, x.height, x.width) == (y.offset_x, y.offset_y, y.height, y.width)
        assert np.array_equal(x.pixels, y.pixels)


def test_colour_float_and_onebit_pages_take_the_device_path():
    """find_lines_all reduces colour / float / already-binarised (bool) pages to uint8 greyscale and
    sends them through the device kernels: same strips as the checker on the original array (whose
    to_onebit takes a bool page as it is)"""
    from oracle import preproc_ref as H
    from text_alignment_amd import alignToOCR as atocr
    grey = _noisy_page(7)
    rgb = np.stack([grey, grey, grey], axis=2)
    flt = grey.astype(np.float32) / 255.0
    onebit = H.to_onebit(grey)
    for page in (rgb, flt, onebit):
        want = H.find_lines(page)
        got = atocr.find_lines_all([page])[0]
        assert want[2] == got[2] and np.array_equal(want[0].ink, got[0].ink)
        assert [s.pixels.shape for s in want[3]] == [s.pixels.shape for s in got[3]]
        assert all(np.array_equal(a.pixels, b.pixels) for a, b in zip(want[3], got[3]))


def test_stage_calls_on_pages_of_different_sizes_and_parameters():
    """One batch of pages of three different sizes through the library's stage calls (ta_pp_*_batch): each page's
    results are what it gives alone and what the checker gives; and the run-filter parameters the stage call takes
    (two rounds, runs of three pixels, no filter at all, no rotation) against the checker's preprocess_images."""
    from oracle import preproc_ref as H
    from text_alignment_amd import preproc_gpu as G
    pages = [_noisy_page(2), np.ascontiguousarray(_noisy_page(4, angle=1.5)[40:, :-60]),
             np.ascontiguousarray(_noisy_page(6, angle=-2.0)[:-80, 25:])]
    assert len({p.shape for p in pages}) == 3
    got = G.find_lines_batch(pages)
    for page, g in zip(pages, got):
        want = H.find_lines(page)
        assert want[2] == g[2] and np.array_equal(want[0].ink, g[0].ink) and np.array_equal(want[1].ink, g[1].ink)
        assert list(want[4]) == list(g[4]) and len(want[3]) == len(g[3]) >= 3
        for x, y in zip(want[3], g[3]):
            assert (x.offset_x, x.offset_y, x.height, x.width) == (y.offset_x, y.offset_y, y.height, y.width)
            assert np.array_equal(x.pixels, y.pixels)
    for kw in (dict(filter_runs=2, filter_runs_amt=2), dict(filter_runs=1, filter_runs_amt=3), dict(filter_runs=0),
               dict(filter_runs=1, filter_runs_amt=1), dict(correct_rotation=False)):
        d, out = G.preprocess_images_batch(pages[1:], **kw)
        for page, (ink, eroded, angle) in zip(pages[1:], out):
            b0, e0, a0 = H.preprocess_images(page, **kw)
            assert a0 == angle, kw
            assert np.array_equal(b0.ink, ink.cpu().numpy().astype(bool)), kw
            assert np.array_equal(e0.ink, eroded.cpu().numpy().astype(bool)), kw


def test_pages_that_are_torch_tensors_are_taken_where_they_lie():
    """A page handed over as a 2-D uint8 torch tensor -- on the device, or in page-locked host memory -- goes through
    the same stages without the staging copy: same angle, planes, peaks and strips as the numpy page, alone, mixed in
    one batch, and through find_lines_all / process_batch's page type checks."""
    from text_alignment_amd import alignToOCR as atocr, preproc_gpu as G
    a, b = _noisy_page(3, angle=1.0), _noisy_page(8)
    want = G.find_lines_batch([a, b])
    on_dev, pinned = torch.from_numpy(b).cuda(), torch.from_numpy(a).pin_memory()
    for got in (G.find_lines_batch([pinned, on_dev]), atocr.find_lines_all([a, on_dev]), atocr.find_lines_all([pinned, b])):
        for w, g in zip(want, got):
            assert w[2] == g[2] and list(w[4]) == list(g[4]) and len(w[3]) == len(g[3]) >= 4
            assert torch.equal(w[0].plane, g[0].plane) and torch.equal(w[1].plane, g[1].plane)
            assert all(np.array_equal(x.pixels, y.pixels) for x, y in zip(w[3], g[3]))
    assert atocr._raw_dim(on_dev).ncols == b.shape[1] and atocr._raw_dim(on_dev).nrows == b.shape[0]
    with pytest.raises(TypeError):
        G.find_lines_batch([on_dev.float()])


def test_byte_plane_kernels_on_aligned_and_unaligned_planes():
    """ta_pp_histogram / ta_pp_threshold / ta_pp_invert walk a plane 16 bytes per lane where it starts on a 16-byte
    boundary and byte by byte where it does not (a caller's view) and over the last n % 16 bytes: same results as
    numpy for both, for sizes around the vector width, for non-0/1 bytes under the inversion."""
    from text_alignment_amd import _native
    lib = _native.lib
    rng = np.random.default_rng(4)
    st = torch.cuda.current_stream().cuda_stream
    base = torch.from_numpy(rng.integers(0, 256, 300000 + 64).astype(np.uint8)).cuda()
    for off in (0, 16, 3, 7):
        for n in (0, 1, 15, 16, 17, 4099, 300000):
            src = base[off:off + n]
            host = src.cpu().numpy()
            hist = torch.full((256,), 7, dtype=torch.int32, device="cuda")
            _native.check(lib.ta_pp_histogram(src.data_ptr() if n else base.data_ptr(), n, hist.data_ptr(), st), "hist")
            assert np.array_equal(hist.cpu().numpy(), np.bincount(host, minlength=256)), (off, n)
            for invert in (0, 1):
                for out_off in (0, 5):
                    outb = torch.full((n + 37,), 9, dtype=torch.uint8, device="cuda")
                    out = outb[out_off:out_off + n]
                    _native.check(lib.ta_pp_threshold(src.data_ptr() if n else base.data_ptr(), n, 131, invert,
                                                      out.data_ptr() if n else outb.data_ptr(), st), "thr")
                    want = (host <= 131) != bool(invert)
                    assert np.array_equal(out.cpu().numpy(), want.astype(np.uint8)), (off, n, invert, out_off)
                    assert int(outb[out_off + n:].min()) == 9 and (out_off == 0 or int(outb[:out_off].min()) == 9)
            work = src.clone() if off == 0 else base.clone()[off:off + n]      # (a clone of the whole keeps the misalignment)
            _native.check(lib.ta_pp_invert(work.data_ptr() if n else base.data_ptr(), n, st), "invert")
            assert np.array_equal(work.cpu().numpy(), (host == 0).astype(np.uint8)), (off, n)


def test_components_over_runs_equal_the_per_pixel_labelling(monkeypatch):
    """The two stages that need connected components find them over RUNS (csrc/ta_preproc.hip, pp_runs_*); the
    per-pixel labelling of rounds 3-5 stays behind a flag of the stage calls.  Same cleaned planes, angles, peaks,
    strips and -- record for record -- component tables from both, on noisy text pages (specks, holes, a tall stroke),
    a blank page, a page of specks, a checkerboard (the most runs a page can have), single-pixel-wide strokes that touch
    only diagonally, and pages whose widths are not multiples of 64."""
    from text_alignment_amd import _native, preproc_gpu as G
    rng = np.random.default_rng(12)
    a, b = _noisy_page(21, angle=0.7), np.ascontiguousarray(_noisy_page(22)[:, :-37])
    blank = np.full((300, 200), 255, np.uint8)
    checker = np.where((np.add.outer(np.arange(160), np.arange(131)) % 2) == 0, 0, 255).astype(np.uint8)
    diag = np.full((400, 333), 255, np.uint8)
    for k in range(300):
        diag[20 + k, 10 + k] = 0                                  # one 8-connected diagonal: 300 pixels, 300 runs
        diag[20 + k, 320 - k] = 0
    diag[200:390:3, 100:250] = 0                                  # stripes
    holes = np.full((500, 400), 255, np.uint8)
    holes[50:450, 50:350] = 0
    for _ in range(200):
        y, x = int(rng.integers(60, 430)), int(rng.integers(60, 330))
        s_ = int(rng.integers(1, 14))
        holes[y:y + s_, x:x + s_] = 255                          # holes of 1 .. 169 pixels: some filled, some not
    pages = [a, b, blank, checker, diag, holes]

    def run(flags):
        monkeypatch.setattr(G, "LABEL_FLAGS", flags)
        d, pre = G.preprocess_images_batch(pages)
        planes = [(i.cpu().numpy(), e.cpu().numpy(), ang) for i, e, ang in pre]
        lines = G.identify_text_lines_batch(d, [(i, e) for i, e, _ in pre], row_sums=d.row_sums)
        return planes, [([(s.offset_x, s.offset_y, s.height, s.width) for s in st], [s.pixels for s in st], list(pk))
                        for st, pk, _ in lines]
    runs, pixels = run(0), run(_native.TA_PP_LABEL_PIXELS)
    for k, (r, p) in enumerate(zip(runs[0], pixels[0])):
        assert r[2] == p[2], k
        assert np.array_equal(r[0], p[0]) and np.array_equal(r[1], p[1]), k
    for k, (r, p) in enumerate(zip(runs[1], pixels[1])):
        assert r[0] == p[0] and r[2] == p[2], k
        assert all(np.array_equal(x, y) for x, y in zip(r[1], p[1])), k
    assert len(runs[1][0][0]) >= 4 and len(runs[1][1][0]) >= 4
    # the component tables themselves: root pixel, area, box of every component of a labelled plane
    d = G._Dev()
    for page in pages:
        plane = torch.from_numpy((page < 128).astype(np.uint8)).cuda()
        h, w = plane.shape
        tables = []
        for flags in (0, _native.TA_PP_LABEL_PIXELS):
            work = torch.empty_like(plane)
            lab = torch.empty(h * w, dtype=torch.int32, device="cuda")
            stats = torch.empty(5 * h * w, dtype=torch.int32, device="cuda")
            table, recs, counts = d.component_buffer(1, 1 << 16)
            ptr = lambda t: np.array([t.data_ptr()], dtype=np.uint64)
            hh, ww, nrows = np.array([h], np.int32), np.array([w], np.int32), np.array([0], np.int32)
            pe, pw, pl, ps, pr = ptr(plane), ptr(work), ptr(lab), ptr(stats), np.zeros(1, np.uint64)
            _native.check(_native.lib.ta_pp_line_components_batch(1, pe.ctypes.data, hh.ctypes.data, ww.ctypes.data, pr.ctypes.data,
                                                                  nrows.ctypes.data, pw.ctypes.data, pl.ctypes.data, ps.ctypes.data,
                                                                  recs.data_ptr(), 1 << 16, counts.data_ptr(), flags, d.stream), "stage")
            tables.append(d.component_tables(1, None, table, 1 << 16)[0])
        assert tables[0].shape == tables[1].shape and np.array_equal(tables[0], tables[1]), page.shape
