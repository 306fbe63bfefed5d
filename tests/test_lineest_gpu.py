"""Device line normaliser (csrc/ta_lineest.hip) against the checker oracle/lineest_ref.py (scipy.ndimage
in float64; ocropy itself is absent, so the checker is a parity-unpinned
restatement): identical centre line and band height, resampled input rows equal to float32 rounding."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _strip(rng, h, w, wobble=0.0):
    """word-like ink blobs around a (possibly curved) baseline, grey-level antialiasing"""
    yy = np.arange(h)[:, None]
    base = h / 2.0 + wobble * np.sin(np.arange(w) / 97.0)[None, :]
    dens = 0.6 * np.exp(-0.5 * ((yy - base) / (h / 7.0)) ** 2)
    ink = rng.random((h, w)) < dens
    gaps = np.zeros(w, bool)
    x = int(rng.integers(5, 40))
    while x < w:
        g = int(rng.integers(8, 30))
        gaps[x:x + g] = True
        x += g + int(rng.integers(40, 120))
    ink[:, gaps] = False
    grey = np.where(ink, rng.integers(0, 90, size=(h, w)), rng.integers(235, 256, size=(h, w)))
    return grey.astype(np.uint8)


def test_device_normaliser_matches_host():
    assert torch.cuda.is_available()
    from oracle import lineest_ref as lineest
    from text_alignment_amd import lineest_gpu
    rng = np.random.default_rng(5)
    shapes = [(44, 1216), (61, 900), (70, 1500), (33, 300), (96, 700), (20, 120), (52, 2000), (45, 64)]
    strips = [_strip(rng, h, w, wobble=(3.0 if k % 2 else 0.0)) for k, (h, w) in enumerate(shapes)]
    strips.append(np.where(strips[0] < 128, 0, 255).astype(np.uint8))        # bilevel, as the page cutter saves them
    x, T, dbg = lineest_gpu.normalize_strips(strips, want_debug=True)
    x = x.cpu().numpy()
    row = 0
    for k, s in enumerate(strips):
        norm = lineest.CenterNormalizer()
        want = lineest.prepare_raw_strip(s, norm)
        assert np.array_equal(dbg["center"][k], norm.center), k
        assert int(dbg["r"][k]) == norm.r, k
        assert T[k] == want.shape[0], (k, T[k], want.shape)
        got = x[row:row + T[k]]
        row += int(T[k])
        assert got.shape == want.shape
        assert float(np.abs(got - want.astype(np.float32)).max()) <= 2e-6, k
    assert row == x.shape[0]


def test_device_normaliser_rejects_what_the_checker_rejects():
    from text_alignment_amd import lineest_gpu
    with pytest.raises(ValueError):
        lineest_gpu.normalize_strips([np.full((30, 100), 255, np.uint8)])
    with pytest.raises(TypeError):
        lineest_gpu.normalize_strips([np.zeros((30, 100), np.float32)])
    x, T, _ = lineest_gpu.normalize_strips([])
    assert x.shape == (0, 48) and len(T) == 0


def test_strips_already_on_the_device_are_normalised_where_they_are():
    """The device preprocessing leaves its strips on the GPU (Strip.device_pixels): the normaliser takes
    them there, alone or mixed with host strips, bit for bit as from host arrays; a constant one is
    refused (after the measuring pass, which finds its minimum and maximum anyway); `.pixels` downloads."""
    from text_alignment_amd import lineest_gpu, page as page_mod
    rng = np.random.default_rng(21)
    strips = [_strip(rng, h, w) for h, w in [(44, 600), (50, 420), (61, 800), (38, 256)]]
    on_dev = [torch.from_numpy(s).cuda() for s in strips]
    x0, T0, _ = lineest_gpu.normalize_strips(strips)
    for mix in (on_dev, [on_dev[0], strips[1], on_dev[2], strips[3]]):
        x1, T1, _ = lineest_gpu.normalize_strips(mix)
        assert np.array_equal(T0, T1) and torch.equal(x0, x1)
    with pytest.raises(ValueError, match="empty or constant"):
        lineest_gpu.normalize_strips([on_dev[0], torch.full((30, 100), 255, dtype=torch.uint8, device="cuda")])
    with pytest.raises(TypeError):
        lineest_gpu.normalize_strips([on_dev[0].t()])                       # not contiguous
    st = page_mod.Strip(3, 4, 44, device_pixels=on_dev[0])
    assert st.width == 600 and page_mod.prepared_line(st)[0] is on_dev[0]
    assert np.array_equal(st.pixels, strips[0])
    # DeviceStrips: pieces of packed buffers, as the page preprocessing leaves them -- neighbours in one buffer (taken
    # as ONE slice), a gap between two, a second buffer, a tensor and a host strip in between
    sizes = [s.size for s in strips]
    buf_a = torch.cat([on_dev[0].reshape(-1), on_dev[1].reshape(-1), torch.zeros(100, dtype=torch.uint8, device="cuda"),
                       on_dev[2].reshape(-1)])
    buf_b = torch.cat([torch.zeros(7, dtype=torch.uint8, device="cuda"), on_dev[3].reshape(-1)])
    ds = [page_mod.DeviceStrip(buf_a, 0, 44, 600), page_mod.DeviceStrip(buf_a, sizes[0], 50, 420),
          page_mod.DeviceStrip(buf_a, sizes[0] + sizes[1] + 100, 61, 800), page_mod.DeviceStrip(buf_b, 7, 38, 256)]
    for mix in (ds, [ds[0], ds[1]] + [on_dev[2], strips[3]], [strips[0], ds[1], ds[2], ds[3]]):
        x1, T1, _ = lineest_gpu.normalize_strips(mix)
        assert np.array_equal(T0, T1) and torch.equal(x0, x1)
    x1, T1, _ = lineest_gpu.normalize_strips(ds[:2])                        # one slice, no copy
    xa, Ta, _ = lineest_gpu.normalize_strips(strips[:2])
    assert np.array_equal(Ta, T1) and torch.equal(xa, x1)
    st = page_mod.Strip(3, 4, 50, device_pixels=ds[1])
    assert st.width == 420 and page_mod.prepared_line(st)[0] is ds[1] and np.array_equal(st.pixels, strips[1])
    with pytest.raises(ValueError, match="outside its buffer"):
        lineest_gpu.normalize_strips([page_mod.DeviceStrip(buf_b, 8, 38, 256)])
    with pytest.raises(TypeError):
        lineest_gpu.normalize_strips([page_mod.DeviceStrip(buf_a.view(2, -1), 0, 44, 600)])


def test_recogniser_takes_raw_strips():
    """LineRecognizer.prepare: raw uint8 strips (device normaliser) mixed with host-prepared lines
    give the rows the checker's normaliser gives, and the same decoded characters."""
    from oracle import lineest_ref as lineest
    from text_alignment_amd import ocr
    rng = np.random.default_rng(9)
    strips = [_strip(rng, h, w) for h, w in [(44, 600), (50, 420), (61, 800), (38, 256), (44, 333)]]
    host = [lineest.prepare_raw_strip(s).astype(np.float32) for s in strips]
    model = ocr.LineModel.random(11, no=40)
    for wts in (model.fwd, model.rev):                         # contractive recurrence: see test_ocr_gpu._tame
        for name in ("WGI", "WGF", "WGO", "WCI"):
            wts[name][:, 49:] *= 0.25
        for name in ("WIP", "WFP", "WOP"):
            wts[name] *= 0.25
    rec = ocr.LineRecognizer(model)
    mixed = [strips[0], host[1], strips[2], strips[3], host[4]]
    st_m, st_h = rec.prepare(mixed), rec.prepare(host)
    assert np.array_equal(st_m["T_host"], st_h["T_host"])
    assert float((st_m["x"] - st_h["x"]).abs().max()) <= 2e-6
    assert rec.recognise(mixed) == rec.recognise(host)
    assert list(rec.last_T) == [h.shape[0] for h in host]
