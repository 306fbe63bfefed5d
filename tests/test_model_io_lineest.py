"""Row N1: .pyrnn.gz loader and line normaliser -- CPU tests.  The
model files of the reference are absent, so the loader is exercised on a pickle this test writes
itself with stand-in classes named like ocrolib's (protocol 2, gzip)."""
import gzip
import pickle
import sys
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _write_fake_pyrnn(path, module_name="ocrolib.lstm"):
    mod = types.ModuleType(module_name)
    names = ["SeqRecognizer", "Stacked", "Parallel", "Reversed", "LSTM", "Softmax", "Codec"]
    cls = {}
    for n in names:
        cls[n] = type(n, (object,), {"__module__": module_name})
        setattr(mod, n, cls[n])
    sys.modules[module_name] = mod
    pkg = module_name.split(".")[0]
    if "." in module_name:
        parent = types.ModuleType(pkg)
        parent.__path__ = []
        setattr(parent, module_name.split(".")[1], mod)
        sys.modules[pkg] = parent
    rng = np.random.default_rng(3)

    def lstm():
        o = cls["LSTM"]()
        for k in ("WGI", "WGF", "WGO", "WCI"):
            setattr(o, k, rng.uniform(-0.5, 0.5, size=(100, 149)))
        for k in ("WIP", "WFP", "WOP"):
            setattr(o, k, rng.uniform(-0.5, 0.5, size=(100,)))
        o.dims = (48, 100)
        return o
    f, r = lstm(), lstm()
    rev = cls["Reversed"](); rev.net = r
    par = cls["Parallel"](); par.nets = [f, rev]
    sm = cls["Softmax"](); sm.W2 = rng.uniform(-1, 1, size=(7, 201))
    st = cls["Stacked"](); st.nets = [par, sm]
    codec = cls["Codec"](); codec.code2char = {0: u"", 1: u" ", 2: u"~", 3: u"a", 4: u"b", 5: u"ā"}
    rec = cls["SeqRecognizer"](); rec.lstm = st; rec.codec = codec; rec.Ni, rec.Ns, rec.No = 48, 100, 7
    try:
        with gzip.open(path, "wb") as fh:
            pickle.dump(rec, fh, protocol=2)
    finally:
        del sys.modules[module_name]
        if "." in module_name:
            sys.modules.pop(pkg, None)
    return f, r, sm.W2


@pytest.mark.parametrize("module_name", ["ocrolib.lstm", "lstm.lstm"])
def test_load_pyrnn_roundtrip(tmp_path, module_name):
    from text_alignment_amd import model_io
    path = str(tmp_path / "m.pyrnn.gz")
    f, r, W2 = _write_fake_pyrnn(path, module_name)
    m = model_io.load_pyrnn(path)
    assert m.no == 7 and m.codec == ["", " ", "~", "a", "b", u"ā", "~"]
    assert np.array_equal(m.fwd["WGI"], f.WGI) and np.array_equal(m.rev["WOP"], r.WOP)
    assert np.array_equal(m.W2, W2)


def test_restricted_unpickler_rejects_other_globals(tmp_path):
    from text_alignment_amd import model_io
    path = str(tmp_path / "evil.pyrnn.gz")
    with gzip.open(path, "wb") as fh:
        pickle.dump(print, fh, protocol=2)            # any global outside the allow-list
    with pytest.raises(pickle.UnpicklingError):
        model_io.load_pyrnn(path)


def test_line_normaliser_checker_shapes_and_polarity():
    """oracle/lineest_ref.py (the checker of csrc/ta_lineest.hip) on a synthetic strip"""
    from oracle import lineest_ref as lineest
    rng = np.random.default_rng(0)
    img = np.ones((70, 300))                           # white page, a dark band of "text"
    band = rng.random((24, 300)) < 0.4
    img[22:46][band] = 0.0
    xs = lineest.prepare_raw_strip((img * 255).astype(np.uint8))
    assert xs.shape[1] == 48 and xs.shape[0] > 32
    assert (xs[:16] == 0).all() and (xs[-16:] == 0).all()
    assert 0.0 <= xs.min() and xs.max() <= 1.0 + 1e-6
    core = xs[16:-16]
    assert core[:, 12:36].mean() > 4 * max(core[:, :6].mean(), 1e-3)      # ink sits in the middle rows
    with pytest.raises(ValueError):
        lineest.prepare_raw_strip(np.full((40, 100), 255, np.uint8))


def test_prepared_lines_hand_raw_strips_to_the_device_normaliser():
    """page.prepared_lines: strips that carry `.prepared` pass through; raw strips are handed on as 2-D
    uint8 images (bool: True = ink -> black on white) for the device normaliser; empty strips and
    pixel types the reference's PNG seam never saw are refused on the host, before any GPU work."""
    from text_alignment_amd import page as page_mod
    rng = np.random.default_rng(3)
    ink = rng.random((50, 300)) < 0.3
    raw = page_mod.Strip(10, 20, 50, pixels=np.where(ink, 0, 255).astype(np.uint8))
    onebit = page_mod.Strip(10, 80, 50, pixels=ink)
    ready = np.zeros((100, 48), np.float32)
    got = page_mod.prepared_lines([raw, page_mod.Strip(0, 0, 48, width=136, prepared=ready), onebit], workers=2)
    assert got[0][0].dtype == np.uint8 and got[0][1] == 300 and got[0][0] is raw.pixels
    assert got[1][0] is ready and got[1][1] == 136
    assert got[2][0].dtype == np.uint8 and np.array_equal(got[2][0], raw.pixels)
    with pytest.raises(ValueError, match="empty or constant"):
        page_mod.prepared_lines([page_mod.Strip(0, 0, 40, pixels=np.zeros((0, 100), np.uint8))])
    # (a CONSTANT strip passes here and is refused, with the same error, by the device normaliser's measuring pass:
    # tests/test_lineest_gpu.py::test_device_normaliser_rejects_what_the_checker_rejects)
    # a strip whose pixels live in a tensor (on the GPU, after the device preprocessing): handed on as it is,
    # `.pixels` reads it back once; assigning pixels drops the tensor
    import torch
    t = torch.from_numpy(np.where(ink, 0, 255).astype(np.uint8))
    lazy = page_mod.Strip(1, 2, 50, device_pixels=t)
    assert lazy.width == 300 and page_mod.prepared_lines([lazy])[0][0] is t
    assert np.array_equal(lazy.pixels, raw.pixels) and lazy.pixels is lazy.pixels
    lazy.pixels = raw.pixels
    assert lazy.device_pixels is None and page_mod.prepared_lines([lazy])[0][0] is raw.pixels
    with pytest.raises(TypeError):
        page_mod.prepared_lines([page_mod.Strip(0, 0, 40, pixels=np.zeros((40, 100), np.float32))])
    with pytest.raises(TypeError):
        page_mod.prepared_lines([page_mod.Strip(0, 0, 40, pixels=np.zeros((40, 100, 3), np.uint8))])


def test_class_count_limit_is_refused_up_front():
    """the output kernel holds a timestep's whole softmax row in one accumulator tile (128 classes):
    a model with more is refused when it is built, not at the first recognise()"""
    from text_alignment_amd import ocr
    ok = ocr.LineModel.random(1, no=128)
    assert ok.no == 128
    with pytest.raises(ValueError, match="2..128 output classes"):
        ocr.LineModel.random(1, no=129)
    with pytest.raises(ValueError):
        ocr.LineModel(ok.fwd, ok.rev, ok.W2[:1], ok.codec[:1])
