"""GPU parity tests of the line recogniser (K3 BiLSTM, K4 output layer + softmax, K5 decode)
against the float64 restatement oracle/ocr_ref_f64.py (parity unpinned:
the reference's OCR arithmetic is third-party and absent).  Tolerance from BASELINE.json's
north_star: logits within 1e-3 of the float64 result; decoded (t, class) lists identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 1e-3


def _tame(om):
    """Contractive variant of the synthetic model: recurrent columns and peepholes scaled by 1/4.
    The SURVEY-spec random weights (U(-0.5, 0.5) on 149 inputs) make the LSTM chaotic -- rounding
    differences of 1e-7 grow to 1e-3 within ~1000 timesteps on some lines, in ANY float32
    implementation -- so long-line parity at 1e-3 is asserted on this stable model, as trained
    recognisers are."""
    for w in (om.fwd, om.rev):
        for k in ("WGI", "WGF", "WGO", "WCI"):
            w[k][:, 49:] *= 0.25
        for k in ("WIP", "WFP", "WOP"):
            w[k] *= 0.25
    return om


def _models(seed, no):
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    om = _tame(R.synthetic_model(seed, no=no))
    pm = ocr.LineModel(om.fwd, om.rev, om.W2, om.codec)
    return R, ocr, om, pm


def _check_lines(R, ocr, om, widths, tol, check_decode=True, precision="f32"):
    pm = ocr.LineModel(om.fwd, om.rev, om.W2, om.codec)
    lines = [R.synthetic_line(8000 + k, width=w) for k, w in enumerate(widths)]
    rec = ocr.LineRecognizer(pm, precision=precision)
    dec, probs, logits, states = rec.recognise(lines, want_probs=True)
    assert rec.recognise(lines, from_probs=True) == dec        # K5 from full probabilities == from summaries
    errs = []
    for k, xs in enumerate(lines):
        ref = R.recognise(om, xs)
        assert states[k].shape == ref["states"].shape
        e_z = float(np.abs(logits[k] - ref["logits"]).max())
        e_p = float(np.abs(probs[k] - ref["probs"]).max())
        errs.append(e_z)
        assert e_z < tol, (k, widths[k], e_z)
        assert e_p < tol, (k, widths[k], e_p)
        if check_decode:
            assert dec[k] == ref["decoded"], (k, widths[k])
            ll = rec.llocs(dec[k], xs.shape[0], widths[k])
            assert ocr.llocs_text(ll) == R.llocs_text(ref["llocs"])
    print("logit errors: max %.3g median %.3g" % (max(errs), float(np.median(errs))))
    return errs


@pytest.mark.parametrize("seed,no", [(7001, 96), (7002, 64)])
def test_spec_model_short_lines(seed, no):
    """SURVEY section 8d weights, lines up to 333 columns: logits within 1e-3, decode identical."""
    assert torch.cuda.is_available()
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    _check_lines(R, ocr, R.synthetic_model(seed, no=no),
                 [1, 2, 3, 17, 31, 40, 64, 65, 100, 128, 200, 256, 300, 333], TOL)


@pytest.mark.parametrize("seed,no", [(7001, 96), (7002, 64)])
def test_stable_model_long_lines(seed, no):
    """Contractive model, lines of the benchmark's widths (up to 2000 columns): 1e-3, decode identical."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    errs = _check_lines(R, ocr, _tame(R.synthetic_model(seed, no=no)),
                        [499, 500, 501, 777, 800, 900, 1000, 1200, 1600, 2000], TOL)
    assert max(errs) < 2e-4


def test_bf16x3_mode():
    """The split-bf16 fast mode: 1e-3 on the contractive model at benchmark widths (decode
    identical); on the chaotic spec model only short lines, at a looser bound."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    errs = _check_lines(R, ocr, _tame(R.synthetic_model(7001, no=96)), [100, 500, 1000, 2000], TOL,
                        precision="bf16x3")
    assert max(errs) < 3e-4
    _check_lines(R, ocr, R.synthetic_model(7001, no=96), [1, 17, 40, 64, 100], 1e-2,
                 check_decode=False, precision="bf16x3")


def test_spec_model_long_lines_bounded():
    """Chaotic spec model on long lines: the typical line is still at float32 noise level; the
    worst line is only bounded loosely (see _tame)."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    errs = _check_lines(R, ocr, R.synthetic_model(7001, no=96), [500, 501, 800, 900, 1000, 1200],
                        5e-2, check_decode=False)
    assert float(np.median(errs)) < 2e-4


def test_group_boundaries_and_order():
    """15, 16, 17 and 33 lines (group edges), results independent of batch composition."""
    R, ocr, om, pm = _models(7001, 96)
    rec = ocr.LineRecognizer(pm)
    base = [R.synthetic_line(9000 + k, width=60 + 7 * k) for k in range(33)]
    ref = [R.recognise(om, xs)["decoded"] for xs in base]
    for cnt in (1, 15, 16, 17, 33):
        dec = rec.recognise(base[:cnt])
        assert dec == ref[:cnt], cnt
    assert rec.recognise([]) == []


def test_input_too_large():
    R, ocr, om, pm = _models(7001, 96)
    rec = ocr.LineRecognizer(pm)
    with pytest.raises(ocr.RecognitionError):
        rec.recognise([np.zeros((5001, 48))])


def test_decode_kernel_on_crafted_probabilities():
    """K5 alone on hand-made outputs: ties go to the earlier t, then the smaller class; a run
    open at the end of the line is emitted."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import _native
    no = 70
    T = 12
    p = np.full((T, no), 0.001, dtype=np.float32)
    p[:, 0] = 0.9
    p[2:5, 0] = 0.1; p[3, 5] = 0.6; p[4, 5] = 0.6; p[3, 69] = 0.6       # tie: (3,5) wins
    p[7, 0] = 0.69; p[7, 1] = 0.3
    p[10:, 0] = 0.2; p[11, 66] = 0.5                                    # run open at the end
    want = R.translate_back(p.astype(np.float64))
    dev = torch.device("cuda")
    pd = torch.from_numpy(p).to(dev)
    ro = torch.zeros(1, dtype=torch.int64, device=dev)
    Td = torch.tensor([T], dtype=torch.int32, device=dev)
    dt = torch.zeros(T, dtype=torch.int32, device=dev); dc = torch.zeros(T, dtype=torch.int32, device=dev)
    dn = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = _native.lib.ta_decode(pd.data_ptr(), ro.data_ptr(), Td.data_ptr(), 1, no, 0.7,
                               dt.data_ptr(), dc.data_ptr(), dn.data_ptr(), ro.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    k = int(dn.item())
    got = list(zip(dt.cpu().numpy()[:k].tolist(), dc.cpu().numpy()[:k].tolist()))
    assert got == want == [(3, 5), (7, 0), (11, 66)]


@pytest.mark.parametrize("no", [5, 16, 17, 33, 80, 112, 128])
def test_class_counts_cover_every_output_tile_variant(no):
    """The output layer is compiled once per number of 16-class tiles (1..8): class counts on both
    sides of the tile boundaries, up to the maximum of 128."""
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    _check_lines(R, ocr, _tame(R.synthetic_model(7100 + no, no=no)), [40, 129, 300], TOL)
