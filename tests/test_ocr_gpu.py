"""GPU parity tests of the line recogniser (K3 BiLSTM, K4 output layer + softmax, K5 decode)
against the float64 restatement oracle/ocr_ref_f64.py (parity unpinned:
the reference's OCR arithmetic is third-party and absent).  Tolerance from BASELINE.json's
north_star: logits within 1e-3 of the float64 result; decoded (t, class) lists identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

TOL = 1e-3


def _models(seed, no):
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import ocr
    om = R.synthetic_model(seed, no=no)
    pm = ocr.LineModel(om.fwd, om.rev, om.W2, om.codec)
    return R, ocr, om, pm


@pytest.mark.parametrize("seed,no", [(7001, 96), (7002, 64)])
def test_lines_vs_f64_oracle(seed, no):
    assert torch.cuda.is_available()
    R, ocr, om, pm = _models(seed, no)
    widths = [1, 2, 40, 100, 333, 800, 1200, 64, 65, 17, 500, 501, 499, 256, 31, 777, 900, 128, 3, 1000]
    lines = [R.synthetic_line(8000 + k, width=w) for k, w in enumerate(widths)]
    rec = ocr.LineRecognizer(pm)
    dec, probs, logits, states = rec.recognise(lines, want_probs=True)
    assert rec.recognise(lines, from_probs=True) == dec        # K5 from full probabilities == from summaries
    worst = 0.0
    errs = []
    for k, xs in enumerate(lines):
        ref = R.recognise(om, xs)
        assert states[k].shape == ref["states"].shape
        e_s = np.abs(states[k] - ref["states"]).max()
        e_z = np.abs(logits[k] - ref["logits"]).max()
        e_p = np.abs(probs[k] - ref["probs"]).max()
        worst = max(worst, e_z)
        errs.append((widths[k], float(e_s), float(e_z), float(e_p)))
        assert e_z < TOL, (k, widths[k], e_z)
        assert e_p < TOL, (k, widths[k], e_p)
        assert dec[k] == ref["decoded"], (k, widths[k])
        ll = rec.llocs(dec[k], xs.shape[0], widths[k])
        assert ocr.llocs_text(ll) == R.llocs_text(ref["llocs"])
    print("max |logit error| =", worst)
    print("per line (width, state err, logit err, prob err):", errs)


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
