"""End-to-end page test (BASELINE.json configs[2] shape: ~30 line strips + one NW alignment):
process() with BOTH kernels live, against the same pipeline driven by the CPU oracles
(float64 OCR restatement + C NW restatement) through the reference-pinned glue."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

VOCAB = ("dominus deus meus alleluia gloria patri et filio spiritui sancto sicut erat in principio "
         "nunc semper saecula saeculorum amen laudate eum omnes gentes quoniam confirmata est "
         "super nos misericordia eius veritas manet aeternum").split()


def _page(seed, nlines, R, page_mod):
    rng = np.random.default_rng(seed)
    strips = []
    for k in range(nlines):
        w = int(rng.integers(150, 420))
        xs = R.synthetic_line(seed * 1000 + k, width=w)
        strips.append(page_mod.Strip(offset_x=40 + int(rng.integers(0, 30)), offset_y=100 + 120 * k,
                                     height=60, width=2 * w, prepared=xs))
    peaks = [130 + 120 * k for k in range(nlines + 1)]
    transcript = " ".join(VOCAB[int(i)] for i in rng.integers(0, len(VOCAB), size=6 * nlines))
    return page_mod.PreparedPage((2200, 3300), (2200, 3300), 0, strips, peaks), transcript


def _expected(page, transcript, om, R, nw_oracle, params=None, rec=None, report=None):
    """The page's JSON from CHECKERS only: float64 recogniser restatement -> .llocs text ->
    oracle/glue_ref.py (pinned to the reference's own outputs, tests/test_oracle_glue.py) with the
    C aligner restatement; no product glue on this side.

    rec (a LineRecognizer): the recogniser mode under test is compared with the float64 restatement
    line by line first (tests/ocr_compare.py).  A line whose decode differs ONLY in decisions that
    the measured probability difference explains (a blank probability within that difference of the
    0.7 threshold, an arg-max within twice it) enters the expected pipeline with the product's
    characters -- no implementation can pin those at the tolerance -- and is counted in `report`;
    any other difference fails here."""
    from oracle import glue_ref
    import ocr_compare
    from text_alignment_amd import latinSyllabification as latsyl          # pinned by tests/test_glue.py
    chars = []
    cmp_ = ocr_compare.compare_lines(R, om, rec, [s.prepared for s in page.strips]) if rec is not None else None
    for k, s in enumerate(page.strips):
        ref = cmp_["refs"][k] if cmp_ else R.recognise(om, s.prepared)
        dec = ref["decoded"]
        if cmp_ is not None:
            assert not cmp_["unexplained"][k], (k, cmp_["unexplained"][k], cmp_["prob_err"][k])
            if cmp_["explained"][k]:
                dec = cmp_["dec"][k]
        scale = float(s.width) / (s.prepared.shape[0] - 32)
        llocs = [(om.codec[c], (t - 16) * scale) for (t, c) in dec]
        lines = R.llocs_text(llocs).split("\n")[:-1]
        chars += glue_ref.chars_from_llocs(lines, s.offset_x, s.offset_y, s.offset_y + s.height)
    if report is not None and cmp_ is not None:
        report.update(lines=len(page.strips), chars=cmp_["chars"], chars_agree=cmp_["chars_agree"],
                      lines_with_explained_differences=sum(1 for e in cmp_["explained"] if e),
                      logit_err_max=max(cmp_["logit_err"]), logit_err_median=float(np.median(cmp_["logit_err"])))
    expanded = glue_ref.expand(chars, latsyl.abbreviations)
    ocr = "".join(b[0] for b in expanded)
    tra_align, ocr_align = nw_oracle.perform_alignment(list(transcript), list(ocr), params)
    js = glue_ref.syllable_json(latsyl.syllabify_text(transcript), expanded, tra_align, ocr_align, page.angle,
                                (page.image.dim.ncols, page.image.dim.nrows), (page.dim.ncols, page.dim.nrows),
                                page.lines_peak_locs)
    return js, ocr


@pytest.mark.parametrize("precision", [None, "f32", "split"])
def test_single_page_process_matches_oracle_pipeline(precision):
    """BASELINE configs[2]: one page of 30 strips through process() with both kernels live, in the
    recogniser's DEFAULT mode (precision=None: whatever ocr.DEFAULT_PRECISION is -- float64 since round 5, the mode
    bench.py times first) and in the opt-in exact-f32 and split modes, free-running on a random-weight model, against the
    checker pipeline.  Measured agreement with the float64 restatement is printed."""
    from oracle import nw_oracle, ocr_ref_f64 as R
    from text_alignment_amd import alignToOCR as atocr, ocr, page as page_mod
    om = R.synthetic_model(7001, no=40)             # small class count: mostly letters come out
    om.W2[0, 0] += 4.0                              # favour blanks -> many short runs -> many characters
    kw = {} if precision is None else {"precision": precision}
    pm = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec), **kw)
    assert pm.mode == {"f32": 0, "split": 1, "f64": 3}[precision or ocr.DEFAULT_PRECISION]
    pg, transcript = _page(3, 30, R, page_mod)
    params = [8, -1, -9, -9, -4, -4]     # cheap mismatches: the random model's text pairs up with the transcript
    res = atocr.process(pg, transcript, pm, seq_align_params=params)
    assert res is not None
    syl_boxes, image, peaks, all_chars = res
    got = atocr.to_JSON_dict(syl_boxes, peaks)
    report = {}
    want, want_ocr = _expected(pg, transcript, om, R, nw_oracle, params, rec=pm, report=report)
    print("page, mode %s: %s" % (precision or ocr.DEFAULT_PRECISION, report))
    assert report["chars_agree"] >= 0.995 * report["chars"]
    if (precision or ocr.DEFAULT_PRECISION) != "split":
        # the one place where product output can reach the expected side (a line whose decode differs "explainably"
        # enters the expected pipeline with the product's characters) is not taken in the default mode nor in
        # float64 mode: every line's decode IS the float64 restatement's
        assert report["lines_with_explained_differences"] == 0 and report["chars_agree"] == report["chars"]
    if (precision or ocr.DEFAULT_PRECISION) == "f64":
        assert report["logit_err_max"] < 1e-4
    assert "".join(c.char for c in all_chars) == want_ocr
    assert got == want
    assert len(got["syl_boxes"]) > 50


def test_process_with_an_ocr_cache_file_the_reference_wrote(tmp_path, capsys):
    """existing_ocr_pickle (alignToOCR.py:225-233): the grid search calls process() 729 times per page with the page's
    OCR cached in ./pik/<name>_boxes.pickle (evaluate_text_alignment.py:159-171).  A cache file in the reference's own
    format (Python 2, class `alignToOCR.CharBox`) is used instead of running the recogniser -- the model argument is
    not even looked at -- and gives the same JSON as the live run."""
    from test_glue import _py2_style_box_pickle
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import alignToOCR as atocr, ocr, page as page_mod
    om = R.synthetic_model(7001, no=40)
    om.W2[0, 0] += 4.0
    pm = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec))
    pg, transcript = _page(5, 6, R, page_mod)
    params = [8, -1, -9, -9, -4, -4]
    live = atocr.process(pg, transcript, pm, seq_align_params=params)
    f = tmp_path / "page_boxes.pickle"
    f.write_bytes(_py2_style_box_pickle("alignToOCR", [(c.char.encode("latin1"), c.ul, c.lr) for c in live[3]]))
    capsys.readouterr()
    cached = atocr.process(pg, transcript, None, seq_align_params=params, existing_ocr_pickle=str(f))
    assert "using pickled ocr results" in capsys.readouterr().out
    assert atocr.to_JSON_dict(cached[0], cached[2]) == atocr.to_JSON_dict(live[0], live[2])
    assert [(c.char, c.ul, c.lr) for c in cached[3]] == [(c.char, c.ul, c.lr) for c in live[3]]
    # a missing file: OCR runs (alignToOCR.py:230-231)
    again = atocr.process(pg, transcript, pm, seq_align_params=params, existing_ocr_pickle=str(tmp_path / "none.pickle"))
    assert "not found - performing ocr instead" in capsys.readouterr().out
    assert atocr.to_JSON_dict(again[0], again[2]) == atocr.to_JSON_dict(live[0], live[2])


def test_process_batch_equals_process_and_sharded_driver():
    from oracle import ocr_ref_f64 as R
    from text_alignment_amd import alignToOCR as atocr, ocr, page as page_mod, sharding
    om = R.synthetic_model(7002, no=40)
    om.W2[0, 0] += 4.0
    rec = ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec))
    pages, trs = zip(*[_page(10 + k, 4 + 3 * k, R, page_mod) for k in range(4)])
    params = [10, -5, -7, -7]
    single = [atocr.to_JSON_dict(*(atocr.process(p, t, rec, seq_align_params=params)[i] for i in (0, 2)))
              for p, t in zip(pages, trs)]
    batch = atocr.process_batch(list(pages), list(trs), rec, seq_align_params=params)
    assert [atocr.to_JSON_dict(b[0], b[2]) for b in batch] == single
    out = sharding.process_pages(list(pages), list(trs), rec, seq_align_params=params)   # world size 1
    assert [out[k] for k in range(4)] == single


def test_config5_shape_64_pages_two_models_sharded_driver():
    """BASELINE configs[4] at one rank: 64 pages, half read with a 96-class (Salzinnes-shaped) model
    and half with a 64-class (St-Gall-shaped) one, through sharding.process_pages (process_batch per
    model + the single fixed-capacity gather); every page's JSON equals process() of that page
    alone, and three pages are rebuilt from the checkers (float64 recogniser + C aligner + glue).
    The recognisers run in the DEFAULT precision, the one bench.py's pages_sharded leg times."""
    from oracle import nw_oracle, ocr_ref_f64 as R
    from text_alignment_amd import alignToOCR as atocr, ocr, page as page_mod, sharding
    oms, recs = [], []
    for seed, no in ((7001, 96), (7002, 64)):
        om = R.synthetic_model(seed, no=no)
        om.W2[0, 0] += 4.0
        om.W2[30:, :] *= 0.25                       # mostly the first classes come out: text-like strings
        oms.append(om)
        recs.append(ocr.LineRecognizer(ocr.LineModel(om.fwd, om.rev, om.W2, om.codec)))      # the default mode
    pages, trs = zip(*[_page(200 + k, 3 + k % 5, R, page_mod) for k in range(64)])
    models = [recs[k % 2] for k in range(64)]
    params = [8, -1, -9, -9, -4, -4]
    out = sharding.process_pages(list(pages), list(trs), models, seq_align_params=params)
    assert sorted(out) == list(range(64))
    for k in range(64):
        res = atocr.process(pages[k], trs[k], models[k], seq_align_params=params)
        assert out[k] == atocr.to_JSON_dict(res[0], res[2]), k
    assert sum(len(out[k]["syl_boxes"]) for k in range(64)) > 300
    for k in (0, 1, 37):
        want, _ = _expected(pages[k], trs[k], oms[k % 2], R, nw_oracle, params, rec=recs[k % 2])
        assert out[k] == want, k


def test_ocr_failure_returns_none(capsys):
    from text_alignment_amd import alignToOCR as atocr, ocr, page as page_mod
    pm = ocr.LineModel.random(1, no=20)
    strip = page_mod.Strip(0, 0, 60, width=100, prepared=np.zeros((5200, 48)))    # too long for the LSTM
    pg = page_mod.PreparedPage((1000, 800), (1000, 800), 0, [strip], [100, 220])
    assert atocr.process(pg, "dominus", pm) is None
    assert "OCRopus failed! Skipping current file." in capsys.readouterr().out
