"""A page's JSON from the CHECKERS alone -- the oracle side of bench.py's page legs (`pages_checked`,
`pages_equal_to_oracle`) and of tools that want to compare a pipelined result with the reference's per-page contract
(alignToOCR.py:187-330) after a timed run.  Nothing here is product code on the compared side:

    text lines  -> oracle/ocr_ref_f64.py   float64 restatement of the recogniser (the product's WEIGHTS, not its kernels)
    .llocs text -> oracle/glue_ref.py      character boxes, abbreviations (pinned to the reference's outputs)
    alignment   -> oracle/nw_oracle.py     C restatement of textSeqCompare.py (pinned to the reference)
    boxes, JSON -> oracle/glue_ref.py

The syllabifier is the product's (pinned to the reference's known answers by tests/test_glue.py), as in
tests/test_page_gpu.py::_expected, which this mirrors without that test's per-line tolerance analysis: a page either
equals the checkers' JSON or it does not.  ~1.5 s of one host core per 30-line page."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def oracle_model(line_model):
    """the checker's model object over the SAME weights as a product ocr.LineModel"""
    from oracle import ocr_ref_f64 as R
    return R.LineModel(line_model.ni, line_model.ns, line_model.no, line_model.fwd, line_model.rev,
                       np.asarray(line_model.W2, dtype=np.float64), list(line_model.codec))


def expected_json(page, transcript, line_model, params=None):
    """to_JSON_dict of `process(page, transcript, model, params)` as the checkers compute it"""
    from oracle import glue_ref, nw_oracle, ocr_ref_f64 as R
    from text_alignment_amd import latinSyllabification as latsyl
    om = oracle_model(line_model)
    chars = []
    for s in page.strips:
        xs = np.asarray(s.prepared, dtype=np.float64)
        dec = R.recognise(om, xs)["decoded"]
        scale = float(s.width) / (xs.shape[0] - 32)
        llocs = [(om.codec[c], (t - 16) * scale) for (t, c) in dec]
        chars += glue_ref.chars_from_llocs(R.llocs_text(llocs).split("\n")[:-1], s.offset_x, s.offset_y, s.offset_y + s.height)
    expanded = glue_ref.expand(chars, latsyl.abbreviations)
    ocr = "".join(b[0] for b in expanded)
    tra_align, ocr_align = nw_oracle.perform_alignment(list(transcript), list(ocr), params)
    return glue_ref.syllable_json(latsyl.syllabify_text(transcript), expanded, tra_align, ocr_align, page.angle,
                                  (page.image.dim.ncols, page.image.dim.nrows), (page.dim.ncols, page.dim.nrows),
                                  page.lines_peak_locs)


def check_pages(got_json, pages, transcripts, line_models, params, which):
    """{"pages_checked", "pages_equal_to_oracle", "checked_page_ids", ...}: pages `which` of a run's per-page JSON
    (got_json[k]: dict as alignToOCR.to_JSON_dict) against expected_json"""
    import time
    t0 = time.perf_counter()
    equal, first_diff = 0, None
    for k in which:
        want = expected_json(pages[k], transcripts[k], line_models[k], params)
        if got_json[k] == want:
            equal += 1
        elif first_diff is None:
            first_diff = {"page": int(k), "boxes_got": len(got_json[k]["syl_boxes"]) if got_json[k] else None,
                          "boxes_want": len(want["syl_boxes"])}
    out = {"pages_checked": len(which), "pages_equal_to_oracle": equal, "checked_page_ids": [int(k) for k in which],
           "checker": "oracle/ocr_ref_f64.py + oracle/glue_ref.py + oracle/nw_oracle.py (tools/pages_check.py), after the timed passes",
           "checker_seconds": time.perf_counter() - t0}
    if first_diff is not None:
        out["first_difference"] = first_diff
    return out
