"""oracle/glue_ref.py (the checker of the page-level GPU tests) against outputs of the imported
reference: tests/golden/llocs.json and tests/golden/glue.json.  CPU only."""
from conftest import load_golden
from oracle import glue_ref, nw_oracle


def test_llocs_restatement_matches_reference():
    for c in load_golden("llocs.json")["cases"]:
        got = []
        for s in c["strips"]:
            got += glue_ref.chars_from_llocs(s["llocs"], s["offset_x"], s["offset_y"], s["offset_y"] + s["height"])
        assert [[b[0], [b[1], b[2]], [b[3], b[4]]] for b in got] == c["chars"]


def test_rotate_restatement_matches_reference():
    for c in load_golden("glue.json")["rotate_cases"]:
        b = glue_ref.unrotate(('x', c["ul"][0], c["ul"][1], c["lr"][0], c["lr"][1]), c["angle"],
                              c["orig_dim"][0], c["orig_dim"][1], c["target_dim"][0], c["target_dim"][1])
        assert [b[1], b[2]] == c["out_ul"] and [b[3], b[4]] == c["out_lr"]


def test_page_glue_restatement_matches_reference_pages():
    """process() -> to_JSON_dict() of the reference on 15 pages, rebuilt from the canned OCR boxes by
    the restatement with oracle/nw_oracle.py as the aligner and the syllabifier's pinned output"""
    from text_alignment_amd import latinSyllabification as latsyl      # pinned by tests/test_glue.py
    g = load_golden("glue.json")
    assert g["abbreviations"] == {k: list(v) for k, v in latsyl.abbreviations.items()}
    for c in g["process_cases"]:
        chars = [(ch, ul[0], ul[1], lr[0], lr[1]) for ch, ul, lr in c["chars"]]
        expanded = glue_ref.expand(chars, latsyl.abbreviations)
        text = ''.join(b[0] for b in expanded)
        assert text == c["expanded_ocr"], c["name"]
        tra, ocr = nw_oracle.perform_alignment(list(c["transcript"]), list(text), c["params"])
        js = glue_ref.syllable_json(latsyl.syllabify_text(c["transcript"]), expanded, tra, ocr, c["angle"],
                                    c["img_dim"], c["raw_dim"], c["peak_locs"])
        js["median_line_spacing"] = float(js["median_line_spacing"])
        assert js == c["json"], c["name"]
