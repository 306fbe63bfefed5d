"""Host-side glue of the page driver against golden vectors captured from the reference's
alignToOCR.process / to_JSON_dict / rotate_bbox and latinSyllabification (tools/gen_golden.py).
The NW step inside process() needs the GPU, so the end-to-end cases are marked gpu; the pure
host pieces run everywhere."""
import json
import pickle

import numpy as np
import pytest

from conftest import load_golden


def test_syllabifier_golden():
    from text_alignment_amd import latinSyllabification as ls
    g = load_golden("glue.json")
    assert ls.syllabify_text(g["syllabify_demo"]["inp"]) == g["syllabify_demo"]["syls"]
    for c in g["syllabify_cases"]:
        assert ls.syllabify_text(c["inp"]) == c["syls"], c
    for c in g["syllabify_words"]:
        assert ls.syllabify_word(c["inp"]) == c["syls"], c
    assert list(ls.abbreviations.items()) == list(g["abbreviations"].items())


def test_syllabifier_vowelless_words_terminate():
    from text_alignment_amd import latinSyllabification as ls
    for w in ("b", "dns", "st", "x1", "qu"):
        assert ls.syllabify_word(w) == [w]          # the reference never returns on these
    assert ls.syllabify_text("dns meus") == ["dns", "me", "us"]


def test_rotate_bbox_golden():
    from text_alignment_amd import alignToOCR as atocr
    from text_alignment_amd.page import Dim
    for c in load_golden("glue.json")["rotate_cases"]:
        r = atocr.rotate_bbox(atocr.CharBox('x', c["ul"], c["lr"]), c["angle"],
                              Dim(*c["orig_dim"]), Dim(*c["target_dim"]))
        assert [int(r.ul[0]), int(r.ul[1])] == c["out_ul"] and [int(r.lr[0]), int(r.lr[1])] == c["out_lr"], c


def test_rotate_bbox_floor_division_on_odd_sizes():
    # Python-2 semantics of the reference (alignToOCR.py:91,95-96): floor division
    from text_alignment_amd import alignToOCR as atocr
    from text_alignment_amd.page import Dim
    r = atocr.rotate_bbox(atocr.CharBox('x', (10, 10), (20, 20)), 0, Dim(1001, 801), Dim(1000, 800))
    assert (int(r.ul[0]), int(r.ul[1])) == (10, 10)      # dx = dy = 0 under floor division
    r = atocr.rotate_bbox(atocr.CharBox('x', (10, 10), (20, 20)), 0, Dim(1000, 800), Dim(1003, 801))
    assert (int(r.ul[0]), int(r.ul[1])) == (12, 11)      # (1000-1003)//2 = -2, (800-801)//2 = -1


def test_rotate_bboxes_equals_rotate_bbox_and_round_semantics():
    from text_alignment_amd import alignToOCR as atocr
    from text_alignment_amd.page import Dim
    rng = np.random.default_rng(0)
    for _ in range(60):
        ang = float(rng.choice([0, 0.5, -1.25, 2.5, 90, 33.3]))
        od = Dim(int(rng.integers(500, 3000)), int(rng.integers(500, 3000)))
        td = Dim(od.ncols - int(rng.integers(-41, 41)), od.nrows - int(rng.integers(-41, 41)))
        boxes = [atocr.CharBox('x', (int(rng.integers(0, 3000)), int(rng.integers(0, 3000))),
                               (int(rng.integers(0, 3000)), int(rng.integers(0, 3000)))) for _ in range(12)]
        one = [atocr.rotate_bbox(b, ang, od, td) for b in boxes]
        many = atocr.rotate_bboxes(boxes, ang, od, td)
        assert [(int(b.ul[0]), int(b.ul[1]), int(b.lr[0]), int(b.lr[1])) for b in one] == \
               [(int(b.ul[0]), int(b.ul[1]), int(b.lr[0]), int(b.lr[1])) for b in many]
    # Python round == numpy round (half to even) on the values chars_from_llocs sees
    for v in list(rng.uniform(0, 5000, 2000).round(1)) + [0.5, 1.5, 2.5, 110.5, 111.5]:
        assert int(round(float(v))) == int(np.round(float(v)))


def test_charbox_and_helpers(tmp_path):
    from text_alignment_amd import alignToOCR as atocr
    b = atocr.CharBox('a', (1, 2), (4, 8))
    assert (b.ulx, b.uly, b.lrx, b.lry, b.width, b.height) == (1, 2, 4, 8, 3, 6)
    assert repr(b) == 'a: (1, 2), (4, 8)'
    gap = atocr.CharBox('_')
    assert gap.ul is None and gap.lr is None and repr(gap) == '_: empty' and not hasattr(gap, 'ulx')
    b2, g2 = pickle.loads(pickle.dumps([b, gap], -1))
    assert b2.lr == (4, 8) and g2.ul is None
    assert atocr.clean_special_chars('a~b~') == 'ab'
    f = tmp_path / "t.txt"
    f.write_text("# comment\ndominus dixit | \nad me\r\n")
    assert atocr.read_file(str(f)) == "dominus dixit  ad me"
    assert atocr.parallel == 2 and atocr.median_line_mult == 2


def _py2_style_box_pickle(module, boxes):
    """Bytes of `pickle.dump(list_of_CharBox, f, -1)` as PYTHON 2 writes it for the reference's class (alignToOCR.py:35-58,
    dumped at :435-436 and evaluate_text_alignment.py:170-171): protocol 2, the class by GLOBAL + NEWOBJ, the default
    state of a __slots__ class -- the pair (None, {slot: value}) -- str values as SHORT_BINSTRING (Python 2 str = bytes)."""
    def s(b):
        return b"U" + bytes([len(b)]) + b
    def i(v):
        v = int(v)
        if 0 <= v < 256:
            return b"K" + bytes([v])                                  # BININT1
        if 0 <= v < 65536:
            return b"M" + v.to_bytes(2, "little")                     # BININT2
        return b"J" + v.to_bytes(4, "little", signed=True)            # BININT (a box whose lr lies left of its ul: negative width)
    out = b"\x80\x02]("
    for ch, ul, lr in boxes:
        out += b"c" + module.encode() + b"\nCharBox\n)\x81N}("
        out += s(b"char") + s(ch)
        if ul is None:
            out += s(b"ul") + b"N" + s(b"lr") + b"N"
        else:
            out += s(b"ul") + i(ul[0]) + i(ul[1]) + b"\x86" + s(b"lr") + i(lr[0]) + i(lr[1]) + b"\x86"
            for name, v in (("ulx", ul[0]), ("uly", ul[1]), ("lrx", lr[0]), ("lry", lr[1]),
                            ("width", lr[0] - ul[0]), ("height", lr[1] - ul[1])):
                out += s(name.encode()) + i(v)
        out += b"u\x86b"
    return out + b"e."


def test_ocr_cache_files_written_by_the_reference_load(tmp_path):
    """existing_ocr_pickle (alignToOCR.py:225-233): the grid search's cache (evaluate_text_alignment.py:159-171).  A
    file the reference wrote names `alignToOCR.CharBox` (or `__main__.CharBox` when it ran as a script), carries
    Python 2 byte strings and the default __slots__ state; it loads into this package's CharBox through an
    unpickler that resolves nothing else."""
    import sys, types
    from text_alignment_amd import alignToOCR as atocr
    boxes = [(b"a", (1, 2), (4, 8)), (b"\xc5\xab", (300, 7), (1000, 47)), (b"_", None, None)]
    for module in ("alignToOCR", "__main__"):
        raw = _py2_style_box_pickle(module, boxes)
        # the bytes are what a class of that shape pickles to: a stand-in module under the reference's name reads them
        fake = types.ModuleType("alignToOCR")
        class CharBox(object):
            __slots__ = ['char', 'ul', 'lr', 'ulx', 'lrx', 'uly', 'lry', 'width', 'height']
        CharBox.__module__, fake.CharBox = "alignToOCR", CharBox
        if module == "alignToOCR":
            sys.modules["alignToOCR"] = fake
            try:
                probe = pickle.loads(raw, encoding="latin1")
            finally:
                del sys.modules["alignToOCR"]
            assert probe[0].lr == (4, 8) and probe[2].ul is None and not hasattr(probe[2], "ulx")
        f = tmp_path / ("%s.pickle" % module.strip("_"))
        f.write_bytes(raw)
        got = atocr.load_ocr_pickle(str(f))
        assert [type(b) for b in got] == [atocr.CharBox] * 3
        assert (got[0].char, got[0].ul, got[0].lr, got[0].width, got[0].height) == ("a", (1, 2), (4, 8), 3, 6)
        assert got[1].char == "\xc5\xab" and got[1].lrx == 1000          # Python 2 bytes read as latin-1, as pickle.load(encoding='latin1')
        assert got[2].ul is None and got[2].lr is None and not hasattr(got[2], "ulx")
    # this package's own files (the output of process(), numpy int16 corners after rotate_bbox)
    own = [atocr.CharBox('x', (np.int16(3), np.int16(4)), (np.int16(30), np.int16(40))), atocr.CharBox('_')]
    f = tmp_path / "own.pickle"
    f.write_bytes(pickle.dumps(own, -1))
    got = atocr.load_ocr_pickle(str(f))
    assert got[0].lr == (30, 40) and got[1].ul is None
    # nothing else resolves: a cache file cannot run code, and it holds a list of CharBox
    f.write_bytes(b"cos\nsystem\n(S'true'\ntR.")
    with pytest.raises(pickle.UnpicklingError):
        atocr.load_ocr_pickle(str(f))
    f.write_bytes(pickle.dumps({"a": 1}, 2))
    with pytest.raises(pickle.UnpicklingError):
        atocr.load_ocr_pickle(str(f))
    bad = _py2_style_box_pickle("alignToOCR", [(b"a", (1, 2), (4, 8))]).replace(b"U\x05width", b"U\x05wedth")
    f.write_bytes(bad)
    with pytest.raises(pickle.UnpicklingError):
        atocr.load_ocr_pickle(str(f))


def test_chars_from_llocs_half_to_even_and_rejects():
    from text_alignment_amd import alignToOCR as atocr
    out = []
    atocr.chars_from_llocs([('a', 10.46), ('~', 20.0), ('', 25.0), ('b', 30.5), ('c', 31.5)], 100, 7, 47, out)
    assert [(c.char, c.ul, c.lr) for c in out] == [
        ('a', (100, 7), (110, 47)),          # 10.46 -> "10.5" -> 110.5 rounds half-to-even to 110
        ('b', (125, 7), (130, 47)),          # '~' and '' advance the left edge but are dropped
        ('c', (130, 7), (132, 47))]


def test_to_json_dict_quantile():
    from text_alignment_amd import alignToOCR as atocr
    d = atocr.to_JSON_dict([atocr.CharBox('do', (50, 90), (68, 130))], [100, 220, 340, 470])
    assert d == {'median_line_spacing': 125.0, 'syl_boxes': [{'syl': 'do', 'ul': [50, 90], 'lr': [68, 130]}]}
    json.dumps({'syl_boxes': d['syl_boxes'], 'median_line_spacing': float(d['median_line_spacing'])})


def test_expand_abbreviations_golden_strings():
    from text_alignment_amd import alignToOCR as atocr
    for c in load_golden("glue.json")["process_cases"]:
        chars = [atocr.CharBox(ch, ul, lr) for ch, ul, lr in c["chars"]]
        out = atocr.expand_abbreviations(chars)
        assert ''.join(x.char for x in out) == c["expanded_ocr"], c["name"]


@pytest.mark.gpu
def test_process_golden_end_to_end():
    """process() -> to_JSON_dict() with the OCR characters canned (as in the golden capture):
    exercises abbreviation expansion, the HIP aligner, gap insertion, syllable grouping, rotation
    and the JSON layout against the reference's own output."""
    from text_alignment_amd import alignToOCR as atocr
    from text_alignment_amd.page import PreparedPage
    g = load_golden("glue.json")
    for c in g["process_cases"]:
        chars = [atocr.CharBox(ch, ul, lr) for ch, ul, lr in c["chars"]]
        page = PreparedPage(c["img_dim"], c["raw_dim"], c["angle"], [], c["peak_locs"])
        saved = atocr.perform_ocr_with_ocropus
        atocr.perform_ocr_with_ocropus = lambda strips, model, wkdir_name=None, parallel=2: list(chars)
        try:
            res = atocr.process(page, c["transcript"], None, seq_align_params=c["params"])
        finally:
            atocr.perform_ocr_with_ocropus = saved
        syl_boxes, image, peaks, all_chars = res
        js = atocr.to_JSON_dict(syl_boxes, peaks)
        js["median_line_spacing"] = float(js["median_line_spacing"])
        assert js == c["json"], c["name"]
        assert ''.join(x.char for x in all_chars) == c["expanded_ocr"]


def test_llocs_parser_matches_reference():
    """tests/golden/llocs.json: the reference's own perform_ocr_with_ocropus (alignToOCR.py:128-184)
    run on canned .llocs files (tools/gen_golden.py: gen_llocs) -- reject / blank classes that still
    advance the position, x.5 positions on odd and even strip offsets (np.round is half-to-even),
    utf-8 characters, an empty line strip."""
    from conftest import load_golden
    from text_alignment_amd import alignToOCR as atocr
    g = load_golden("llocs.json")
    assert len(g["cases"]) >= 6
    ties = 0
    for c in g["cases"]:
        got = []
        for s in c["strips"]:
            llocs = atocr.parse_llocs_text("".join(line + "\n" for line in s["llocs"]))
            ties += sum(1 for _, x in llocs if abs(x * 2 - round(x * 2)) < 1e-9 and abs(x - round(x)) > 0.25)
            atocr.chars_from_llocs(llocs, s["offset_x"], s["offset_y"], s["offset_y"] + s["height"], got)
        assert [[b.char, [int(b.ul[0]), int(b.ul[1])], [int(b.lr[0]), int(b.lr[1])]] for b in got] == c["chars"]
    assert ties >= 20            # the fixture does exercise the half-to-even cases


def test_token_encoding_and_list_rebuild_fast_paths_equal_the_token_by_token_forms():
    """textSeqCompare.encode_tokens / ops_to_alignment (host side of perform_alignment, no GPU): single-character strings
    take numpy passes, anything else -- bigrams (textSeqCompare.py:185-186), numbers, mixed -- the token-by-token dict;
    both number tokens by first appearance, so ids (and with them alignments) never depend on the path.  The rebuilt
    lists are the reference's: a column shows the next unconsumed token or '_' (textSeqCompare.py:116-162)."""
    from text_alignment_amd import textSeqCompare as tsc

    def by_token(*seqs):
        ids, out = {}, []
        for seq in seqs:
            out.append([ids.setdefault(tok, len(ids)) for tok in seq])
        return out, ids

    def rebuild(ops, t, o):
        tra, oc, i, j = [], [], 0, 0
        for op in ops:
            tra.append(t[i] if op != 2 else '_'); oc.append(o[j] if op != 1 else '_')
            i += op != 2; j += op != 1
        return tra, oc
    rng = np.random.default_rng(9)
    for trial in range(120):
        n, m = (int(v) for v in rng.integers(0, 30, 2))
        t = [chr(int(c)) for c in rng.integers(97, 104, n)]
        o = [chr(int(c)) for c in rng.integers(97, 106, m)]
        if trial % 4 == 1:
            t = list(zip(t, t[1:]))                                  # bigram tuples
        if trial % 4 == 2:
            o = [ord(c) for c in o]                                  # numbers
        if trial % 4 == 3 and t:
            t = list(u"dūß€\U0001F600 _~"[:len(t)])       # beyond Latin-1, beyond the BMP, the gap marker itself
        (a, b), ids = tsc.encode_tokens(t, o)
        (a2, b2), ids2 = by_token(t, o)
        assert a.tolist() == a2 and b.tolist() == b2 and ids == ids2, (t, o)
        assert a.dtype == np.int32 and b.dtype == np.int32
        i = j = 0
        ops = []
        while i < len(t) or j < len(o):
            op = int(rng.choice([k for k, ok in ((0, i < len(t) and j < len(o)), (1, i < len(t)), (2, j < len(o))) if ok]))
            ops.append(op); i += op != 2; j += op != 1
        assert tsc.ops_to_alignment(np.array(ops, dtype=np.uint8), t, o) == rebuild(ops, t, o)
    assert tsc.ops_to_alignment(np.zeros(0, np.uint8), [], []) == ([], [])
    assert tsc.encode_tokens([], [])[0][0].tolist() == []
    # tokens the code-point pass cannot encode (a lone surrogate: UnicodeEncodeError, a ValueError) go token by token
    # like any other hashable, and a generator is read once, by whichever path numbers it
    (a, b), ids = tsc.encode_tokens(list(u"ab\ud800a"), iter(u"ba"))
    assert (a.tolist(), b.tolist(), ids) == ([0, 1, 2, 0], [1, 0], {u"a": 0, u"b": 1, u"\ud800": 2})
    (a, b), ids = tsc.encode_tokens((c for c in "abca"), (c for c in ("bc", "a")))
    assert (a.tolist(), b.tolist()) == ([0, 1, 2, 0], [3, 0])
