"""text_alignment_amd/page_batch.py (the array form of the page glue process_batch runs) against the
reference's own outputs (tests/golden/glue.json, llocs.json) and against the object-by-object
functions of alignToOCR.py.  CPU only: the aligner here is oracle/nw_oracle.py."""
import numpy as np
import pytest

from conftest import load_golden


def _ops(tra, ocr):
    return np.array([1 if b == '_' and a != '_' else (2 if a == '_' else 0) for a, b in zip(tra, ocr)], dtype=np.uint8)


def test_array_glue_reproduces_reference_pages():
    from oracle import nw_oracle
    from text_alignment_amd import alignToOCR as atocr, latinSyllabification as latsyl, page_batch as pb
    from text_alignment_amd.page import Dim
    g = load_golden("glue.json")
    for c in g["process_cases"]:
        chars = [ch for ch, _, _ in c["chars"]]
        boxes = np.array([[ul[0], ul[1], lr[0], lr[1]] for _, ul, lr in c["chars"]], dtype=np.int64).reshape(-1, 4)
        if any(len(ch) != 1 for ch in chars):
            continue
        text, idx = pb.expand_abbreviations(''.join(chars), np.arange(len(chars)), latsyl.abbreviations)
        assert text == c["expanded_ocr"], c["name"]
        tr = c["transcript"]
        tra, ocr = nw_oracle.perform_alignment(list(tr), list(text), c["params"])
        syls = latsyl.syllabify_text(tr)
        assert pb.plain_page(tr, syls)
        which, sb = pb.syllable_boxes_arrays(tr, syls, _ops(tra, ocr), boxes[idx])
        sb = pb.rotate_boxes(sb, -1 * c["angle"], Dim(*c["img_dim"]), Dim(*c["raw_dim"]))
        named = [s for s in syls if len(s) >= 1]
        seq = pb.BoxSeq([named[w] for w in which], sb, atocr.CharBox)
        js = atocr.to_JSON_dict(seq, c["peak_locs"])
        js["median_line_spacing"] = float(js["median_line_spacing"])
        assert js == c["json"], c["name"]


def test_chars_of_batch_matches_reference_llocs():
    """the decoder's (t, class) arrays -> boxes, on the reference's llocs fixture: positions are put
    back into timestep form (scale 1: raw width == T - 32) so that x = t - 16 reproduces the file"""
    from text_alignment_amd import page_batch as pb
    g = load_golden("llocs.json")
    for c in g["cases"]:
        codec, dec_t, dec_c, dec_n, T, xmin, ymin, ymax = ["", "~"], [], [], [], [], [], [], []
        for s in c["strips"]:
            n = 0
            for line in s["llocs"]:
                ch, x = line.split("\t")
                if ch not in codec:
                    codec.append(ch)
                # x has one decimal: carry it as t = 10 x + 16 on a line whose scale is 1/10
                dec_t.append(int(round(float(x) * 10)) + 16)
                dec_c.append(codec.index(ch))
                n += 1
            dec_n.append(n); T.append(100032)
            xmin.append(s["offset_x"]); ymin.append(s["offset_y"]); ymax.append(s["offset_y"] + s["height"])
        off = np.concatenate([[0], np.cumsum(dec_n)[:-1]]).astype(np.int64)
        cps = pb.codec_code_points(codec)
        line, cp, boxes = pb.chars_of_batch(np.array(dec_t), np.array(dec_c), np.array(dec_n, dtype=np.int64), off,
                                            np.array(T), np.full(len(T), 10000), np.array(xmin), np.array(ymin),
                                            np.array(ymax), cps, 16)
        got = [[chr(int(a)), [int(b[0]), int(b[1])], [int(b[2]), int(b[3])]] for a, b in zip(cp, boxes)]
        assert got == c["chars"]
        # the array form (rounds 3-5) against the same fixture
        line2, cp2, boxes2 = pb.chars_of_batch_numpy(np.array(dec_t), np.array(dec_c), np.array(dec_n, dtype=np.int64), off,
                                                    np.array(T), np.full(len(T), 10000), np.array(xmin), np.array(ymin),
                                                    np.array(ymax), cps, 16)
        assert np.array_equal(line, line2) and np.array_equal(cp, cp2) and np.array_equal(boxes, boxes2)
    assert pb.codec_code_points(["", " ", "~", "ab"]) is None


def test_native_character_loop_equals_the_array_form_on_random_batches():
    """ta_host_chars_of_batch (one native call) against chars_of_batch_numpy on random decoder outputs: ragged lines, empty
    lines, dropped classes at line starts and ends, negative positions (t < pad), scales that put x on .x5 boundaries
    (where the reference's "%.1f" decides) and on exact halves after the offset (round half to even)"""
    from text_alignment_amd import page_batch as pb
    rng = np.random.default_rng(17)
    for trial in range(60):
        nlines = int(rng.integers(0, 40))
        dec_n = rng.integers(0, 30, size=nlines).astype(np.int64)
        if trial % 5 == 0 and nlines:
            dec_n[rng.integers(0, nlines)] = 0
        cap = dec_n + rng.integers(0, 5, size=nlines)
        dec_off = np.concatenate([[0], np.cumsum(cap)[:-1]]).astype(np.int64) if nlines else np.zeros(0, np.int64)
        total = int(cap.sum()) if nlines else 0
        T = rng.integers(40, 3000, size=nlines).astype(np.int64)
        dec_t = np.zeros(max(total, 1), np.int32); dec_c = np.zeros(max(total, 1), np.int32)
        ncls = 12
        for b in range(nlines):
            ts = np.sort(rng.integers(0, T[b], size=dec_n[b]))
            dec_t[dec_off[b]:dec_off[b] + dec_n[b]] = ts
            dec_c[dec_off[b]:dec_off[b] + dec_n[b]] = rng.integers(0, ncls, size=dec_n[b])
        # widths that make (t - 16) * raw_w / (T - 32) land on multiples of 0.05 and 0.25 often
        raw_w = np.where(rng.random(nlines) < 0.5, (T - 32) * rng.integers(1, 4, size=nlines) // rng.choice([1, 2, 4, 20], size=nlines),
                         rng.integers(10, 4000, size=nlines)).astype(np.int64)
        raw_w = np.maximum(raw_w, 1)
        x_min = rng.integers(-5, 300, size=nlines).astype(np.int64)
        y_min = rng.integers(0, 3000, size=nlines).astype(np.int64)
        y_max = y_min + rng.integers(20, 90, size=nlines)
        cps = rng.integers(97, 123, size=ncls).astype(np.int64)
        cps[[0, 2]] = -1                                                   # '' and '~'
        a = pb.chars_of_batch(dec_t, dec_c, dec_n, dec_off, T, raw_w, x_min, y_min, y_max, cps, 16)
        b_ = pb.chars_of_batch_numpy(dec_t, dec_c, dec_n, dec_off, T, raw_w, x_min, y_min, y_max, cps, 16)
        assert all(np.array_equal(u, v) for u, v in zip(a, b_)), trial
    # counts that point outside the decoder's arrays (they come from the device) are refused, not read
    from text_alignment_amd import _native
    with pytest.raises(_native.NativeArgumentError, match="outside the decoder"):
        pb.chars_of_batch(np.zeros(4, np.int32), np.zeros(4, np.int32), np.array([3, 9]), np.array([0, 3]), np.array([100, 100]),
                          np.array([50, 50]), np.zeros(2, np.int64), np.zeros(2, np.int64), np.ones(2, np.int64), np.array([97]), 16)


def test_array_glue_equals_object_glue_on_random_pages():
    """random OCR strings with gaps on both sides, two text lines, abbreviations: the array path and
    alignToOCR.align_page give the same boxes and the same syllable indices"""
    from oracle import nw_oracle
    from text_alignment_amd import alignToOCR as atocr, latinSyllabification as latsyl, page_batch as pb
    from text_alignment_amd.page import Dim
    rng = np.random.default_rng(5)
    words = "dominus deus meus alleluia gloria patri et filio cuius eius in excelsis laudate dns alla".split()
    pages, all_boxes, nbox = [], [], 0
    for trial in range(40):
        tr = " ".join(words[int(i)] for i in rng.integers(0, len(words) - 2, size=int(rng.integers(3, 30))))
        noisy = []
        for ch in tr:
            u = rng.random()
            if u < 0.1:
                continue
            noisy.append(ch if u < 0.8 else "abcdeilmnostu^ "[int(rng.integers(0, 15))])
            if rng.random() < 0.08:
                noisy.append("xq"[int(rng.integers(0, 2))])
        if trial % 5 == 0:
            noisy[2:2] = list("dns")
        boxes = np.array([[30 + 20 * k, 100 if k < len(noisy) // 2 else 220, 48 + 20 * k, 140 if k < len(noisy) // 2 else 262]
                          for k in range(len(noisy))], dtype=np.int64).reshape(-1, 4)
        objs = [atocr.CharBox(ch, b[0:2], b[2:4]) for ch, b in zip(noisy, boxes)]
        angle = float(rng.uniform(-3, 3))
        dims = (Dim(1000, 800), Dim(980, 790))
        text, idx = pb.expand_abbreviations(''.join(noisy), np.arange(len(noisy)), latsyl.abbreviations)
        expanded = atocr.expand_abbreviations(list(objs))
        assert text == ''.join(o.char for o in expanded)
        al = nw_oracle.perform_alignment(list(tr), list(text), None)
        want_idx = []
        want, _ = atocr.align_page(tr, expanded, angle, dims[0], dims[1], None, alignment=al, indices=want_idx,
                                   expanded=True)
        which, sb = pb.syllable_boxes_arrays(tr, latsyl.syllabify_text(tr), _ops(*al), boxes[idx])
        sb = pb.rotate_boxes(sb, -1 * angle, dims[0], dims[1])
        assert which.tolist() == want_idx
        assert sb.tolist() == [[int(b.ulx), int(b.uly), int(b.lrx), int(b.lry)] for b in want]
        pages.append((tr, latsyl.syllabify_text(tr), _ops(*al), idx + nbox, angle, want_idx, sb))
        all_boxes.append(boxes)
        nbox += len(boxes)
    # the same pages through the many-pages form, three at a time and all at once
    all_boxes = np.concatenate(all_boxes)
    for chunk in (3, len(pages)):
        for a in range(0, len(pages), chunk):
            part = pages[a:a + chunk]
            got = pb.syllable_boxes_batch([p[0] for p in part], [p[1] for p in part], [p[2] for p in part],
                                          [p[3] for p in part], all_boxes, [p[4] for p in part],
                                          [dims[0]] * len(part), [dims[1]] * len(part))
            for p, (which, sb) in zip(part, got):
                assert which.tolist() == p[5] and sb.tolist() == p[6].tolist()
    assert pb.syllable_boxes_batch([], [], [], [], all_boxes, [], [], []) == []


def test_syllable_span_fast_path_equals_sequential_search():
    """page_batch._syllable_spans_fast (array arithmetic) against the reference's one-by-one search
    (alignToOCR.py:297-324 with str.find standing for the regex): equal wherever it applies, and it
    declines (None) whenever its premise does not hold."""
    from text_alignment_amd import page_batch as pb
    from text_alignment_amd import latinSyllabification as latsyl
    rng = np.random.default_rng(3)
    words = "dominus deus meus alleluia gloria patri et filio a e spiritui sancto".split()

    def slow(tr, syls):
        cur, first, last = 0, [], []
        for syl in syls:
            if not syl:
                continue
            p = tr.find(syl, cur)
            assert p >= 0
            cur = p + len(syl)
            first.append(p); last.append(cur - 1)
        return first, last

    taken = 0
    for k in range(200):
        tr = " ".join(words[int(i)] for i in rng.integers(0, len(words), size=int(rng.integers(1, 40))))
        if k % 5 == 1:
            tr = tr.replace(" ", "  ", 2)                      # runs of spaces
        syls = [s for w in tr.split() for s in latsyl.syllabify_word(w)] if hasattr(latsyl, "syllabify_word") \
            else latsyl.syllabify_text(tr)
        if k % 7 == 2:
            syls = syls[:len(syls) // 2] + [""] + syls[len(syls) // 2:]     # empty syllables are skipped
        got = pb._syllable_spans_fast(tr, syls)
        want = slow(tr, syls)
        if got is not None:
            taken += 1
            assert got[0].tolist() == want[0] and got[1].tolist() == want[1], (tr, syls)
    assert taken > 150
    # premise broken: a syllable list that does not cover the text, a syllable across a space, a tab
    assert pb._syllable_spans_fast("do mi nus", ["do", "mi"]) is None
    assert pb._syllable_spans_fast("do mi", ["dom", "i"]) is None
    assert pb._syllable_spans_fast("do\tmi", ["do", "mi"]) is None
    assert pb._syllable_spans_fast("", []) is None


def test_chunk_plan_of_the_page_pipeline():
    """alignToOCR.plan_chunks: pages grouped by recogniser, runs of C pages, no sliver at the end of a group, every page
    exactly once and in order within its group"""
    from text_alignment_amd import alignToOCR as atocr
    for n_a, n_b, C in [(64, 0, 16), (32, 32, 16), (20, 0, 16), (24, 0, 16), (25, 7, 16), (1, 1, 16), (0, 0, 16), (100, 3, 32)]:
        groups = [("A", list(range(0, 2 * n_a, 2)))] + ([("B", list(range(1, 2 * n_b, 2)))] if n_b else [])
        chunks = atocr.plan_chunks(groups, C)
        for name, ks in groups:
            mine = [c for c in chunks if c[0] == name]
            assert [k for _, c in mine for k in c] == ks                       # all pages, in order, once
            sizes = [len(c) for _, c in mine]
            assert all(s == C for s in sizes[:-1]) or len(sizes) == 1 or sizes[:-2] == [C] * (len(sizes) - 2)
            if ks:
                assert sizes[-1] >= min(len(ks), C // 2) and max(sizes) < C + C // 2 + (C % 2 == 0)
            else:
                assert sizes == [0]
        # groups are not interleaved: the pipeline changes recogniser once per group
        assert [c[0] for c in chunks] == sorted([c[0] for c in chunks])
        # a half-size LEADING chunk (round 6): taken from the first group only, and only while more than a chunk and a half
        # of it remains; every page still exactly once and in order
        led = atocr.plan_chunks(groups, C, (C // 2,))
        for name, ks in groups:
            assert [k for nm, c in led if nm == name for k in c] == ks
        if n_a > C + C // 2:
            assert len(led[0][1]) == C // 2 and led[0][0] == "A"
            assert [len(c) for _, c in led[1:]] == [len(c) for _, c in atocr.plan_chunks(
                [("A", groups[0][1][C // 2:])] + groups[1:], C)]
        else:
            assert [len(c) for _, c in led] == [len(c) for _, c in chunks]
    assert [len(c) for _, c in atocr.plan_chunks([("A", list(range(64)))], 16, (8,))] == [8, 16, 16, 16, 8]


def test_native_syllable_union_equals_the_array_form():
    """ta_host_syllable_boxes against page_batch._syllable_union_numpy on random alignments: gaps on both sides, syllables
    with no OCR character under them, syllables spanning two text lines (the lower line's boxes only), several pages end to end"""
    from text_alignment_amd import page_batch as pb
    rng = np.random.default_rng(23)
    for trial in range(40):
        ncol = int(rng.integers(5, 400))
        ops = rng.choice([0, 0, 0, 1, 2], size=ncol).astype(np.uint8)
        nt, no = int((ops != 2).sum()), int((ops != 1).sum())
        if nt == 0:
            continue
        nboxes = no + 7
        boxes = np.zeros((nboxes, 4), np.int64)
        boxes[:, 0] = rng.integers(0, 2000, nboxes); boxes[:, 2] = boxes[:, 0] + rng.integers(1, 60, nboxes)
        boxes[:, 1] = rng.choice([90, 210, 330], nboxes); boxes[:, 3] = boxes[:, 1] + 40
        idx = rng.permutation(nboxes)[:no].astype(np.int64)
        # disjoint ascending syllable ranges over the transcript characters
        cuts = np.sort(rng.choice(np.arange(nt + 1), size=min(nt + 1, int(rng.integers(2, 30))), replace=False))
        first = cuts[:-1][::2].astype(np.int64)
        last = (cuts[1:][::2] - 1).astype(np.int64)
        keep = last >= first
        first, last = first[keep], last[keep]
        if len(first) == 0:
            continue
        low_a, box_a = pb._syllable_union(ops, idx, boxes, first, last)
        low_b, box_b = pb._syllable_union_numpy(ops, idx, boxes, first, last)
        assert np.array_equal(low_a, low_b), trial
        present = low_a > np.iinfo(np.int64).min
        assert np.array_equal(box_a[present], box_b[present]), trial
    with pytest.raises(AssertionError, match="not same length"):
        pb._syllable_union(np.array([0, 0, 1], np.uint8), np.array([0], np.int64), np.zeros((3, 4), np.int64),
                           np.array([0], np.int64), np.array([1], np.int64))
