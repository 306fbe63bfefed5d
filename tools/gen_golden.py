#!/usr/bin/env python3
"""tools/gen_golden.py -- capture golden vectors from the imported reference.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's textSeqCompare / latinSyllabification / alignToOCR
modules with harness-side stand-ins for the packages the reference imports but the hot
path does not need (unidecode: unused import at textSeqCompare.py:2; gamera: image toolkit
used only by preprocessing, alignToOCR.py:3-5), calls them on seeded inputs and writes the
inputs + outputs as small JSON fixtures under tests/golden/.  Only data is written: no
reference source text is copied.

    python tools/gen_golden.py            # everything except the 4096^2 case
    python tools/gen_golden.py --big      # also the 4096^2 case (~70 s)
    python tools/gen_golden.py --speed    # compare oracle/nw_ref_py.py speed with the reference
"""
import argparse
import builtins
import hashlib
import json
import os
import sys
import tempfile
import time
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True          # never drop __pycache__ into the reference tree

from tools.synth import ALPHA, synth_pair  # noqa: E402


# ----------------------------------------------------------------------------- harness
class _Point(object):
    def __init__(self, x, y):
        self.x, self.y = x, y


class _Dim(object):
    def __init__(self, ncols, nrows):
        self.ncols, self.nrows = ncols, nrows


class FakeImage(object):
    def __init__(self, ncols, nrows):
        self.dim = _Dim(ncols, nrows)
        self.ncols, self.nrows = ncols, nrows


def import_reference():
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    un = types.ModuleType("unidecode")
    un.unidecode = lambda s: s
    sys.modules["unidecode"] = un
    gam = types.ModuleType("gamera")
    core = types.ModuleType("gamera.core")
    core.init_gamera = lambda: None
    core.Point = _Point
    core.Dim = _Dim
    core.RGBPixel = lambda *a: a
    plugins = types.ModuleType("gamera.plugins")
    iu = types.ModuleType("gamera.plugins.image_utilities")
    iu.union_images = lambda imgs: None
    gam.core = core
    gam.plugins = plugins
    plugins.image_utilities = iu
    sys.modules.update({"gamera": gam, "gamera.core": core, "gamera.plugins": plugins,
                        "gamera.plugins.image_utilities": iu})
    builtins.reload = lambda mod: mod
    builtins.unicode = str
    import textSeqCompare as tsc
    import latinSyllabification as latsyl
    import alignToOCR as atocr
    return tsc, latsyl, atocr


def sha16(tra, ocr):
    return hashlib.sha256(("".join(tra) + "|" + "".join(ocr)).encode()).hexdigest()[:16]


def to_ops(tra, ocr):
    ops = []
    for a, b in zip(tra, ocr):
        ops.append(1 if b == '_' and a != '_' else (2 if a == '_' else 0))
    return ops


def rle(ops):
    out = []
    for op in ops:
        if out and out[-1][0] == op:
            out[-1][1] += 1
        else:
            out.append([op, 1])
    return out


SYSTEMS = [None, [10, -5, -7, -7], [5, -10, -2, -7, 0, -5], [11, -4, -2, -2, 0, 0],
           [1, -1, -1, -1], [3, -3, 0, 0, 0, 0], [8, -4, -7, -7, -3, 0],
           [2, -1, 1, -3, -1, 1], [0, 0, 0, 0, 0, 0], [4, -6, -9, -1, -2, -4]]


# ----------------------------------------------------------------------------- NW fixtures
def gen_nw_kat(tsc):
    cases = []
    s1 = 'Lorem ipsum dolor sit amet, consectetur adipiscing elit '
    s2 = 'LoLorem fipsudolor ..... sit eamet, c.nnr adizisdcing eelitellit'
    b1 = [s1[2 * x] + s1[2 * x + 1] for x in range(len(s1) // 2)]
    b2 = [s2[2 * x] + s2[2 * x + 1] for x in range(len(s2) // 2)]
    a, b = tsc.perform_alignment(b1, b2, scoring_system=[10, -5, -7, -7])
    cases.append(dict(name="KAT-1 bigram demo", transcript=b1, ocr=b2, scoring=[10, -5, -7, -7],
                      tra_align=a, ocr_align=b))
    a, b = tsc.perform_alignment(list(s1), list(s2))
    cases.append(dict(name="KAT-2 demo sentences as chars", transcript=list(s1), ocr=list(s2),
                      scoring=None, tra_align=a, ocr_align=b))
    minis = [("abc", "abc", None), ("", "abc", None), ("abc", "", None), ("", "", None),
             ("aaaa", "aa", None), ("aa", "aaaa", None), ("abcd", "xbcy", None),
             ("gloria in excelsis deo", "glorla inexcelsls xx deo", None),
             ("gloria in excelsis deo", "glorla inexcelsls xx deo", [5, -10, -2, -7, 0, -5]),
             ("abab", "baba", [1, -1, -1, -1]), ("abcabc", "cab", [3, -3, 0, 0, 0, 0]),
             ("a", "a", None), ("a", "b", None), ("a", "", None), ("", "b", None),
             ("ab_c", "a_bc", None),
             ("abcd", "xbcy", [8.5, -4.25, -7, -7, -3, 0]),
             ("dominus dixit ad me", "dns dixlt ad rne", [8, -4, -7.5, -6.5, -2.5, -0.5])]
    for t, o, sc in minis:
        a, b = tsc.perform_alignment(list(t), list(o), scoring_system=sc)
        cases.append(dict(name="mini %r/%r %r" % (t, o, sc), transcript=list(t), ocr=list(o),
                          scoring=sc, tra_align=a, ocr_align=b))
    # callable scoring function (textSeqCompare.py:27-29); the fixture names the function
    vow = set("aeiouy")
    fns = {
        "vowel_class": lambda a, b: 6 if a == b else (1 if (a in vow) == (b in vow) else -5),
        "ord_distance": lambda a, b: 5 - abs(ord(a) - ord(b)),
    }
    for fname, fn in fns.items():
        for t, o in [("gloria in excelsis deo", "glorla inexcelsls xx deo"),
                     ("alleluia", "allaluya"), ("abc", "")]:
            a, b = tsc.perform_alignment(list(t), list(o), scoring_system=[fn, -7, -6, -2, -1])
            cases.append(dict(name="callable %s %r/%r" % (fname, t, o), transcript=list(t),
                              ocr=list(o), scoring_fn=fname, scoring_gaps=[-7, -6, -2, -1],
                              tra_align=a, ocr_align=b))
    # error behaviour (textSeqCompare.py:41-42)
    errs = []
    for bad in ([1, 2, 3], [1, 2, 3, 4, 5], [1, 2, 3, 4, 5, 6, 7], []):
        try:
            tsc.perform_alignment(list("ab"), list("ab"), scoring_system=bad)
            errs.append(dict(scoring=bad, raises=None))
        except ValueError as e:
            errs.append(dict(scoring=bad, raises="ValueError", message=str(e)))
    return dict(cases=cases, errors=errs)


def gen_nw_random(tsc, count=320, seed=20260101):
    rng = np.random.default_rng(seed)
    cases = []
    for k in range(count):
        asz = [2, 4, 27][k % 3]
        n = int(rng.integers(0, 41))
        m = int(rng.integers(0, 41))
        t = [ALPHA[int(c)] for c in rng.integers(0, asz, size=n)]
        if k % 2 == 0:       # noisy copy
            o = []
            for ch in t:
                u = rng.random()
                if u < 0.1:
                    continue
                o.append(ALPHA[int(rng.integers(0, asz))] if u < 0.25 else ch)
                if rng.random() < 0.08:
                    o.append(ALPHA[int(rng.integers(0, asz))])
            o = (o + [ALPHA[int(c)] for c in rng.integers(0, asz, size=m)])[:m]
        else:
            o = [ALPHA[int(c)] for c in rng.integers(0, asz, size=m)]
        sc = SYSTEMS[int(rng.integers(0, len(SYSTEMS)))]
        a, b = tsc.perform_alignment(list(t), list(o), scoring_system=sc)
        cases.append(dict(t="".join(t), o="".join(o), scoring=sc, tra="".join(a), ocr="".join(b)))
    return dict(seed=seed, cases=cases)


def gen_nw_synth(tsc, big=False):
    specs = [(64, 64, 1234, None), (500, 500, 1234, None), (500, 500, 1235, None),
             (300, 700, 1236, None), (700, 300, 1237, None), (257, 255, 1238, None),
             (1, 300, 1239, None), (300, 1, 1240, None), (255, 1025, 1241, None),
             (200, 200, 99, [11, -10, -2, -7, 0, -5]), (200, 200, 99, [5, -4, -7, -2, -5, 0]),
             (513, 511, 77, [10, -5, -7, -7]), (640, 640, 78, [1, -1, -1, -1]),
             (1024, 1000, 1242, None), (2048, 2048, 1234, None)]
    if big:
        specs.append((4096, 4096, 1234, None))
    out = []
    for n, m, seed, sc in specs:
        t, o = synth_pair(n, m, seed)
        t0 = time.perf_counter()
        arg = np.array(sc) if (sc is not None and seed == 99) else sc   # ndarray form, as evaluate_text_alignment.py:192 passes rows
        a, b = tsc.perform_alignment(t, o, scoring_system=arg)
        dt = time.perf_counter() - t0
        ops = to_ops(a, b)
        out.append(dict(n=n, m=m, seed=seed, scoring=sc, align_len=len(a), sha16=sha16(a, b),
                        ops_rle=rle(ops), ref_seconds=round(dt, 3), t_head="".join(t[:20])))
        print("synth", n, m, seed, sc, len(a), sha16(a, b), "%.2fs" % dt, flush=True)
    return dict(cases=out)


# ----------------------------------------------------------------------------- glue fixtures
def run_process(atocr, transcript, chars, peak_locs, angle, img_dim, raw_dim, params=None):
    """alignToOCR.process() with preprocessing and the OCR seam replaced by canned data."""
    atocr.preproc.preprocess_images = lambda raw: (FakeImage(*img_dim), None, angle)
    atocr.preproc.identify_text_lines = lambda a, b: ([], list(peak_locs), None)
    boxes = [atocr.CharBox(c, ul, lr) for (c, ul, lr) in chars]
    atocr.perform_ocr_with_ocropus = lambda strips, model, wkdir_name, parallel: list(boxes)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            res = atocr.process(FakeImage(*raw_dim), transcript, None, seq_align_params=params,
                                wkdir_name="wk_golden")
        finally:
            os.chdir(cwd)
    syl_boxes, image, lines_peak_locs, all_chars = res
    js = atocr.to_JSON_dict(syl_boxes, lines_peak_locs)
    js["median_line_spacing"] = float(js["median_line_spacing"])
    expanded = "".join(c.char for c in all_chars)
    return js, expanded


def layout_chars(text, per_line=20, x0=50, dx=20, w=18, y0=90, dy=120, h=40):
    chars = []
    for k, ch in enumerate(text):
        line, col = divmod(k, per_line)
        ul = (x0 + dx * col, y0 + dy * line)
        chars.append((ch, ul, (ul[0] + w, ul[1] + h)))
    return chars


def gen_glue(latsyl, atocr):
    out = {}
    words = 'quaecumque ejus michi antiphonum assistens alleluya dixit extra exhibeamus'
    out["syllabify_demo"] = dict(inp=words, syls=latsyl.syllabify_text(words))
    # more syllabification cases; every word has a vowel (vowel-less words hang the
    # reference at latinSyllabification.py:71 and are never fed to it)
    texts = ["dominus dixit ad me filius meus es tu alleluia", "euouae cuius eius",
             "gloria in excelsis deo et in terra pax hominibus bonae voluntatis",
             "sanctus sanctus sanctus dominus deus sabaoth", "christe eleison kyrie",
             "a e i o u y", "", "  ave  maria ", "quia fecit michi magna qui potens est",
             "exaudi nos domine quoniam benigna est misericordia tua",
             "laudate dominum omnes gentes", "phtha thronus flos frater stella"]
    out["syllabify_cases"] = [dict(inp=t, syls=latsyl.syllabify_text(t)) for t in texts]
    out["syllabify_words"] = [dict(inp=w, syls=latsyl.syllabify_word(w)) for w in
                              ["alleluia", "quoniam", "sanctus", "excelsis", "exhibeamus", "aeterna",
                               "cuius", "eius", "euouae", "a", "ya", "ix", "patris", "spiritui",
                               "saeculorum", "christus", "psalmus", "thronum", "ejus", "michi"]]
    out["abbreviations"] = {k: v for k, v in latsyl.abbreviations.items()}

    # KAT-4 and variants: process() -> to_JSON_dict()
    cases = []
    tr = 'dominus dixit ad me filius meus es tu alleluia'
    oc = 'dns dixlt ad rne fllius rneus es tu alla'
    base = dict(transcript=tr, chars=layout_chars(oc), peak_locs=[100, 220, 340, 470],
                angle=0, img_dim=[1000, 800], raw_dim=[1000, 800], params=None)
    variants = [("KAT-4", base)]
    v = dict(base); v["params"] = [10, -5, -7, -7]; variants.append(("KAT-4 len4 params", v))
    v = dict(base); v["angle"] = 2.5; v["img_dim"] = [1040, 860]; v["raw_dim"] = [1000, 800]
    variants.append(("rotated 2.5deg padded", v))
    v = dict(base); v["angle"] = -1.25; v["img_dim"] = [1020, 812]; v["raw_dim"] = [1000, 800]
    variants.append(("rotated -1.25deg padded", v))
    tr2 = 'gloria in excelsis deo et in terra pax hominibus'
    oc2 = 'glorla inexcelsls xx de^ et ln terrā pax homlnibus'
    v = dict(base); v["transcript"] = tr2; v["chars"] = layout_chars(oc2, per_line=17)
    v["peak_locs"] = [95, 214, 333, 455, 570]
    variants.append(("abbrev ^ and macron, 3 lines", v))
    tr3 = 'laudate dominum omnes gentes laudate eum omnes populi'
    oc3 = 'laudatedūs omnesgentes lauda te eum oēs populi'
    v = dict(base); v["transcript"] = tr3; v["chars"] = layout_chars(oc3, per_line=12, dy=100)
    v["peak_locs"] = [80, 180, 290, 395, 500]
    variants.append(("abbrev dus overlap, narrow lines", v))
    tr4 = 'sanctus sanctus sanctus'
    oc4 = 'xx'
    v = dict(base); v["transcript"] = tr4; v["chars"] = layout_chars(oc4)
    variants.append(("almost empty ocr", v))
    for name, spec in variants:
        js, expanded = run_process(atocr, spec["transcript"],
                                   [(c, tuple(ul), tuple(lr)) for (c, ul, lr) in spec["chars"]],
                                   spec["peak_locs"], spec["angle"], spec["img_dim"],
                                   spec["raw_dim"], spec["params"])
        cases.append(dict(name=name, transcript=spec["transcript"],
                          chars=[[c, list(ul), list(lr)] for (c, ul, lr) in spec["chars"]],
                          peak_locs=spec["peak_locs"], angle=spec["angle"], img_dim=spec["img_dim"],
                          raw_dim=spec["raw_dim"], params=spec["params"], json=js,
                          expanded_ocr=expanded))
        print("glue", name, len(js["syl_boxes"]), "boxes", flush=True)
    # seeded random pages: transcript from a word list, OCR = noisy copy, 18 chars per line
    vocab = ("dominus deus meus alleluia gloria patri et filio spiritui sancto sicut erat in principio "
             "nunc semper saecula saeculorum amen laudate eum omnes gentes quoniam confirmata est "
             "super nos misericordia eius veritas manet aeternum").split()
    rng = np.random.default_rng(4242)
    for k in range(8):
        nw = int(rng.integers(6, 30))
        tr = " ".join(vocab[int(i)] for i in rng.integers(0, len(vocab), size=nw))
        oc = []
        for ch in tr:
            u = rng.random()
            if u < 0.06:
                continue
            oc.append("abcdefghilmnorstu "[int(rng.integers(0, 18))] if u < 0.18 else ch)
            if rng.random() < 0.04:
                oc.append("il.t"[int(rng.integers(0, 4))])
        oc = "".join(oc)
        per = int(rng.integers(14, 28))
        nlines = (len(oc) + per - 1) // per
        peaks = [100 + 120 * i for i in range(nlines + 1)]
        chars = layout_chars(oc, per_line=per)
        js, expanded = run_process(atocr, tr, chars, peaks, 0, [1200, 1600], [1200, 1600], None)
        cases.append(dict(name="random page %d" % k, transcript=tr,
                          chars=[[c, list(ul), list(lr)] for (c, ul, lr) in chars],
                          peak_locs=peaks, angle=0, img_dim=[1200, 1600], raw_dim=[1200, 1600],
                          params=None, json=js, expanded_ocr=expanded))
    out["process_cases"] = cases

    # rotate_bbox on its own.  Dimensions are even and differ by even amounts so that the
    # reference's Python-2 integer divisions (alignToOCR.py:91,95-96) and this Python-3 run agree.
    rots = []
    for (ul, lr, ang, od, td) in [((50, 90), (68, 130), 0, (1000, 800), (1000, 800)),
                                  ((50, 90), (68, 130), 3.0, (1040, 860), (1000, 800)),
                                  ((400, 300), (480, 352), -2.0, (1100, 900), (1000, 800)),
                                  ((10, 10), (990, 790), 0.5, (1010, 808), (1000, 800)),
                                  ((123, 457), (223, 499), 90, (800, 800), (800, 800)),
                                  ((5, 5), (25, 45), 0.5, (1000, 800), (1000, 800)),
                                  ((500, 400), (501, 401), 45, (1000, 800), (1000, 800))]:
        cb = atocr.CharBox('x', ul, lr)
        r = atocr.rotate_bbox(cb, ang, _Dim(*od), _Dim(*td))
        rots.append(dict(ul=list(ul), lr=list(lr), angle=ang, orig_dim=list(od), target_dim=list(td),
                         out_ul=[int(r.ul[0]), int(r.ul[1])], out_lr=[int(r.lr[0]), int(r.lr[1])]))
    out["rotate_cases"] = rots
    return out


def gen_preproc():
    """Pure-numpy helpers of textAlignPreprocessing.py (no Gamera call inside them):
    moving_avg_filter :147, calculate_peak_prominence :59, find_peak_locations :113,
    vertically_coincide :38 -- captured on seeded projection profiles."""
    import textAlignPreprocessing as preproc
    rng = np.random.default_rng(777)
    out = {"profiles": [], "coincide": []}
    for k in range(6):
        nrows = int(rng.integers(300, 800))
        nlines = int(rng.integers(3, 14))
        y = np.zeros(nrows)
        centers = np.sort(rng.integers(40, nrows - 40, size=nlines))
        for c0 in centers:                      # text lines: bumps of ink in the row projection
            wdt = float(rng.integers(6, 20))
            amp = float(rng.integers(100, 900))
            y += amp * np.exp(-0.5 * ((np.arange(nrows) - c0) / wdt) ** 2)
        y += rng.integers(0, 30, size=nrows)
        y = np.floor(y)
        fs = [30, 30, 10, 5][k % 4]
        sm = preproc.moving_avg_filter(y, fs)
        peaks = preproc.find_peak_locations(sm)
        peaks_tol = preproc.find_peak_locations(sm, tol=0.5)
        ranked = preproc.find_peak_locations(sm, ranked=True)
        proms = [float(preproc.calculate_peak_prominence(sm, i)) for i in range(0, nrows, 7)]
        out["profiles"].append(dict(data=[int(v) for v in y], filter_size=fs,
                                    smoothed=[float(v) for v in sm], peaks=[int(p) for p in peaks],
                                    peaks_tol05=[int(p) for p in peaks_tol],
                                    ranked=[[int(a), float(b)] for a, b in ranked],
                                    prominence_every7=proms))
    for k in range(60):
        a = [int(rng.integers(0, 500)), int(rng.integers(0, 500)), int(rng.integers(1, 120)), int(rng.integers(1, 80))]
        out["coincide"].append(dict(args=a, result=bool(preproc.vertically_coincide(*a))))
    for flat in ([5, 5, 5, 5], [1, 2, 3, 4, 5], [0, 3, 3, 0, 1, 0], [0, 0, 0]):
        out["profiles"].append(dict(data=flat, filter_size=0, smoothed=None,
                                    peaks=[int(p) for p in preproc.find_peak_locations(np.array(flat, float))],
                                    peaks_tol05=None, ranked=None, prominence_every7=None))
    return out


MEI_DOC = """<?xml version="1.0" encoding="UTF-8"?>
<mei xmlns="http://www.music-encoding.org/ns/mei" meiversion="3.9.9">
<music><facsimile><surface xml:id="surf1">
%s
</surface></facsimile>
<body><mdiv><score><section><staff n="1"><layer n="1">
%s
</layer></staff></section></score></mdiv></body></music></mei>
"""


def mei_case(seed, nsyl):
    """A synthetic Neon-style MEI page: neume components with zones, one <syllable> per neume,
    and text boxes below some of them (others share a text box or have none)."""
    rng = np.random.default_rng(seed)
    zones, syls, boxes = [], [], []
    x, y, zid = 100, 300, 0
    for k in range(nsyl):
        ncs = []
        for _ in range(int(rng.integers(1, 4))):
            zid += 1
            w, h = int(rng.integers(20, 40)), int(rng.integers(20, 40))
            yy = y + int(rng.integers(-30, 30))
            zones.append('<zone xml:id="z%d" ulx="%d" uly="%d" lrx="%d" lry="%d"/>' % (zid, x, yy, x + w, yy + h))
            ncs.append('<nc xml:id="nc%d" facs="z%d"/>' % (zid, zid))
            x += w + int(rng.integers(0, 10))
        syls.append('<syllable xml:id="syl%d"><neume xml:id="n%d">%s</neume></syllable>' % (k, k, "".join(ncs)))
        x += int(rng.integers(10, 60))
        if x > 1500:
            x, y = 100, y + 400
    # text boxes: one per two or three neumes, placed a line below; some stretches have none
    text = "do mi nus de us me us al le lu ia glo ri a pa tri".split()
    bx, by, k = 100, 300, 0
    while by < y + 1 and k < 60:
        w = int(rng.integers(60, 220))
        if rng.random() < 0.8:
            boxes.append([text[k % len(text)], [bx, by + 110], [bx + w, by + 170]])
        bx += w + int(rng.integers(0, 40))
        k += 1
        if bx > 1500:
            bx, by = 100, by + 400
    return MEI_DOC % ("\n".join(zones), "\n".join(syls)), boxes


def gen_mei():
    """writeToMEI.add_text_to_mei_file (writeToMEI.py:41-145) on synthetic MEI documents.  The
    reference indexes syllable boxes as (text, ul, lr) sequences; ids come from np.random, seeded."""
    import xml.etree.ElementTree as ET
    sys.modules["xml.etree.cElementTree"] = ET            # removed in Python 3.9
    import writeToMEI as mei
    cases = []
    for seed, nsyl, spacing in [(1, 12, 240.0), (2, 30, 250.5), (3, 45, 180.0), (4, 6, 400.0)]:
        doc, boxes = mei_case(seed, nsyl)
        ET.register_namespace('', 'http://www.music-encoding.org/ns/mei')
        root = ET.fromstring(doc)
        tree = ET.ElementTree(root)
        np.random.seed(1000 + seed)
        syl_boxes = [(b[0], tuple(b[1]), tuple(b[2])) for b in boxes]
        tree, all_bboxes, assign_lines = mei.add_text_to_mei_file(tree, syl_boxes, spacing)
        cases.append({"seed": 1000 + seed, "doc": doc, "boxes": boxes, "med_line_spacing": spacing,
                      "xml": ET.tostring(tree.getroot(), encoding="unicode"),
                      "all_bboxes": all_bboxes, "assign_lines": assign_lines})
    broken = '<mei xmlns="http://www.music-encoding.org/ns/mei" meiversion="3.9.9"><a xlink:href="x"/></mei>'
    return {"cases": cases, "repair": {"in": broken, "out": mei.repair_xml(broken)},
            "intersect": [[a, b, c, d, mei.intersect(a, b, c, d)] for a, b, c, d in
                          [((0, 0), (10, 10), (5, 5), (20, 8)), ((0, 0), (10, 10), (10, 10), (20, 20)),
                           ((0, 0), (4, 9), (1, 2), (3, 5)), ((5, 5), (6, 6), (0, 0), (1, 1))]]}


def gen_llocs(atocr):
    """perform_ocr_with_ocropus (alignToOCR.py:128-184) itself, with the shell-out to ocropus-rpred
    (:147) replaced by a stub that drops canned `_i.llocs` files (the tool's output format,
    "%s\\t%.1f\\n" per character) where the reference then reads them: pins the .llocs -> CharBox
    parser -- right-edge to left-edge shift, `~` / empty classes that still advance the position,
    np.round half-to-even on x + offset_x, utf-8 -- to the reference."""
    rng = np.random.default_rng(77)
    alphabet = list("abcdefghilmnopqrstuv") + [" ", " ", "~", "", "\u016b", "\u0113", "^", "9", "_"]

    class Strip(object):
        def __init__(self, ox, oy, h):
            self.offset_x, self.offset_y, self.height = ox, oy, h
            self.saved = None

        def save_image(self, path):
            self.saved = path

    cases = []
    for c in range(6):
        nstrips = [3, 1, 5, 2, 4, 1][c]
        strips, files = [], []
        for k in range(nstrips):
            strips.append(Strip(int(rng.integers(0, 90)), 100 + 120 * k + int(rng.integers(0, 7)),
                                int(rng.integers(30, 70))))
            x, lines = 0.0, []
            nchar = 0 if (c == 3 and k == 1) else int(rng.integers(1, 25))
            for _ in range(nchar):
                x += float(rng.integers(1, 60)) / 2.0 if rng.random() < 0.6 else float(rng.integers(1, 400)) / 10.0
                lines.append("%s\t%.1f" % (alphabet[int(rng.integers(0, len(alphabet)))], x))
            files.append(lines)
        wk = "wk_llocs_%d" % c

        def fake_check_call(cmd, shell=False, _files=files, _wk=wk):
            assert shell and "ocropus-rpred" in cmd and "--llocs" in cmd and _wk in cmd
            for k, lines in enumerate(_files):
                with open("./%s/_%d.llocs" % (_wk, k), "w", encoding="utf-8") as f:
                    f.write("".join(line + "\n" for line in lines))
        os.makedirs(wk)
        real = atocr.subprocess.check_call
        atocr.subprocess.check_call = fake_check_call
        try:
            chars = atocr.perform_ocr_with_ocropus(strips, "some_model.pyrnn.gz", wk, parallel=2)
        finally:
            atocr.subprocess.check_call = real
        assert [s.saved for s in strips] == ["./%s/_%d.png" % (wk, k) for k in range(nstrips)]
        cases.append({"strips": [{"offset_x": s.offset_x, "offset_y": s.offset_y, "height": s.height,
                                  "llocs": lines} for s, lines in zip(strips, files)],
                      "chars": [[b.char, [int(b.ul[0]), int(b.ul[1])], [int(b.lr[0]), int(b.lr[1])]] for b in chars]})
    return {"cases": cases}


def speed_check(tsc):
    from oracle import nw_ref_py
    for n, m in [(500, 500), (1000, 1000)]:
        t, o = synth_pair(n, m, 1234)
        t0 = time.perf_counter(); r1 = tsc.perform_alignment(t, o); t1 = time.perf_counter()
        r2 = nw_ref_py.perform_alignment(t, o); t2 = time.perf_counter()
        assert r1 == r2
        print("n=%d m=%d reference %.2fs (%.3g cells/s)  port %.2fs (%.3g cells/s)  ratio %.2f"
              % (n, m, t1 - t0, n * m / (t1 - t0), t2 - t1, n * m / (t2 - t1), (t2 - t1) / (t1 - t0)))


def dump(name, obj):
    path = os.path.join(GOLD, name)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(obj, f, ensure_ascii=False, indent=None, separators=(",", ":"))
        f.write("\n")
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--speed", action="store_true")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    tsc, latsyl, atocr = import_reference()
    os.makedirs(GOLD, exist_ok=True)
    if args.speed:
        speed_check(tsc)
        return
    only = set(args.only.split(",")) if args.only else None
    if not only or "kat" in only:
        dump("nw_kat.json", gen_nw_kat(tsc))
    if not only or "random" in only:
        dump("nw_random_small.json", gen_nw_random(tsc))
    if not only or "glue" in only:
        dump("glue.json", gen_glue(latsyl, atocr))
    if not only or "preproc" in only:
        dump("preproc.json", gen_preproc())
    if not only or "mei" in only:
        dump("mei.json", gen_mei())
    if not only or "llocs" in only:
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as scratch:
            os.chdir(scratch)
            try:
                out = gen_llocs(atocr)
            finally:
                os.chdir(cwd)
        dump("llocs.json", out)
    if not only or "synth" in only:
        dump("nw_synth.json", gen_nw_synth(tsc, big=args.big))


if __name__ == "__main__":
    main()
