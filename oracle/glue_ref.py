"""CPU restatement of the page glue around the two kernels -- TEST INFRASTRUCTURE (the checker of
the page-level GPU tests), never imported by the product.

Follows the reference step by step on plain tuples instead of CharBox objects:
  chars_from_llocs    alignToOCR.py:155-182   .llocs lines of one strip -> character boxes
  expand              alignToOCR.py:251-264   abbreviation expansion on the box list
  syllable_json       alignToOCR.py:267-351   gap insertion, per-syllable regex search, box union,
                                              un-rotation (rotate_bbox :90-125), to_JSON_dict
A box is (char, ulx, uly, lrx, lry); a gap marker is (char, None, None, None, None).

Pinned (tests/test_oracle_glue.py, CPU) to outputs of the imported reference: tests/golden/llocs.json
(perform_ocr_with_ocropus on canned .llocs files) and tests/golden/glue.json (process() ->
to_JSON_dict() on 15 pages, rotate_bbox cases).  The alignment and the syllable list are INPUTS here:
they come from oracle/nw_oracle.py and from the syllabifier, which have fixtures of their own.
"""
import re

import numpy as np


def chars_from_llocs(lines, x_min, y_min, y_max):
    """alignToOCR.py:160-182: each character runs from the previous character's position to its
    own; '~' and '' are dropped but still move the position."""
    out = []
    left = x_min
    for line in lines:
        fields = line.rstrip('\n').split('\t')
        right = int(np.round(float(fields[1]) + x_min))          # half-to-even, alignToOCR.py:170
        if fields[0] not in ('~', ''):
            out.append((fields[0].replace('~', ''), left, y_min, right, y_max))
        left = right
    return out


def expand(chars, abbreviations):
    """alignToOCR.py:251-264: while the OCR string contains an abbreviation, splice in its
    expansion; letter k of the abbreviation lends its box to every letter of segment k."""
    chars = list(chars)
    for abb in abbreviations.keys():
        while True:
            text = ''.join(str(c[0]) for c in chars)
            at = text.find(abb)
            if at < 0:
                break
            spliced = []
            for k, segment in enumerate(abbreviations[abb]):
                donor = chars[at + k]
                spliced.extend((letter,) + tuple(donor[1:]) for letter in segment)
            chars[at:at + len(abb)] = spliced
    return chars


def unrotate(box, angle_deg, orig_cols, orig_rows, target_cols, target_rows):
    """rotate_bbox (alignToOCR.py:90-125) with the reference's Python 2 integer divisions."""
    px, py = orig_cols // 2, orig_rows // 2
    dx, dy = (orig_cols - target_cols) // 2, (orig_rows - target_rows) // 2
    a = angle_deg * np.pi / 180
    s, c = np.sin(a), np.cos(a)
    pts = []
    for x, y in ((box[1], box[2]), (box[3], box[4])):
        x, y = x - px, y - py
        pts.append(np.round([(x * c) - (y * s) + (px - dx), (x * s) + (y * c) + (py - dy)]).astype('int16'))
    return (box[0], int(pts[0][0]), int(pts[0][1]), int(pts[1][0]), int(pts[1][1]))


def syllable_json(syls, chars, tra_align, ocr_align, angle, image_dim, raw_dim, lines_peak_locs):
    """alignToOCR.py:267-351 after the aligner has run: `chars` are the expanded OCR boxes,
    (tra_align, ocr_align) the alignment of the transcript against their text; *_dim = (ncols, nrows)."""
    tra = ''.join(tra_align)
    ocr = ''.join(ocr_align)
    row = list(chars)
    for k, ch in enumerate(ocr):                                  # :285-287
        if ch == '_':
            row.insert(k, ('_', None, None, None, None))
    assert len(row) == len(tra), 'all_chars not same length as alignment: {} vs {}'.format(len(row), len(tra))
    found = []
    cursor = 0
    for syl in syls:                                              # :297-324
        if len(syl) < 1:
            continue
        pattern = syl if len(syl) == 1 else syl[0] + syl[1:-1].replace('', '_*') + syl[-1]
        hit = re.search(pattern, tra[cursor:])
        lo, hi = hit.start() + cursor, hit.end() + cursor
        cursor = hi
        boxes = [b for b in row[lo:hi] if b[3] is not None]
        if not boxes:
            continue
        if len(set(b[2] for b in boxes)) > 1:
            lowest = max(b[2] for b in boxes)
            boxes = [b for b in boxes if b[2] == lowest]
        found.append((syl, min(b[1] for b in boxes), min(b[2] for b in boxes),
                      max(b[3] for b in boxes), max(b[4] for b in boxes)))
    found = [unrotate(b, -1 * angle, image_dim[0], image_dim[1], raw_dim[0], raw_dim[1]) for b in found]   # :327-328
    return {'median_line_spacing': np.quantile(np.diff(lines_peak_locs), 0.75),                          # :338
            'syl_boxes': [{'syl': b[0], 'ul': [b[1], b[2]], 'lr': [b[3], b[4]]} for b in found]}
