# -*- coding: utf-8 -*-
"""Array form of the page glue between the two kernels, for batches of pages.

`alignToOCR.process` does for ONE page what the reference does (alignToOCR.py:247-330), object by
object: a CharBox per OCR character, a list splice per abbreviation, a regular-expression search per
syllable.  That is ~7 ms of interpreter time per page -- fifty times the kernels' share.  This
module computes the same results for a whole batch on numpy arrays:

  decoded (t, class) arrays of every line -> character code points + boxes    (alignToOCR.py:160-182)
  abbreviation expansion on the page's string + an index array               (alignToOCR.py:251-264)
  one NW launch for all pages on integer token ids                           (alignToOCR.py:273)
  gap insertion / syllable search / box union per page, vectorised           (alignToOCR.py:285-324)
  un-rotation of all boxes of a page at once                                 (alignToOCR.py:327-328)

The per-syllable search needs no regular expression here: the pattern the reference builds for a
syllable (first letter, then every further letter with `_*` in front, alignToOCR.py:299-304) matches
in the aligned transcript exactly where the plain syllable occurs in the transcript ITSELF (the
aligned transcript minus its gap markers), so a `str.find` on the transcript plus the map from
transcript positions to alignment columns gives the same (start, end).  Pages whose syllables or
transcript contain characters that mean something to `re`, or a literal '_', take the object path
(`alignToOCR.align_page`), as do recogniser codecs with multi-character entries.

Results are exposed through `BoxSeq`, a read-only sequence that builds CharBox objects when they
are asked for, so callers that only need the arrays (the sharded driver, the benchmark) never pay
for a hundred thousand small objects.
"""
import numpy as np

try:
    from collections.abc import Sequence
except ImportError:                      # pragma: no cover
    from collections import Sequence

_META = set(r'\.^$*+?{}[]|()_')


class BoxSeq(Sequence):
    """A sequence of CharBox over parallel arrays: chars (list of str) and an int array [k, 4] of
    (ulx, uly, lrx, lry); element access builds the CharBox the reference would have built."""

    def __init__(self, chars, boxes, make):
        self.chars, self.boxes, self._make = chars, np.asarray(boxes).reshape(-1, 4), make

    def __len__(self):
        return len(self.chars)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return [self[i] for i in range(*k.indices(len(self)))]
        if k < 0:
            k += len(self)
        b = self.boxes[k]
        return self._make(self.chars[k], b[0:2], b[2:4])


def edge_positions(x, x_min):
    """int(np.round(float('%.1f' % x) + x_min)) of the reference (alignToOCR.py:167-170), for arrays:
    x as the .llocs file carries it (one decimal), then round half to even."""
    v = np.asarray(x, dtype=np.float64)
    t = v * 10.0
    f = np.floor(t)
    frac = t - f
    one_dec = np.where(frac > 0.5, f + 1.0, f) / 10.0
    for i in np.nonzero(np.abs(frac - 0.5) < 1e-6)[0]:          # (near-)ties: let printf decide
        one_dec[i] = float('%.1f' % v[i])
    return np.rint(one_dec + x_min).astype(np.int64)


def codec_code_points(codec):
    """code point of each class's character, -1 for the classes the reference drops ('~' and ''),
    None if some entry is longer than one character (object path)."""
    cps = np.full(len(codec), -1, dtype=np.int64)
    for k, s in enumerate(codec):
        if len(s) > 1:
            return None
        if len(s) == 1 and s != '~':
            cps[k] = ord(s)
    return cps


def chars_of_batch(dec_t, dec_c, dec_n, dec_off, T, raw_w, x_min, y_min, y_max, cps, pad):
    """All characters of all lines, reading order (line, then position): (line of each character, code point, boxes
    [k, 4]) with the dropped classes removed -- ONE native call (ta_host_chars_of_batch, host arithmetic in the library;
    chars_of_batch_numpy below is the same in array operations and its cross-check, tests/test_page_batch.py)."""
    from . import _native
    nlines = len(dec_n)
    dec_n = np.ascontiguousarray(dec_n, dtype=np.int64)
    total = int(dec_n.sum()) if nlines else 0
    if total == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 4), np.int64)
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)            # noqa: E731
    dec_t, dec_c = np.ascontiguousarray(dec_t, dtype=np.int32), np.ascontiguousarray(dec_c, dtype=np.int32)
    dec_off, T, raw_w, x_min, y_min, y_max, cps = (i64(a) for a in (dec_off, T, raw_w, x_min, y_min, y_max, cps))
    line, cp, boxes = np.empty(total, np.int64), np.empty(total, np.int64), np.empty((total, 4), np.int64)
    count = np.zeros(1, np.int64)
    _native.check(_native.lib.ta_host_chars_of_batch(
        dec_t.ctypes.data, dec_c.ctypes.data, dec_n.ctypes.data, dec_off.ctypes.data, T.ctypes.data, raw_w.ctypes.data,
        x_min.ctypes.data, y_min.ctypes.data, y_max.ctypes.data, cps.ctypes.data, len(cps), int(pad), nlines,
        min(len(dec_t), len(dec_c)), line.ctypes.data, cp.ctypes.data, boxes.ctypes.data, count.ctypes.data), "ta_host_chars_of_batch")
    k = int(count[0])
    return line[:k], cp[:k], boxes[:k]


def chars_of_batch_numpy(dec_t, dec_c, dec_n, dec_off, T, raw_w, x_min, y_min, y_max, cps, pad):
    """chars_of_batch in numpy array operations (the form rounds 3-5 ran; kept as the native loop's cross-check).

    dec_*: the decoder's output arrays (entry i of line b at dec_off[b] + i, dec_n[b] entries);
    T, raw_w, x_min, y_min, y_max: per line.  Returns (line of each character, code point, boxes
    [k, 4]) with the dropped classes removed -- they still move the left edge of what follows."""
    nlines = len(dec_n)
    line = np.repeat(np.arange(nlines), dec_n)
    if line.size == 0:
        return line, np.zeros(0, np.int64), np.zeros((0, 4), np.int64)
    first = np.zeros(nlines + 1, dtype=np.int64)
    np.cumsum(dec_n, out=first[1:])
    within = np.arange(line.size) - first[line]
    src = dec_off[line] + within
    t, c = dec_t[src].astype(np.float64), dec_c[src]
    scale = raw_w.astype(np.float64) / (T - 2 * pad)
    right = edge_positions((t - pad) * scale[line], x_min[line].astype(np.float64))
    left = np.empty_like(right)
    left[1:] = right[:-1]
    starts = first[:-1][dec_n > 0]
    left[starts] = x_min[line[starts]]
    cp = cps[c]
    keep = cp >= 0
    boxes = np.stack([left, y_min[line], right, y_max[line]], axis=1)
    return line[keep], cp[keep], boxes[keep]


def expand_abbreviations(text, idx, abbreviations):
    """alignToOCR.py:251-264 on (string, index array): every occurrence of an abbreviation is
    replaced by its expansion, letter k of the abbreviation lending its box (index) to the letters of
    segment k.  Returns (text, idx)."""
    if not any(abb in text for abb in abbreviations):
        return text, idx
    idx = list(idx)
    for abb, segments in abbreviations.items():
        at = text.find(abb)
        while at != -1:
            ins_t, ins_i = [], []
            for k, segment in enumerate(segments):
                ins_t.append(segment)
                ins_i.extend([idx[at + k]] * len(segment))
            text = text[:at] + ''.join(ins_t) + text[at + len(abb):]
            idx[at:at + len(abb)] = ins_i
            at = text.find(abb)
    return text, np.asarray(idx, dtype=np.int64)


def plain_page(transcript, syls):
    """can this page's syllable search be done without `re`?"""
    return not (_META & set(transcript))


def syllable_boxes_arrays(transcript, syls, ops, box_of_ocr):
    """alignToOCR.py:285-324 on arrays.  ops: the alignment columns (0 pair, 1 transcript token over
    a gap, 2 gap over an OCR token); box_of_ocr: [m, 4] boxes of the (expanded) OCR characters.
    Returns (which, boxes [k, 4]): for each syllable that is aligned to at least one OCR character,
    its index among the non-empty syllables and the union of the boxes on its lowest text line."""
    ops = np.asarray(ops)
    has_t = ops != 2
    has_o = ops != 1
    col_of_t = np.flatnonzero(has_t)                     # alignment column of transcript character k
    ocr_at_col = np.cumsum(has_o) - 1                    # OCR index of a column that has one
    assert len(col_of_t) == len(transcript), 'all_chars not same length as alignment'
    ncol = len(ops)
    ulx = np.full(ncol + 1, np.iinfo(np.int64).max, dtype=np.int64)
    uly_min = ulx.copy()
    lrx = np.full(ncol + 1, np.iinfo(np.int64).min, dtype=np.int64)
    lry = lrx.copy()
    uly_max = lrx.copy()
    cols = np.flatnonzero(has_o)
    b = box_of_ocr[ocr_at_col[cols]]
    ulx[cols], uly_min[cols], lrx[cols], lry[cols], uly_max[cols] = b[:, 0], b[:, 1], b[:, 2], b[:, 3], b[:, 1]
    starts, ends, which = [], [], []
    cur = 0
    k = -1
    for syl in syls:
        if len(syl) < 1:
            continue
        k += 1
        p = transcript.find(syl, cur)
        if p < 0:
            raise AttributeError("'NoneType' object has no attribute 'start'")     # as re.search(...).start()
        cur = p + len(syl)
        starts.append(col_of_t[p])
        ends.append(col_of_t[cur - 1] + 1)
        which.append(k)
    if not starts:
        return np.zeros(0, np.int64), np.zeros((0, 4), np.int64)
    starts, ends, which = np.asarray(starts), np.asarray(ends), np.asarray(which)
    bounds = np.stack([starts, ends], axis=1).reshape(-1)          # reduceat over [s0, e0, s1, e1, ...]
    low = np.maximum.reduceat(uly_max, bounds)[0::2]               # lowest text line under the syllable
    present = low > np.iinfo(np.int64).min
    # keep the boxes on that line only (alignToOCR.py:318-320)
    # syllable of each column: +1 at a syllable's first column, -1 after its last (ranges are disjoint
    # and in order), running sum > 0 inside; the syllable's index is the number of starts so far - 1
    begun = np.bincount(starts, minlength=ncol + 2)                # (np.add.at costs 0.2 ms per call on 3 000 indices)
    mark = begun - np.bincount(ends, minlength=ncol + 2)
    inside = np.cumsum(mark)[:ncol + 1] > 0
    seg_of_col = np.cumsum(begun)[:ncol + 1] - 1
    on_line = inside & (uly_max == low[np.maximum(seg_of_col, 0)])
    big, small = np.iinfo(np.int64).max, np.iinfo(np.int64).min
    out = np.stack([np.minimum.reduceat(np.where(on_line, ulx, big), bounds)[0::2],
                    np.minimum.reduceat(np.where(on_line, uly_min, big), bounds)[0::2],
                    np.maximum.reduceat(np.where(on_line, lrx, small), bounds)[0::2],
                    np.maximum.reduceat(np.where(on_line, lry, small), bounds)[0::2]], axis=1)
    return which[present], out[present]


def rotate_boxes(boxes, angle, orig_dim, target_dim):
    """rotate_bbox (alignToOCR.py:90-125) over an int array [k, 4]; the reference's Python 2
    divisions are floor divisions."""
    if len(boxes) == 0:
        return np.zeros((0, 4), dtype=np.int64)
    px, py = orig_dim.ncols // 2, orig_dim.nrows // 2
    dx = (orig_dim.ncols - target_dim.ncols) // 2
    dy = (orig_dim.nrows - target_dim.nrows) // 2
    a = angle * np.pi / 180
    s, c = np.sin(a), np.cos(a)
    x = boxes[:, 0::2] - px
    y = boxes[:, 1::2] - py
    nx = (x * c) - (y * s) + (px - dx)
    ny = (x * s) + (y * c) + (py - dy)
    rx = np.round(nx).astype('int16').astype(np.int64)
    ry = np.round(ny).astype('int16').astype(np.int64)
    return np.stack([rx[:, 0], ry[:, 0], rx[:, 1], ry[:, 1]], axis=1)


def _syllable_spans_fast(tr, syls):
    """(first, last) transcript positions of every non-empty syllable when the syllables are exactly
    the transcript's non-space characters in order, each inside one word -- what the syllabifier
    produces for plain text -- or None.  Then the reference's sequential search (alignToOCR.py:297-
    324: each syllable is looked for from where the one before ended) can only find syllable i at
    the i-th run of those characters: between the cursor and that run there are only spaces, and a
    syllable does not start with one."""
    syl_ne = [x for x in syls if x]
    if not syl_ne or ''.join(syl_ne) != tr.replace(' ', ''):
        return None
    ns = np.flatnonzero(np.frombuffer(tr.encode('utf-32-le'), dtype=np.uint32) != 32)
    lens = np.fromiter(map(len, syl_ne), dtype=np.int64, count=len(syl_ne))
    end = np.cumsum(lens)
    first, last = ns[end - lens], ns[end - 1]
    if not np.array_equal(last - first, lens - 1):        # a syllable that spans a space: search for it
        return None
    return first, last


def _syllable_union_numpy(ops, idx_all, boxes, first_t, last_t):
    """(low, box [k, 4]) per syllable in array operations -- the form rounds 3-5 ran, kept as the cross-check of the native
    loop (ta_host_syllable_boxes): the alignment columns of all pages are laid end to end, a syllable is a column range,
    and one reduceat per quantity serves every syllable of every page"""
    has_t = ops != 2
    has_o = ops != 1
    col_of_t = np.flatnonzero(has_t)
    ncol = len(ops)
    big, small = np.iinfo(np.int64).max, np.iinfo(np.int64).min
    ulx = np.full(ncol + 1, big, dtype=np.int64)
    uly_min = ulx.copy()
    lrx = np.full(ncol + 1, small, dtype=np.int64)
    lry = lrx.copy()
    uly_max = lrx.copy()
    cols = np.flatnonzero(has_o)
    b = boxes[idx_all]
    ulx[cols], uly_min[cols], lrx[cols], lry[cols], uly_max[cols] = b[:, 0], b[:, 1], b[:, 2], b[:, 3], b[:, 1]
    starts, ends = col_of_t[first_t], col_of_t[last_t] + 1
    bounds = np.stack([starts, ends], axis=1).reshape(-1)
    low = np.maximum.reduceat(uly_max, bounds)[0::2]
    begun = np.bincount(starts, minlength=ncol + 2)                # (np.add.at costs 0.2 ms per call on 3 000 indices)
    mark = begun - np.bincount(ends, minlength=ncol + 2)
    inside = np.cumsum(mark)[:ncol + 1] > 0
    seg_of_col = np.cumsum(begun)[:ncol + 1] - 1
    on_line = inside & (uly_max == low[np.maximum(seg_of_col, 0)])
    out = np.stack([np.minimum.reduceat(np.where(on_line, ulx, big), bounds)[0::2],
                    np.minimum.reduceat(np.where(on_line, uly_min, big), bounds)[0::2],
                    np.maximum.reduceat(np.where(on_line, lrx, small), bounds)[0::2],
                    np.maximum.reduceat(np.where(on_line, lry, small), bounds)[0::2]], axis=1)
    return low, out


def _syllable_union(ops, idx_all, boxes, first_t, last_t):
    """(low, box [k, 4]) per syllable: ONE native call (ta_host_syllable_boxes, host arithmetic in the library)"""
    from . import _native
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    idx_all = np.ascontiguousarray(idx_all, dtype=np.int64)
    boxes = np.ascontiguousarray(boxes, dtype=np.int64).reshape(-1, 4)
    first_t, last_t = np.ascontiguousarray(first_t, dtype=np.int64), np.ascontiguousarray(last_t, dtype=np.int64)
    k = len(first_t)
    low, out = np.empty(k, np.int64), np.empty((k, 4), np.int64)
    rc = _native.lib.ta_host_syllable_boxes(ops.ctypes.data, len(ops), idx_all.ctypes.data, len(idx_all), boxes.ctypes.data,
                                            len(boxes), first_t.ctypes.data, last_t.ctypes.data, k, low.ctypes.data, out.ctypes.data)
    if rc != 0 and b"not same length" in _native.lib.ta_last_error():
        raise AssertionError('all_chars not same length as alignment')      # the reference's assert, alignToOCR.py:291
    _native.check(rc, "ta_host_syllable_boxes")
    return low, out


def syllable_boxes_batch(transcripts, syls_list, ops_list, idx_list, boxes, angles, image_dims, raw_dims):
    """syllable_boxes_arrays + rotate_boxes for MANY pages at once (the per-page versions spend their time in numpy's
    per-call overhead on 2000-element arrays): the alignment columns of all pages are laid end to end, a syllable is a
    range of transcript characters, and one native loop serves every syllable of every page (_syllable_union).
    idx_list[k]: indices into `boxes` of page k's (expanded) OCR characters.  Returns per page (which, boxes [k, 4])."""
    npages = len(transcripts)
    if npages == 0:
        return []
    ops = np.concatenate([np.asarray(o) for o in ops_list]) if ops_list else np.zeros(0, np.uint8)
    idx_all = np.concatenate(idx_list) if idx_list else np.zeros(0, np.int64)
    toff = np.zeros(npages + 1, dtype=np.int64)
    np.cumsum([len(t) for t in transcripts], out=toff[1:])
    assert int((ops != 2).sum()) == toff[-1] and int((ops != 1).sum()) == len(idx_all), 'all_chars not same length as alignment'
    small = np.iinfo(np.int64).min
    first_t, last_t, which, page = [], [], [], []
    for k, (tr, syls) in enumerate(zip(transcripts, syls_list)):
        base = int(toff[k])
        found = _syllable_spans_fast(tr, syls)
        if found is None:                                 # the reference's search, one syllable at a time
            cur, first_k, last_k = 0, [], []
            for syl in syls:
                if len(syl) < 1:
                    continue
                p = tr.find(syl, cur)
                if p < 0:
                    raise AttributeError("'NoneType' object has no attribute 'start'")     # as re.search(...).start()
                cur = p + len(syl)
                first_k.append(p)
                last_k.append(cur - 1)
            found = (np.asarray(first_k, dtype=np.int64), np.asarray(last_k, dtype=np.int64))
        first_t.append(found[0] + base)
        last_t.append(found[1] + base)
        which.append(np.arange(len(found[0]), dtype=np.int64))
        page.append(np.full(len(found[0]), k, dtype=np.int64))
    empty = [(np.zeros(0, np.int64), np.zeros((0, 4), np.int64))] * npages
    which = np.concatenate(which) if which else np.zeros(0, np.int64)
    if len(which) == 0:
        return list(empty)
    page, first_t, last_t = np.concatenate(page), np.concatenate(first_t), np.concatenate(last_t)
    low, out = _syllable_union(ops, idx_all, boxes, first_t, last_t)
    present = low > small
    which, page, out = which[present], page[present], out[present]
    # un-rotation (rotate_bbox, alignToOCR.py:90-125) with each box's own page geometry
    px = np.array([d.ncols // 2 for d in image_dims], dtype=np.int64)[page]
    py = np.array([d.nrows // 2 for d in image_dims], dtype=np.int64)[page]
    dx = np.array([(a.ncols - b.ncols) // 2 for a, b in zip(image_dims, raw_dims)], dtype=np.int64)[page]
    dy = np.array([(a.nrows - b.nrows) // 2 for a, b in zip(image_dims, raw_dims)], dtype=np.int64)[page]
    ang = (-1 * np.asarray(angles, dtype=np.float64))[page] * np.pi / 180
    sn, cs = np.sin(ang)[:, None], np.cos(ang)[:, None]
    x = out[:, 0::2] - px[:, None]
    y = out[:, 1::2] - py[:, None]
    nx = (x * cs) - (y * sn) + (px - dx)[:, None]
    ny = (x * sn) + (y * cs) + (py - dy)[:, None]
    rx = np.round(nx).astype('int16').astype(np.int64)
    ry = np.round(ny).astype('int16').astype(np.int64)
    rot = np.stack([rx[:, 0], ry[:, 0], rx[:, 1], ry[:, 1]], axis=1)
    cuts = np.searchsorted(page, np.arange(npages + 1))
    return [(which[cuts[k]:cuts[k + 1]], rot[cuts[k]:cuts[k + 1]]) for k in range(npages)]
