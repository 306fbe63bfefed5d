# -*- coding: utf-8 -*-
"""Page driver -- drop-in for the reference module of the same name (reference
alignToOCR.py:24-351): same `process` / `perform_ocr_with_ocropus` / `to_JSON_dict` /
`CharBox` / `rotate_bbox` / `read_file` surface, same `syl_boxes` JSON.

What changed underneath (and nothing else): the OCR seam no longer writes PNG strips and shells
out to `ocropus-rpred` (alignToOCR.py:131-147) -- text-line strips go through the HIP line
recogniser in-process (text_alignment_amd.ocr) -- and the transcript/OCR alignment runs in the
HIP Needleman-Wunsch kernels (text_alignment_amd.textSeqCompare).  The glue between the two
kernels (abbreviation expansion, gap insertion, syllable grouping, rotation, JSON) is host-side
string/box work restated from the reference.

Gamera (the reference's image toolkit, alignToOCR.py:3-5) is not needed: a page arrives either
as a `text_alignment_amd.page.PreparedPage` carrying its text-line strips, or as a raw text-layer
image (numpy array), which `preproc` (text_alignment_amd.textAlignPreprocessing over the HIP kernels
of csrc/ta_preproc.hip) binarises, deskews and cuts into lines -- the two calls `process` makes at alignToOCR.py:216-218.
"""
import io
import json  # noqa: F401  (callers json.dump the result of to_JSON_dict, alignToOCR.py:434)
import os
import pickle
import re

import numpy as np

from . import latinSyllabification as latsyl
from . import textSeqCompare as tsc
from . import page as page_mod
from . import textAlignPreprocessing as preproc

parallel = 2                # kept for signature compatibility (alignToOCR.py:24): one GPU batch
median_line_mult = 2        # alignToOCR.py:25


class CharBox(object):
    __slots__ = ['char', 'ul', 'lr', 'ulx', 'lrx', 'uly', 'lry', 'width', 'height']

    def __init__(self, char, ul=None, lr=None):
        self.char = char
        if (ul is None) or (lr is None):        # a gap marker: no geometry (alignToOCR.py:41-44)
            self.ul = None
            self.lr = None
            return
        self.ul, self.lr = tuple(ul), tuple(lr)
        self.ulx, self.uly = ul[0], ul[1]
        self.lrx, self.lry = lr[0], lr[1]
        self.width = lr[0] - ul[0]
        self.height = lr[1] - ul[1]

    def __repr__(self):
        if self.ul and self.lr:
            return '{}: {}, {}'.format(self.char, self.ul, self.lr)
        return '{}: empty'.format(self.char)

    # __slots__ classes pickle by slot; unset geometry of gap markers is skipped
    def __getstate__(self):
        return {k: getattr(self, k) for k in self.__slots__ if hasattr(self, k)}

    def __setstate__(self, state):
        # a pickle the reference wrote (alignToOCR.py:435-436: its CharBox has no __getstate__) carries the
        # default state of a __slots__ class, the pair (None, {slot: value})
        if isinstance(state, tuple) and len(state) == 2:
            state = dict(state[0] or {}, **(state[1] or {}))
        for k, v in state.items():
            if k not in self.__slots__:
                raise pickle.UnpicklingError("CharBox has no slot %r" % (k,))
            setattr(self, k, v)


class _BoxUnpickler(pickle.Unpickler):
    """Loader of the OCR cache files of alignToOCR.py:225-233 (written at :435-436 and by
    evaluate_text_alignment.py:170-171 with protocol 2): a list of CharBox and nothing else.  The class may be
    named after the reference's module (`alignToOCR`, or `__main__` when the reference ran as a script) or
    after this one; coordinates may be numpy scalars (rotate_bbox's int16).  No other global resolves, so a
    cache file cannot run code -- the same rule as the model loader's (model_io.RestrictedUnpickler)."""
    _BOX_MODULES = ('alignToOCR', '__main__', 'text_alignment_amd.alignToOCR')

    def find_class(self, module, name):
        if name == 'CharBox' and module in self._BOX_MODULES:
            return CharBox
        if module in ('numpy.core.multiarray', 'numpy._core.multiarray') and name == 'scalar':
            import numpy.core.multiarray as ma
            return ma.scalar
        if module == 'numpy' and name == 'dtype':
            return np.dtype
        if (module, name) in (('copy_reg', '_reconstructor'), ('copyreg', '_reconstructor')):
            import copyreg
            return copyreg._reconstructor
        if (module, name) in (('__builtin__', 'object'), ('builtins', 'object')):
            return object
        if (module, name) == ('_codecs', 'encode'):       # how Python 3 writes bytes at protocol 2
            import _codecs
            return _codecs.encode
        raise pickle.UnpicklingError('global %s.%s is not allowed in an OCR cache file' % (module, name))


def load_ocr_pickle(path):
    """list[CharBox] from a cache file this package or the reference (Python 2, latin-1 strings) wrote"""
    with open(path, 'rb') as f:
        chars = _BoxUnpickler(f, encoding='latin1').load()
    if not isinstance(chars, list) or not all(isinstance(c, CharBox) for c in chars):
        raise pickle.UnpicklingError('an OCR cache file holds a list of CharBox')
    return chars


def clean_special_chars(inp):
    '''drops the reject class '~' from OCR output (alignToOCR.py:61-72)'''
    return inp.replace('~', '')


def read_file(fname):
    '''plaintext transcript of a page -> one string (alignToOCR.py:75-87)'''
    with open(fname, 'r') as f:
        rows = f.readlines()
    text = ' '.join(r for r in rows if not r[0] == '#')
    for junk in ('\n', '\r', '| '):
        text = text.replace(junk, '')
    return text


def rotate_bbox(cbox, angle, orig_dim, target_dim, radians=False):
    '''rotate a box about the centre of `orig_dim` and shift by half the size difference to
    `target_dim` (alignToOCR.py:90-125).  The reference runs on Python 2, where the three
    divisions at alignToOCR.py:91,95-96 are integer floor divisions.'''
    px, py = orig_dim.ncols // 2, orig_dim.nrows // 2
    dx = (orig_dim.ncols - target_dim.ncols) // 2
    dy = (orig_dim.nrows - target_dim.nrows) // 2
    if not radians:
        angle = angle * np.pi / 180
    s, c = np.sin(angle), np.cos(angle)

    def turn(x, y):
        x, y = x - px, y - py
        return (x * c) - (y * s) + (px - dx), (x * s) + (y * c) + (py - dy)

    ulx, uly = turn(cbox.ulx, cbox.uly)
    lrx, lry = turn(cbox.lrx, cbox.lry)
    new_ul = np.round([ulx, uly]).astype('int16')
    new_lr = np.round([lrx, lry]).astype('int16')
    return CharBox(cbox.char, new_ul, new_lr)


def rotate_bboxes(boxes, angle, orig_dim, target_dim, radians=False):
    """rotate_bbox over a whole page's boxes at once: the same float64 arithmetic element by
    element (one numpy pass instead of four np.round calls per box)."""
    if not boxes:
        return []
    px, py = orig_dim.ncols // 2, orig_dim.nrows // 2
    dx = (orig_dim.ncols - target_dim.ncols) // 2
    dy = (orig_dim.nrows - target_dim.nrows) // 2
    if not radians:
        angle = angle * np.pi / 180
    s, c = np.sin(angle), np.cos(angle)
    pts = np.array([[b.ulx, b.uly, b.lrx, b.lry] for b in boxes])
    x = pts[:, 0::2] - px
    y = pts[:, 1::2] - py
    nx = (x * c) - (y * s) + (px - dx)
    ny = (x * s) + (y * c) + (py - dy)
    rx = np.round(nx).astype('int16')
    ry = np.round(ny).astype('int16')
    return [CharBox(b.char, np.array([rx[k, 0], ry[k, 0]]), np.array([rx[k, 1], ry[k, 1]]))
            for k, b in enumerate(boxes)]


# --------------------------------------------------------------------------- OCR seam
_recognizers = {}


def _recognizer_for(ocropus_model):
    """`ocropus_model` as the reference passes it is a model file path (alignToOCR.py:390-405);
    a LineModel or a ready LineRecognizer is accepted as well."""
    from . import ocr
    if isinstance(ocropus_model, ocr.LineRecognizer):
        return ocropus_model
    key = id(ocropus_model) if not isinstance(ocropus_model, str) else os.path.abspath(ocropus_model)
    if key not in _recognizers:
        if isinstance(ocropus_model, str):
            from . import model_io
            model = model_io.load_pyrnn(ocropus_model)
        else:
            model = ocropus_model
        _recognizers[key] = (ocropus_model, ocr.LineRecognizer(model))
    return _recognizers[key][1]


def _edge_positions(xs, x_min):
    """int(np.round(float('%.1f' % x) + x_min)) of the reference (alignToOCR.py:167-170) for a
    whole line at once: x as the .llocs file carries it (one decimal), then round half to even."""
    v = np.asarray(xs, dtype=np.float64)
    t = v * 10.0
    f = np.floor(t)
    frac = t - f
    one_dec = np.where(frac > 0.5, f + 1.0, f) / 10.0
    for i in np.nonzero(np.abs(frac - 0.5) < 1e-6)[0]:          # (near-)ties: let printf decide
        one_dec[i] = float('%.1f' % v[i])
    return np.rint(one_dec + x_min).astype(np.int64).tolist()


def parse_llocs_text(text):
    """The `.llocs` wire format ocropus-rpred writes and the reference reads back
    (alignToOCR.py:157-170): one "char<TAB>x" line per decoded character, x with one decimal ->
    [(char, x)] as chars_from_llocs takes it."""
    out = []
    for line in text.split('\n'):
        if line == '':
            continue
        lsp = line.split('\t')
        out.append((lsp[0], float(lsp[1])))
    return out


def chars_from_llocs(llocs, x_min, y_min, y_max, all_chars):
    """One strip's (char, x) list -> CharBoxes appended to all_chars (alignToOCR.py:160-182).
    ocropus reports the RIGHT edge of each character, so a character's box runs from the
    previous character's position to its own; '~' and '' still advance the position."""
    if not llocs:
        return
    prev_xpos = x_min
    for (ch, _), cur_xpos in zip(llocs, _edge_positions([x for _, x in llocs], x_min)):
        if not (ch == '~' or ch == ''):
            all_chars.append(CharBox(clean_special_chars(ch), (prev_xpos, y_min), (cur_xpos, y_max)))
        prev_xpos = cur_xpos


def perform_ocr_with_ocropus(cc_strips, ocropus_model, wkdir_name=None, parallel=parallel):
    """Recognise every text-line strip of a page and return its characters in reading order
    (strip order, then x), as reference alignToOCR.py:128-184 does through `ocropus-rpred`.

    cc_strips: objects with offset_x, offset_y, height and either `prepared` (a (T, 48) array,
    ink = 1, already normalised and padded) or `pixels` (raw 2-D uint8 strip, white background:
    normalised on the device, csrc/ta_lineest.hip).  wkdir_name and `parallel` (the reference's
    temp directory and number of ocropus worker processes) are accepted and unused: all strips of
    the page go to the GPU in one batch and no host-side image work is left to spread.
    """
    from . import ocr
    rec = _recognizer_for(ocropus_model)
    prepared = page_mod.prepared_lines(list(cc_strips), workers=parallel)
    lines = [xs for xs, _ in prepared]
    widths = [w for _, w in prepared]
    decoded = rec.recognise(lines)
    all_chars = []
    for k, (strip, raw_w, dec) in enumerate(zip(cc_strips, widths, decoded)):
        llocs = rec.llocs(dec, int(rec.last_T[k]), raw_w)
        chars_from_llocs(llocs, strip.offset_x, strip.offset_y, strip.offset_y + strip.height, all_chars)
    return all_chars


# --------------------------------------------------------------------------- alignment glue
def expand_abbreviations(all_chars):
    """Replace every occurrence of an abbreviation in the OCR character list by its expansion;
    character k of the abbreviation lends its box to all letters of segment k
    (alignToOCR.py:251-264).  String positions index the character list, as in the reference."""
    ocr_str = ''.join(str(x.char) for x in all_chars)
    for abb, segments in latsyl.abbreviations.items():
        idx = ocr_str.find(abb)
        while idx != -1:
            ins = []
            for k, segment in enumerate(segments):
                donor = all_chars[k + idx]
                ins += [CharBox(ch, donor.ul, donor.lr) for ch in segment]
            all_chars = all_chars[:idx] + ins + all_chars[idx + len(abb):]
            ocr_str = ''.join(str(x.char) for x in all_chars)        # only after a replacement
            idx = ocr_str.find(abb)
    return all_chars


def syllable_boxes(syls, tra_align, all_chars, indices=None):
    """For each syllable of the transcript, the union of the OCR character boxes it is aligned
    to (alignToOCR.py:297-324).  `indices`, if given, receives for every emitted box the index of
    its syllable among the non-empty syllables of the transcript."""
    out = []
    offset = 0
    which = -1
    for syl in syls:
        if len(syl) < 1:
            continue
        which += 1
        if len(syl) == 1:
            pattern = syl
        else:
            pattern = syl[0] + syl[1:-1].replace('', '_*') + syl[-1]
        hit = re.search(pattern, tra_align[offset:])
        start, end = hit.start() + offset, hit.end() + offset
        offset = end
        boxes = [b for b in all_chars[start:end] if b.lr is not None]
        if not boxes:
            continue                          # aligned to nothing in the OCR
        if len(set(b.uly for b in boxes)) > 1:
            lowest = max(b.uly for b in boxes)       # spans two text lines: keep the lower one
            boxes = [b for b in boxes if b.uly == lowest]
        ul = (min(b.ulx for b in boxes), min(b.uly for b in boxes))
        lr = (max(b.lrx for b in boxes), max(b.lry for b in boxes))
        out.append(CharBox(syl, ul, lr))
        if indices is not None:
            indices.append(which)
    return out


def align_page(transcript, all_chars, angle, image_dim, raw_dim, seq_align_params=None,
               alignment=None, indices=None, expanded=False):
    """Everything `process` does after OCR (alignToOCR.py:247-328), on plain data.  `alignment`
    may carry a precomputed (tra_align, ocr_align) of the transcript against the expanded OCR
    string (the batched driver aligns all pages in one launch)."""
    all_chars = list(all_chars) if expanded else expand_abbreviations(list(all_chars))
    ocr = ''.join(x.char for x in all_chars)
    all_chars_copy = list(all_chars)

    if alignment is None:
        alignment = tsc.perform_alignment(list(transcript), list(ocr),
                                          scoring_system=seq_align_params, verbose=False)
    tra_align, ocr_align = alignment
    tra_align = ''.join(tra_align)
    ocr_align = ''.join(ocr_align)
    syls = latsyl.syllabify_text(transcript)

    # gaps of the OCR side get placeholder boxes (alignToOCR.py:285-292; a literal '_' in the OCR
    # text counts as a gap there too and trips the same assertion)
    ngaps = ocr_align.count('_')
    assert len(all_chars) + ngaps == len(tra_align), 'all_chars not same length as alignment: ' \
        '{} vs {}'.format(len(all_chars) + ngaps, len(tra_align))
    boxes = iter(all_chars)
    all_chars = [CharBox('_') if ch == '_' else next(boxes) for ch in ocr_align]

    syl_boxes = syllable_boxes(syls, tra_align, all_chars, indices)
    syl_boxes = rotate_bboxes(syl_boxes, -1 * angle, image_dim, raw_dim)
    return syl_boxes, all_chars_copy


def find_lines_all(pages, workers=1):
    """preprocessing + text-line finding of every page: (image_bin, image_eroded, angle, strips,
    peak locations) per page.  Page images (greyscale, colour or onebit, any numeric type: reduced to
    uint8 greyscale by preproc.to_grey_u8) go through the device kernels (preproc_gpu,
    csrc/ta_preproc.hip), several pages per batch so that the pipeline's data-dependent host decisions
    share their waits for the device; PreparedPages pass through.  `workers` -- the reference's
    `parallel` -- is accepted and unused: there is no host-side image work left to spread."""
    return preproc.find_lines_many(list(pages))


def _raw_dim(pg):
    """`raw_image.dim` of the reference (alignToOCR.py:328) for whatever stands for the raw page
    here: an object with .dim (PreparedPage, a Gamera-like image), or a bare pixel array /
    an object with .pixels, whose size is its shape."""
    dim = getattr(pg, "dim", None)
    if dim is not None and hasattr(dim, "ncols"):
        return dim
    px = getattr(pg, "pixels", pg)
    if not hasattr(px, "shape") or type(px).__module__.split(".")[0] != "torch":   # (a tensor -- possibly on the device -- has its shape)
        px = np.asarray(px)
    if len(px.shape) < 2:
        raise TypeError("a page is a PreparedPage, an image object with .dim, or a 2-D / 3-D pixel array")
    return page_mod.Dim(int(px.shape[1]), int(px.shape[0]))


def _process_batch_objects(rec, pages, raw_dims, found, strips_per_page, lines, widths, transcripts,
                           seq_align_params, indices_out):
    """process_batch, object by object (CharBox lists, alignToOCR.py:247-330 per page): the path for
    scoring callables / non-integral numbers and for codecs with multi-character entries."""
    decoded = rec.decoded(rec._last_state)
    chars_per_page, k = [], 0
    for strips in strips_per_page:
        all_chars = []
        for strip in strips:
            llocs = rec.llocs(decoded[k], int(rec.last_T[k]), widths[k])
            chars_from_llocs(llocs, strip.offset_x, strip.offset_y, strip.offset_y + strip.height, all_chars)
            k += 1
        chars_per_page.append(expand_abbreviations(all_chars))
    pairs = [(list(tr), [c.char for c in chars]) for tr, chars in zip(transcripts, chars_per_page)]
    alignments = tsc.perform_alignment_batch(pairs, seq_align_params)
    results = []
    for raw_dim, f, tr, chars, al in zip(raw_dims, found, transcripts, chars_per_page, alignments):
        image, angle, lp = f[0], f[2], f[4]
        idx = [] if indices_out is not None else None
        syl_boxes, all_chars_copy = align_page(tr, chars, angle, image.dim, raw_dim,
                                               seq_align_params, alignment=al, indices=idx, expanded=True)
        if indices_out is not None:
            indices_out.append(idx)
        results.append((syl_boxes, image, lp, all_chars_copy))
    return results


# process_batch runs batches larger than this as a pipeline over chunks of PIPELINE_CHUNK_PAGES pages (three stages per chunk,
# see the loop): while the recogniser kernels of chunk k run, the copy pool stages the rows of chunk k + 1 and this thread
# does the later stages of chunks k - 2 and k - 3 (characters and the aligner launch; syllable boxes).  One host thread
# besides the copy pool; results are those of the unchunked call.
# (Plain module attributes: this module reads no environment variables; tools/switches.py sets them for timing experiments.)
PIPELINE_CHUNK_PAGES = 16
PIPELINE_CHUNK_PAGES_RAW = 16        # raw strips: the device normaliser in front (32 while every chunk's first stage WAITED for
                                     # its measuring pass; round 6 enqueues it and waits a stage later: 16 is faster, 970 against 885)
PIPELINE_CHUNK_PAGES_IMAGES = 16     # page images: preprocessing in front, itself in batches of pages on page threads of its own (64 until
                                     # round 6: no faster -- 447-564 against 542-557 pages/s -- and its 1 920-line recogniser batch took a
                                     # 17.7 GB scratch buffer from the caching allocator and gave it back on every call).  Measured twice and
                                     # not kept: the line finding of the NEXT chunk started ahead on a thread of its own -- 551-575
                                     # pages/s against 625-636 without while the page threads launched from Python, 750-787 against
                                     # 802-807 with a stage per library call and four page threads (same box each; bound to the GPU's
                                     # NUMA node as bench.py binds a rank: 731-789 against 801-854, three runs each, alternating): the
                                     # stages of the chunks in flight are Python, and more threads of it contend for one interpreter lock
_side_streams = {}
WAIT_SECONDS = [0.0]                 # wall seconds the calling thread has spent WAITING for the device inside process_batch (a
                                     # running total: callers take differences): a pass's wall time minus this is its host work
TWO_STREAMS = True                   # consecutive chunks' recogniser kernels on two compute streams


def _ocr_streams(device):
    """two compute streams for the recogniser kernels of consecutive chunks: on ONE stream the projection of chunk k + 1
    cannot start before the last workgroup of chunk k's recurrence has finished, and a chunk's recurrence is one round of
    workgroups whose CUs free up one by one as the shorter lines end (mean / longest line = 0.71)"""
    import torch
    key = ("ocr", device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _side_streams:
        _side_streams[key] = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
    return _side_streams[key]


def _nw_stream(device):
    """the aligner's launches of a chunk go to a side stream: on the recogniser's stream they would queue behind the
    next chunk's recurrence and the host would wait for it"""
    import torch
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


# The call's FIRST chunk is half a chunk: nothing can be finished on the host before the first chunk's characters are back,
# so its latency stands in front of the whole pipeline (a 16-page chunk: ~12 ms until decoded, an 8-page one ~6).  Measured
# on 64 pages of page-locked rows: chunks 8, 16, 16, 16, 8 give 2 125-2 140 pages/s against 1 937 for 4 x 16 (device rows
# 2 223-2 239 against 2 095-2 111; pageable rows the same within their noise); a first chunk of 4 pages is too small to
# fill the chip (1 790), halving the last chunk as well gains nothing (profiles/r06_lead_chunk.txt).
LEAD_CHUNK_DIVISOR = 2


def plan_chunks(groups, C, lead=()):
    """[(recogniser, page indices)] in pipeline order: every group of pages (one recogniser each, in order of first
    appearance) cut into runs of C pages; a last run shorter than C / 2 joins the run before it (no sliver of a chunk
    whose kernels would not cover the next chunk's host stage).  lead: sizes of the call's FIRST chunks (taken from the
    first group while more than C pages of it remain)"""
    chunks = []
    lead = list(lead)
    for rec, ks in groups:
        a = 0
        while lead and len(ks) - a > C + lead[0]:
            chunks.append((rec, ks[a:a + lead[0]]))
            a += lead.pop(0)
        lead = []
        while True:
            b = len(ks) if len(ks) - (a + C) < C // 2 else min(a + C, len(ks))
            chunks.append((rec, ks[a:b]))
            a = b
            if a >= len(ks):
                break
    return chunks


def process_batch(pages, transcripts, ocropus_model, seq_align_params=None, indices_out=None,
                  parallel=parallel, arrays_out=None):
    """`process` for many pages at once: the strips of ALL pages go through the line recogniser
    in one batch (large batches: in chunks of PIPELINE_CHUNK_PAGES pages, host and device overlapped), the
    transcript/OCR alignments of a chunk's pages run in one NW launch, and the glue in
    between runs on arrays (text_alignment_amd.page_batch) -- the shape in which a GPU is worth
    using.  Per page the result equals process(page, transcript, model, seq_align_params).  `ocropus_model` is one model
    for all pages or a list with one per page.
    Returns a list of (syl_boxes, image, lines_peak_locs, all_chars); the two box lists are
    sequences that build their CharBox objects on access.  indices_out, if given, receives per page
    the index of each box's syllable among the transcript's non-empty syllables; arrays_out the
    boxes themselves as an int array [k, 4] (ulx, uly, lrx, lry)."""
    pages, transcripts = list(pages), list(transcripts)
    n = len(pages)
    # one model for all pages, or one per page (the reference's two manuscripts have a model each, alignToOCR.py:390-405):
    # pages are grouped by recogniser, every chunk has one, and the pipeline runs on across the groups
    if isinstance(ocropus_model, (list, tuple)):
        if len(ocropus_model) != n:
            raise ValueError("a list of models needs one entry per page")
        recs = [_recognizer_for(m) for m in ocropus_model]
    else:
        recs = [_recognizer_for(ocropus_model)] * n
    groups = {}
    for k, r in enumerate(recs):
        groups.setdefault(id(r), (r, []))[1].append(k)
    if not groups:
        groups[0] = (_recognizer_for(ocropus_model), [])
    # pages that still need the preprocessing kernels (page images) take larger chunks: their line finding waits for the
    # device several times per batch of pages, and those waits queue behind a previous chunk's recogniser
    # (measured at 64 pages, chunks of 8 / 16 / 32: normalised strips 1 410 / 1 515 / 1 515 pages/s, raw strips -- whose
    # normaliser returns its data-dependent widths with a device wait per chunk -- 670 / 866 / 1 005; page images ~450-500
    # whatever the chunk)
    images = any(not isinstance(pg, page_mod.PreparedPage) for pg in pages)
    raw = not images and any(st.prepared is None for pg in pages for st in getattr(pg, "strips", ()))
    C = PIPELINE_CHUNK_PAGES_IMAGES if images else (PIPELINE_CHUNK_PAGES_RAW if raw else PIPELINE_CHUNK_PAGES)
    lead = (C // LEAD_CHUNK_DIVISOR,) if LEAD_CHUNK_DIVISOR > 1 and not (images or raw) else ()   # (raw strips: measured, no gain)
    chunks = plan_chunks(list(groups.values()), C, lead)
    out_res, out_idx, out_arr = [None] * n, [None] * n, [None] * n
    def begin(job):
        rec, ks = job
        ctx = _pb_begin(rec, [pages[k] for k in ks], [transcripts[k] for k in ks], seq_align_params, parallel)
        ctx["page_ids"] = ks
        return ctx

    def collect(ctx):
        idx, arr = [], []
        res = _pb_finish_b(ctx, idx, arr)
        for j, k in enumerate(ctx["page_ids"]):
            out_res[k] = res[j]
            out_idx[k], out_arr[k] = idx[j], arr[j]

    def deliver():
        # one entry per page, in page order, whichever path each chunk took (a chunk whose alignment does not fit the
        # integer kernels goes object by object on its own; callers zip these lists with the pages)
        if indices_out is not None:
            indices_out.extend(out_idx)
        if arrays_out is not None:
            arrays_out.extend(out_arr)
        return out_res
    if len(chunks) == 1:
        ctx = begin(chunks[0])
        _pb_launch(ctx)
        _pb_transcripts(ctx)
        _pb_finish_a(ctx)
        collect(ctx)
        return deliver()
    import torch
    device = chunks[0][0].device
    streams = _ocr_streams(device)
    caller = torch.cuda.current_stream(device)
    for st_ in streams:
        st_.wait_stream(caller)                                  # whatever the caller enqueued before this call
    # Three stages per chunk: (1) begin + launch -- the recogniser's kernels enqueued; (2) _pb_finish_a, once the chunk's
    # characters are back: abbreviations and the chunk's ONE aligner launch; (3) collect, once the alignment columns are
    # back: syllable boxes.  The device must never run dry, so a new chunk is launched BEFORE any second stage of the
    # iteration, and neither later stage may make the host wait: stage 2 is taken for the chunk launched two iterations
    # ago (its recogniser is over by the time the chunk before this one is running), stage 3 for the chunk whose aligner
    # launch was made one iteration ago (small kernels that slip onto CUs the recurrences leave free).
    # (Round 5's order -- stage 2 of the oldest chunk, THEN this chunk's launch, stage 3 behind it -- left the GPU idle for
    # ~2 ms per chunk while the host did stage 2: timeline in profiles/r06_pages_timeline_pinned.txt.)
    # The first stage's host half of chunk c + 1 (row layout, the pool's staging copies STARTED) is taken right behind
    # chunk c's launch: the copies then run under this iteration's later stages and the next launch finds them done.
    return _pb_pipeline(chunks, begin, collect, deliver, streams, caller)


def _pb_pipeline(chunks, begin, collect, deliver, streams, caller):
    """the loop of process_batch over its chunks (see the comment there)"""
    import torch
    flight, aligned = [], []
    nxt = begin(chunks[0])
    for c, job in enumerate(chunks):
        ctx = nxt
        lane = streams[c % 2] if TWO_STREAMS else caller
        # begin() ran on the CALLER's stream: for raw strips and page images it enqueued device work there (the normaliser's
        # kernels write the rows, the metadata uploads, the zero-fill of the decoder's outputs) that this chunk's recogniser
        # reads -- the chunk's compute stream takes it all in before its first kernel.  (Host rows: nothing was enqueued
        # there and the wait is on an idle stream.)  Nothing else is ever put on the caller's stream inside this loop,
        # so this never orders a chunk behind another chunk's kernels.
        if lane is not caller:
            lane.wait_stream(caller)
        with torch.cuda.stream(lane):
            _pb_launch(ctx)
        nxt = begin(chunks[c + 1]) if c + 1 < len(chunks) else None
        _pb_transcripts(ctx)
        flight.append(ctx)
        if aligned:
            collect(aligned.pop(0))
        if len(flight) > 2:
            oldest = flight.pop(0)
            _pb_finish_a(oldest)
            aligned.append(oldest)
    for st_ in streams:
        caller.wait_stream(st_)
    for ctx in flight:                                            # the drain: every aligner launch as soon as its characters are
        _pb_finish_a(ctx)                                         # back, the box assembly of a chunk under the next one's launch
        if aligned:
            collect(aligned.pop(0))
        aligned.append(ctx)
    for ctx in aligned:
        collect(ctx)
    return deliver()


def _timed_wait(event):
    import time
    t0 = time.perf_counter()
    event.synchronize()
    WAIT_SECONDS[0] += time.perf_counter() - t0


def _pb_begin(rec, pages, transcripts, seq_align_params, workers):
    """first stage of process_batch for one chunk, host part: line finding, the layout of the chunk's rows, the staging
    copies STARTED (pool threads), and the host work that needs no OCR result"""
    from . import page_batch as pb
    raw_dims = [_raw_dim(pg) for pg in pages]            # bad page types fail before any GPU work
    found = find_lines_all(list(pages), workers=workers)
    strips_per_page = [f[3] for f in found]
    all_strips = [st for strips in strips_per_page for st in strips]
    prepared = page_mod.prepared_lines(all_strips, workers=workers)
    lines = [xs for xs, _ in prepared]
    widths = [w for _, w in prepared]
    st = rec.prepare(lines, defer=True)
    return {"rec": rec, "pages": pages, "transcripts": transcripts, "params": seq_align_params, "raw_dims": raw_dims,
            "found": found, "strips_per_page": strips_per_page, "all_strips": all_strips, "lines": lines,
            "widths": widths, "st": st, "cps": pb.codec_code_points(rec.model.codec)}


def _pb_transcripts(ctx):
    """the host work of a chunk that needs no OCR result -- syllables and code points of the transcripts -- done AFTER the
    chunk's kernels have been enqueued: nothing the device is waiting for stands behind it"""
    if "syls_all" not in ctx:
        ctx["syls_all"] = [latsyl.syllabify_text(tr) for tr in ctx["transcripts"]]
        ctx["t_cp"] = [np.frombuffer(tr.encode('utf-32-le'), dtype='<u4').astype(np.int64) for tr in ctx["transcripts"]]


def _pb_launch(ctx):
    """first stage, device part: the rows' transfer, the recogniser kernels and the download of the decoded characters
    -- everything enqueued, nothing waited for but the staging copies"""
    import torch
    rec, st = ctx["rec"], ctx["st"]
    rec.complete(st)
    rec.last_T = st["T_host"]
    rec._last_state = st
    rec.run(st)
    # the decoder's outputs come back through pinned buffers behind an event of their own: a plain .cpu() issued later
    # would queue behind whatever the stream has been given since (the next chunk's kernels)
    host = {}
    for key in ("dec_t", "dec_c", "dec_n"):
        host[key] = torch.empty(st[key].shape, dtype=st[key].dtype, pin_memory=True)
        host[key].copy_(st[key], non_blocking=True)
    ctx["host"] = host
    ctx["done"] = torch.cuda.Event()
    ctx["done"].record()


def _pb_finish_a(ctx):
    """second stage, first half: characters and boxes of every line, abbreviations, ONE NW launch for the chunk's pages
    and the download of its alignment columns STARTED"""
    import torch
    from . import page_batch as pb
    rec, pages, transcripts, seq_align_params = ctx["rec"], ctx["pages"], ctx["transcripts"], ctx["params"]
    raw_dims, found, strips_per_page, all_strips = ctx["raw_dims"], ctx["found"], ctx["strips_per_page"], ctx["all_strips"]
    _pb_transcripts(ctx)
    lines, widths, st, syls_all, t_cp, cps = ctx["lines"], ctx["widths"], ctx["st"], ctx["syls_all"], ctx["t_cp"], ctx["cps"]
    params, fn = tsc.parse_scoring_system(seq_align_params)
    ctx["nw"] = None
    if fn is not None or cps is None or not tsc._is_integral(params):
        return

    # ---- every character of every line: code points + boxes (alignToOCR.py:160-182) ----
    nlines = len(all_strips)
    _timed_wait(ctx["done"])
    dec_t = ctx["host"]["dec_t"].numpy()
    dec_c = ctx["host"]["dec_c"].numpy()
    rec.check_status(ctx["host"]["dec_n"].numpy())       # the device's status word travels behind the counts
    dec_n = ctx["host"]["dec_n"].numpy()[:nlines].astype(np.int64) if nlines else np.zeros(0, np.int64)
    x_min = np.array([s.offset_x for s in all_strips], dtype=np.int64)
    y_min = np.array([s.offset_y for s in all_strips], dtype=np.int64)
    y_max = y_min + np.array([s.height for s in all_strips], dtype=np.int64)
    line, cp, boxes = pb.chars_of_batch(dec_t, dec_c, dec_n, st["row_start_host"], st["T_host"],
                                        np.asarray(widths, dtype=np.int64), x_min, y_min, y_max, cps, ocr_pad())
    first_line = np.zeros(len(pages) + 1, dtype=np.int64)
    np.cumsum([len(s) for s in strips_per_page], out=first_line[1:])
    first_char = np.searchsorted(line, first_line)         # characters of page k: first_char[k] .. first_char[k+1]

    # ---- abbreviations, then one NW launch for all pages (alignToOCR.py:251-276) ----
    texts, idxs = [], []
    for k in range(len(pages)):
        a, b = int(first_char[k]), int(first_char[k + 1])
        text = cp[a:b].astype('<u4').tobytes().decode('utf-32-le')
        text, idx = pb.expand_abbreviations(text, np.arange(a, b), latsyl.abbreviations)
        texts.append(text)
        idxs.append(idx)
    o_cp = [np.frombuffer(tx.encode('utf-32-le'), dtype='<u4').astype(np.int64) for tx in texts]
    alphabet = np.unique(np.concatenate(t_cp + o_cp)) if (t_cp or o_cp) else np.zeros(0, np.int64)
    # (the aligner's inputs come from the host and its buffers are the side stream's own: nothing to wait for)
    with torch.cuda.stream(_nw_stream(rec.device)):
        try:
            batch = tsc.NWBatch([np.searchsorted(alphabet, a).astype(np.int32) for a in t_cp],
                                [np.searchsorted(alphabet, a).astype(np.int32) for a in o_cp],
                                [int(v) for v in params])
            batch.run()
            batch.fetch_begin()
            ctx["nw"] = batch
        except OverflowError:
            pass
    ctx["texts"], ctx["idxs"], ctx["boxes"] = texts, idxs, boxes


def _pb_finish_b(ctx, indices_out, arrays_out):
    """second stage, second half: the alignment columns (waited for here), syllable boxes (alignToOCR.py:277-328)"""
    from . import page_batch as pb
    rec, pages, transcripts, seq_align_params = ctx["rec"], ctx["pages"], ctx["transcripts"], ctx["params"]
    raw_dims, found, strips_per_page = ctx["raw_dims"], ctx["found"], ctx["strips_per_page"]
    lines, widths, st, syls_all = ctx["lines"], ctx["widths"], ctx["st"], ctx["syls_all"]
    if ctx["nw"] is None:                 # a scoring callable / non-integral numbers / a multi-character codec / too large
        rec._last_state, rec.last_T = st, st["T_host"]
        res = _process_batch_objects(rec, pages, raw_dims, found, strips_per_page, lines, widths,
                                     transcripts, seq_align_params, indices_out)
        if arrays_out is not None:            # the same [k, 4] arrays as the array path hands over: one entry per page
            for r in res:
                arrays_out.append(np.array([[b.ulx, b.uly, b.lrx, b.lry] for b in r[0]], dtype=np.int64).reshape(-1, 4))
        return res
    if getattr(ctx["nw"], "_fetched", None) is not None:
        _timed_wait(ctx["nw"]._fetched)
    all_ops = ctx["nw"].results()
    ctx["nw"] = None
    texts, idxs, boxes = ctx["texts"], ctx["idxs"], ctx["boxes"]

    # ---- syllable boxes (alignToOCR.py:277-328): all plain pages in one set of array operations ----
    plain = [k for k in range(len(pages)) if pb.plain_page(transcripts[k], syls_all[k])]
    batched = dict(zip(plain, pb.syllable_boxes_batch(
        [transcripts[k] for k in plain], [syls_all[k] for k in plain], [all_ops[k] for k in plain],
        [idxs[k] for k in plain], boxes, [found[k][2] for k in plain], [found[k][0].dim for k in plain],
        [raw_dims[k] for k in plain])))
    results = []
    for k, (raw_dim, f, tr) in enumerate(zip(raw_dims, found, transcripts)):
        image, angle, lp = f[0], f[2], f[4]
        chars_seq = pb.BoxSeq(texts[k], boxes[idxs[k]], CharBox)          # (a str is a sequence of its characters)
        if k in batched:
            which, sb = batched[k]
            named = [s for s in syls_all[k] if len(s) >= 1]
            syl_seq = pb.BoxSeq([named[w] for w in which], sb, CharBox)
            which = which.tolist()
        else:                                             # characters `re` would interpret: object path
            which = []
            al = tsc.ops_to_alignment(all_ops[k], list(tr), list(texts[k]))
            syl_boxes, _ = align_page(tr, list(chars_seq), angle, image.dim, raw_dim, seq_align_params,
                                      alignment=al, indices=which, expanded=True)
            sb = np.array([[b.ulx, b.uly, b.lrx, b.lry] for b in syl_boxes], dtype=np.int64).reshape(-1, 4)
            syl_seq = syl_boxes
        if indices_out is not None:
            indices_out.append(which)
        if arrays_out is not None:
            arrays_out.append(sb)
        results.append((syl_seq, image, lp, chars_seq))
    return results


def ocr_pad():
    from . import ocr
    return ocr.PAD


def process(raw_image,
            transcript,
            ocropus_model,
            seq_align_params=None,
            wkdir_name='wkdir_ocropy',
            parallel=parallel,
            median_line_mult=median_line_mult,
            existing_ocr_pickle=None,
            existing_preproc_images=None,
            verbose=True):
    '''
    given a text layer @raw_image and a string transcript @transcript, performs OCR on the text
    lines and aligns the results to the transcript text (reference alignToOCR.py:187-330).
    Returns (syl_boxes, image, lines_peak_locs, all_chars), or None when OCR fails.
    '''
    raw_dim = _raw_dim(raw_image)
    image, eroded, angle, cc_strips, lines_peak_locs = find_lines_all([raw_image], workers=1)[0]

    all_chars = []
    if existing_ocr_pickle:
        try:
            all_chars = load_ocr_pickle(existing_ocr_pickle)
            print('using pickled ocr results in {}...'.format(existing_ocr_pickle))
        except IOError:
            print('Pickle file {} not found - performing ocr instead'.format(existing_ocr_pickle))
        except AttributeError:
            print('Pickle error: re-performing ocr')

    if not all_chars:
        from . import ocr
        try:
            all_chars = perform_ocr_with_ocropus(cc_strips, ocropus_model, wkdir_name=wkdir_name,
                                                 parallel=parallel)
        except ocr.RecognitionError:
            print('OCRopus failed! Skipping current file.')
            return None

    syl_boxes, all_chars_copy = align_page(transcript, all_chars, angle, image.dim, raw_dim,
                                           seq_align_params)
    return syl_boxes, image, lines_peak_locs, all_chars_copy


def to_JSON_dict(syl_boxes, lines_peak_locs):
    '''
    output of process() -> the dict the MEI-encoding job consumes (alignToOCR.py:333-351).
    'median_line_spacing' is the 75th percentile of the line gaps, as in the reference.
    '''
    data = {'median_line_spacing': np.quantile(np.diff(lines_peak_locs), 0.75), 'syl_boxes': []}
    for s in syl_boxes:
        data['syl_boxes'].append({'syl': s.char,
                                  'ul': [int(s.ul[0]), int(s.ul[1])],
                                  'lr': [int(s.lr[0]), int(s.lr[1])]})
    return data
