"""Gamera-free page containers and the two preprocessing calls `process` makes.

The reference obtains (image, eroded, angle) and (cc_strips, lines_peak_locs, _) from Gamera-
based preprocessing (reference alignToOCR.py:216-218, textAlignPreprocessing.py:160-285), which
is upstream of the hot path and out of scope here.  A
`PreparedPage` carries those results in plain numpy so that `process` keeps its call surface:
`preprocess_images` / `identify_text_lines` below simply hand them over.
"""
import numpy as np


class Dim(object):
    """ncols x nrows, the two attributes rotate_bbox reads (alignToOCR.py:91-96)."""

    def __init__(self, ncols, nrows):
        self.ncols, self.nrows = int(ncols), int(nrows)


class Image(object):
    def __init__(self, ncols, nrows):
        self.dim = Dim(ncols, nrows)
        self.ncols, self.nrows = self.dim.ncols, self.dim.nrows


class Strip(object):
    """One text-line strip: position on the (deskewed) page plus its pixels.

    offset_x, offset_y, height: as the reference reads them (alignToOCR.py:160-162).
    prepared: (T, 48) float array, ink = 1, normalised to height 48 and padded by 16 columns on
        each side (what ocropus-rpred feeds its network); or
    pixels: raw (H, W) strip with white background, normalised on demand by lineest.
    width: raw strip width in pixels (sets the scale of the reported character positions).
    """

    def __init__(self, offset_x, offset_y, height, width=None, prepared=None, pixels=None):
        self.offset_x, self.offset_y, self.height = int(offset_x), int(offset_y), int(height)
        self.prepared = None if prepared is None else np.asarray(prepared)
        self.pixels = None if pixels is None else np.asarray(pixels)
        if width is None:
            if self.pixels is not None:
                width = self.pixels.shape[1]
            elif self.prepared is not None:
                width = self.prepared.shape[0] - 32
        self.width = int(width)


class PreparedPage(object):
    """A text-layer page after preprocessing: deskewed image size, raw image size, the rotation
    angle that was applied, the text-line strips and the line peak locations."""

    def __init__(self, image_dim, raw_dim, angle, strips, lines_peak_locs):
        self.image = Image(*image_dim)
        self.dim = Dim(*raw_dim)              # `raw_image.dim` (alignToOCR.py:328)
        self.angle = angle
        self.strips = list(strips)
        self.lines_peak_locs = list(lines_peak_locs)
        self.image._page = self


def preprocess_images(raw_image):
    if not isinstance(raw_image, PreparedPage):
        raise NotImplementedError(
            "image preprocessing (Gamera, textAlignPreprocessing.py:160-195) is outside the hot "
            "path: pass a text_alignment_amd.page.PreparedPage")
    return raw_image.image, None, raw_image.angle


def identify_text_lines(image, eroded):
    pg = image._page
    return pg.strips, pg.lines_peak_locs, None


def prepared_line(strip):
    """(xs, raw_width): the (T, 48) network input of a strip and its raw pixel width."""
    if getattr(strip, "prepared", None) is not None:
        xs = np.asarray(strip.prepared)
        return xs, int(getattr(strip, "width", xs.shape[0] - 32))
    from . import lineest
    xs = lineest.prepare_raw_strip(strip.pixels)
    return xs, int(strip.pixels.shape[1])


# ---- host-side line normalisation in worker processes -----------------------------------------
# The reference hands its strips to `ocropus-rpred -Q <parallel>` (alignToOCR.py:142-147), whose
# worker processes spend most of their time in the line normaliser; here the recogniser runs on the
# GPU and `parallel` keeps its meaning for the part that is still host work (lineest, ~50 ms per
# raw strip in scipy).  Workers are spawned (never forked: the parent may hold a GPU context), only
# import numpy/scipy, and stay alive for the next page.
_pool = None
_pool_size = 0


def _normaliser_pool(workers):
    global _pool, _pool_size
    if _pool is None or _pool_size != workers:
        close_pool()
        import atexit
        import multiprocessing
        _pool = multiprocessing.get_context("spawn").Pool(workers)
        _pool_size = workers
        atexit.register(close_pool)
    return _pool


def close_pool():
    global _pool, _pool_size
    if _pool is not None:
        _pool.terminate()
        _pool.join()
        _pool, _pool_size = None, 0


def map_host(fn, items, workers=1, min_batch=2):
    """[fn(x) for x in items] for a picklable module-level fn of pure host work; raw page arrays
    go to the worker pool when there are enough of them, PreparedPages (nothing to compute) and
    small jobs stay in-process."""
    heavy = [k for k, it in enumerate(items) if not isinstance(it, PreparedPage)]
    if workers > 1 and len(heavy) >= min_batch:
        out = [None] * len(items)
        # workers get the bare pixel array (callers' page objects need not be picklable)
        done = _normaliser_pool(int(workers)).map(fn, [getattr(items[k], "pixels", items[k]) for k in heavy])
        for k, r in zip(heavy, done):
            out[k] = r
        for k, it in enumerate(items):
            if out[k] is None:
                out[k] = fn(it)
        return out
    return [fn(it) for it in items]


def _is_constant(px):
    """px.max() == px.min(), decided from ~64 samples whenever they already differ (two full
    reductions per strip were 10 % of a page's host time)"""
    flat = px.reshape(-1)
    probe = flat[::max(1, flat.size // 64)]
    if probe.min() != probe.max():
        return False
    return bool(px.max() == px.min())


def prepared_lines(strips, workers=1, min_batch=4, device_normaliser=True):
    """[(line, raw_width)] for a list of strips, `line` being what LineRecognizer.prepare takes.
    Strips that carry `.prepared` pass through.  Raw uint8 greyscale strips are handed on as they
    are when `device_normaliser` is set (the recogniser normalises them on the GPU); anything else
    (colour, float) is normalised on the host, in `workers` processes when there are enough."""
    out = [None] * len(strips)
    raw = []
    for k, strip in enumerate(strips):
        if getattr(strip, "prepared", None) is not None:
            out[k] = prepared_line(strip)
            continue
        px = np.asarray(strip.pixels)
        if device_normaliser and px.dtype == np.uint8 and px.ndim == 2:
            if px.size == 0 or _is_constant(px):
                raise ValueError("empty or constant text-line image")
            out[k] = (px, int(px.shape[1]))
        else:
            raw.append(k)
    if workers > 1 and len(raw) >= min_batch:
        from . import lineest
        pool = _normaliser_pool(int(workers))
        done = pool.map(lineest.prepare_raw_strip, [strips[k].pixels for k in raw],
                        chunksize=max(1, len(raw) // (4 * int(workers))))
        for k, xs in zip(raw, done):
            out[k] = (xs, int(strips[k].pixels.shape[1]))
    else:
        for k in raw:
            out[k] = prepared_line(strips[k])
    return out
