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


class RowBlock(object):
    """Prepared rows ((T, 48) float32, ink = 1, padded -- what `Strip.prepared` holds) of MANY text lines in ONE block
    of memory the GPU reads where it lies: page-locked host memory (kind "pinned": a batch's rows cross PCIe in one DMA
    transfer per block, straight from here, no staging copy) or device memory (kind "device": nothing moves at all).
    A loader that decodes / normalises strips fills `block.host[a:b]` (pinned) or `block.tensor[a:b]` (device) and
    hands `block.span(a, b)` to the strip.  The recogniser gathers the lines into its own row order on the device
    (csrc/ta_rows.hip); results are those of the same rows passed as numpy arrays.

    The caller keeps the rows unchanged until the call that consumes them has returned (as with any input array)."""
    WIDTH = 48

    def __init__(self, rows, kind="pinned", device=None):
        import torch
        rows = max(int(rows), 1)
        if kind == "pinned":
            self.tensor = torch.empty((rows, self.WIDTH), dtype=torch.float32, pin_memory=True)
            self.host = self.tensor.numpy()
        elif kind == "device":
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            self.tensor = torch.empty((rows, self.WIDTH), dtype=torch.float32, device=dev)
            self.host = None
        else:
            raise ValueError("a RowBlock is 'pinned' (page-locked host memory) or 'device'")
        self.kind, self.rows, self.used = kind, rows, 0

    @classmethod
    def from_lines(cls, lines, kind="pinned", device=None):
        """(block, spans): the given (T, 48) arrays laid out one after the other in a new block"""
        import torch
        total = sum(int(ln.shape[0]) for ln in lines)
        block = cls(total, kind, device)
        spans = []
        for ln in lines:
            sp = block.take(int(ln.shape[0]))
            if kind == "pinned":
                block.host[sp.start:sp.stop] = ln
            else:
                block.tensor[sp.start:sp.stop] = torch.from_numpy(np.ascontiguousarray(ln, dtype=np.float32)).to(block.tensor.device)
            spans.append(sp)
        return block, spans

    def take(self, nrows):
        """the next `nrows` unused rows of the block as a span"""
        if self.used + nrows > self.rows:
            raise ValueError("RowBlock of %d rows is full" % self.rows)
        self.used += int(nrows)
        return RowSpan(self, self.used - int(nrows), self.used)

    def span(self, start, stop):
        if not 0 <= start <= stop <= self.rows:
            raise ValueError("rows %d..%d are outside the block" % (start, stop))
        return RowSpan(self, int(start), int(stop))


class RowSpan(object):
    """rows [start, stop) of a RowBlock: one text line's prepared (T, 48) rows.  Looks like an array where the host glue
    looks (shape, ndim, dtype, np.asarray -- which DOWNLOADS a device span)."""
    __slots__ = ("block", "start", "stop")
    ndim = 2
    dtype = np.dtype(np.float32)

    def __init__(self, block, start, stop):
        self.block, self.start, self.stop = block, start, stop

    @property
    def shape(self):
        return (self.stop - self.start, RowBlock.WIDTH)

    def numpy(self):
        if self.block.kind == "pinned":
            return self.block.host[self.start:self.stop]
        return self.block.tensor[self.start:self.stop].cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.stop - self.start


class DeviceStrip(object):
    """A raw strip that lies in a larger uint8 device buffer (the packed buffer the page preprocessing cuts a batch's
    strips into): rows x cols bytes from element `start` of `buffer`.  Stands where a 2-D uint8 device tensor would
    (`.shape`, `.dtype`, `.cpu()`), without a tensor view per strip -- a page has ~30 strips and the normaliser wants
    them as ONE run of pixels anyway (lineest_gpu.measure_strips_begin joins neighbours in the same buffer)."""
    __slots__ = ("buffer", "start", "shape")

    def __init__(self, buffer, start, rows, cols):
        self.buffer, self.start, self.shape = buffer, int(start), (int(rows), int(cols))

    @property
    def dtype(self):
        return self.buffer.dtype

    def tensor(self):
        h, w = self.shape
        return self.buffer[self.start:self.start + h * w].view(h, w)

    def cpu(self):
        return self.tensor().cpu()


class Strip(object):
    """One text-line strip: position on the (deskewed) page plus its pixels.

    offset_x, offset_y, height: as the reference reads them (alignToOCR.py:160-162).
    prepared: (T, 48) float array, ink = 1, normalised to height 48 and padded by 16 columns on
        each side (what ocropus-rpred feeds its network) -- or a RowSpan: those rows
        inside a RowBlock (page-locked or device memory), taken by the recogniser where they lie; or
    pixels: raw (H, W) uint8 strip with white background, normalised on the device (lineest_gpu); or
    device_pixels: the same strip as a 2-D uint8 tensor that already lives on the GPU, or a DeviceStrip (what the
        device preprocessing cuts, preproc_gpu.identify_text_lines_batch) -- `.pixels` then downloads it on
        first use, the recogniser takes it where it is.
    width: raw strip width in pixels (sets the scale of the reported character positions).
    """

    def __init__(self, offset_x, offset_y, height, width=None, prepared=None, pixels=None, device_pixels=None):
        self.offset_x, self.offset_y, self.height = int(offset_x), int(offset_y), int(height)
        self.prepared = prepared if (prepared is None or isinstance(prepared, RowSpan)) else np.asarray(prepared)
        self._pixels = None if pixels is None else np.asarray(pixels)
        self.device_pixels = device_pixels
        if width is None:
            if self._pixels is not None:
                width = self._pixels.shape[1]
            elif device_pixels is not None:
                width = device_pixels.shape[1]
            elif self.prepared is not None:
                width = self.prepared.shape[0] - 32
        self.width = int(width)

    @property
    def pixels(self):
        if self._pixels is None and self.device_pixels is not None:
            self._pixels = self.device_pixels.cpu().numpy()
        return self._pixels

    @pixels.setter
    def pixels(self, value):
        self._pixels = None if value is None else np.asarray(value)
        self.device_pixels = None


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

    @classmethod
    def from_rows(cls, image_dim, raw_dim, angle, block, spans, boxes, lines_peak_locs):
        """A page whose text lines' prepared rows lie in a RowBlock: spans[k] = (first row, end row) of line k in
        `block` (or a RowSpan), boxes[k] = (offset_x, offset_y, height, raw width in pixels) of its strip."""
        strips = []
        for sp, (ox, oy, h, w) in zip(spans, boxes):
            if not isinstance(sp, RowSpan):
                sp = block.span(int(sp[0]), int(sp[1]))
            strips.append(Strip(ox, oy, h, width=w, prepared=sp))
        return cls(image_dim, raw_dim, angle, strips, lines_peak_locs)


def preprocess_images(raw_image):
    if not isinstance(raw_image, PreparedPage):
        raise NotImplementedError(
            "image preprocessing (Gamera, textAlignPreprocessing.py:160-195) is outside the hot "
            "path: pass a text_alignment_amd.page.PreparedPage")
    return raw_image.image, None, raw_image.angle


def identify_text_lines(image, eroded):
    pg = image._page
    return pg.strips, pg.lines_peak_locs, None


def raw_strip_pixels(strip):
    """The 2-D uint8 greyscale image of a raw strip (white background), as the PNG the reference
    saves for ocropus-rpred (alignToOCR.py:131-132: Gamera writes onebit / greyscale PNGs).  A bool
    strip (True = ink) becomes black on white.  Anything else -- colour, float -- is not something
    the reference's seam ever saw and is refused: convert it to uint8 greyscale first.  An EMPTY strip is refused here; a
    constant (blank) one by the device normaliser's measuring pass, with the same ValueError -- it reduces every strip
    to its minimum and maximum anyway, and the host's own look at 1 920 strips per pass was 10 % of a raw page's host time."""
    px = np.asarray(strip.pixels)
    if px.dtype == bool and px.ndim == 2:
        px = np.where(px, 0, 255).astype(np.uint8)
    if px.dtype != np.uint8 or px.ndim != 2:
        raise TypeError("a raw text-line strip is a 2-D uint8 greyscale image (bool: True = ink); got %s, %d-D"
                        % (px.dtype, px.ndim))
    if px.size == 0:
        raise ValueError("empty or constant text-line image")
    return px               # (a CONSTANT strip is found by the normaliser's measuring pass, which reduces every strip to its
                            # minimum and maximum anyway, and raises the same error: lineest_gpu.measure_strips_end)


def prepared_line(strip):
    """(line, raw_width): what LineRecognizer.prepare takes for a strip -- its (T, 48) network input
    if the strip carries one, else its raw uint8 pixels (host array or device tensor; normalised on the
    device, csrc/ta_lineest.hip) -- and its raw pixel width."""
    if getattr(strip, "prepared", None) is not None:
        xs = strip.prepared if isinstance(strip.prepared, RowSpan) else np.asarray(strip.prepared)
        return xs, int(getattr(strip, "width", xs.shape[0] - 32))
    dp = getattr(strip, "device_pixels", None)
    if dp is not None:                  # cut on the GPU: stays there (an empty or constant one is refused by
        return dp, int(dp.shape[1])     # the normaliser, which measures it anyway)
    px = raw_strip_pixels(strip)
    return px, int(px.shape[1])


def prepared_lines(strips, workers=1):
    """[(line, raw_width)] for a list of strips, `line` being what LineRecognizer.prepare takes:
    strips that carry `.prepared` pass through, raw strips are handed on as uint8 images (the
    recogniser normalises them on the GPU).  `workers` -- the reference's `parallel`, its number of
    ocropus-rpred worker processes (alignToOCR.py:24, :142-147) -- is accepted and unused: there is no
    host-side normalisation left to spread."""
    return [prepared_line(strip) for strip in strips]
