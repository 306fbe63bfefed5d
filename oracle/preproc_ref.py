"""oracle/preproc_ref.py -- numpy / scipy.ndimage restatement of the reference's page preprocessing
and text-line finding (reference textAlignPreprocessing.py:38-285).  TEST INFRASTRUCTURE ONLY (never
imported by text_alignment_amd/): the checker of csrc/ta_preproc.hip + preproc_gpu.py
(tests/test_preproc_gpu.py, tests/test_preprocessing.py).

Two kinds of code, with different parity status:

* The projection / peak-finding numerics (`moving_avg_filter` :147-157, `calculate_peak_prominence`
  :59-110, `find_peak_locations` :113-144, `vertically_coincide` :38-56 of the reference) are plain
  numpy in the reference too; restated here and PINNED to golden vectors captured from the imported
  reference (tests/golden/preproc.json, tests/test_preprocessing.py).
* The image operations the reference delegates to the Gamera C++ toolkit (`to_onebit`, `despeckle`,
  `cc_analysis`, `rotation_angle_projections`, `rotate`, `filter_short_runs`, `filter_narrow_runs`,
  `projection_rows`, `draw_line`, `subimage`; reference :167-195, :212-253) are re-expressed with
  scipy.ndimage from Gamera's documented behaviour.  Gamera (requirements.txt:4) is absent here:
  PARITY UNPINNED -- same pipeline, same parameters, not bit-checked against Gamera.

Images are numpy arrays; a "onebit" image is a bool array with True = ink (Gamera's black).
"""
import numpy as np
from scipy import ndimage


# PARAMETERS FOR PREPROCESSING (reference textAlignPreprocessing.py:12-16)
saturation_thresh = 0.9
sat_area_thresh = 150
despeckle_amt = 100
noise_area_thresh = 100

# PARAMETERS FOR TEXT LINE SEGMENTATION (reference :18-22)
filter_size = 30
prominence_tolerance = 0.70
collision_strip_scale = 1
remove_capitals_scale = 10000

_EIGHT = np.ones((3, 3), dtype=bool)          # Gamera labels connected components 8-connected


# --------------------------------------------------------------------------- pinned numerics
def vertically_coincide(hline_position, comp_offset, comp_nrows, collision,
                        collision_scale=collision_strip_scale):
    """True if any part of a component (rows comp_offset .. comp_offset + comp_nrows) lies within
    the horizontal strip of height `collision` centred on hline_position (reference :38-56)."""
    collision *= collision_strip_scale
    top, bottom = comp_offset, comp_offset + comp_nrows
    strip_top = hline_position - int(collision / 2)
    strip_bottom = hline_position + int(collision / 2)
    above = top < strip_top and bottom < strip_top
    below = top > strip_bottom and bottom > strip_bottom
    return (not above and not below)


def coincide_mask(hline_position, comp_offsets, comp_nrows, collision):
    """vertically_coincide for many components at once (arrays of offsets and heights)"""
    collision = collision * collision_strip_scale
    top = np.asarray(comp_offsets)
    bottom = top + np.asarray(comp_nrows)
    strip_top = hline_position - int(collision / 2)
    strip_bottom = hline_position + int(collision / 2)
    above = (top < strip_top) & (bottom < strip_top)
    below = (top > strip_bottom) & (bottom > strip_bottom)
    return ~above & ~below


def calculate_peak_prominence(data, index, data_max=None):
    '''log of the prominence of the peak at `index`: isolated peaks score high, peaks in the
    foothills of larger ones low (reference :59-110).  `data_max` may carry max(data) when many
    indices of the same array are scored.'''
    here = data[index]
    if (index == 0 or index == len(data) - 1 or data[index - 1] > here or data[index + 1] > here or
            (data[index - 1] == here and data[index + 1] == here)):
        return 0
    if here == (max(data) if data_max is None else data_max):
        return np.log(here)
    higher = np.nonzero(np.asarray(data) > here)[0]                 # indices of everything above this peak
    cut = int(np.searchsorted(higher, index))
    nearest_right = higher[cut] if cut < len(higher) else np.inf
    nearest_left = higher[cut - 1] if cut > 0 else -np.inf
    nearest = nearest_left if (nearest_right - index) > (index - nearest_left) else nearest_right
    lo, hi = min(nearest, index), max(nearest, index)
    key_col = min(data[int(lo):int(hi)])
    return np.log(data[index] - key_col + 1)


def find_peak_locations(data, tol=prominence_tolerance, ranked=False):
    '''indices of the prominent peaks of a row projection (reference :113-144)'''
    data_max = max(data) if len(data) else None
    # only local maxima can score: find them in one array pass (the same test
    # calculate_peak_prominence starts with), everything else has prominence 0
    d = np.asarray(data)
    cand = np.zeros(len(d), dtype=bool)
    if len(d) > 2:
        mid, left, right = d[1:-1], d[:-2], d[2:]
        cand[1:-1] = ~((left > mid) | (right > mid) | ((left == mid) & (right == mid)))
    proms = [(i, calculate_peak_prominence(data, i, data_max) if cand[i] else 0) for i in range(len(data))]
    top = max([p[1] for p in proms])
    if top == 0 or len(proms) == 0:
        return []
    proms = [(i, v / top) for i, v in proms]
    peaks = [p for p in proms if p[1] > tol]
    # both corners of a flat-topped peak are prominent: drop the first of two equal neighbours
    dupes = [peaks[i] for i in range(len(peaks) - 2) if peaks[i][1] == peaks[i + 1][1]]
    for d in dupes:
        peaks.remove(d)
    if ranked:
        peaks.sort(key=lambda p: p[1] * -1)
        return peaks
    return [p[0] for p in peaks]


def moving_avg_filter(data, filter_size=filter_size):
    '''moving average over filter_size samples to either side; the ends stay zero (reference :147-157)'''
    smoothed = np.zeros(len(data))
    data = np.asarray(data)
    n = len(data)
    if n > 2 * filter_size and data.dtype.kind in "iub":
        # integer samples (a row projection): window sums are exact in any order, so a running
        # sum gives bit for bit what np.mean gives per window
        c = np.concatenate([[0], np.cumsum(data.astype(np.int64))])
        win = c[2 * filter_size + 1:] - c[:n - 2 * filter_size]
        smoothed[filter_size:n - filter_size] = win / float(2 * filter_size + 1)
        return smoothed
    for k in range(filter_size, n - filter_size):
        smoothed[k] = np.mean(data[k - filter_size: k + filter_size + 1])
    return smoothed


# --------------------------------------------------------------------------- image operations
def otsu_threshold(grey):
    """Otsu's threshold of a uint8 image (Gamera's to_onebit on a greyscale image)."""
    hist = np.bincount(grey.ravel(), minlength=256).astype(np.float64)
    total = hist.sum()
    cum = np.cumsum(hist)
    mean_cum = np.cumsum(hist * np.arange(256))
    mean_all = mean_cum[-1]
    with np.errstate(divide='ignore', invalid='ignore'):
        between = (mean_all * cum - mean_cum * total) ** 2 / (cum * (total - cum))
    between[~np.isfinite(between)] = 0
    return int(np.argmax(between))


def to_grey_u8(image):
    """RGB / greyscale of any numeric type -> 2-D uint8 greyscale (what the threshold works on)"""
    a = np.asarray(image)
    if a.ndim == 3:
        a = a[..., :3].mean(axis=2)
    if a.dtype != np.uint8:
        a = np.clip(a * (255.0 if a.max() <= 1.0 else 1.0), 0, 255).astype(np.uint8)
    return a


def to_onebit(image):
    """RGB / greyscale / bool array -> bool array, True = ink."""
    a = np.asarray(image)
    if a.dtype == bool:
        return a.copy()
    a = to_grey_u8(a)
    return a <= otsu_threshold(a)


def despeckle(onebit, size):
    """remove ink components of fewer than `size` pixels"""
    lab, n = ndimage.label(onebit, structure=_EIGHT)
    if n == 0:
        return onebit
    area = np.bincount(lab.ravel(), minlength=n + 1)
    keep = area >= size
    keep[0] = False
    return keep[lab]


def components(onebit):
    """[(label slice pair, label id)] of the 8-connected ink components, plus the label image"""
    lab, n = ndimage.label(onebit, structure=_EIGHT)
    return lab, ndimage.find_objects(lab)


def rotation_angle_projections(onebit, lo=-6.0, hi=6.0, coarse=0.25, fine=0.05):
    """angle in [lo, hi] degrees whose rotation (as `rotate` below applies it) makes the row
    projection sharpest (largest variance): coarse sweep, then a fine sweep around the best.
    The projection of the rotated page is formed directly from the ink coordinates -- pixel (y, x)
    lands on row cy + (y - cy) cos a - (x - cx) sin a -- instead of rotating the image once per
    candidate angle (60 rotations of a page cost seconds; this costs milliseconds)."""
    step = max(1, int(max(onebit.shape) / 1200))          # large pages: every step-th row and column
    small = onebit[::step, ::step]
    ys, xs = np.nonzero(small)
    if ys.size == 0:
        return 0.0
    h, w = small.shape
    cy, cx = (h - 1) / 2.0, (w - 1) / 2.0
    dy, dx = ys - cy, xs - cx

    def score(ang):
        a = np.deg2rad(ang)
        rows = np.rint(cy + dy * np.cos(a) - dx * np.sin(a)).astype(np.int64)
        rows = rows[(rows >= 0) & (rows < h)]
        return float(np.var(np.bincount(rows, minlength=h)))
    grid = np.arange(lo, hi + 1e-9, coarse)
    best = grid[int(np.argmax([score(a) for a in grid]))]
    grid = np.arange(best - coarse, best + coarse + 1e-9, fine)
    best = grid[int(np.argmax([score(a) for a in grid]))]
    return float(np.round(best, 3))


def reported_angle(skew):
    """The angle `preprocess_images` hands to `process`.  `process` maps syllable boxes back onto the
    raw page with rotate_bbox(box, -angle, ...) (reference alignToOCR.py:327-328), whose rotation
    x' = x cos a - y sin a, y' = x sin a + y cos a (image coordinates, y down; alignToOCR.py:104-112)
    turns the opposite way from scipy.ndimage.rotate(img, a).  `rotate` below (and the device
    kernel) deskew with scipy's sense, so the angle that makes rotate_bbox(-angle) the exact inverse
    of the deskewing is minus the scipy angle that was applied."""
    return -skew if skew != 0 else 0.0


def rotate(onebit, angle):
    """rotate about the centre, growing the canvas to hold the whole page (as Gamera's rotate;
    alignToOCR.rotate_bbox undoes exactly this padding, reference alignToOCR.py:93-96)"""
    if angle == 0:
        return onebit.copy()
    rot = ndimage.rotate(onebit.astype(np.float32), angle, reshape=True, order=1, mode='constant', cval=0.0)
    return rot > 0.5


def _filter_runs(onebit, length, axis):
    """remove ink runs shorter than `length` along `axis`: a morphological opening with a
    `length` x 1 line (what scipy.ndimage.binary_opening computes, here with shifted views: a pixel
    survives iff it lies in a window of `length` consecutive ink pixels)"""
    if length <= 1:
        return onebit
    a = np.moveaxis(np.asarray(onebit, dtype=bool), axis, 0)
    n = a.shape[0]
    out = np.zeros_like(a)
    if n >= length:
        full = a[:n - length + 1].copy()                  # full[i]: a[i .. i+length-1] all ink
        for k in range(1, length):
            full &= a[k:n - length + 1 + k]
        for k in range(length):
            out[k:n - length + 1 + k] |= full
    return np.moveaxis(out, 0, axis)


def filter_short_runs(onebit, length):      # vertical runs (Gamera: filter_short_runs)
    return _filter_runs(onebit, length, 0)


def filter_narrow_runs(onebit, length):     # horizontal runs (Gamera: filter_narrow_runs)
    return _filter_runs(onebit, length, 1)


# --------------------------------------------------------------------------- the two entry points
class Dim(object):
    def __init__(self, ncols, nrows):
        self.ncols, self.nrows = int(ncols), int(nrows)


class BinImage(object):
    """A onebit page image: `.ink` (bool, True = ink) and the `dim` / `ncols` / `nrows` the glue reads."""

    def __init__(self, ink):
        self.ink = ink
        self.dim = Dim(ink.shape[1], ink.shape[0])
        self.ncols, self.nrows = self.dim.ncols, self.dim.nrows


class Strip(object):
    """One text-line strip as the reference reads it (alignToOCR.py:160-162) plus its pixels (ink black
    on white, as the saved PNG)."""

    def __init__(self, offset_x, offset_y, height, width, pixels):
        self.offset_x, self.offset_y, self.height, self.width = int(offset_x), int(offset_y), int(height), int(width)
        self.pixels = pixels


def preprocess_images(input_image, despeckle_amt=despeckle_amt, filter_runs=1, filter_runs_amt=2,
                      correct_rotation=True):
    '''denoise and deskew the text layer before text-line segmentation (reference :160-195).
    Returns (image_bin, image_eroded, angle).'''
    ink = to_onebit(getattr(input_image, "pixels", input_image))
    ink = despeckle(ink, despeckle_amt)
    ink = ~despeckle(~ink, despeckle_amt)                      # fill small holes
    lab, objs = components(ink)
    for k, sl in enumerate(objs):                              # drop components taller than the threshold
        if sl is not None and sat_area_thresh < (sl[0].stop - sl[0].start):
            ink[sl][lab[sl] == k + 1] = False
    skew = rotation_angle_projections(ink, -6, 6)
    if correct_rotation:
        ink = rotate(ink, skew)
    eroded = ink.copy()
    for _ in range(filter_runs):
        eroded = filter_short_runs(eroded, filter_runs_amt)
        eroded = filter_narrow_runs(eroded, filter_runs_amt)
    return BinImage(ink), BinImage(eroded), reported_angle(skew)


def find_lines(input_image):
    """preprocess_images + identify_text_lines of one page in one call (a unit of host work the
    batched page driver can hand to a worker process): (image_bin, image_eroded, angle, line
    strips, peak locations)."""
    image_bin, image_eroded, angle = preprocess_images(input_image)
    strips, peaks, _ = identify_text_lines(image_bin, image_eroded)
    return image_bin, image_eroded, angle, strips, peaks


def identify_text_lines(image_bin, image_eroded):
    '''text lines of a preprocessed page (reference :198-285): peaks of the smoothed row
    projection; a white line at the projection minimum between neighbouring peaks; connected
    components; per peak the union of the components a strip around the peak touches.
    Returns (line_strips, peak_locations, smoothed_projection).'''
    ink = image_eroded.ink.copy()
    project = ink.sum(axis=1)
    smoothed = moving_avg_filter(project, filter_size)
    peaks = find_peak_locations(smoothed)
    for a, b in zip(peaks[:-1], peaks[1:]):
        idx = int(np.argmin(smoothed[a:b])) + a
        ink[max(idx - 1, 0):idx + 1, :] = False                # 2-pixel white line
    lab, objs = components(ink)
    comps = []
    for k, sl in enumerate(objs):
        if sl is None:
            continue
        area = int((lab[sl] == k + 1).sum())
        if area > noise_area_thresh:
            comps.append((sl[1].start, sl[0].start, sl[1].stop - 1, sl[0].stop - 1))   # ulx, uly, lrx, lry
    if not comps:
        return [], peaks, smoothed
    heights = [c[3] - c[1] + 1 for c in comps]
    med = np.median(heights)
    comps = [c for c, h in zip(comps, heights) if h < med * remove_capitals_scale]
    cc_median_height = np.median([c[3] - c[1] + 1 for c in comps])
    strips = []
    box = np.asarray(comps, dtype=np.int64)
    for loc in peaks:
        hit = box[coincide_mask(loc, box[:, 1], box[:, 3] - box[:, 1] + 1, cc_median_height)]
        if not len(hit):
            continue
        ulx, uly = int(hit[:, 0].min()), int(hit[:, 1].min())
        lrx, lry = int(hit[:, 2].max()), int(hit[:, 3].max())
        sub = image_bin.ink[uly:lry + 1, ulx:lrx + 1]
        pixels = np.where(sub, 0, 255).astype(np.uint8)        # as the saved PNG: ink black on white
        strips.append(Strip(ulx, uly, lry - uly + 1, lrx - ulx + 1, pixels))
    return strips, peaks, smoothed
