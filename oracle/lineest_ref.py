"""oracle/lineest_ref.py -- float64 scipy restatement of `ocropus-rpred`'s line normaliser.
TEST INFRASTRUCTURE ONLY (never imported by text_alignment_amd/): the checker of
csrc/ta_lineest.hip (tests/test_lineest_gpu.py).

PARITY UNPINNED.  Restates ocropy 1.3.3's `CenterNormalizer` (measure / dewarp / normalize) and
`prepare_line` as recorded in SURVEY.md Appendix B.0-B.2 -- what `ocropus-rpred` does to each PNG
strip that reference alignToOCR.py:131-147 hands it, before the LSTM sees it.  The arithmetic is
third-party (ocropy==1.3.3, reference requirements.txt:2), absent from /root/reference and not
installed; no reference test touches it.  Row N1 of section 8f.
"""
import numpy as np
from scipy.ndimage import affine_transform, gaussian_filter, uniform_filter

TARGET_HEIGHT = 48
PAD = 16


def read_gray(pixels):
    """B.0: uint8 -> [0,1]; colour -> mean over channels; white ~ 1, ink ~ 0."""
    a = np.asarray(pixels)
    if a.dtype == np.uint8:
        a = a / 255.0
    a = a.astype(np.float64)
    if a.ndim == 3:
        a = a.mean(axis=2)
    return a


class CenterNormalizer(object):
    def __init__(self, target_height=TARGET_HEIGHT, params=(4, 1.0, 0.3)):
        self.target_height = target_height
        self.range, self.smoothness, self.extra = params

    def measure(self, line):
        """`line`: ink-positive image (ink > 0).  Finds the smoothed centre line and the
        half-height r of the band to cut out."""
        h, w = line.shape
        smoothed = gaussian_filter(line, (h * 0.5, h * self.smoothness), mode='constant')
        smoothed += 0.001 * uniform_filter(smoothed, (h * 0.5, w), mode='constant')
        a = np.argmax(smoothed, axis=0)
        a = gaussian_filter(a, h * self.extra)        # filter applied to the integer array
        self.center = np.array(a, 'i')
        deltas = np.abs(np.arange(h)[:, np.newaxis] - self.center[np.newaxis, :])
        self.mad = np.mean(deltas[line != 0])
        self.r = int(1 + self.range * self.mad)

    def dewarp(self, img, cval=0, dtype=np.dtype('f')):
        h, w = img.shape
        padded = np.vstack([cval * np.ones((self.r, w)), img, cval * np.ones((self.r, w))])
        center = self.center + self.r
        cols = [padded[center[i] - self.r:center[i] + self.r, i] for i in range(w)]
        return np.array(cols, dtype=dtype).T

    def normalize(self, img, order=1, dtype=np.dtype('f'), cval=0):
        dewarped = self.dewarp(img, cval=cval, dtype=dtype)
        h, w = dewarped.shape
        scale = self.target_height * 1.0 / h
        out = affine_transform(dewarped, np.eye(2) / scale, order=order,
                               output_shape=(self.target_height, int(scale * w)),
                               mode='constant', cval=cval)
        return np.array(out, dtype=dtype)


def prepare_line(line, pad=PAD):
    """B.2: ink -> 1, transpose to (W', 48), `pad` zero rows before and after."""
    line = line * 1.0 / np.amax(line)
    line = np.amax(line) - line
    line = line.T
    if pad > 0:
        w = line.shape[1]
        line = np.vstack([np.zeros((pad, w)), line, np.zeros((pad, w))])
    return line


def prepare_raw_strip(pixels, normalizer=None):
    """Raw strip (white background) -> (T, 48) network input, as ocropus-rpred's process1."""
    line = read_gray(pixels)
    if line.size == 0 or np.amax(line) == np.amin(line):
        raise ValueError("empty or constant text-line image")
    norm = normalizer or CenterNormalizer()
    temp = np.amax(line) - line
    temp = temp * 1.0 / np.amax(temp)
    norm.measure(temp)
    line = norm.normalize(line, cval=np.amax(line))
    return prepare_line(line, PAD)
