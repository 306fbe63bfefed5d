"""The tolerance-aware decode comparison used by the page-level GPU tests (tests/ocr_compare.py),
on crafted probabilities: differences within the probability tolerance are explained, real
disagreements are not."""
import numpy as np

import ocr_compare as oc
from oracle import ocr_ref_f64 as R


def _probs():
    T, no = 12, 5
    p = np.full((T, no), 0.02)
    p[:, 0] = 0.92
    p[3, 0] = 0.6999; p[3, 2] = 0.25                     # blank probability 1e-4 under the threshold
    p[7, 0] = 0.1; p[7, 1] = 0.45; p[7, 3] = 0.4499      # two classes 1e-4 apart
    p[10, 0] = 0.2; p[10, 4] = 0.7                       # a clear character
    return p


def test_explained_and_unexplained_differences():
    p = _probs()
    ref = R.translate_back(p)
    assert ref == [(3, 0), (7, 1), (10, 4)]
    assert oc.decode_differences(p, ref, ref, 1e-3) == ([], [])
    ex, un = oc.decode_differences(p, ref, [(7, 1), (10, 4)], 1e-3)          # the fragile run vanished
    assert ex == [(3, 0)] and un == []
    ex, un = oc.decode_differences(p, ref, [(3, 0), (7, 3), (10, 4)], 1e-3)  # the close arg-max flipped
    assert sorted(ex) == [(7, 1), (7, 3)] and un == []
    ex, un = oc.decode_differences(p, ref, [(3, 0), (7, 1), (10, 3)], 1e-3)  # a clear character differs
    assert ex == [] and sorted(un) == [(10, 3), (10, 4)]
    ex, un = oc.decode_differences(p, ref, ref + [(5, 1)], 1e-3)             # a character out of a blank stretch
    assert un == [(5, 1)]
    # the same differences are NOT explained at a tolerance below their margins
    ex, un = oc.decode_differences(p, ref, [(7, 1), (10, 4)], 1e-5)
    assert un == [(3, 0)]
