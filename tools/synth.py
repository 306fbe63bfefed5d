"""Seeded synthetic inputs shared by tests, tools/gen_golden.py and bench.py (an input generator,
not part of the oracle: it lives outside oracle/ so that only checkers import that package).

`synth_pair` is the generator of SURVEY.md Appendix D (the recipe the golden sha256 values
were captured with); it models an OCR read of a chant transcript with ~80 % character
agreement (reference README.md:24) over the 27-symbol alphabet parse_cantus_csv.clean
leaves (reference parse_cantus_csv.py:5-13).  TEST/BENCH INFRASTRUCTURE, not product code.
"""
import numpy as np

ALPHA = "abcdefghijklmnopqrstuvwxyz "


def synth_pair_ids(n, m, seed, p_del=0.08, p_sub=0.12, p_ins=0.06):
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 27, size=n)
    u = rng.random(size=(n, 2))
    s = rng.integers(0, 27, size=(n, 2))
    o = []
    for k in range(n):
        if u[k, 0] < p_del:
            pass
        elif u[k, 0] < p_del + p_sub:
            o.append(int(s[k, 0]))
        else:
            o.append(int(t[k]))
        if u[k, 1] < p_ins:
            o.append(int(s[k, 1]))
    pad = rng.integers(0, 27, size=m)
    o = (o + [int(x) for x in pad])[:m]
    return np.asarray(t, dtype=np.int32), np.asarray(o, dtype=np.int32)


def synth_pair(n, m, seed, **kw):
    t, o = synth_pair_ids(n, m, seed, **kw)
    return [ALPHA[c] for c in t], [ALPHA[c] for c in o]
