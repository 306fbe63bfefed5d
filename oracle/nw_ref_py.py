"""oracle/nw_ref_py.py -- behavioural CPU port of the reference aligner.  TEST/BENCH INFRASTRUCTURE ONLY.

A pure-Python restatement of textSeqCompare.perform_alignment (reference
textSeqCompare.py:13-177) that keeps the reference's *cost profile*: six dense float64
numpy matrices (textSeqCompare.py:45-50) filled by a per-cell interpreter loop with
numpy scalar reads/writes (textSeqCompare.py:62-88), then a pointer walk
(textSeqCompare.py:96-170).  The reference's own .py files may not travel to the GPU
box, so bench.py times THIS port there as `cpu_baseline` (kind "port").  In the build
container it is checked against the imported reference for identical outputs and for
speed within +-15 % (tools/gen_golden.py --speed).

Pinned by tests/test_oracle_nw.py against tests/golden/nw_*.json.
"""
import numpy as np

FLOOR = -1e100           # textSeqCompare.py:55,60
EDGE_STEP = -1           # module-global gap_extend, textSeqCompare.py:9 (used at :54-59)
DEFAULT_SYS = [8, -4, -7, -7, -3, 0]


def _scoring(scoring_system):
    # textSeqCompare.py:24-42
    if scoring_system is None:
        scoring_system = DEFAULT_SYS
    k = len(scoring_system)
    if k == 5 and callable(scoring_system[0]):
        return (scoring_system[0],) + tuple(scoring_system[-4:])
    if k == 6:
        hit, miss = scoring_system[0], scoring_system[1]
        return (lambda a, b: hit if a == b else miss,) + tuple(scoring_system[-4:])
    if k == 4:
        hit, miss, opn, ext = (scoring_system[q] for q in range(4))
        return (lambda a, b: hit if a == b else miss, opn, opn, ext, ext)
    raise ValueError('scoring_system {} invalid'.format(scoring_system))


def perform_alignment(transcript, ocr, scoring_system=None, verbose=False):
    score, open_x, open_y, ext_x, ext_y = _scoring(scoring_system)
    rows = list(transcript) + [' ']        # sentinel only sizes the tables (textSeqCompare.py:21-22)
    cols = list(ocr) + [' ']
    R, C = len(rows), len(cols)
    shape = (R, C)
    main, gap_y, gap_x = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    from_main, from_y, from_x = np.zeros(shape), np.zeros(shape), np.zeros(shape)

    for r in range(R):                      # textSeqCompare.py:53-56
        main[r][0] = EDGE_STEP * r
        gap_x[r][0] = FLOOR
        gap_y[r][0] = EDGE_STEP * r
    for c in range(C):                      # textSeqCompare.py:57-60
        main[0][c] = EDGE_STEP * c
        gap_x[0][c] = EDGE_STEP * c
        gap_y[0][c] = FLOOR

    for r in range(1, R):                   # textSeqCompare.py:62-88
        for c in range(1, C):
            cand = [main[r - 1][c - 1], gap_x[r - 1][c - 1], gap_y[r - 1][c - 1]]
            top = max(cand)
            main[r][c] = top + score(rows[r - 1], cols[c - 1])
            from_main[r][c] = int(cand.index(top))

            cand = [main[r][c - 1] + open_y + ext_y,
                    gap_x[r][c - 1] + open_y + ext_y,
                    gap_y[r][c - 1] + ext_y]
            top = max(cand)
            gap_y[r][c] = top
            from_y[r][c] = int(cand.index(top))

            cand = [main[r - 1][c] + open_x + ext_x,
                    gap_x[r - 1][c] + ext_x,
                    gap_y[r - 1][c] + open_x + ext_x]
            top = max(cand)
            gap_x[r][c] = top
            from_x[r][c] = int(cand.index(top))

    # walk back from the corner; the start state is from_main of the corner cell
    # (textSeqCompare.py:100-102), and the corner pair itself is forced (:105-106)
    r, c = R - 1, C - 1
    state = from_main[r][c]
    out_t, out_o = [rows[r]], [cols[c]]
    while r > 0 and c > 0:                  # textSeqCompare.py:110-145
        if state == 0:
            out_t.append(rows[r - 1]); out_o.append(cols[c - 1])
            state = from_main[r][c]; r -= 1; c -= 1
        elif state == 1:
            out_t.append(rows[r - 1]); out_o.append('_')
            state = from_x[r][c]; r -= 1
        elif state == 2:
            out_t.append('_'); out_o.append(cols[c - 1])
            state = from_y[r][c]; c -= 1
    while c > 0:                            # textSeqCompare.py:154-158
        out_t.append('_'); out_o.append(cols[c - 1]); c -= 1
    while r > 0:                            # textSeqCompare.py:160-164
        out_t.append(rows[r - 1]); out_o.append('_'); r -= 1

    out_t = out_t[-1:0:-1]                  # textSeqCompare.py:167-168
    out_o = out_o[-1:0:-1]
    if verbose:
        for a, b in zip(out_t, out_o):
            print('{} {} {}'.format(a, b, 'O' if a == b else ('~' if '_' not in (a, b) else ' ')))
    return out_t, out_o
