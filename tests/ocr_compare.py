"""Tolerance-aware comparison of two decodes of one text line (test helper, not a test).

The line recogniser's parity statement is on probabilities (BASELINE.json north_star: logits within
1e-3 of the float64 restatement).  translate_back then thresholds the blank
probability at 0.7 and takes an arg-max per run, so a decision whose margin is smaller than the
probability difference between two implementations cannot be pinned by ANY implementation at that
tolerance.  `decode_differences` separates such decisions from real disagreements."""
import numpy as np

THRESHOLD = 0.7


def decode_differences(ref_probs, ref_dec, got_dec, eps, thr=THRESHOLD):
    """The (t, class) items on which two decodes of one line differ -> (explained, unexplained).

    ref_probs: the oracle's (T, No) probabilities; eps: the largest probability difference measured
    between the two implementations on this line.  A differing item is EXPLAINED when, inside the
    maximal interval of timesteps around it that could be non-blank under a perturbation of eps
    (blank < thr + eps), either (i) some blank probability lies within eps of the threshold (the
    segmentation into runs can change) or (ii) the two largest entries lie within 2 eps of each other
    (the arg-max of the run can change).  Anything else is a real disagreement."""
    diff = sorted(set(map(tuple, ref_dec)) ^ set(map(tuple, got_dec)))
    if not diff:
        return [], []
    blank = np.asarray(ref_probs)[:, 0]
    T = blank.shape[0]
    maybe = blank < thr + eps
    fragile = np.abs(blank - thr) < eps
    explained, unexplained = [], []
    for (t, c) in diff:
        if not (0 <= t < T) or not maybe[t]:
            unexplained.append((t, c))
            continue
        a = b = t
        while a > 0 and maybe[a - 1]:
            a -= 1
        while b + 1 < T and maybe[b + 1]:
            b += 1
        ok = bool(fragile[a:b + 1].any())
        if not ok:
            flat = np.sort(np.asarray(ref_probs)[a:b + 1].reshape(-1))[::-1]
            ok = len(flat) > 1 and float(flat[0] - flat[1]) < 2 * eps
        (explained if ok else unexplained).append((t, c))
    return explained, unexplained


def compare_lines(R, om, rec, lines, max_prob_err=None):
    """Every line through `rec` (free-running, the kernels' own state all the way) and through the
    float64 oracle: returns a dict with per-line logit / probability errors, the product's and the
    oracle's decodes and the explained / unexplained decode differences."""
    dec, probs, logits, states = rec.recognise(lines, want_probs=True)
    refs = [R.recognise(om, xs) for xs in lines]
    out = dict(dec=dec, refs=refs, logit_err=[], prob_err=[], explained=[], unexplained=[], chars=0, chars_agree=0)
    for k in range(len(lines)):
        ez = float(np.abs(logits[k] - refs[k]["logits"]).max())
        ep = float(np.abs(probs[k] - refs[k]["probs"]).max())
        out["logit_err"].append(ez)
        out["prob_err"].append(ep)
        ex, un = decode_differences(refs[k]["probs"], refs[k]["decoded"], dec[k], max(ep, 1e-6) * 1.5)
        out["explained"].append(ex)
        out["unexplained"].append(un)
        out["chars"] += len(refs[k]["decoded"])
        out["chars_agree"] += len(set(map(tuple, dec[k])) & set(map(tuple, refs[k]["decoded"])))
    return out
