"""Pin the CPU oracle (oracle/nw_oracle.c, oracle/nw_ref_py.py) to golden vectors captured
from the imported reference (textSeqCompare.py:13-177) by tools/gen_golden.py."""
import hashlib

import numpy as np
import pytest

from conftest import kat_scoring, load_golden, unrle
from oracle import nw_oracle, nw_ref_py
from tools.synth import synth_pair, synth_pair_ids


def _sha16(tra, ocr):
    return hashlib.sha256(("".join(tra) + "|" + "".join(ocr)).encode()).hexdigest()[:16]


@pytest.mark.parametrize("impl", [nw_oracle, nw_ref_py], ids=["c", "py"])
def test_kat(impl):
    g = load_golden("nw_kat.json")
    for c in g["cases"]:
        tra, ocr = impl.perform_alignment(c["transcript"], c["ocr"], kat_scoring(c))
        assert tra == c["tra_align"], c["name"]
        assert ocr == c["ocr_align"], c["name"]
    for e in g["errors"]:
        if e["raises"]:
            with pytest.raises(ValueError) as ei:
                impl.perform_alignment(list("ab"), list("ab"), e["scoring"])
            assert str(ei.value) == e["message"]


def test_survey_kat_strings():
    # SURVEY.md Appendix D, KAT-2 (captured in the survey session)
    s1 = 'Lorem ipsum dolor sit amet, consectetur adipiscing elit '
    s2 = 'LoLorem fipsudolor ..... sit eamet, c.nnr adizisdcing eelitellit'
    tra, ocr = nw_oracle.perform_alignment(list(s1), list(s2))
    assert "".join(tra) == '__Lorem _ipsum dolor______ sit _amet, consectetur adipis_cing _elit ____'
    assert "".join(ocr) == 'LoLorem fipsu__dolor ..... sit eamet, c.n______nr adizisdcing eelitellit'


@pytest.mark.parametrize("impl", [nw_oracle, nw_ref_py], ids=["c", "py"])
def test_random_small(impl):
    g = load_golden("nw_random_small.json")
    assert len(g["cases"]) >= 200
    for c in g["cases"]:
        tra, ocr = impl.perform_alignment(list(c["t"]), list(c["o"]), c["scoring"])
        assert "".join(tra) == c["tra"] and "".join(ocr) == c["ocr"], c


def test_inputs_not_mutated():
    t, o = list("abcd"), list("xbcy")
    nw_oracle.perform_alignment(t, o)
    assert t == list("abcd") and o == list("xbcy")


def test_synth_generator_pinned():
    t, _ = synth_pair(64, 64, 1234)
    assert "".join(t[:20]) == '   keychdiodvgviv zh'


def test_synth_c_oracle():
    g = load_golden("nw_synth.json")
    for c in g["cases"]:
        t, o = synth_pair(c["n"], c["m"], c["seed"])
        assert "".join(t[:20]) == c["t_head"]
        tra, ocr = nw_oracle.perform_alignment(t, o, c["scoring"])
        assert len(tra) == c["align_len"]
        assert _sha16(tra, ocr) == c["sha16"], (c["n"], c["m"], c["seed"])
        t_ids, o_ids = synth_pair_ids(c["n"], c["m"], c["seed"])
        params, _ = nw_oracle.parse_scoring(c["scoring"])
        ops = nw_oracle.align_ids(t_ids, o_ids, params)
        assert list(ops) == unrle(c["ops_rle"])


def test_synth_py_port_small():
    g = load_golden("nw_synth.json")
    for c in g["cases"]:
        if c["n"] * c["m"] > 70_000:
            continue
        t, o = synth_pair(c["n"], c["m"], c["seed"])
        tra, ocr = nw_ref_py.perform_alignment(t, o, c["scoring"])
        assert _sha16(tra, ocr) == c["sha16"]


def test_pointer_matrix_consistency():
    # the packed pointer bytes the kernel tests compare against: PM | PX<<2 | PY<<4
    t_ids, o_ids = synth_pair_ids(37, 53, 5)
    ops, ptr, sc = nw_oracle.align_ids(t_ids, o_ids, [8, -4, -7, -7, -3, 0], want_ptr=True)
    assert ptr.shape == (38, 54)
    assert (ptr[0, :] == 0).all() and (ptr[:, 0] == 0).all()
    assert ((ptr & 3) < 3).all() and (((ptr >> 2) & 3) < 3).all() and (((ptr >> 4) & 3) < 3).all()
    assert np.isfinite(sc[0])
