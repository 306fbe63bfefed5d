# -*- coding: utf-8 -*-
"""Rule-based Latin syllabifier -- behavioural restatement of the reference module of the same
name (reference latinSyllabification.py:5-174), used on the `process` surface
(alignToOCR.py:251, :277).  Pure host-side string work; nothing here is accelerated.

One deliberate deviation: a non-empty word without any vowel or diphthong ('dns', 'st', 'b')
makes the reference's merge loop spin forever (latinSyllabification.py:71); here such a word is
returned as a single syllable.
"""
import re

consonant_groups = ['qu', 'ch', 'ph', 'fl', 'fr', 'st', 'br', 'cr', 'cl', 'pr', 'tr', 'ct', 'th']
diphthongs = ['ae', 'au', 'ei', 'oe', 'ui', 'ya', 'ex', 'ix']
vowels = ['a', 'e', 'i', 'o', 'u', 'y']

# insertion order is the order process() tries them in (latinSyllabification.py:9-19)
abbreviations = {
    u'dns': ['do', 'mi', 'nus'],
    u'dūs': ['do', 'mi', 'nus'],
    u'dne': ['do', 'mi', 'ne'],
    u'alla': ['al', 'le', 'lu', 'ia'],
    u'^': ['us'],
    u'ā': ['am'],
    u'ē': ['em'],
    u'ū': ['um'],
    u'ō': ['om']
}

_FIXED = {'euouae': ['e', 'u', 'o', 'u', 'ae'], 'cuius': ['cu', 'ius'], 'eius': ['e', 'ius']}
_NUCLEI = set(vowels + diphthongs)


def _units(word):
    """Cut a word into units: consonant groups first, then diphthongs (each list in order, each
    pass only on text no earlier pass claimed), then single letters
    (latinSyllabification.py:37-63).  Returns [(text, claimed_by_a_pass)]."""
    pieces = [(word, False)]
    for unit in consonant_groups + diphthongs:
        nxt = []
        for text, claimed in pieces:
            if claimed or '*' in text:
                nxt.append((text, claimed))
                continue
            parts = text.split(unit)
            for k, part in enumerate(parts):
                if part:
                    nxt.append((part, False))
                if k + 1 < len(parts):
                    nxt.append((unit, True))
        pieces = nxt
    out = []
    for text, claimed in pieces:
        if claimed:
            out.append(text)
        else:
            out.extend(text)
    return out


_memo = {}


def syllabify_word(inp):
    '''
    Units that are a vowel or a diphthong seed a syllable; every other unit first sticks to the
    seed right after it, and what is still loose then sticks to the syllable before it
    (latinSyllabification.py:65-107).
    '''
    if inp in _FIXED:
        return list(_FIXED[inp])
    hit = _memo.get(inp)
    if hit is not None:
        return list(hit)
    units = _units(inp)
    seeded = [u in _NUCLEI for u in units]
    if units and not any(seeded):
        return [inp]                      # the reference would never return (see module docstring)
    while not all(seeded):
        for forward in (True, False):
            merged, flags = [], []
            k = 0
            while k < len(units):
                if k + 1 < len(units):
                    a, b = seeded[k], seeded[k + 1]
                    if (forward and b and not a) or (not forward and a and not b):
                        merged.append(units[k] + units[k + 1])
                        flags.append(True)
                        k += 2
                        continue
                merged.append(units[k])
                flags.append(seeded[k])
                k += 1
            units, seeded = merged, flags
    if len(_memo) < 100000:
        _memo[inp] = tuple(units)
    return units


def syllabify_text(input):
    syls = []
    memo = _memo
    for word in input.split(' '):
        hit = memo.get(word)                  # (chant texts repeat their words: most of them are here already)
        syls.extend(hit if hit is not None else syllabify_word(word))
    return syls
