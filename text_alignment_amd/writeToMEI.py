"""Attach aligned syllable text to an MEI file -- host side, pure
ElementTree.  Counterpart of the reference module of the same name (writeToMEI.py:14-145), fixed to
what `alignToOCR.process` returns today (a 4-tuple whose syllable boxes are CharBox objects; the
reference still indexes them as (text, ul, lr) sequences, writeToMEI.py:97-98 -- both forms are
accepted here).  Pinned to golden vectors captured from the imported reference
(tests/golden/mei.json).  The PIL page rendering of the reference's __main__ (writeToMEI.py:183-214)
is a debugging aid and is not rebuilt.
"""
import xml.etree.ElementTree as ET

import numpy as np

XML_NS = '{http://www.w3.org/XML/1998/namespace}'
MEI_URI = 'http://www.music-encoding.org/ns/mei'
MEI_NS = '{%s}' % MEI_URI


def intersect(ul1, lr1, ul2, lr2):
    """area of the intersection of two boxes given by corner pairs, or False (writeToMEI.py:14-20)"""
    d1 = min(lr1[1], lr2[1]) - max(ul1[1], ul2[1])
    d0 = min(lr1[0], lr2[0]) - max(ul1[0], ul2[0])
    return d1 * d0 if (d1 > 0 and d0 > 0) else False


def generate_id():
    """xml:id of the form m-8-4-4-4-12 hex digits drawn from np.random (writeToMEI.py:24-30; the
    draws come in the same order and ranges, so a seeded run reproduces the reference's ids)"""
    parts = [hex(np.random.randint(0, 16 ** digits))[2:] for digits in (8, 4, 4, 4, 12)]
    return 'm-' + '-'.join(parts)


def repair_xml(xml_input):
    """declare the xlink prefix pitch-finding output forgets (writeToMEI.py:33-37)"""
    at = xml_input.index('meiversion')
    return xml_input[:at] + 'xmlns:xlink="http://www.w3.org/1999/xlink" ' + xml_input[at:]


def parse_mei(raw_xml):
    """MEI text -> ElementTree, with the namespace handling of writeToMEI.py:166-174"""
    ET.register_namespace('', MEI_URI)
    try:
        root = ET.fromstring(raw_xml)
    except ET.ParseError:
        root = ET.fromstring(repair_xml(raw_xml))
    return ET.ElementTree(root)


def _fields(box):
    """(text, ul, lr) of a syllable box: a CharBox or the reference's older 3-sequence"""
    if hasattr(box, 'char'):
        return box.char, box.ul, box.lr
    return box[0], box[1], box[2]


def add_text_to_mei_file(tree, syls_boxes, med_line_spacing):
    """For every <syllable> (one neume each on input) find the text box its neume hangs over --
    the neume's bounding box is pushed down by half a line and the syllable box with the largest
    overlap wins -- then merge runs of neumes over the same text (or over none) into one
    <syllable> carrying a <syl> and a new <zone> (writeToMEI.py:41-145).
    Returns (tree, neume bounding boxes, neume-to-text lines for visualisation)."""
    root = tree.getroot()
    parent_of = {child: parent for parent in tree.iter() for child in parent}
    surface = root.findall('.//%ssurface' % MEI_NS)[0]
    zone_of = {z.attrib[XML_NS + 'id']: z.attrib for z in root.findall('.//%szone' % MEI_NS)}
    boxes = [_fields(b) for b in syls_boxes]

    all_bboxes, assign_lines, emptied = [], [], []
    current = None              # <syllable> element neumes are being gathered into
    prev_hit = None             # index of the text box the previous neume hit (None: no text)
    last_assigned = None        # last text box any neume hit
    for se in root.findall('.//%ssyllable' % MEI_NS):
        neume = se[0]
        if current is None or len(current) == 0:
            current = se
        assert 'neume' in neume.tag
        parts = [zone_of[nc.attrib['facs']] for nc in neume.findall(MEI_NS + 'nc')]
        ulx = min(int(z['ulx']) for z in parts)
        uly = min(int(z['uly']) for z in parts)
        lrx = max(int(z['lrx']) for z in parts)
        lry = max(int(z['lry']) for z in parts)
        all_bboxes.append([ulx, uly, lrx, lry])

        probe_ul, probe_lr = (ulx, uly + med_line_spacing / 2), (lrx, lry + med_line_spacing)
        hit, best = None, 0
        for k, (_, ul, lr) in enumerate(boxes):          # first box of maximal overlap, as max()
            area = intersect(ul, lr, probe_ul, probe_lr)
            if area > best:
                hit, best = k, area
        if hit is not None:
            last_assigned = hit

        same_text = hit is not None and prev_hit is not None and \
            (hit == prev_hit or boxes[hit] == boxes[prev_hit])
        if hit is None or same_text:
            current.append(neume)                        # belongs to the syllable in progress
            emptied.append(se)
        else:
            current = se
            syl = ET.Element('syl')
            syl.text = boxes[hit][0]
            current.insert(0, syl)
            zone = ET.SubElement(surface, '%szone' % MEI_NS)
            new_id = generate_id()
            current.set('facs', new_id)
            zone.set(XML_NS + 'id', new_id)
            zone.set('lrx', str(lrx))
            zone.set('lry', str(lry))
            zone.set('ulx', str(ulx))
            zone.set('uly', str(uly))
        if last_assigned is not None:
            ul = boxes[last_assigned][1]
            assign_lines.append([ulx, uly, ul[0], ul[1]])
        prev_hit = hit

    for el in emptied:
        parent_of[el].remove(el)
    return tree, all_bboxes, assign_lines


def write_mei(raw_xml, syl_boxes, lines_peak_locs, out_path=None):
    """The reference's per-page flow (writeToMEI.py:162-181) after `process`: parse, attach the
    text with the 75th-percentile line spacing (as to_JSON_dict, alignToOCR.py:338), serialise."""
    tree = parse_mei(raw_xml)
    spacing = np.quantile(np.diff(lines_peak_locs), 0.75)
    tree, _, _ = add_text_to_mei_file(tree, syl_boxes, spacing)
    if out_path is not None:
        tree.write(out_path)
    return tree
