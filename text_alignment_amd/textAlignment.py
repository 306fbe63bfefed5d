"""Rodan job wrapper, counterpart of the reference's
textAlignment.py:1-63: one text-layer image + one transcript in, the syl_boxes JSON out.

Rodan is a server framework that is not a dependency of this package: the task class is only
defined when `rodan` is importable; `run_alignment` is the body of `run_my_task` on plain paths and
is what the tests exercise.  Unlike the reference (textAlignment.py:56 unpacks three values from a
`process` that returns four), this follows the current 4-tuple.
"""
import json

from . import alignToOCR as align

DEFAULT_MODEL = './models/salzinnes_model-00054500.pyrnn.gz'        # alignToOCR.py:390-405


def load_text_layer(path):
    """PNG -> uint8 array for textAlignPreprocessing (Gamera's load_image in the reference)"""
    import numpy as np
    from PIL import Image
    return np.asarray(Image.open(path).convert('L'))


def run_alignment(image_path, transcript_path, out_json_path, ocropus_model=DEFAULT_MODEL,
                  seq_align_params=None):
    transcript = align.read_file(transcript_path)
    raw_image = load_text_layer(image_path)
    result = align.process(raw_image, transcript, ocropus_model, seq_align_params=seq_align_params,
                           wkdir_name='test', verbose=False)
    if result is None:
        return False
    syl_boxes, _, lines_peak_locs, _ = result
    with open(out_json_path, 'w') as f:
        json.dump(align.to_JSON_dict(syl_boxes, lines_peak_locs), f)
    return True


try:
    from rodan.jobs.base import RodanTask
except ImportError:                 # not running inside Rodan
    RodanTask = None

def _port(name, mime):
    """one mandatory single-resource Rodan port"""
    return {'name': name, 'resource_types': [mime], 'minimum': 1, 'maximum': 1, 'is_list': False}


if RodanTask is not None:
    class textAlignment(RodanTask):
        # job metadata as Rodan shows it (names and port layout of reference textAlignment.py:7-49)
        name = 'Text Alignment'
        author = 'Timothy de Reuse'
        description = ('Finds the position on the page of every syllable of a known transcript, '
                       'given the text layer of the page image (MI355X build).')
        enabled, category, interactive = True, 'text', False
        settings = {
            'title': 'Text Alignment Settings',
            'type': 'object',
            'required': ['MEI Version'],
            'properties': {'MEI Version': {
                'type': 'string', 'enum': ['4.0.0', '3.9.9'], 'default': '3.9.9',
                'description': '3.9.9 is the unofficial MEI flavour Neon reads; 4.0.0 the released standard'}},
        }
        input_port_types = [_port('Text Layer', 'image/rgba+png'), _port('Transcript', 'text/plain')]
        output_port_types = [_port('JSON', 'application/JSON')]

        def run_my_task(self, inputs, settings, outputs):
            def path(ports, name):
                return ports[name][0]['resource_path']
            return run_alignment(path(inputs, 'Text Layer'), path(inputs, 'Transcript'), path(outputs, 'JSON'))
