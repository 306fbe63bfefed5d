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

if RodanTask is not None:
    class textAlignment(RodanTask):
        name = 'Text Alignment'
        author = 'Timothy de Reuse'
        description = 'Given a text layer image and plaintext of some text on that page, finds the ' \
                      'positions of each syllable of text in the image (MI355X build).'
        enabled = True
        category = 'text'
        interactive = False
        settings = {
            'title': 'Text Alignment Settings',
            'type': 'object',
            'required': ['MEI Version'],
            'properties': {
                'MEI Version': {
                    'enum': ['4.0.0', '3.9.9'],
                    'type': 'string',
                    'default': '3.9.9',
                    'description': 'Specifies the MEI version, 3.9.9 is the old unofficial MEI standard used by Neon',
                },
            },
        }
        input_port_types = [
            {'name': 'Text Layer', 'resource_types': ['image/rgba+png'], 'minimum': 1, 'maximum': 1, 'is_list': False},
            {'name': 'Transcript', 'resource_types': ['text/plain'], 'minimum': 1, 'maximum': 1, 'is_list': False},
        ]
        output_port_types = [
            {'name': 'JSON', 'resource_types': ['application/JSON'], 'minimum': 1, 'maximum': 1, 'is_list': False},
        ]

        def run_my_task(self, inputs, settings, outputs):
            return run_alignment(inputs['Text Layer'][0]['resource_path'],
                                 inputs['Transcript'][0]['resource_path'],
                                 outputs['JSON'][0]['resource_path'])
