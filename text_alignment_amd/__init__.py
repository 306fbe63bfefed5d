"""text_alignment_amd -- MI355X-native hot path of DDMAL/text_alignment.

Drop-in surface (same names, arguments and return values as the reference):
    text_alignment_amd.textSeqCompare.perform_alignment   (reference textSeqCompare.py:13)
    text_alignment_amd.alignToOCR.process / to_JSON_dict / CharBox / perform_ocr_with_ocropus
                                                          (reference alignToOCR.py:35-351)
All arithmetic runs in hand-written HIP kernels (csrc/) behind the C ABI of
include/text_alignment_amd.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
