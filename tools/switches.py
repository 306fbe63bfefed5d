"""Experiment switches of the timing / profiling tools.

The product modules read no environment variables: their tunables are plain module attributes
(text_alignment_amd.ocr.FORCE_GROUP, .F64_CLASS_PIPELINE, ...; alignToOCR.PIPELINE_CHUNK_PAGES, ...).
The tools under tools/ are run under rocprofv3 and from shell scripts (tools/profile_round.sh), where the
environment is the only handle -- so THEY translate it, once, after importing the product:

    TA_OCR_GROUP=4|16          ocr.FORCE_GROUP            lines per workgroup of the recurrence (f32 / f64 modes)
    TA_OCR_CLASS_SPLIT=0|1     ocr.FORCE_CLASS_SPLIT      split mode: K3 + K4 per length class on side streams
    TA_OCR_F64_PIPE=0|1        ocr.F64_CLASS_PIPELINE     f64 mode: projection / recurrence pipelined per length class
    TA_OCR_F64_CUTS=a,b        ocr.F64_CLASS_CUTS         ... its cuts (shares of the groups)
    TA_OCR_COPY_THREADS=n      ocr.COPY_THREADS           pool threads that stage pageable rows (before the pool's first use)
    TA_PAGE_CHUNK=n            alignToOCR.PIPELINE_CHUNK_PAGES          pages per pipeline chunk (normalised rows)
    TA_PAGE_CHUNK_RAW=n        alignToOCR.PIPELINE_CHUNK_PAGES_RAW      ... raw strips
    TA_PAGE_CHUNK_IMAGES=n     alignToOCR.PIPELINE_CHUNK_PAGES_IMAGES   ... page images
    TA_PB_TWO_STREAMS=0|1      alignToOCR.TWO_STREAMS     consecutive chunks on two compute streams
    TA_PB_LEAD_DIVISOR=n       alignToOCR.LEAD_CHUNK_DIVISOR   the call's first chunk is 1/n of a chunk (1: a whole one)
    TA_BIND=1                  sharding.bind_to_gpu_node()   the placement bench.py gives a rank (cores of the GPU's NUMA node)
    TA_SWITCH_INTERVAL=s       sys.setswitchinterval      seconds a thread may hold the interpreter lock against a waiting one
    TA_PP_LABEL_PIXELS=1       preproc_gpu.LABEL_FLAGS    connected components by a label per pixel (rounds 3-5) instead of over runs
    TA_PP_BATCH=n              textAlignPreprocessing.PAGES_PER_BATCH   page images: pages per preprocessing batch
    TA_PP_THREADS=n            textAlignPreprocessing.PAGE_THREADS      ... batches in flight (a host thread and a stream each)
"""
import os


def apply(environ=None):
    """set the product modules' attributes from TA_* variables; returns what was set"""
    env = os.environ if environ is None else environ
    from text_alignment_amd import alignToOCR as atocr, ocr
    done = {}

    def put(mod, attr, value):
        setattr(mod, attr, value)
        done["%s.%s" % (mod.__name__.rsplit(".", 1)[-1], attr)] = value
    v = env.get("TA_OCR_GROUP")
    if v in ("4", "16"):
        put(ocr, "FORCE_GROUP", int(v))
    v = env.get("TA_OCR_CLASS_SPLIT")
    if v in ("0", "1"):
        put(ocr, "FORCE_CLASS_SPLIT", v == "1")
    v = env.get("TA_OCR_F64_PIPE")
    if v in ("0", "1"):
        put(ocr, "F64_CLASS_PIPELINE", v == "1")
    v = env.get("TA_OCR_F64_CUTS")
    if v:
        put(ocr, "F64_CLASS_CUTS", tuple(float(x) for x in v.split(",") if x)[:3])
    for var, attr in (("TA_PAGE_CHUNK", "PIPELINE_CHUNK_PAGES"), ("TA_PAGE_CHUNK_RAW", "PIPELINE_CHUNK_PAGES_RAW"),
                      ("TA_PAGE_CHUNK_IMAGES", "PIPELINE_CHUNK_PAGES_IMAGES")):
        v = env.get(var)
        if v and v.isdigit() and int(v) > 0:
            put(atocr, attr, int(v))
    for var, attr in (("TA_PB_TWO_STREAMS", "TWO_STREAMS"),):
        v = env.get(var)
        if v in ("0", "1"):
            put(atocr, attr, v == "1")
    v = env.get("TA_OCR_COPY_THREADS")
    if v and v.isdigit() and int(v) > 0:
        put(ocr, "COPY_THREADS", int(v))
    v = env.get("TA_PB_LEAD_DIVISOR")
    if v and v.isdigit():
        put(atocr, "LEAD_CHUNK_DIVISOR", int(v))
    if env.get("TA_BIND") == "1":                    # the placement bench.py gives a rank: cores of the GPU's NUMA node, one torch thread
        from text_alignment_amd import sharding
        done["bind"] = sharding.bind_to_gpu_node()
    v = env.get("TA_SWITCH_INTERVAL")
    if v:
        import sys
        sys.setswitchinterval(float(v))
        done["sys.switchinterval"] = float(v)
    if env.get("TA_PP_LABEL_PIXELS") == "1":
        from text_alignment_amd import _native, preproc_gpu
        put(preproc_gpu, "LABEL_FLAGS", _native.TA_PP_LABEL_PIXELS)
    from text_alignment_amd import textAlignPreprocessing as preproc
    for var, attr in (("TA_PP_BATCH", "PAGES_PER_BATCH"), ("TA_PP_THREADS", "PAGE_THREADS")):
        v = env.get(var)
        if v and v.isdigit() and int(v) > 0:
            put(preproc, attr, int(v))
    return done
