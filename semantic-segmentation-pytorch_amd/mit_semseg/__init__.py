"""
MIT CSAIL Semantic Segmentation -- MI355X (gfx950) native build of the encoder->decoder hot path.

Drop-in for the `mit_semseg` import surface used by train.py / eval.py of
CSAILVision/semantic-segmentation-pytorch (mit_semseg/__init__.py:1-5): same package, class and
argument names, state-dict keys and error behaviour; all arithmetic in libsemseg_hip.so.
"""

__version__ = '1.0.0+mi355x'
