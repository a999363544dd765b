"""
MIT CSAIL Semantic Segmentation -- MI355X (gfx950) native build of the encoder->decoder hot path.

Drop-in for the `mit_semseg` import surface used by train.py / eval.py / test.py of
CSAILVision/semantic-segmentation-pytorch (mit_semseg/__init__.py:1-5): `models`, `lib.nn`, `lib.utils`, `config`, `dataset`,
`utils` -- same package, class and argument names, state-dict keys and error behaviour; all arithmetic in libsemseg_hip.so.
One process per GPU: the reference's single-process `--gpus 0-7` recipe is served by this build's own train.py / eval_multipro.py,
which spawn the ranks (mit_semseg/drivers.py).
"""

__version__ = '1.0.0+mi355x'
