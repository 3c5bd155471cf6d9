"""raw2logit_amd -- the raw-Bayer -> RGB ISP hot path of aiaudit-org/raw2logit as hand-written
gfx950 (MI355X) kernels behind the reference's own Python API.

    from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing, RawToRGB, raw2rgb
    from raw2logit_amd.processing.pipeline_numpy import RawProcessingPipeline, StaticProcessing

or, unchanged reference imports (train.py:23-24), through the top-level ``processing`` package of this
repository."""
from . import _lib  # noqa: F401

__version__ = '0.1.0'
