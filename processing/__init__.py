"""Top-level alias so that the reference's imports work unchanged against this repository:

    from processing.pipeline_torch import raw2rgb, RawToRGB, ParametrizedProcessing, NNProcessing   # train.py:24
    from processing.pipeline_numpy import RawProcessingPipeline                                     # train.py:23
"""
