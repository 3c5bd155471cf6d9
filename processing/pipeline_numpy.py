from raw2logit_amd.processing.pipeline_numpy import (  # noqa: F401
    RawProcessingPipeline, StaticProcessing, processing)
