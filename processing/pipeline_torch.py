from raw2logit_amd.processing.pipeline_torch import *  # noqa: F401,F403
from raw2logit_amd.processing.pipeline_torch import (  # noqa: F401
    K_G, K_RB, K_BLUR, K_SHARP, M_RGB_2_YUV, M_YUV_2_RGB, DEFAULT_CAMERA_PARAMS, RawToRGB, NNProcessing,
    ParametrizedProcessing, Debayer, raw2rgb, append_additive_layer)
