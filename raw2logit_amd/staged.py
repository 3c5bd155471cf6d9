"""Staged (track_stages=True) execution of ParametrizedProcessing -- placeholder until the per-stage
kernels land; fails loudly instead of silently falling back to ATen ops."""


def staged_forward(module, raw):
    raise NotImplementedError(
        'track_stages=True (materialised per-stage tensors with retain_grad, reference '
        'pipeline_torch.py:197-221) is not built yet in raw2logit_amd; use track_stages=False')
