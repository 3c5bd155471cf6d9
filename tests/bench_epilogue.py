"""Weak augmentation after the ISP: separate permutation kernel (+ its inverse in the backward) vs the forward's output
epilogue (R2L_STEP_EPI_*), fwd + bwd step time, BatchNorm train."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raw2logit_amd import cameras, augmentation as aug
from raw2logit_amd.processing.pipeline_torch import ParametrizedProcessing
dev = 'cuda'
for B, S in ((64, 512), (128, 256), (64, 256)):
    raw = torch.randint(0, 4096, (B, S, S), device=dev, dtype=torch.int32).to(torch.float32) / 4095.0
    cot = torch.randn((B, 3, S, S), device=dev)
    m = ParametrizedProcessing(cameras.DRONE, batch_norm_output=True).to(dev).train()
    params = list(m.parameters())
    m.fuse_rot90 = True     # 'epilogue' = the kernels' own stores for rotations too (the module's default sends them to the
                            # permutation kernel, which transposes through LDS tiles: the 'separate' column)
    for (h, v, k) in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 0, 2), (0, 0, 1), (1, 0, 3)):
        res = []
        for mode in ('separate', 'epilogue'):
            def step():
                for p in params:
                    p.grad = None
                if mode == 'epilogue' and (h or v or k):
                    m.__dict__['_epilogue'] = (h, v, k)
                    y = m(raw)
                else:
                    y = aug.flip_rot(m(raw), h, v, k)
                y.backward(cot)
            for _ in range(30):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                step()
            torch.cuda.synchronize()
            res.append(1e3 * (time.perf_counter() - t0) / 50)
        print(f'{B:4d}x{S}^2  hflip={h} vflip={v} k={k}   separate kernel(s) {res[0]:.4f} ms/step   output epilogue {res[1]:.4f} ms/step')
