"""Drop-in for the reference's ``utils/augmentation.py`` (weak set) with the pixel moves on the GPU.

``augmentation_weak`` = RandomHorizontalFlip, RandomVerticalFlip, RandomRotate90 (utils/augmentation.py:70-74),
applied to the processor's output batch between ISP and classifier (model.py:79-81) and, for segmentation, to
the masks with the same random state (``retain_state`` / ``mask_transform``, :36-67).  The random draws are made
on the host exactly where torchvision / the reference make them (``torch.rand(1) < p``, ``random.randint``), so a
seeded run takes the same decisions; the three moves of a call are then fused into ONE permutation kernel
(``r2l_augment``) instead of up to three passes, with the inverse permutation as its VJP.

``augmentation_strong`` additionally needs torchvision's RandomRotation / RandomAdjustSharpness (absent from
the image, resampling ops outside the ISP path): not built -- ``get_augmentation('strong')`` raises."""
import random

import numpy as np
import torch

from . import _lib
from ._lib import ptr
from .functional import _f32c


class _FlipRot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, hflip, vflip, k):
        x = _f32c(x, 'x')
        H, W = x.shape[-2:]
        N = x.numel() // (H * W)
        lib, stream = _lib.library_for(x)
        out_shape = tuple(x.shape[:-2]) + ((W, H) if (k & 1) else (H, W))
        y = torch.empty(out_shape, dtype=torch.float32, device=x.device)
        lib.check(lib.r2l_augment(ptr(x), ptr(y), N, H, W, int(hflip), int(vflip), int(k), 0, stream), 'r2l_augment')
        ctx.meta = (N, H, W, int(hflip), int(vflip), int(k), tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        N, H, W, hflip, vflip, k, shape = ctx.meta
        g = _f32c(g, 'g')
        lib, stream = _lib.library_for(g)
        gx = torch.empty(shape, dtype=torch.float32, device=g.device)
        lib.check(lib.r2l_augment(ptr(g), ptr(gx), N, H, W, hflip, vflip, k, 1, stream), 'r2l_augment(inverse)')
        return gx, None, None, None


def flip_rot(x, hflip=False, vflip=False, k=0):
    """rot90^k(vflip(hflip(x))) over the last two axes, k as in ``x.rot90(k, dims=(-1, -2))``; one kernel."""
    if not (hflip or vflip or (k & 3)):
        return x
    return _FlipRot.apply(x, bool(hflip), bool(vflip), int(k) & 3)


class _Pending:
    """moves decided by the transforms of one ComposeState call, applied together at the end"""

    def __init__(self):
        self.hflip = self.vflip = False
        self.k = 0

    def flush(self, x):
        x = flip_rot(x, self.hflip, self.vflip, self.k)
        self.__init__()
        return x


class RandomHorizontalFlip:
    """torchvision.transforms.RandomHorizontalFlip: one draw per call, the whole batch flips (:71)."""

    def __init__(self, p=0.5):
        self.p = p

    def decide(self, pending):
        if torch.rand(1) < self.p:
            if pending.k:            # a flip after a rotation does not commute: apply what is pending first
                return True
            pending.hflip = not pending.hflip
        return False

    def __call__(self, x):
        return flip_rot(x, hflip=bool(torch.rand(1) < self.p))

    def __repr__(self):
        return f'{self.__class__.__name__}(p={self.p})'


class RandomVerticalFlip(RandomHorizontalFlip):
    """torchvision.transforms.RandomVerticalFlip (:72)."""

    def decide(self, pending):
        if torch.rand(1) < self.p:
            if pending.k:
                return True
            pending.vflip = not pending.vflip
        return False

    def __call__(self, x):
        return flip_rot(x, vflip=bool(torch.rand(1) < self.p))


class RandomRotate90:  # Note: not the same as T.RandomRotation(90)
    """utils/augmentation.py:8-14."""

    def decide(self, pending):
        pending.k = (pending.k + random.randint(0, 3)) & 3
        return False

    def __call__(self, x):
        return flip_rot(x, k=random.randint(0, 3))

    def __repr__(self):
        return self.__class__.__name__


class _PhiloxNoise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, std, seed, offset):
        x = _f32c(x, 'x')
        lib, stream = _lib.library_for(x)
        y = torch.empty_like(x)
        lib.check(lib.r2l_add_noise_philox(ptr(x), ptr(y), float(std), int(seed), int(offset), x.numel(), stream),
                  'r2l_add_noise_philox')
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None, None, None          # the noise does not depend on x


def add_gaussian_noise(x, std, seed, offset=0):
    """x + std * N(0,1), the deviates generated inside the kernel (Philox4x32-10 keyed by `seed`, Box-Muller): a pure
    function of (seed, offset, element index)"""
    return _PhiloxNoise.apply(x, std, seed, offset)


class AddGaussianNoise:
    """utils/augmentation.py:17-31: x + randn_like(x) * std.  The deviates come from an in-kernel Philox generator
    (no noise tensor, one pass over x); its 63-bit seed is drawn from torch's CPU generator, which
    ``set_global_seed`` seeds -- a seeded run reproduces its noise (same distribution as the reference's
    torch.randn_like, not the same stream)."""

    def __init__(self, std=0.01):
        self.std = std

    def __call__(self, x):
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        return add_gaussian_noise(x, self.std, seed)

    def __repr__(self):
        return self.__class__.__name__ + f'(std={self.std})'


def set_global_seed(seed):
    """utils/augmentation.py:34-37."""
    torch.random.manual_seed(seed)
    np.random.seed(seed % (2**32 - 1))
    random.seed(seed)


class ComposeState:
    """utils/augmentation.py:40-67: a Compose that can replay its random state for the masks."""

    def __init__(self, transforms):
        self.transforms = []
        self.mask_transforms = []
        for t in transforms:
            apply_for_mask = True
            if isinstance(t, tuple):
                t, apply_for_mask = t
            self.transforms.append(t)
            if apply_for_mask:
                self.mask_transforms.append(t)
        self.seed = None

    def _enter(self, retain_state):
        """the reference's seed bookkeeping at the head of a call (:51-57)"""
        if self.seed is not None:   # retain previous state
            set_global_seed(self.seed)
        if retain_state:    # save state for next call
            self.seed = self.seed or torch.seed()
            set_global_seed(self.seed)
        else:
            self.seed = None    # reset / ignore state

    def arm(self, processor, retain_state=False):
        """Make the NEXT ``processor(x)`` call return the augmented batch: the draws of this transform list are made NOW
        (the processor consumes no host randomness, so a seeded run takes the same decisions as the reference, which
        draws after the processor call) and handed to the processor as a one-shot output epilogue -- the fused kernels
        then write the flipped / rotated output themselves instead of a separate 24 B/px pass, and the backward reads
        ``grad_out`` through the same map.  The ``self(x, retain_state=...)`` call that follows in model.py:79-81 returns
        its argument unchanged (once).  In LitModel.forward that is ONE added line in front of ``self.processor(x)``:

            if self.augmentation is not None and apply_augmentation_step:
                self.augmentation.arm(self.processor, retain_state=self.is_segmentation_task)

        Returns False (and arms nothing) for transform lists that hold anything but flips / rot90, or an order of them
        that does not commute into one permutation.  ``processor.buffer['processed_rgb']`` of an armed call holds the
        augmented output."""
        self._armed = None
        # only a processor that pops `_epilogue` itself may be armed (ParametrizedProcessing declares it); RawToRGB,
        # NNProcessing, nn.Identity (train.py:173-202) get the separate permutation kernel in __call__ as before
        if not getattr(processor, 'supports_output_epilogue', False):
            return False
        if not all(hasattr(t, 'decide') for t in self.transforms):
            return False
        state = (torch.random.get_rng_state(), np.random.get_state(), random.getstate(), self.seed)
        self._enter(retain_state)
        pending = _Pending()
        for t in self.transforms:
            if t.decide(pending):           # needs a flush in between: not one permutation -- undo the draws
                torch.random.set_rng_state(state[0])
                np.random.set_state(state[1])
                random.setstate(state[2])
                self.seed = state[3]
                return False
        processor.__dict__['_epilogue'] = (pending.hflip, pending.vflip, pending.k)
        self._armed = processor
        return True

    def __call__(self, x, retain_state=False, mask_transform=False):
        armed = getattr(self, '_armed', None)
        if armed is not None and not mask_transform:
            self._armed = None
            # consumed: the processor's epilogue already applied this call's moves.  Still there (the processor raised
            # before its pop, or was never called): the draws are made, apply them here so that image and mask agree
            left = armed.__dict__.pop('_epilogue', None)
            return x if left is None else flip_rot(x, *left)
        self._enter(retain_state)
        transforms = self.transforms if not mask_transform else self.mask_transforms
        pending = _Pending()
        for t in transforms:
            if hasattr(t, 'decide'):          # flip / rot90: drawn now (reference order), moved once at the end
                if t.decide(pending):         # the move does not commute with what is pending: flush, then redo
                    x = pending.flush(x)
                    if isinstance(t, RandomVerticalFlip):
                        pending.vflip = True
                    else:
                        pending.hflip = True
            else:
                x = t(pending.flush(x))
        return pending.flush(x)


augmentation_weak = ComposeState([
    RandomHorizontalFlip(),
    RandomVerticalFlip(),
    RandomRotate90(),
])


def get_augmentation(type):
    """utils/augmentation.py:87-93."""
    if type == 'none':
        return None
    if type == 'weak':
        return augmentation_weak
    if type == 'strong':
        raise NotImplementedError("augmentation 'strong' needs torchvision's RandomRotation / RandomAdjustSharpness "
                                  "(resampling ops, not installed): only 'none' and 'weak' are built")
