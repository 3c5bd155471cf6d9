"""Plain-torch stand-ins for the task models of BASELINE configs 4 / 5 (test / bench infrastructure).

The reference takes its classifier from torchvision (`resnet_model`, model.py:15-23: resnet18 with a replaced
fc) and its segmenter from segmentation_models_pytorch (`smp.UnetPlusPlus`, train.py:218-225) with
`smp.losses.DiceLoss(mode='binary', from_logits=True)` (train.py:236).  Neither package is in the image and the
task models are OUT of the hot-path scope: these are the same architectures family
written out in plain torch so that the ISP can be measured inside a whole training step.  They run on
ATen / MIOpen; nothing here is part of raw2logit_amd."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)), inplace=True)
        y = self.bn2(self.conv2(y))
        return F.relu(y + (x if self.down is None else self.down(x)), inplace=True)


class ResNet18(nn.Module):
    """torchvision's resnet18 layout (He et al. 2016): 7x7/2 stem, 3x3/2 max-pool, 4 stages of 2 basic blocks
    (64, 128, 256, 512 channels), global average pool, fc -> n_classes (model.py:15-23 replaces fc)."""

    def __init__(self, n_classes=16, in_channels=3):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64),
                                  nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1))
        layers, cin = [], 64
        for cout, stride in ((64, 1), (128, 2), (256, 2), (512, 2)):
            layers += [BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1)]
            cin = cout
        self.layers = nn.Sequential(*layers)
        self.fc = nn.Linear(512, n_classes)

    def forward(self, x):
        x = self.layers(self.stem(x))
        return self.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


def _double_conv(cin, cout):
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                         nn.Conv2d(cout, cout, 3, 1, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class SmallUNet(nn.Module):
    """U-Net (Ronneberger et al. 2015), 4 resolutions, one output channel of logits (classes=1, activation=None
    like train.py:218-225)."""

    def __init__(self, in_channels=3, width=32):
        super().__init__()
        w = width
        self.e1, self.e2, self.e3, self.e4 = (_double_conv(in_channels, w), _double_conv(w, 2 * w),
                                              _double_conv(2 * w, 4 * w), _double_conv(4 * w, 8 * w))
        self.u3, self.u2, self.u1 = (nn.ConvTranspose2d(8 * w, 4 * w, 2, 2), nn.ConvTranspose2d(4 * w, 2 * w, 2, 2),
                                     nn.ConvTranspose2d(2 * w, w, 2, 2))
        self.d3, self.d2, self.d1 = _double_conv(8 * w, 4 * w), _double_conv(4 * w, 2 * w), _double_conv(2 * w, w)
        self.head = nn.Conv2d(w, 1, 1)

    def forward(self, x):
        e1 = self.e1(x)
        e2 = self.e2(F.max_pool2d(e1, 2))
        e3 = self.e3(F.max_pool2d(e2, 2))
        e4 = self.e4(F.max_pool2d(e3, 2))
        d3 = self.d3(torch.cat([self.u3(e4), e3], 1))
        d2 = self.d2(torch.cat([self.u2(d3), e2], 1))
        d1 = self.d1(torch.cat([self.u1(d2), e1], 1))
        return self.head(d1)


def dice_loss(logits, target, eps=1e-7):
    """smp.losses.DiceLoss(mode='binary', from_logits=True) as train.py:236 uses it: soft Dice of sigmoid(logits)
    over the whole batch, 1 - score, zero when the batch has no positive pixel."""
    p = F.logsigmoid(logits).exp().reshape(logits.shape[0], 1, -1)
    t = target.reshape(target.shape[0], 1, -1).to(p.dtype)
    inter = (p * t).sum((0, 2))
    card = (p + t).sum((0, 2))
    score = (2.0 * inter) / card.clamp_min(eps)
    return ((1.0 - score) * (t.sum((0, 2)) > 0).to(p.dtype)).mean()


class IspTask(nn.Module):
    """processor -> classifier, the composition of LitModel.forward (model.py:77-83; no augmentation here)"""

    def __init__(self, processor, classifier):
        super().__init__()
        self.processor = processor
        self.classifier = classifier

    def forward(self, x):
        return self.classifier(self.processor(x))
