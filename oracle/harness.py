"""Tiny LitModel-style harness (processor -> classifier -> loss -> Adam step) used to pin
"the classifier head sees identical logits" (golden set G4).

Mirrors the composition of the reference's LitModel.forward / update_step / configure_optimizers
(model.py:77-83, :85-125, :144-146): x = processor(x); x = classifier(x); loss = CE(logits, y);
Adam over processor + classifier parameters.  Test infrastructure only."""
import torch
import torch.nn as nn


class TinyClassifier(nn.Module):
    """small deterministic conv head standing in for resnet_model (model.py:15-23)."""

    def __init__(self, n_classes=4):
        super().__init__()
        self.conv = nn.Conv2d(3, 8, kernel_size=3, stride=2, padding=1)
        self.fc = nn.Linear(8, n_classes)

    def forward(self, x):
        x = torch.relu(self.conv(x))
        x = x.mean(dim=(2, 3))
        return self.fc(x)


def make_classifier(seed=1234, n_classes=4):
    g = torch.Generator().manual_seed(seed)
    m = TinyClassifier(n_classes)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.empty(p.shape).uniform_(-0.3, 0.3, generator=g))
    return m


def train_step(processor, classifier, raw, labels, lr=1e-3):
    """one LitModel.training_step + optimizer step; returns (logits, loss) before the step."""
    params = list(processor.parameters()) + list(classifier.parameters())
    opt = torch.optim.Adam(params, lr=lr)
    opt.zero_grad()
    logits = classifier(processor(raw))
    loss = nn.functional.cross_entropy(logits, labels)
    loss.backward()
    opt.step()
    return logits.detach(), loss.detach()
