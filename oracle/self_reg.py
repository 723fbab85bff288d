"""TEST INFRASTRUCTURE ONLY -- fp32 CPU restatement of the reference's self-regularisation loss
(models/losses.py:6-217), pinned by tests/golden/selfreg_cases.npz (outputs of the reference itself).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

The reference picks the channel subsets of LSRC from `hash(str(idx) + "feature") % 10000` (losses.py:126), a value
that changes with the interpreter's string-hash seed; here the seeds are an argument (`seeds[idx]`), and the
fixture stores the values the generating interpreter produced.
"""
import torch
import torch.nn.functional as F


def default_seeds(n):
    """The reference's expression (losses.py:126), evaluated in this interpreter."""
    return [hash(str(idx) + "feature") % 10000 for idx in range(n)]


def channel_subset(seed, channels, keep):
    """losses.py:129-130 / :136-137: first `keep` entries of a seeded CPU permutation (without touching the global
    generator, which the reference reseeds as a side effect)."""
    g = torch.Generator().manual_seed(int(seed))
    return torch.randperm(channels, generator=g)[:keep]


def dice_loss(predictions, targets, smooth=1e-6):
    """losses.py:18-41: per-sample Dice with squared terms in the denominator, on sigmoid(predictions)."""
    p = torch.sigmoid(predictions)
    inter = torch.sum(p * targets, dim=[1, 2, 3])
    union = torch.sum(p * p, dim=[1, 2, 3]) + torch.sum(targets * targets, dim=[1, 2, 3])
    return 1 - torch.mean((2.0 * inter + smooth) / (union + smooth))


def cross_entropy(predictions, targets):
    """losses.py:58: nn.CrossEntropyLoss on [B,C,H,W] logits with same-shape (class-probability) targets."""
    return F.cross_entropy(predictions, targets)


def lsrc(features, seeds):
    """losses.py:81-155: mean over the first n-1 features of mse(feature, adaptive_avg_pool(last feature)),
    with the wider of the two cut down to the narrower one's channel count by a seeded permutation."""
    n = len(features)
    if n <= 1:
        return torch.tensor(0.0)
    final = features[-1].detach()
    total = torch.zeros(())
    for idx in range(n - 1):
        f = features[idx].detach()
        pooled = F.adaptive_avg_pool2d(final, f.shape[2:])
        if f.size(1) != pooled.size(1):
            keep = min(f.size(1), pooled.size(1))
            if f.size(1) > keep:
                f = torch.index_select(f, 1, channel_subset(seeds[idx], f.size(1), keep))
            if pooled.size(1) > keep:
                pooled = torch.index_select(pooled, 1, channel_subset(seeds[idx] + 1, pooled.size(1), keep))
        total = total + F.mse_loss(f, pooled)
    return total / max(1, n - 1)


def lifd(features, orders=None):
    """losses.py:157-217: per feature, order the channels by mean of the spatially L2-normalised maps
    (descending), and take the mse between the first and the second half of the ordered channels.
    `orders[idx]`, when given, replaces the argsort (to check a result whose ordering resolved channels of
    equal-to-rounding importance differently)."""
    n = len(features)
    if n <= 1:
        return torch.tensor(0.0)
    total = torch.zeros(())
    for idx in range(n - 1):
        f = features[idx].detach()
        norm = torch.norm(f, p=2, dim=(2, 3), keepdim=True)
        importance = (f / (norm + 1e-8)).mean([0, 2, 3])
        order = torch.argsort(importance, descending=True)
        if orders is not None:
            order = torch.as_tensor(orders[idx], dtype=torch.long)
        fs = torch.index_select(f, 1, order)
        mid = fs.shape[1] // 2
        if mid > 0:
            total = total + F.mse_loss(fs[:, :mid], fs[:, mid:2 * mid])
    return total / max(1, n - 1)


def channel_order(feature):
    """The ordering LIFD uses for one feature, ties broken towards the lower channel index."""
    norm = torch.norm(feature, p=2, dim=(2, 3), keepdim=True)
    importance = (feature / (norm + 1e-8)).mean([0, 2, 3])
    return torch.argsort(importance, descending=True, stable=True), importance


def self_reg_loss(predictions, targets, features, seeds):
    """losses.py:43-79: 0.4*CE + 0.6*Dice + 0.015*(LSRC + LIFD); the two auxiliaries enter as constants
    (computed under no_grad and read back as Python floats at :68-69)."""
    ce = cross_entropy(predictions, targets)
    dl = dice_loss(predictions, targets)
    with torch.no_grad():
        a = lsrc(features, seeds).item()
        b = lifd(features).item()
    return 0.4 * ce + 0.6 * dl + a * 0.015 + b * 0.015
