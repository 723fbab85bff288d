"""CPU oracle for the U-Net / LUNeXt / clDice hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32, CPU) restatement of the mathematics of the
reference hot path.  It exists so that the HIP path can be checked on a GPU box
where the reference itself is not present.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``unet_amd/`` does.

Parity pin: the reference holds no tests or golden vectors of its own.  The
oracle is pinned by fixtures under ``tests/golden/`` that were produced by
importing the reference modules in the build container
(``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` replays them.

What is restated (reference file:line):
  * ``RefUNet``            models/unet.py:6-176      (layer table, forward order)
  * ``RefLUNeXt``          models/lunext.py:6-259    (dw block, gate, forward order)
  * ``soft_erode`` ...     losses/cl_dice.py:12-46   (soft skeleton morphology)
  * ``soft_dice``          losses/cl_dice.py:73-94
  * ``RefSoftDiceClDice``  losses/cl_dice.py:97-125
  * ``probs_from_logits``  trainer/training_utils.py:146 (sigmoid + clamp glue)

The modules are built from layer tables rather than hand-written attribute
lists; the construction ORDER matches the reference's ``__init__`` so that the
same ``torch.manual_seed`` produces bit-identical default initialisation and
the same ``state_dict`` key order.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# U-Net (models/unet.py:19-78 declares the layers in exactly this order)
# --------------------------------------------------------------------------
# (name, kind, cin, cout); kind 'c3' = conv3x3 pad1 + BatchNorm, 'up' = convT 2x2 s2
def unet_layer_table(in_channels=3, out_channels=1):
    t = []
    widths = [64, 128, 256, 512, 1024]
    cin = in_channels
    for lvl, w in enumerate(widths, start=1):          # unet.py:19-45
        t.append((f"conv{lvl}_1", "c3", cin, w))
        t.append((f"conv{lvl}_2", "c3", w, w))
        cin = w
    for k, w in enumerate([512, 256, 128, 64], start=1):  # unet.py:51-76
        t.append((f"upconv{k}", "up", 2 * w, w))
        t.append((f"conv_up{k}_1", "c3", 2 * w, w))
        t.append((f"conv_up{k}_2", "c3", w, w))
    t.append(("final_conv", "c1", 64, out_channels))     # unet.py:78
    return t


def bn_name(conv_name):
    # conv3_2 -> bn3_2, conv_up2_1 -> bn_up2_1  (unet.py:20,53)
    return conv_name.replace("conv", "bn", 1)


# names whose post-ReLU output is appended to intermediate_features (unet.py:100-170)
UNET_FEATURE_LAYERS = (
    [f"conv{l}_{i}" for l in (1, 2, 3, 4) for i in (1, 2)]
    + [f"conv_up{k}_{i}" for k in (1, 2, 3) for i in (1, 2)]
    + ["conv_up4_2"]
)


class RefUNet(nn.Module):
    def __init__(self, in_channels=3, out_channels=1):
        super().__init__()
        self.table = unet_layer_table(in_channels, out_channels)
        for name, kind, ci, co in self.table:
            if kind == "c3":
                setattr(self, name, nn.Conv2d(ci, co, 3, padding=1))
                setattr(self, bn_name(name), nn.BatchNorm2d(co))
            elif kind == "up":
                setattr(self, name, nn.ConvTranspose2d(ci, co, 2, stride=2))
            else:
                setattr(self, name, nn.Conv2d(ci, co, 1))
        self.intermediate_features = []

    def _cbr(self, name, x):
        y = getattr(self, name)(x)
        y = getattr(self, bn_name(name))(y)
        y = F.relu(y)
        if name in UNET_FEATURE_LAYERS:
            self.intermediate_features.append(y)
        return y

    def forward(self, x):
        self.intermediate_features = []
        skips = []
        for lvl in (1, 2, 3, 4):                        # unet.py:99-124
            x = self._cbr(f"conv{lvl}_2", self._cbr(f"conv{lvl}_1", x))
            skips.append(x)
            x = F.max_pool2d(x, 2, 2)
        x = self._cbr("conv5_2", self._cbr("conv5_1", x))   # unet.py:127-128
        for k in (1, 2, 3, 4):                          # unet.py:132-169
            up = getattr(self, f"upconv{k}")(x)
            x = torch.cat([skips[4 - k], up], dim=1)   # skip first, then upsampled
            x = self._cbr(f"conv_up{k}_2", self._cbr(f"conv_up{k}_1", x))
        return self.final_conv(x)                       # logits, no sigmoid (unet.py:173-174)


# --------------------------------------------------------------------------
# LUNeXt (models/lunext.py)
# --------------------------------------------------------------------------
class RefDWBlock(nn.Module):
    """relu(BN(depthwise3x3(x)) + x)   lunext.py:6-30"""

    def __init__(self, c):
        super().__init__()
        self.depthwise = nn.Conv2d(c, c, 3, padding=1, groups=c)
        self.bn = nn.BatchNorm2d(c)

    def forward(self, x):
        return F.relu(self.bn(self.depthwise(x)) + x)


class RefGate(nn.Module):
    """sigmoid(W1 g)*tanh(W2 g) + sigmoid(W3 (g+x))*x   lunext.py:67-122.

    ``output_conv`` is declared (lunext.py:74) but never called (lunext.py:120),
    so it exists only as state and never receives a gradient.
    """

    def __init__(self, c):
        super().__init__()
        self.sigmoid_conv = nn.Conv2d(c, c, 1)
        self.tanh_conv = nn.Conv2d(c, c, 1)
        self.combine_conv = nn.Conv2d(c, c, 1)
        self.output_conv = nn.Conv2d(c, c, 1)

    def forward(self, g, x):
        if g.shape[2:] != x.shape[2:]:                 # lunext.py:91-92
            g = F.interpolate(g, size=x.shape[2:], mode="bilinear", align_corners=True)
        e = torch.sigmoid(self.sigmoid_conv(g)) * torch.tanh(self.tanh_conv(g))
        a = torch.sigmoid(self.combine_conv(g + x))
        return a * x + e


class RefLUNeXt(nn.Module):
    def __init__(self, in_channels=3, out_channels=1):
        super().__init__()
        chans = [in_channels, 64, 128, 256, 512]
        for lvl in range(1, 5):                         # lunext.py:137-154
            c = chans[lvl - 1]
            setattr(self, f"conv{lvl}_1", RefDWBlock(c))
            setattr(self, f"conv{lvl}_2", RefDWBlock(c))
            setattr(self, f"downconv{lvl}", nn.Conv2d(c, chans[lvl], 2, stride=2))
        self.conv5_1 = RefDWBlock(512)                  # lunext.py:157-158
        self.conv5_2 = RefDWBlock(512)
        for k in range(1, 5):                           # lunext.py:161-182
            ci, co = chans[5 - k], chans[4 - k]
            setattr(self, f"upconv{k}", nn.ConvTranspose2d(ci, co, 2, stride=2))
            setattr(self, f"att{k}", RefGate(co))
            setattr(self, f"conv_up{k}_1", RefDWBlock(co))
            setattr(self, f"conv_up{k}_2", RefDWBlock(co))
        self.final_conv = nn.Conv2d(in_channels, out_channels, 1)   # lunext.py:185

    def forward(self, x):
        skips = []
        for lvl in range(1, 5):                         # lunext.py:199-216
            x = getattr(self, f"conv{lvl}_2")(getattr(self, f"conv{lvl}_1")(x))
            skips.append(x)
            x = getattr(self, f"downconv{lvl}")(x)
        x = self.conv5_2(self.conv5_1(x))
        for k in range(1, 5):                           # lunext.py:223-254
            skip = skips[4 - k]
            up = getattr(self, f"upconv{k}")(x)
            if up.shape != skip.shape:
                up = F.interpolate(up, size=skip.shape[2:], mode="bilinear", align_corners=False)
            x = getattr(self, f"att{k}")(up, skip)
            x = getattr(self, f"conv_up{k}_2")(getattr(self, f"conv_up{k}_1")(x))
        return self.final_conv(x)


# --------------------------------------------------------------------------
# clDice (losses/cl_dice.py)
# --------------------------------------------------------------------------
SKEL_ITERS = 10   # cl_dice.py:103 instantiates SoftSkeletonize(num_iter=10); ctor arg iter_ is ignored


def soft_erode(img):
    """5-point-cross minimum; out-of-image neighbours are ignored (cl_dice.py:15-17)."""
    p1 = -F.max_pool2d(-img, (3, 1), (1, 1), (1, 0))
    p2 = -F.max_pool2d(-img, (1, 3), (1, 1), (0, 1))
    return torch.min(p1, p2)


def soft_dilate(img):
    """3x3 maximum; out-of-image neighbours are ignored (cl_dice.py:27)."""
    return F.max_pool2d(img, (3, 3), (1, 1), (1, 1))


def soft_open(img):
    return soft_dilate(soft_erode(img))                 # cl_dice.py:33


def soft_skel(img, num_iter=SKEL_ITERS):
    """cl_dice.py:35-46"""
    skel = F.relu(img - soft_open(img))
    for _ in range(num_iter):
        img = soft_erode(img)
        delta = F.relu(img - soft_open(img))
        skel = skel + F.relu(delta - skel * delta)
    return skel


def soft_dice(y_true, y_pred, smooth=1.0):
    """cl_dice.py:90-94 (sums are global over the batch)."""
    y_true = y_true.clamp(0.0, 1.0)
    y_pred = y_pred.clamp(0.0, 1.0)
    inter = (y_true * y_pred).sum()
    return 1.0 - (2.0 * inter + smooth) / (y_true.sum() + y_pred.sum() + smooth)


class RefSoftDiceClDice(nn.Module):
    """cl_dice.py:97-125; argument order is (y_true, y_pred), y_pred are probabilities."""

    def __init__(self, iter_=3, alpha=0.5, smooth=1.0, exclude_background=False):
        super().__init__()
        self.alpha, self.smooth = alpha, smooth

    def forward(self, y_true, y_pred):
        y_true = y_true.clamp(0.0, 1.0)
        y_pred = y_pred.clamp(0.0, 1.0)
        dice = soft_dice(y_true, y_pred)
        sp, st = soft_skel(y_pred), soft_skel(y_true)
        tprec = ((sp * y_true).sum() + self.smooth) / (sp.sum() + self.smooth)
        tsens = ((st * y_pred).sum() + self.smooth) / (st.sum() + self.smooth)
        cl = 1.0 - 2.0 * (tprec * tsens) / (tprec + tsens)
        return (1.0 - self.alpha) * dice + self.alpha * cl


def probs_from_logits(logits):
    """training_utils.py:146"""
    return torch.sigmoid(logits).clamp(1e-6, 1 - 1e-6)


# --------------------------------------------------------------------------
# synthetic inputs shared by tests / bench (data loaders give fp32 images in
# [0,1] and {0,1} masks: data_helpers/utils.py:58-62,106)
# --------------------------------------------------------------------------
def synth_batch(b, c, h, w, seed_img=1234, seed_mask=4321):
    gi = torch.Generator().manual_seed(seed_img)
    gm = torch.Generator().manual_seed(seed_mask)
    x = torch.rand(b, c, h, w, generator=gi)
    m = (torch.rand(b, 1, h, w, generator=gm) > 0.5).float()
    return x, m
