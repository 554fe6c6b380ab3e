"""The Deschaintre et al. SVBRDF network, as the reference wires it
(development/multiImage_pytorch/models.py), re-stated table-driven on stock torch.nn modules.

Architecture (models.py:208-320): an 8-level U-Net of stride-2 4x4 convolutions and
nearest-upsample + two padded 4x4 convolutions, LeakyReLU(0.2) pre-activations, affine
InstanceNorm, skip connections, dropout on the three innermost decoder levels -- plus a "global
track": a vector that is updated after every level from [track, channel means of the level's
conv output] by Linear+SELU and injected back into the next level through a bias-free Linear
added to every pixel.  ``SingleViewModel`` (models.py:322-346) maps one photo to 9 channels;
``MultiViewModel`` (models.py:348-411, never wired into the reference's main.py) runs the
generator on N photos, max-pools feature maps and tracks over the photos and refines with three
3x3 convolutions.  Both end in tanh followed by the head decode; with ``decode=False`` they
return the 9 encoded channels for ``losses.FusedHeadLoss`` (the decode then runs in the kernel).

The convolutions stay on stock PyTorch-ROCm (MIOpen/hipBLASLt) by design (north star).  State-dict
keys differ from the reference's; ``convert_reference_state_dict`` / ``convert_to_reference_state_dict`` translate
a ``checkpoint.tar``'s ``model_state_dict`` (persistence.py:52-69) in either direction.
"""
import math

import torch
import torch.nn as nn

from .. import losses

NGF = 64
ENCODER_WIDTHS = (1, 2, 4, 8, 8, 8, 8, 8)           # x NGF, levels 1..8                 models.py:240-247
DECODER_WIDTHS = (8, 8, 8, 8, 4, 2, 1)              # x NGF, levels 8..2 (level 1 -> out) models.py:249-256
DROPOUT_LEVELS = (8, 7, 6)                          #                                    models.py:249-251


def _init_conv(m, scale=0.02):
    nn.init.normal_(m.weight, 0.0, scale)            # models.py:24-26
    return m


def _init_linear(m, scale):
    nn.init.normal_(m.weight, 0.0, scale * math.sqrt(1.0 / m.in_features))   # models.py:20
    if m.bias is not None:
        nn.init.zeros_(m.bias)
    return m


class TrackedConv(nn.Module):
    """(pre-activation) -> conv -> channel means -> (instance norm) -> + Linear(track) per pixel
    (models.py:47-80 InterconnectedConvLayer + :31-45 MergeLayer)"""

    def __init__(self, conv, channels, norm, activation):
        super().__init__()
        self.activation = nn.LeakyReLU(0.2) if activation else None
        self.conv = conv
        self.norm = nn.InstanceNorm2d(channels, 1e-5, affine=True) if norm else None
        self.inject = _init_linear(nn.Linear(channels, channels, bias=False), 0.01)

    def forward(self, x, track):
        if self.activation is not None:
            x = self.activation(x)
        x = self.conv(x)
        mean = x.mean(dim=(2, 3))
        if self.norm is not None:
            x = self.norm(x)
        if track is not None:
            x = x + self.inject(track)[:, :, None, None]
        return x, mean


def _down(cin, cout):
    return _init_conv(nn.Conv2d(cin, cout, 4, stride=2, padding=1, bias=False))            # models.py:97


def _up(cin, cout):
    return nn.Sequential(nn.UpsamplingNearest2d(scale_factor=2.0), nn.ZeroPad2d((1, 2, 1, 2)),
                         _init_conv(nn.Conv2d(cin, cout, 4, bias=False)), nn.ZeroPad2d((1, 2, 1, 2)),
                         _init_conv(nn.Conv2d(cout, cout, 4, bias=False)))                  # models.py:120-126


class TrackUpdate(nn.Module):
    """track <- SELU(Linear([track, means]))   (models.py:185-206 GlobalTrackLayer)"""

    def __init__(self, cin, cout):
        super().__init__()
        self.fc = _init_linear(nn.Linear(cin, cout, bias=True), 1.0)
        self.selu = nn.SELU()

    def forward(self, means, track):
        return self.selu(self.fc(means if track is None else torch.cat((track, means), dim=1)))


class Generator(nn.Module):
    def __init__(self, out_channels, ngf=NGF, use_coords=True):
        super().__init__()
        self.use_coords = use_coords
        cin = 3 + (2 if use_coords else 0)
        enc_w = [ngf * w for w in ENCODER_WIDTHS]
        dec_w = [ngf * w for w in DECODER_WIDTHS] + [out_channels]
        self.enc = nn.ModuleList()
        prev = cin
        for level, w in enumerate(enc_w, start=1):
            self.enc.append(TrackedConv(_down(prev, w), w, norm=1 < level < 8, activation=level > 1))
            prev = w
        self.dec = nn.ModuleList()
        self.drop = nn.ModuleList()
        for k, w in enumerate(dec_w):                                  # decoder level 8-k
            level = 8 - k
            din = enc_w[-1] if level == 8 else 2 * dec_w[k - 1]        # skip connection doubles the input
            self.dec.append(TrackedConv(_up(din, w), w, norm=level > 1, activation=True))
            self.drop.append(nn.Dropout(0.5) if level in DROPOUT_LEVELS else nn.Identity())
        # global track: encoder side feeds level l+1, decoder side feeds the next (outer) level
        self.track_enc = nn.ModuleList([TrackUpdate(cin, enc_w[1])] +
                                       [TrackUpdate(2 * enc_w[l], enc_w[l + 1]) for l in range(1, 7)] +
                                       [TrackUpdate(2 * enc_w[7], dec_w[0])])
        self.track_dec = nn.ModuleList([TrackUpdate(2 * dec_w[k], dec_w[k + 1]) for k in range(7)] +
                                       [TrackUpdate(2 * dec_w[7], out_channels)])

    @staticmethod
    def _coords(x):
        """two extra input channels holding the pixel coordinates (models.py:161-183)"""
        B, _, H, W = x.shape
        xs = torch.linspace(-1, 1, W, device=x.device, dtype=x.dtype).view(1, 1, 1, W).expand(B, 1, H, W)
        ys = -torch.linspace(-1, 1, W, device=x.device, dtype=x.dtype).view(1, 1, W, 1).expand(B, 1, H, W)
        return torch.cat((x, xs, ys), dim=1)

    def forward(self, x):
        """models.py:276-318: enc1 ignores the track; the first track update sees the INPUT's channel
        means, every later one the means of the level's convolution output."""
        if self.use_coords:
            x = self._coords(x)
        input_means = x.mean(dim=(2, 3))
        down = [None] * 8
        down[0], _ = self.enc[0](x, None)
        track = self.track_enc[0](input_means, None)
        for l in range(1, 8):                                               # enc2..enc8
            down[l], m = self.enc[l](down[l - 1], track)
            track = self.track_enc[l](m, track)
        up = down[7]
        for k in range(8):                                                  # dec8..dec1
            level = 8 - k
            inp = up if level == 8 else torch.cat((up, down[level - 1]), dim=1)
            up, m = self.dec[k](inp, track)
            up = self.drop[k](up)
            track = self.track_dec[k](m, track)
        return up, track


class SingleViewModel(nn.Module):
    """models.py:322-346.  forward(input [B,3,H,W] or [B,N,3,H,W] -> first photo) -> [B,12,H,W] maps,
    or the [B,9,H,W] tanh output with decode=False (for losses.FusedHeadLoss)."""

    def __init__(self, use_coords=True, decode=True):
        super().__init__()
        self.generator = Generator(9, use_coords=use_coords)
        self.activation = nn.Tanh()
        self.decode = decode
        # Present in the reference but never reached by a gradient (enc1 gets no track; the last track
        # update's output is dropped by the single-view head): frozen so that DDP's reducer does not wait
        # for them.  Adam never moved them in the reference either.
        for p in list(self.generator.enc[0].inject.parameters()) + list(self.generator.track_dec[7].parameters()):
            p.requires_grad_(False)

    def forward(self, input):
        if input.dim() == 5:
            input = input[:, 0]
        encoded = self.activation(self.generator(input)[0])
        return losses.decode_head(encoded) if self.decode else encoded


class _Feature(TrackedConv):
    """3x3 feature convolution on the pooled maps (models.py:141-159 ConvFeatureLayer)"""

    def __init__(self, cin, cout, norm, activation):
        super().__init__(_init_conv(nn.Conv2d(cin, cout, 3, stride=1, padding=1, bias=False)), cout, norm, activation)


class MultiViewModel(nn.Module):
    """models.py:348-411: shared generator over the N photos, max-pool over photos, three refinement
    convolutions with their own global track."""

    def __init__(self, use_coords=True, decode=True):
        super().__init__()
        self.generator = Generator(64, use_coords=use_coords)
        widths = (64, 32, 9)
        self.pool_inject = _init_linear(nn.Linear(64, 64, bias=False), 0.01)        # models.py:365 MergeLayer
        self.tracks = nn.ModuleList([TrackUpdate(2 * 64, widths[0]), TrackUpdate(2 * widths[0], widths[1]),
                                     TrackUpdate(2 * widths[1], widths[2])])
        self.convs = nn.ModuleList([_Feature(64, widths[0], True, False), _Feature(widths[0], widths[1], True, True),
                                    _Feature(widths[1], widths[2], False, True)])
        self.activation = nn.Tanh()
        self.decode = decode
        for p in self.generator.enc[0].inject.parameters():      # enc1 gets no track (see SingleViewModel)
            p.requires_grad_(False)

    def forward(self, input):
        B, N = input.shape[:2]
        maps, tracks = self.generator(input.reshape((B * N,) + tuple(input.shape[2:])))   # one batched pass
        maps = maps.view((B, N) + tuple(maps.shape[1:])).max(dim=1)[0]
        track = tracks.view(B, N, -1).max(dim=1)[0]
        x = maps + self.pool_inject(track)[:, :, None, None]
        means = maps.mean(dim=(2, 3))
        for upd, conv in zip(self.tracks, self.convs):
            track = upd(means, track)
            x, means = conv(x, track)
        encoded = self.activation(x)
        return losses.decode_head(encoded) if self.decode else encoded


def convert_reference_state_dict(ref_state):
    """state dict of the reference's SingleViewModel / MultiViewModel (a ``checkpoint.tar``'s
    ``model_state_dict``, persistence.py:52-69) -> state dict for the classes above."""
    import re
    out = {}
    for key, value in ref_state.items():
        k = key
        m = re.match(r"generator\.enc(\d)\.conv\.(.*)", k)
        if m:
            lvl, rest = int(m.group(1)) - 1, m.group(2)
            rest = rest.replace("merge.fully_connected.", "inject.")
            out["generator.enc.%d.%s" % (lvl, rest)] = value
            continue
        m = re.match(r"generator\.dec(\d)\.deconv\.(.*)", k)
        if m:
            idx, rest = 8 - int(m.group(1)), m.group(2)
            rest = rest.replace("merge.fully_connected.", "inject.")
            out["generator.dec.%d.%s" % (idx, rest)] = value
            continue
        m = re.match(r"generator\.gte(\d)\.fully_connected\.(.*)", k)
        if m:
            out["generator.track_enc.%d.fc.%s" % (int(m.group(1)) - 1, m.group(2))] = value
            continue
        m = re.match(r"generator\.gtd(\d)\.fully_connected\.(.*)", k)
        if m:
            out["generator.track_dec.%d.fc.%s" % (8 - int(m.group(1)), m.group(2))] = value
            continue
        m = re.match(r"gt(\d)\.fully_connected\.(.*)", k)                 # multi-view refinement tracks
        if m:
            out["tracks.%d.fc.%s" % (int(m.group(1)) - 1, m.group(2))] = value
            continue
        m = re.match(r"conv(\d)\.conv\.(.*)", k)                          # multi-view refinement convolutions
        if m:
            out["convs.%d.%s" % (int(m.group(1)) - 1, m.group(2).replace("merge.fully_connected.", "inject."))] = value
            continue
        m = re.match(r"merge\.fully_connected\.(.*)", k)
        if m:
            out["pool_inject.%s" % m.group(1)] = value
            continue
        raise KeyError("unexpected key in reference state dict: %s" % key)
    return out


def convert_to_reference_state_dict(state):
    """inverse of ``convert_reference_state_dict``: a state dict of the classes above -> the reference's key names
    (so that a model trained here loads into the reference's SingleViewModel / MultiViewModel)."""
    import re
    out = {}
    for key, value in state.items():
        m = re.match(r"generator\.enc\.(\d)\.(.*)", key)
        if m:
            rest = m.group(2).replace("inject.", "merge.fully_connected.")
            out["generator.enc%d.conv.%s" % (int(m.group(1)) + 1, rest)] = value
            continue
        m = re.match(r"generator\.dec\.(\d)\.(.*)", key)
        if m:
            rest = m.group(2).replace("inject.", "merge.fully_connected.")
            out["generator.dec%d.deconv.%s" % (8 - int(m.group(1)), rest)] = value
            continue
        m = re.match(r"generator\.track_enc\.(\d)\.fc\.(.*)", key)
        if m:
            out["generator.gte%d.fully_connected.%s" % (int(m.group(1)) + 1, m.group(2))] = value
            continue
        m = re.match(r"generator\.track_dec\.(\d)\.fc\.(.*)", key)
        if m:
            out["generator.gtd%d.fully_connected.%s" % (8 - int(m.group(1)), m.group(2))] = value
            continue
        m = re.match(r"tracks\.(\d)\.fc\.(.*)", key)
        if m:
            out["gt%d.fully_connected.%s" % (int(m.group(1)) + 1, m.group(2))] = value
            continue
        m = re.match(r"convs\.(\d)\.(.*)", key)
        if m:
            out["conv%d.conv.%s" % (int(m.group(1)) + 1, m.group(2).replace("inject.", "merge.fully_connected."))] = value
            continue
        m = re.match(r"pool_inject\.(.*)", key)
        if m:
            out["merge.fully_connected.%s" % m.group(1)] = value
            continue
        raise KeyError("unexpected key: %s" % key)
    return out
