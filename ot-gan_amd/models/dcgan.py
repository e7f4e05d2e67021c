"""DCGAN-style critic and generator of OT-GAN on the HIP layer kernels.

Plugin surface of the reference's `models/dcgan.py` (module-level `generator` and
`discriminator` templates with shared parameters, selected by `--model dcgan`):

    discriminator(x, init=False, nonlinearity='crelu', ema=None, **kw) -> [B, 32768] unit rows
        reference models/dcgan.py:7-22
    generator(batch_size, init=False, nonlinearity='crelu', ema=None, **kw) -> [B,32,32,3] in (-1,1)
        reference models/dcgan.py:28-52

Extra keyword arguments (not in the reference): `noise` supplies the U(-1,1) latent instead
of drawing it (parity tests), `device` places a freshly drawn latent.
"""
import torch

from .. import ops
from ..utils import nn

# (filters, stride, pre-activated) per critic convolution; all 5x5  (models/dcgan.py:11-14)
_CRITIC = ((128, 1, False), (256, 2, True), (512, 2, True), (1024, 2, True))
# output channels of the three upsampling generator convolutions, before the GLU (:39,43,47)
_GEN = (2 * 512, 2 * 256, 2 * 128)


def disc_spec(x, init=False, nonlinearity='crelu', ema=None, **kwargs):
    with nn.arg_scope([nn.conv2d, nn.dense], counters={}, init=init, weight_norm=True, ema=ema):
        for filters, s, act in _CRITIC:
            x = nn.conv2d(x, filters, filter_size=[5, 5], stride=[s, s],
                          pre_activation=nonlinearity if act else None)
        # CReLU, flatten (h, w, c), L2-normalise: the critic's feature vector
        return nn.feature_head(x)


discriminator = nn.make_template('discriminator', disc_spec)


def gen_spec(batch_size, init=False, nonlinearity='crelu', ema=None, noise=None, device=None,
             image_size=32, **kwargs):
    """`image_size` (added; the reference hard-codes 32x32, train.py:52,67): the stem starts at
    image_size/8 so that the three upsampling stages end at image_size (64 -> BASELINE config 5)."""
    base = image_size // 8
    if noise is None:
        # models/dcgan.py:30 -- fresh uniform(-1, 1) latent on every call
        # U(-1, 1) (models/dcgan.py:30) in one launch: the same Philox draws and the same fp32 arithmetic as rand() * 2 - 1
        noise = torch.empty((batch_size, 100), device=device or 'cuda').uniform_(-1.0, 1.0)
    with nn.arg_scope([nn.conv2d, nn.dense], counters={}, init=init, weight_norm=True, ema=ema):
        x = nn.glu(nn.dense(noise, 2 * base * base * 1024, pre_activation=None))   # split along axis 1
        x = ops.carry_amax(x.view(noise.shape[0], base, base, 1024), x)   # (the GLU's amax record survives the reshape)
        for filters in _GEN:
            # nearest-neighbour x2 (fused into the conv's gather) -> 5x5 conv -> gated linear unit
            # (glu_hint: the layer's output kernel writes the gated product as well, nn.glu picks it up)
            x = nn.glu(nn.conv2d(x, filters, filter_size=[5, 5], pre_activation=None, upsample=True, glu_hint=True))
        return nn.tanh(nn.conv2d(x, 3, filter_size=[5, 5], pre_activation=None, init_scale=0.1))


generator = nn.make_template('generator', gen_spec)
