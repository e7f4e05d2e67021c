"""DenseNet critic and generator of OT-GAN on the HIP layer kernels.

Plugin surface of the reference's `models/densenet.py` (selected by `--model densenet`):

    discriminator(x, init=False, layers_per_block=16, filters_per_layer=16,
                  nonlinearity='crelu', ema=None, **kw) -> [B, 7296] unit rows
        reference models/densenet.py:7-45
    generator(batch_size, init=False, layers_per_block=16, filters_per_layer=16,
              nonlinearity='crelu', ema=None, **kw) -> [B,32,32,3] in (-1,1)
        reference models/densenet.py:51-88

Dense blocks grow in place in one NHWC buffer (`nn.dense_block`) instead of re-concatenating
the feature list before every layer.  Extra keyword arguments: `noise` (list of the four
U(-1,1) tensors the reference draws at :53-56) and `device`.
"""
import torch

from ..utils import nn


def disc_spec(x, init=False, layers_per_block=16, filters_per_layer=16, nonlinearity='crelu', ema=None,
              **kwargs):
    with nn.arg_scope([nn.conv2d, nn.dense, nn.dense_block], counters={}, init=init, weight_norm=True,
                      ema=ema):
        # (grow: the dense block that follows appends its L * F outputs behind this layer's, in the same buffer)
        room = layers_per_block * filters_per_layer
        x = nn.conv2d(x, 2 * filters_per_layer, pre_activation=None, grow=room)
        for _stage in range(3):
            feats = nn.dense_block(x, layers_per_block, filters_per_layer, pre_activation=nonlinearity)
            width = sum(int(t.shape[-1]) for t in feats)
            # transition: stride-2 conv over the whole concatenation, halving the channels (:18-21)
            x = nn.conv2d(feats, width // 2, pre_activation=nonlinearity, stride=[2, 2], grow=room if _stage < 2 else 0)
        return nn.feature_head(x)


discriminator = nn.make_template('discriminator', disc_spec)


def gen_spec(batch_size, init=False, layers_per_block=16, filters_per_layer=16, nonlinearity='crelu',
             ema=None, noise=None, device=None, **kwargs):
    F = filters_per_layer
    if noise is None:
        dev = device or 'cuda'
        # (one launch per draw: the same Philox draws and fp32 arithmetic as rand() * 2 - 1)
        noise = [torch.empty(shape, device=dev).uniform_(-1.0, 1.0)
                 for shape in ((batch_size, 100), (batch_size, 8, 8, F), (batch_size, 16, 16, F),
                               (batch_size, 32, 32, F))]
    B = noise[0].shape[0]
    with nn.arg_scope([nn.conv2d, nn.dense, nn.dense_block], counters={}, init=init, weight_norm=True,
                      ema=ema):
        x = nn.dense(noise[0], 8 * 8 * F, pre_activation=None).view(B, 8, 8, F)
        feats = nn.dense_block([x, noise[1]], layers_per_block, F, pre_activation=nonlinearity)
        for scale in (2, 3):
            # upsample: concatenate, nearest-neighbour x2 (folded into the conv), halve channels (:67-73)
            width = sum(int(t.shape[-1]) for t in feats)
            x = nn.conv2d(feats, width // 2, pre_activation=nonlinearity, upsample=True, grow=F + layers_per_block * F)
            feats = nn.dense_block([x, noise[scale]], layers_per_block, F, pre_activation=nonlinearity)
        return nn.tanh(nn.conv2d(feats, 3, pre_activation=nonlinearity, init_scale=0.1))


generator = nn.make_template('generator', gen_spec)
