"""Parameter schema of the inference model and a seeded synthetic state dict.

The schema reproduces, key for key and in order, the reference checkpoint layout
(`RDMNet.state_dict()`, experiments/model_infer.py:26-107; 497 entries) so that the reference's
`weights/rdmnet.pth.tar` (`state['model']`, geotransformer/engine/base_tester.py:97-107) loads
unchanged.  The trained blob is not shipped with the reference (.MISSING_LARGE_BLOBS), so tests and
the bench use `synthetic_state_dict`: values drawn from a seeded numpy PCG64 stream with the
reference's init scales (kaiming-uniform Linear / KPConv, non-trivial norm affines).  The same dict
is loaded into the reference model when golden vectors are generated.
"""
import math
import os
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def _linear(d, name, cin, cout):
    d[name + '.weight'] = (cout, cin)
    d[name + '.bias'] = (cout,)


def _norm(d, name, c):
    d[name + '.weight'] = (c,)
    d[name + '.bias'] = (c,)


def _kpconv(d, name, k, cin, cout):
    d[name + '.weights'] = (k, cin, cout)
    d[name + '.bias'] = (cout,)
    d[name + '.kernel_points'] = (k, 3)


def _unary(d, name, cin, cout):
    _linear(d, name + '.mlp', cin, cout)
    _norm(d, name + '.norm.norm', cout)


def _residual(d, name, k, cin, cout):
    mid = cout // 4
    if cin != mid:
        _unary(d, name + '.unary1', cin, mid)
    _kpconv(d, name + '.KPConv', k, mid, mid)
    _norm(d, name + '.norm_conv.norm', mid)
    _unary(d, name + '.unary2', mid, cout)
    if cin != cout:
        _unary(d, name + '.unary_shortcut', cin, cout)


def encoder_blocks(cfg):
    """(name, kind, cin, cout, level, strided) for the 14 KPConv blocks (experiments/backbone.py:7-70)."""
    c = cfg.backbone.init_dim
    blocks = [('encoder1_1', 'conv', cfg.backbone.input_dim, c, 0, False), ('encoder1_2', 'res', c, 2 * c, 0, False)]
    width = 2 * c
    for stage in range(2, 6):
        lvl = stage - 1
        blocks.append((f'encoder{stage}_1', 'res', width, width, lvl - 1, True))
        blocks.append((f'encoder{stage}_2', 'res', width, 2 * width, lvl, False))
        blocks.append((f'encoder{stage}_3', 'res', 2 * width, 2 * width, lvl, False))
        width *= 2
    return blocks


def _transformer(d, name, cin, cout, hidden, heads, num_layers):
    _linear(d, name + '.embedding.proj', 3, hidden // 2)
    _linear(d, name + '.in_proj', cin, hidden)
    for i in range(2 * num_layers):
        p = f'{name}.transformer.layers.{i}'
        for proj in ('proj_q', 'proj_k', 'proj_v'):
            _linear(d, f'{p}.attention.attention.{proj}', hidden, hidden)
        if i % 2 == 0:  # 'self' layers carry the (unused) rotary div_term buffer
            d[f'{p}.attention.attention.pos_encoder.div_term'] = (1, 1, 1, hidden // heads)
        _linear(d, f'{p}.attention.linear', hidden, hidden)
        _norm(d, f'{p}.attention.norm', hidden)
        _linear(d, f'{p}.output.expand', hidden, 2 * hidden)
        _linear(d, f'{p}.output.squeeze', 2 * hidden, hidden)
        _norm(d, f'{p}.output.norm', hidden)
    _linear(d, name + '.out_proj', hidden, cout)


def schema(cfg):
    """OrderedDict name -> shape, in the reference's state_dict order."""
    d = OrderedDict()
    k, gn = cfg.backbone.kernel_size, cfg.backbone.group_norm
    for name, kind, cin, cout, _, _ in encoder_blocks(cfg):
        if kind == 'conv':
            _kpconv(d, f'encoder.{name}.KPConv', k, cin, cout)
            _norm(d, f'encoder.{name}.norm.norm', cout)
        else:
            _residual(d, f'encoder.{name}', k, cin, cout)
    c = cfg.backbone.init_dim
    _unary(d, 'decoder.decoder4', 20 * c + 1, 16 * c)
    _unary(d, 'decoder.decoder3', 24 * c, 8 * c)
    _linear(d, 'decoder.decoder2.mlp', 12 * c, cfg.backbone.output_dim + 1)
    t = cfg.thdroformer
    _transformer(d, 'transformer', t.input_dim, t.output_dim, t.hidden_dim, t.num_heads, t.num_layers)
    if cfg.Vote.model_use_vote:
        pre = t.output_dim
        for i, width in enumerate(cfg.Vote.MLPS):
            _linear(d, f'vote.mlp_modules.{3 * i}', pre, width)
            _norm(d, f'vote.mlp_modules.{3 * i + 1}', width)
            pre = width
        _linear(d, 'vote.ctr_reg', pre, 3 + t.output_dim)
        _norm(d, 'vote.out_proj.0', t.output_dim)
        _linear(d, 'proj_n2n_score', t.output_dim, 1)
        _transformer(d, 'transformer2', t.input_dim2, t.output_dim, t.hidden_dim, t.num_heads, t.num_layers2)
    _linear(d, 'proj_n2p_score', t.output_dim, 1)
    d['optimal_transport.alpha'] = ()
    return d


def kernel_disposition():
    """The 15-point 'center' disposition KPConv ships (k_015_center_3D), unit radius, f64 [15,3]."""
    return np.load(os.path.join(_HERE, 'data', 'kpconv_k15_center.npy'))


def kpconv_radius(cfg, level):
    return cfg.backbone.init_radius * (2 ** level)


def kpconv_sigma(cfg, level):
    return cfg.backbone.init_sigma * (2 ** level)


def synthetic_state_dict(cfg, seed=0):
    """name -> float32 numpy array.  Deterministic for a given numpy version and seed."""
    rng = np.random.Generator(np.random.PCG64(seed))
    levels = {f'encoder.{n}.KPConv.kernel_points': lvl for n, _, _, _, lvl, _ in encoder_blocks(cfg)}
    out = OrderedDict()
    shapes = schema(cfg)
    for name, shape in shapes.items():
        if name.endswith('kernel_points'):
            # same recipe as the reference's loader (kpconv/kernel_points.py:426-455): noise, scale, z-rotation
            theta = rng.random() * 2 * np.pi
            c, s = np.cos(theta), np.sin(theta)
            rot = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32)
            pts = kernel_disposition().astype(np.float32) + rng.normal(scale=0.01, size=shape)
            v = np.matmul(kpconv_radius(cfg, levels[name]) * pts, rot)
        elif name.endswith('div_term'):
            dm = shape[-1]
            v = np.repeat(np.exp(np.arange(0, dm, 2, dtype=np.float32) * (-math.log(10000.0) / dm)), 2).reshape(shape)
        elif name.endswith('KPConv.weights'):
            bound = 1.0 / math.sqrt(shape[1] * shape[2])
            v = rng.uniform(-bound, bound, size=shape)
        elif name == 'optimal_transport.alpha':
            v = np.asarray(1.0)
        elif name.endswith('.weight') and len(shape) == 2:
            bound = 1.0 / math.sqrt(shape[1])
            v = rng.uniform(-bound, bound, size=shape)
        elif name.endswith('.weight'):  # norm scale
            v = rng.uniform(0.8, 1.2, size=shape)
        elif name.endswith('.bias'):
            prev = name[:-5] + ('.weights' if name.endswith('KPConv.bias') else '.weight')
            wshape = shapes[prev]
            if len(wshape) == 1:  # norm shift
                v = rng.uniform(-0.1, 0.1, size=shape)
            else:
                fan_in = wshape[1] * wshape[2] if len(wshape) == 3 else wshape[1]
                bound = 1.0 / math.sqrt(fan_in)
                v = rng.uniform(-bound, bound, size=shape)
        else:
            raise KeyError(name)
        out[name] = np.ascontiguousarray(v, dtype=np.float32).reshape(shape)
    return out
