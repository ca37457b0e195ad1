"""`RDMNet` -- host-side mirror of the reference's inference model on the HIP kernels.

Same operator API as the reference (experiments/model_infer.py:26-107, 109-354, 357-359):
`create_model(cfg)`, `model.load_state_dict(state['model'])` with the reference's 497 checkpoint
keys, `model(data_dict) -> output_dict` with the reference's keys.  Every tensor op runs in
librdmnet_hip.so through rdmnet_amd.ops; torch provides device memory and the stream.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import ops, weights
from .ops import ACT_LEAKY, ACT_NONE, ACT_RELU, pad4


def _dev_linear(w, b, device):
    """nn.Linear weight [out, in] -> B operand [pad4(in), pad4(out)] (zero padded), bias [out]."""
    out_f, in_f = w.shape
    bt = torch.zeros((pad4(in_f), pad4(out_f)), dtype=torch.float32)
    bt[:in_f, :out_f] = torch.from_numpy(np.ascontiguousarray(w)).t()
    return bt.to(device), torch.from_numpy(np.ascontiguousarray(b)).to(device), in_f, out_f


class RDMNet:
    def __init__(self, cfg, device=None):
        self.cfg = cfg
        self.device = torch.device(device) if device is not None else None
        self.training = False
        self._schema = weights.schema(cfg)
        self._state = None   # name -> numpy float32
        self._w = None       # prepared device tensors
        self.use_vote = bool(cfg.Vote.inference_use_vote and cfg.Vote.model_use_vote)

    # ------------------------------------------------------------------ nn.Module-like surface
    def cuda(self, device=None):
        self.device = torch.device('cuda', torch.cuda.current_device() if device is None else device)
        self._w = None
        return self

    def eval(self):
        self.training = False
        return self

    def state_dict(self):
        if self._state is None:
            self._state = weights.synthetic_state_dict(self.cfg, seed=0)
        return OrderedDict((k, torch.from_numpy(v.copy())) for k, v in self._state.items())

    def load_state_dict(self, state, strict=True):
        new = OrderedDict()
        missing = [k for k in self._schema if k not in state]
        unexpected = [k for k in state if k not in self._schema]
        if strict and (missing or unexpected):
            raise RuntimeError(f'Error(s) in loading state_dict: missing {missing[:5]} unexpected {unexpected[:5]}')
        for k, shape in self._schema.items():
            if k not in state:
                continue
            v = state[k]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if tuple(v.shape) != tuple(shape):
                raise RuntimeError(f'size mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}')
            new[k] = np.ascontiguousarray(v, dtype=np.float32)
        self._state = new
        self._w = None
        return self

    # ------------------------------------------------------------------ weight preparation
    def _prepare(self):
        if self._w is not None:
            return self._w
        if self.device is None:
            self.cuda()
        if self._state is None:
            self.state_dict()
        S, dev, W = self._state, self.device, {}

        def lin(name):
            W[name] = _dev_linear(S[name + '.weight'], S[name + '.bias'], dev)

        def vec(name):
            W[name] = torch.from_numpy(S[name]).to(dev)

        for name in S:
            if name.endswith('KPConv.weights'):
                k, cin, cout = S[name].shape
                kdim = 16 if cin == 1 else k * cin
                b = torch.zeros((pad4(kdim), pad4(cout)), dtype=torch.float32)
                b[:k * cin, :cout] = torch.from_numpy(S[name]).reshape(k * cin, cout)
                W[name] = (b.to(dev), cin, cout)
            elif name.endswith('.weight') and S[name].ndim == 2:
                lin(name[:-7])
            elif name.endswith('.bias') and (name[:-5] + '.weight') in S and S[name[:-5] + '.weight'].ndim == 2:
                continue
            else:
                vec(name)
        self._w = W
        return W

    # ------------------------------------------------------------------ building blocks
    def _linear(self, name, x, act=ACT_NONE, out=None):
        b, bias, in_f, out_f = self._w[name]
        return ops.gemm(x, b, pad4(in_f), out_f, bias=bias, act=act, out=out)

    def _gn(self, name, x, act=ACT_NONE, residual=None, want_positive=False):
        return ops.group_norm(x, self._w[name + '.norm.weight'], self._w[name + '.norm.bias'],
                              self.cfg.backbone.group_norm, act=act, residual=residual, want_positive=want_positive)

    def _kpconv(self, name, x, x_pos, q, s, idx, sigma, width=None):
        b, cin, cout = self._w[name + '.weights']
        wf, nn = ops.kpconv_gather(q, s, x, x_pos, idx, self._w[name + '.kernel_points'], sigma, width)
        return ops.gemm(wf, b, b.shape[0], cout, bias=self._w[name + '.bias'], rowdiv=nn)

    def _unary(self, name, x, act=ACT_LEAKY, residual=None, want_positive=False):
        return self._gn(name + '.norm', self._linear(name + '.mlp', x), act=act, residual=residual,
                        want_positive=want_positive)

    def _conv_block(self, name, x, x_pos, q, s, idx, sigma, width):
        y = self._kpconv(name + '.KPConv', x, x_pos, q, s, idx, sigma, width)
        return self._gn(name + '.norm', y, act=ACT_LEAKY)

    def _residual_block(self, name, x, x_pos, q, s, idx, sigma, strided, width):
        W = self._w
        if (name + '.unary1.mlp') in W:
            y, y_pos = self._unary(name + '.unary1', x, want_positive=True)
        else:
            y, y_pos = x, (x_pos if x_pos is not None else ops.row_positive(x))
        y = self._kpconv(name + '.KPConv', y, y_pos, q, s, idx, sigma, width)
        y = self._gn(name + '.norm_conv', y, act=ACT_LEAKY)
        sc = ops.gather_max(x, idx, width) if strided else x
        if (name + '.unary_shortcut.mlp') in W:
            sc = self._unary(name + '.unary_shortcut', sc, act=ACT_NONE)
        # leaky_relu(unary2(y) + shortcut): the add and the activation ride on unary2's GroupNorm apply
        return self._unary(name + '.unary2', y, act=ACT_LEAKY, residual=sc)

    def encoder(self, data, taps=None):
        """experiments/backbone.py:72-107."""
        cfg = self.cfg
        P, x = data['points'], data['features']
        widths = data.get('_widths', {})
        x_pos = ops.row_positive(x)
        feats = []
        for name, kind, _, _, lvl, strided in weights.encoder_blocks(cfg):
            out_lvl = lvl + 1 if strided else lvl
            idx = data['subsampling'][lvl] if strided else data['neighbors'][lvl]
            width = widths.get(('subsampling' if strided else 'neighbors', lvl))
            sigma = weights.kpconv_sigma(cfg, lvl)
            if kind == 'conv':
                x = self._conv_block('encoder.' + name, x, x_pos, P[out_lvl], P[lvl], idx, sigma, width)
            else:
                x = self._residual_block('encoder.' + name, x, x_pos, P[out_lvl], P[lvl], idx, sigma, strided, width)
            x_pos = None
            if taps is not None:
                taps['encoder.' + name] = x
            if name.endswith('_3') or name == 'encoder1_2':
                feats.append(x)
        return feats

    def decoder(self, feats, data):
        """experiments/backbone.py:118-151."""
        up = data['upsampling']
        l4 = self._unary('decoder.decoder4', ops.upsample_concat(feats[4], up[3], feats[3]))
        l3 = self._unary('decoder.decoder3', ops.upsample_concat(l4, up[2], feats[2]))
        return self._linear('decoder.decoder2.mlp', ops.upsample_concat(l3, up[1], feats[1]))

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, data_dict, taps=None):
        raise NotImplementedError('assembled in stages; see forward_backbone for the encoder/decoder slice')

    __call__ = forward


def create_model(cfg):
    """experiments/model_infer.py:357-359."""
    return RDMNet(cfg)
