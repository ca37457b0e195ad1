"""`RDMNet` -- host-side mirror of the reference's inference model on the HIP kernels.

Same operator API as the reference (experiments/model_infer.py:26-107, 109-354, 357-359):
`create_model(cfg)`, `model.load_state_dict(state['model'])` with the reference's 497 checkpoint
keys, `model(data_dict) -> output_dict` with the reference's keys.  Every tensor op runs in
librdmnet_hip.so through rdmnet_amd.ops; torch provides device memory and the stream.
"""
import os
import threading
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


class _Node(torch.nn.Module):
    """A name in the reference's module tree: holds parameters / buffers and child nodes, nothing else."""


class RDMNet(torch.nn.Module):
    """A torch.nn.Module like the reference's (experiments/model_infer.py:26-107): `.cuda() / .to() / .eval()`,
    `.parameters()`, `.state_dict()` with the reference's 497 keys in the reference's order, strict
    `load_state_dict`, wrappable by DistributedDataParallel as geotransformer/engine/base_tester.py:113 does.  The
    parameter tree is built from rdmnet_amd.weights.schema (generic container nodes: the arithmetic is not torch's);
    `kernel_points` are buffers, as in modules/kpconv/kpconv.py:64-65."""

    def __init__(self, cfg, device=None):
        super().__init__()
        self.cfg = cfg
        self._schema = weights.schema(cfg)
        init = weights.synthetic_state_dict(cfg, seed=0)  # the reference initialises randomly too; a checkpoint follows
        for key, shape in self._schema.items():
            node, parts = self, key.split('.')
            for part in parts[:-1]:
                if part not in node._modules:
                    node.add_module(part, _Node())
                node = node._modules[part]
            value = torch.from_numpy(np.ascontiguousarray(init[key], dtype=np.float32)).reshape(tuple(shape))
            if parts[-1] == 'kernel_points':
                node.register_buffer(parts[-1], value)
            else:
                node.register_parameter(parts[-1], torch.nn.Parameter(value, requires_grad=True))  # as the reference's; DDP refuses a module without one
        self._w = None        # prepared device tensors of the per-op path
        self._np_state = None  # name -> numpy float32 (what the kernels' weight preparation reads)
        self.use_vote = bool(cfg.Vote.inference_use_vote and cfg.Vote.model_use_vote)
        self.attention_bf16 = bool(getattr(cfg.thdroformer, 'attention_bf16', False))
        self._tls = threading.local()  # .profile: list -> per-KPConv-layer HIP-event records (bench.py)
        # native engines by (device, stream), least recently used first; at most `max_engines` are kept (each owns an arena of
        # >= 3 GiB of HBM): a caller that keeps creating streams recycles engines instead of accumulating them
        self._engines, self._engines_state, self._engines_lock = OrderedDict(), None, threading.Lock()
        self.max_engines = 8
        self.pairs_in_flight = 1       # scheduling hint handed to the native engines (Engine.set_pairs_in_flight): set it before the first forward
        self.fast_path = True          # forward(data_dict) as one native call; False = the per-op mirror
        self.device = None
        if device is not None:
            self.to(device)

    # ------------------------------------------------------------------ nn.Module plumbing
    def _apply(self, fn, *args, **kwargs):  # .cuda() / .to() / .float(): parameters moved -> prepared copies are stale
        out = super()._apply(fn, *args, **kwargs)
        self._w = self._np_state = None
        p = next(self.parameters())
        self.device = p.device if p.is_cuda else None
        return out

    def load_state_dict(self, state, strict=True, **kwargs):
        """torch's loader (missing / unexpected keys and size mismatches raise RuntimeError when strict, like
        base_tester.py:97-107 expects); numpy arrays are accepted as values."""
        state = OrderedDict((k, torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v) for k, v in state.items())
        out = super().load_state_dict(state, strict=strict, **kwargs)
        self._w = self._np_state = None
        return out

    @property
    def _state(self):
        if self._np_state is None:
            sd = self.state_dict()
            # (np.array keeps 0-d entries 0-d -- optimal_transport.alpha -- which np.ascontiguousarray would not)
            self._np_state = OrderedDict((k, np.array(sd[k].detach().cpu().numpy(), dtype=np.float32, order='C'))
                                         for k in self._schema)
        return self._np_state

    def set_thread_profile(self, records):
        """Per-thread list that receives one HIP-event record per KPConv layer (None disables)."""
        self._tls.profile = records

    # ------------------------------------------------------------------ weight preparation
    def _prepare(self):
        if self._w is not None:
            return self._w
        if self.device is None:
            self.cuda()
        S, dev, W = self._state, self.device, {}

        def lin(name):
            W[name] = _dev_linear(S[name + '.weight'], S[name + '.bias'], dev)
            w = S[name + '.weight']
            if w.shape[0] in (128, 256) and pad4(w.shape[1]) % 16 == 0:  # checkpoint layout for the fused transformer kernels
                wt = torch.zeros((w.shape[0], pad4(w.shape[1])), dtype=torch.float32)
                wt[:, :w.shape[1]] = torch.from_numpy(np.ascontiguousarray(w))
                W[name + '.wt'] = wt.to(dev)

        def vec(name):
            W[name] = torch.from_numpy(S[name]).to(dev)

        for name in S:
            if name.endswith('KPConv.weights'):
                k, cin, cout = S[name].shape
                kdim = 16 if cin == 1 else k * cin
                b = torch.zeros((pad4(kdim), pad4(cout)), dtype=torch.float32)
                b[:k * cin, :cout] = torch.from_numpy(S[name]).reshape(k * cin, cout)
                W[name] = (b.to(dev), cin, cout)
                if ops.kpconv_fused_enabled() and ops.kpconv_fused_supported(cin, cout):  # the one-kernel form of the fine levels
                    W[name + '.packed'] = torch.from_numpy(ops.kpconv_pack_weights(S[name])).to(dev)
            elif name.endswith('.weight') and S[name].ndim == 2:
                lin(name[:-7])
            elif name.endswith('.bias') and (name[:-5] + '.weight') in S and S[name[:-5] + '.weight'].ndim == 2:
                continue
            else:
                vec(name)
        # fused attention projections: q|k|v for self layers, q and k|v for cross layers
        for name in [n for n in S if n.endswith('attention.attention.proj_q.weight')]:
            p = name[:-len('.attention.attention.proj_q.weight')]
            a = p + '.attention.attention'
            wq, wk, wv = S[a + '.proj_q.weight'], S[a + '.proj_k.weight'], S[a + '.proj_v.weight']
            bq, bk, bv = S[a + '.proj_q.bias'], S[a + '.proj_k.bias'], S[a + '.proj_v.bias']
            W[p + '.qkv'] = _dev_linear(np.concatenate([wq, wk, wv], 0), np.concatenate([bq, bk, bv]), dev)[:2]
            W[p + '.q'] = _dev_linear(wq, bq, dev)[:2]
            W[p + '.kv'] = _dev_linear(np.concatenate([wk, wv], 0), np.concatenate([bk, bv]), dev)[:2]
        self._w = W
        return W

    # ------------------------------------------------------------------ building blocks
    def _linear(self, name, x, act=ACT_NONE, out=None):
        b, bias, in_f, out_f = self._w[name]
        return ops.gemm(x, b, pad4(in_f), out_f, bias=bias, act=act, out=out)

    def _gn(self, name, x, act=ACT_NONE, residual=None, want_positive=False):
        return ops.group_norm(x, self._w[name + '.norm.weight'], self._w[name + '.norm.bias'],
                              self.cfg.backbone.group_norm, act=act, residual=residual, want_positive=want_positive)

    def _kpconv(self, name, norm, x, x_pos, q, s, idx, sigma, width=None, pool_src=None, order=None):
        """KPConv + its GroupNorm + LeakyReLU (gather kernel, then the weight GEMM whose epilogue emits the
        GroupNorm statistics).  `norm` = parameter prefix of the GroupNorm that follows the convolution."""
        b, cin, cout = self._w[name + '.weights']
        prof = getattr(self._tls, 'profile', None)
        if prof is not None:
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
        packed = self._w.get(name + '.weights.packed')
        if packed is not None:
            y = ops.kpconv_fused_group_norm(q, s, x, x_pos, idx, self._w[name + '.kernel_points'], sigma, packed,
                                            self._w[name + '.bias'], cout, self._w[norm + '.norm.weight'],
                                            self._w[norm + '.norm.bias'], self.cfg.backbone.group_norm, width=width, act=ACT_LEAKY,
                                            order=order)
            if prof is not None:
                e1.record()
        else:
            wf, nn = ops.kpconv_gather(q, s, x, x_pos, idx, self._w[name + '.kernel_points'], sigma, width)
            if prof is not None:
                e1.record()
            y = ops.linear_group_norm(wf, b, b.shape[0], cout, self._w[name + '.bias'], self._w[norm + '.norm.weight'],
                                      self._w[norm + '.norm.bias'], self.cfg.backbone.group_norm, rowdiv=nn, act=ACT_LEAKY)
        pooled = ops.gather_max(pool_src, idx, width) if pool_src is not None else None
        if prof is not None:
            e2.record()
            m, h = idx.shape
            # SURVEY.md §8d algorithmic bytes of one KPConv layer (padded slots counted, int64 indices, fp32):
            #   M*H*(8 + 12 + 4*C_in) + 4*M*C_out   (+ M*H*(8 + 4*C_block_in) for the strided shortcut pool)
            pooled_channels = pool_src.shape[1] if pool_src is not None else 0
            nbytes = m * h * (8 + 12 + 4 * cin) + 4 * m * cout + (m * h * (8 + 4 * pooled_channels) if pooled_channels else 0)
            prof.append({'name': name, 'm': m, 'h': h, 'cin': cin, 'cout': cout, 'bytes': nbytes,
                         'gather_bytes': m * h * (8 + 12 + 4 * cin), 'events': (e0, e1, e2), 'pooled': pooled_channels})
        return (y, pooled) if pool_src is not None else y

    def _unary(self, name, x, act=ACT_LEAKY, residual=None, want_positive=False):
        b, bias, in_f, out_f = self._w[name + '.mlp']
        return ops.linear_group_norm(x, b, pad4(in_f), out_f, bias, self._w[name + '.norm.norm.weight'],
                                     self._w[name + '.norm.norm.bias'], self.cfg.backbone.group_norm, act=act,
                                     residual=residual, want_positive=want_positive)

    def _conv_block(self, name, x, x_pos, q, s, idx, sigma, width, order=None):
        return self._kpconv(name + '.KPConv', name + '.norm', x, x_pos, q, s, idx, sigma, width, order=order)

    def _residual_block(self, name, x, x_pos, q, s, idx, sigma, strided, width, order=None):
        W = self._w
        if (name + '.unary1.mlp') in W:
            y, y_pos = self._unary(name + '.unary1', x, want_positive=True)
        else:
            y, y_pos = x, (x_pos if x_pos is not None else ops.row_positive(x))
        if strided:
            y, sc = self._kpconv(name + '.KPConv', name + '.norm_conv', y, y_pos, q, s, idx, sigma, width, pool_src=x, order=order)
        else:
            y, sc = self._kpconv(name + '.KPConv', name + '.norm_conv', y, y_pos, q, s, idx, sigma, width, order=order), x
        if (name + '.unary_shortcut.mlp') in W:
            sc = self._unary(name + '.unary_shortcut', sc, act=ACT_NONE)
        # leaky_relu(unary2(y) + shortcut): the add and the activation ride on unary2's GroupNorm apply
        return self._unary(name + '.unary2', y, act=ACT_LEAKY, residual=sc)

    def run_encoder(self, data, taps=None):
        """experiments/backbone.py:72-107."""
        cfg = self.cfg
        P, x = data['points'], data['features']
        widths = data.get('_widths', {})
        x_pos = ops.row_positive(x)
        feats = []
        # the query order of the one-kernel KPConv layers: the cell-sorted records of each level's search grid (radius r_0 2^l),
        # exactly what the native engine builds -- the records are a function of the points alone, so both paths group the
        # same rows into the same workgroups and produce the same GroupNorm partials
        orders = {}

        def order_of(level):
            if level not in orders:
                orders[level] = ops.radius_grid_records(P[level], data['lengths'][level], cfg.backbone.init_radius * 2 ** level)
            return orders[level]
        for name, kind, _, _, lvl, strided in weights.encoder_blocks(cfg):
            out_lvl = lvl + 1 if strided else lvl
            idx = data['subsampling'][lvl] if strided else data['neighbors'][lvl]
            width = widths.get(('subsampling' if strided else 'neighbors', lvl))
            sigma = weights.kpconv_sigma(cfg, lvl)
            if kind == 'conv':
                x = self._conv_block('encoder.' + name, x, x_pos, P[out_lvl], P[lvl], idx, sigma, width, order=order_of(out_lvl))
            else:
                x = self._residual_block('encoder.' + name, x, x_pos, P[out_lvl], P[lvl], idx, sigma, strided, width,
                                         order=order_of(out_lvl) if out_lvl <= 2 else None)
            x_pos = None
            if taps is not None:
                taps['encoder.' + name] = x
            if name.endswith('_3') or name == 'encoder1_2':
                feats.append(x)
        return feats

    def run_decoder(self, feats, data):
        """experiments/backbone.py:118-151."""
        up, W, groups = data['upsampling'], self._w, self.cfg.backbone.group_norm

        def stage(name, coarse, idx, skip, norm=True):
            b, bias, in_f, out_f = W[name + '.mlp']
            g, be = (W[name + '.norm.norm.weight'], W[name + '.norm.norm.bias']) if norm else (None, None)
            return ops.decoder_stage(coarse, idx, skip, b, out_f, bias, g, be, groups, act=ACT_LEAKY)

        l4 = stage('decoder.decoder4', feats[4], up[3], feats[3])
        l3 = stage('decoder.decoder3', l4, up[2], feats[2])
        return stage('decoder.decoder2', l3, up[1], feats[1], norm=False)

    # ------------------------------------------------------------------ 3DRoFormer
    def _attention_tail(self, p, hid, x, out):
        """Output projection, residual LayerNorm, FFN, residual LayerNorm (thdroformer.py:142-173,
        vanilla_transformer.py:69-103, output_layer.py:6-21); `out` = rows of the stacked state."""
        W = self._w
        lo, l1, l2 = p + '.attention.linear', p + '.output.expand', p + '.output.squeeze'
        wts = [W.get(k + '.wt') for k in (lo, l1, l2)]
        if all(w is not None for w in wts) and [tuple(w.shape) for w in wts] == [(128, 128), (256, 128), (128, 256)]:
            if (p + '.tail_packed') not in W:  # the three matrices in the kernel's operand order, once (as the native engine keeps them)
                W[p + '.tail_packed'] = ops.attention_tail_pack_weights(*wts)
            return ops.attention_tail_packed(hid, x, W[p + '.tail_packed'], W[lo][1], W[p + '.attention.norm.weight'],
                                             W[p + '.attention.norm.bias'], W[l1][1], W[l2][1], W[p + '.output.norm.weight'],
                                             W[p + '.output.norm.bias'], out=out)
        y = self._linear_ln(p + '.attention.linear', p + '.attention.norm', hid, x)
        z = self._linear(p + '.output.expand', y, act=ACT_RELU)
        return self._linear_ln(p + '.output.squeeze', p + '.output.norm', z, y, out=out)

    def _linear_ln(self, lin, norm, x, residual, out=None):
        """LayerNorm(Linear(x) + residual): one fused launch at the transformer width, two launches otherwise."""
        W = self._w
        b, bias, in_f, out_f = W[lin]
        if lin + '.wt' in W and out_f == 128:
            return ops.linear_layer_norm(x, W[lin + '.wt'], pad4(in_f), out_f, bias, W[norm + '.weight'], W[norm + '.bias'],
                                         residual=residual, out=out)
        h = self._linear(lin, x)
        return ops.layer_norm(h, W[norm + '.weight'], W[norm + '.bias'], residual=residual, out=out)

    def _thdroformer(self, name, pts4, x, n0, num_layers, out):
        """rdmnet/thdroformer/thdroformer.py:266-347 on the STACKED [ref; src] rows (same op sequence as
        the native engine): shared-weight ops run once on all rows, attention per cloud; cross layers are
        sequential -- src attends to the UPDATED ref features (:244-245)."""
        W, heads = self._w, self.cfg.thdroformer.num_heads
        N = x.shape[0]
        n1 = N - n0
        emb = self._linear(name + '.embedding.proj', pts4)
        f = self._linear(name + '.in_proj', x)
        d = f.shape[1]
        for i in range(2 * num_layers):
            p = f'{name}.transformer.layers.{i}'
            fnew, hid = ops.feat_empty(N, d, x.device), ops.feat_empty(N, d, x.device)
            if i % 2 == 0:
                qkv = ops.gemm(f, W[p + '.qkv'][0], d, 3 * d, bias=W[p + '.qkv'][1])
                q, k, v = qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:]
                ops.rope(q, k, emb)
                ops.attention_self_pair(q, k, v, n0, heads, out=hid, bf16=self.attention_bf16)  # both clouds, one launch
                self._attention_tail(p, hid, f, fnew)
            else:
                q = ops.gemm(f, W[p + '.q'][0], d, d, bias=W[p + '.q'][1])
                kv1 = ops.gemm(f[n0:], W[p + '.kv'][0], d, 2 * d, bias=W[p + '.kv'][1])
                ops.attention(q[:n0], kv1[:, :d], kv1[:, d:], heads, out=hid[:n0], bf16=self.attention_bf16)
                self._attention_tail(p, hid[:n0], f[:n0], fnew[:n0])
                kv0 = ops.gemm(fnew[:n0], W[p + '.kv'][0], d, 2 * d, bias=W[p + '.kv'][1])
                ops.attention(q[n0:], kv0[:, :d], kv0[:, d:], heads, out=hid[n0:], bf16=self.attention_bf16)
                self._attention_tail(p, hid[n0:], f[n0:], fnew[n0:])
            f = fnew
        self._linear(name + '.out_proj', f, out=out)

    @staticmethod
    def _pts4(pts):
        """[n,3] -> [n,4] zero padded (K of the positional Linear must be a multiple of 4)."""
        out = torch.zeros((pts.shape[0], 4), dtype=torch.float32, device=pts.device)
        out[:, :3] = pts
        return out

    # ------------------------------------------------------------------ forward
    def _engine(self):
        """The native engine of the CURRENT STREAM (an engine is not re-entrant and its work is ordered by the stream it
        runs on; bench.py drives one forward per host thread, each on its own stream), built from this module's state
        dict on first use and kept in an LRU cache of `max_engines` entries (>= 3 GiB of HBM each; release_engines() frees them;
        an evicted engine's arena is freed at once -- the shared weights are reference counted by the library and stay
        as long as any engine uses them)."""
        return self._engine_group(1)[0]

    def _engine_group(self, n):
        """The current stream's engines for a lock-step group of n data_dicts (`forward([d0, d1, ...])`): the stream's engine of
        `_engine()` first, further ones created on demand (they share its weights; one cache entry per stream whatever n)."""
        from . import engine as engine_mod
        if self.device is None:
            self.cuda()
        key = (self.device.index, torch.cuda.current_stream(self.device).cuda_stream)
        with self._engines_lock:
            if self._engines_state is not self._state:  # parameters changed (load_state_dict / .to()): rebuild lazily
                self._engines.clear()
                self._engines_state = self._state
            group = self._engines.get(key)
            if group is not None:
                self._engines.move_to_end(key)
                if len(group) >= n:
                    return group[:n]
            owner = next((g[0] for (d, _), g in self._engines.items() if d == self.device.index and g), None)
        group = list(group or [])
        with torch.cuda.device(self.device):
            while len(group) < n:
                eng = engine_mod.Engine(self.cfg, self._state, device=self.device, share_with=owner or (group[0] if group else None))
                eng.keep_taps(True)
                eng.set_pairs_in_flight(self.pairs_in_flight)
                group.append(eng)
        with self._engines_lock:
            self._engines[key] = group
            while len(self._engines) > max(int(self.max_engines), 1):
                self._engines.popitem(last=False)  # the least recently used stream's engines (their arenas are freed with them)
        return group[:n]

    def release_engines(self):
        """Drops every cached native engine (and its HBM arena); the next forward builds the calling stream's anew."""
        with self._engines_lock:
            self._engines.clear()

    def engine(self):
        """The calling thread's native engine (for rdmnet_amd.collate.registration_collate_fn_stack_mode(..., engine=...))."""
        return self._engine()

    def engine_group(self, n):
        """The calling thread's engines for a lock-step group of n pairs (rdmnet_amd.collate.registration_collate_lockstep(...,
        engines=...); `forward([d0, ...])` runs on the same ones)."""
        return self._engine_group(n)

    @torch.no_grad()
    def forward(self, data_dict, taps=None):
        """experiments/model_infer.py:109-354 (inference).  `data_dict` as produced by the collate
        (rdmnet_amd.collate or the reference's), tensors on the GPU; returns the reference's 31-key output_dict.

        Two host paths over the same kernels, bit-identical results (tests/test_engine_gpu.py): by default the whole
        forward is ONE native call (rdm_engine_forward issues the ~550 launches from C++); with a `taps` dictionary
        the per-op mirror below runs instead and records the stage tensors the parity tests compare.

        A LIST of data_dicts (round 6; up to 8) runs as one lock-step group on the current stream
        (rdm_engine_forward_lockstep: identical kernels of the pairs as one grouped launch) and returns the list of
        output_dicts -- each `torch.equal` to what `forward(data_dict)` returns for it alone; `model(data_dict)` stays batch 1
        like the reference's."""
        if isinstance(data_dict, (list, tuple)):
            if taps is not None or not self.fast_path:
                return [self._forward_per_op(d, taps if taps is not None else None) for d in data_dict]
            return self._forward_native_group(list(data_dict))
        if taps is None and self.fast_path:
            return self._forward_native(data_dict)
        return self._forward_per_op(data_dict, taps)

    def _forward_native(self, data_dict):
        eng = self._engine()
        return self._output_dict(eng, eng.forward(data_dict), data_dict)

    def _forward_native_group(self, data_dicts):
        from . import engine as engine_mod
        if len(data_dicts) == 0:
            return []
        if len(data_dicts) > 8:
            raise ValueError('a lock-step group carries at most 8 data_dicts')
        if len(data_dicts) == 1:
            return [self._forward_native(data_dicts[0])]
        group = self._engine_group(len(data_dicts))
        results = engine_mod.Engine.forward_lockstep(group, data_dicts)
        return [self._output_dict(e, r, d) for e, r, d in zip(group, results, data_dicts)]

    def _output_dict(self, eng, res, data_dict):
        """The reference's output_dict (model_infer.py:117-352) from the engine's result and stage tensors."""
        cfg, k_pts = self.cfg, self.cfg.model.num_points_in_patch
        n_c, n_f, n_0 = (int(res.level_ref_sizes[i]) for i in (4, 1, 0))
        m_r, B = int(res.n_ref_nodes), int(res.n_node_correspondences)
        pts_c, pts_f, pts = data_dict['points'][-1], data_dict['points'][1], data_dict['points'][0]
        out = dict(ori_ref_points_c=pts_c[:n_c], ori_src_points_c=pts_c[n_c:], ref_points_f=pts_f[:n_f],
                   src_points_f=pts_f[n_f:], ref_points=pts[:n_0], src_points=pts[n_0:])
        names = ['p2p_scores', 'decoder', 'nodes', 'feats_c', 'ref_node_corr_indices', 'src_node_corr_indices',
                 'ref_node_corr_knn_points', 'src_node_corr_knn_points', 'ref_node_corr_knn_masks',
                 'src_node_corr_knn_masks', 'matching_scores', 'ref_corr_points', 'src_corr_points', 'corr_scores',
                 'estimated_transform'] + (['vote_xyz', 'node_scores'] if self.use_vote else ['n2p_scores'])
        t = eng.tensors(names)  # one allocation, one batched copy out of the engine's arena
        p2p, dec, nodes, fn = t['p2p_scores'][:, 0], t['decoder'], t['nodes'], t['feats_c']
        out.update(ref_p2p_scores_c=p2p[:n_f], src_p2p_scores_c=p2p[n_f:])
        if self.use_vote:
            shifted, sc = t['vote_xyz'], t['node_scores']
            out.update(shifted_ref_points_c=shifted[:n_c], shifted_src_points_c=shifted[n_c:],
                       ref_n2p_scores_c=sc[:m_r, 0], src_n2p_scores_c=sc[m_r:, 0],
                       ref_n2n_scores_c=sc[:m_r, 1], src_n2n_scores_c=sc[m_r:, 1])
        else:
            n2p = t['n2p_scores'][:, 0]
            out.update(ref_n2p_scores_c=n2p[:n_c], src_n2p_scores_c=n2p[n_c:])
        out.update(ref_points_c=nodes[:m_r], src_points_c=nodes[m_r:], ref_feats_c=fn[:m_r], src_feats_c=fn[m_r:],
                   ref_feats_f=dec[:n_f, :cfg.backbone.output_dim], src_feats_f=dec[n_f:, :cfg.backbone.output_dim],
                   ref_node_corr_indices=t['ref_node_corr_indices'][:, 0], src_node_corr_indices=t['src_node_corr_indices'][:, 0],
                   ref_node_corr_knn_points=t['ref_node_corr_knn_points'].reshape(B, k_pts, 3),
                   src_node_corr_knn_points=t['src_node_corr_knn_points'].reshape(B, k_pts, 3),
                   ref_node_corr_knn_masks=t['ref_node_corr_knn_masks'].view(torch.bool),
                   src_node_corr_knn_masks=t['src_node_corr_knn_masks'].view(torch.bool),
                   matching_scores=t['matching_scores'].reshape(B, k_pts + 1, k_pts + 1),
                   ref_corr_points=t['ref_corr_points'], src_corr_points=t['src_corr_points'],
                   corr_scores=t['corr_scores'][:, 0], estimated_transform=t['estimated_transform'])
        return out

    @torch.no_grad()
    def _forward_per_op(self, data_dict, taps=None):
        W, cfg, dev = self._prepare(), self.cfg, self.device
        t = cfg.thdroformer
        taps = taps if taps is not None else {}
        out = {}
        L = data_dict['lengths']
        lens = torch.stack([L[-1], L[1], L[0]]).cpu()  # one sync: the ref/src split of three levels
        n_c, n_f, n_0 = int(lens[0, 0]), int(lens[1, 0]), int(lens[2, 0])
        pts_c, pts_f, pts = data_dict['points'][-1], data_dict['points'][1], data_dict['points'][0]
        N_c = pts_c.shape[0]
        out.update(ori_ref_points_c=pts_c[:n_c], ori_src_points_c=pts_c[n_c:], ref_points_f=pts_f[:n_f],
                   src_points_f=pts_f[n_f:], ref_points=pts[:n_0], src_points=pts[n_0:])

        feats = self.run_encoder(data_dict, taps)
        f_c = feats[-1]
        taps['feats_c_enc'] = f_c

        # transformer #1 -> stacked [ref; src] (256 features + n2p logit in column 256)
        buf_c = ops.feat_empty(N_c, t.output_dim + 1, dev)
        pts_c4 = self._pts4(pts_c)
        self._thdroformer('transformer', pts_c4, f_c, n_c, t.num_layers, buf_c[:, :t.output_dim])
        x_c = buf_c[:, :t.output_dim]
        taps['t1_ref'], taps['t1_src'] = x_c[:n_c], x_c[n_c:]
        self._linear('proj_n2p_score', x_c, out=buf_c[:, t.output_dim:])
        n2p = ops.sigmoid_column(buf_c[:, t.output_dim:])

        feats[-1] = buf_c
        dec = self.run_decoder(feats, data_dict)
        taps['decoder'] = dec
        feats_f = dec[:, :cfg.backbone.output_dim]
        p2p = ops.sigmoid_column(dec[:, cfg.backbone.output_dim:])
        out.update(ref_p2p_scores_c=p2p[:n_f], src_p2p_scores_c=p2p[n_f:])

        flags = torch.zeros(8, dtype=torch.int32, device=dev)  # [max_count, status, n_ref, n_src, p2n status...]
        if self.use_vote:
            # vote layer (rdmnet/vote/vote.py:83-117)
            h = x_c
            for i in range(len(cfg.Vote.MLPS)):
                h = self._linear(f'vote.mlp_modules.{3 * i}', h)
                h = ops.layer_norm(h, W[f'vote.mlp_modules.{3 * i + 1}.weight'], W[f'vote.mlp_modules.{3 * i + 1}.bias'],
                                   act=ACT_RELU)
            off = self._linear('vote.ctr_reg', h)
            shifted = ops.vote_shift(pts_c, off, cfg.Vote.MAX_TRANSLATE_RANGE)
            vfeats = ops.layer_norm(x_c, W['vote.out_proj.0.weight'], W['vote.out_proj.0.bias'], residual=off[:, 3:])
            taps['vote_xyz'], taps['vote_feats'] = shifted, vfeats
            out.update(shifted_ref_points_c=shifted[:n_c], shifted_src_points_c=shifted[n_c:])
            n2n = ops.sigmoid_column(self._linear('proj_n2n_score', vfeats))

            # NMS (vote.py:13-40): neighbours of the shifted nodes, greedy sweep, order-preserving compaction
            nms_idx = ops.radius_search_device(shifted, shifted, L[-1], L[-1], cfg.Vote.NMS_radius,
                                               cfg.neighbor_limits[-1], flags)
            keep = ops.nms(nms_idx, flags)
            taps['nms_mask'], taps['nms_idx'] = keep, nms_idx
            order = torch.empty((N_c,), dtype=torch.int32, device=dev)
            ops.compact_indices(keep, 0, n_c, order, flags[2:])
            ops.compact_indices(keep, n_c, N_c, order[n_c:], flags[3:])
            fl = flags.cpu()  # sync: kept-node counts size everything downstream
            if int(fl[1]) != 0:
                raise RuntimeError('radius search (NMS): internal error (status word set)')
            m_r, m_s = int(fl[2]), int(fl[3])
            sel = torch.cat([order[:m_r], order[n_c:n_c + m_s]]).to(torch.int64)
            nodes = ops.gather_rows(shifted, sel)
            ref_c, src_c = nodes[:m_r], nodes[m_r:]
            sel_feats = ops.gather_rows(vfeats, sel)
            scores3 = ops.gather_rows(torch.stack([n2p, n2n], 1), sel)
            out.update(ref_n2p_scores_c=scores3[:m_r, 0], src_n2p_scores_c=scores3[m_r:, 0],
                       ref_n2n_scores_c=scores3[:m_r, 1], src_n2n_scores_c=scores3[m_r:, 1],
                       ref_points_c=ref_c, src_points_c=src_c)

            # transformer #2 on the surviving nodes
            buf2 = ops.feat_empty(m_r + m_s, t.output_dim, dev)
            nodes4 = self._pts4(nodes)
            self._thdroformer('transformer2', nodes4, sel_feats, m_r, t.num_layers2, buf2)
            taps['t2_ref'], taps['t2_src'] = buf2[:m_r], buf2[m_r:]
        else:
            # infer.py:119-120 (Mulran) disables the vote layer; model_infer.py:179-246 then leaves
            # ref_points_c undefined.  Defined as: superpoints = un-shifted coarse points, features = first
            # transformer's output (SURVEY.md §7 hard part 7; same definition in oracle/forward.py).
            m_r, m_s = n_c, N_c - n_c
            ref_c, src_c, buf2 = pts_c[:n_c], pts_c[n_c:], x_c
            out.update(ref_n2p_scores_c=n2p[:n_c], src_n2p_scores_c=n2p[n_c:], ref_points_c=ref_c, src_points_c=src_c)
        fn = ops.l2_normalize(buf2)
        rfn, sfn = fn[:m_r], fn[m_r:]
        out.update(ref_feats_c=rfn, src_feats_c=sfn)

        # point-to-node grouping + coarse matching
        k_pts = cfg.model.num_points_in_patch
        r_nmask, r_knn, r_kmask = ops.point_to_node(pts_f[:n_f], ref_c, k_pts, flags[4:])
        s_nmask, s_knn, s_kmask = ops.point_to_node(pts_f[n_f:], src_c, k_pts, flags[4:])
        taps.update(ref_node_masks=r_nmask, src_node_masks=s_nmask, ref_knn=r_knn, src_knn=s_knn,
                    ref_knn_masks=r_kmask, src_knn_masks=s_kmask)
        out.update(ref_feats_f=feats_f[:n_f], src_feats_f=feats_f[n_f:])
        r_sel, s_sel, node_scores, n_sel = ops.coarse_matching_features(rfn, sfn, r_nmask, s_nmask,
                                                                        cfg.coarse_matching.num_correspondences,
                                                                        cfg.coarse_matching.dual_normalization)
        B = int(n_sel.item())  # sync: number of patch correspondences (256 unless the clouds are tiny)
        r_sel, s_sel, node_scores = r_sel[:B], s_sel[:B], node_scores[:B]
        taps['node_corr_scores'] = node_scores
        out.update(ref_node_corr_indices=r_sel, src_node_corr_indices=s_sel)

        # patches: indices, masks, points, features
        r_idx, s_idx = ops.gather_rows(r_knn, r_sel), ops.gather_rows(s_knn, s_sel)            # [B,k] i64
        r_pm, s_pm = ops.gather_rows(r_kmask, r_sel), ops.gather_rows(s_kmask, s_sel)          # [B,k] u8
        r_pts = ops.gather_rows(pts_f[:n_f], r_idx.view(-1)).view(B, k_pts, 3)
        s_pts = ops.gather_rows(pts_f[n_f:], s_idx.view(-1)).view(B, k_pts, 3)
        c_f = cfg.backbone.output_dim
        out.update(ref_node_corr_knn_points=r_pts, src_node_corr_knn_points=s_pts,
                   ref_node_corr_knn_masks=r_pm.bool(), src_node_corr_knn_masks=s_pm.bool())
        # index_select of the patch features + einsum / sqrt(c) (model_infer.py:291-311) as one call: the gathers happen in the
        # GEMM's operand loads
        patch_scores = ops.patch_scores(feats_f[:n_f], r_idx, feats_f[n_f:], s_idx, rowdiv=self._sqrt_c(c_f, k_pts))
        taps['patch_scores'] = patch_scores
        ms = ops.sinkhorn(patch_scores, r_pm, s_pm, W['optimal_transport.alpha'], cfg.model.num_sinkhorn_iterations)
        out['matching_scores'] = ms

        fm = cfg.fine_matching
        rc, sc, cs, T, counts = ops.lgr(ms, r_pts, s_pts, r_pm, s_pm, fm.acceptance_radius, fm.correspondence_threshold,
                                        fm.num_refinement_steps)
        # sync: number of correspondences + status words (grouping capacity; the collate's 13 radius searches, whose
        # status nobody has read yet unless the collate ran with exact_shapes=True)
        cflags = data_dict.get('_flags')
        parts = [counts, flags[4:5]] + ([cflags[:, 1].to(torch.int32)] if cflags is not None else [])
        cn = torch.cat(parts).cpu()
        if int(cn[3]) != 0:
            raise RuntimeError('point_to_node: a node owns more than 4096 points')
        if cn.numel() > 4 and int(cn[4:].max()) != 0:
            raise RuntimeError('radius search: a search of the collate reported an internal error (status word set)')
        C = int(cn[0])
        taps['lgr'] = {'n_hypotheses': int(cn[1]), 'best': int(cn[2])}
        out.update(ref_corr_points=rc[:C], src_corr_points=sc[:C], corr_scores=cs[:C], estimated_transform=T)
        return out

    def _sqrt_c(self, c, rows):
        key = ('sqrt_c', c, rows)
        if key not in self._w:
            self._w[key] = torch.full((rows,), float(c) ** 0.5, dtype=torch.float32, device=self.device)
        return self._w[key]


def create_model(cfg):
    """experiments/model_infer.py:357-359."""
    return RDMNet(cfg)
