"""ctypes front-end of the native engine (rdm_engine_*): one call per scan pair."""
import ctypes

import numpy as np
import torch

from . import _lib, weights


class EngineConfig(ctypes.Structure):
    _fields_ = [('num_stages', ctypes.c_int), ('kernel_size', ctypes.c_int), ('group_norm', ctypes.c_int),
                ('init_voxel_size', ctypes.c_float), ('init_radius', ctypes.c_float), ('init_sigma', ctypes.c_float),
                ('neighbor_limits', ctypes.c_int * 5), ('out_dim', ctypes.c_int), ('num_heads', ctypes.c_int),
                ('num_layers', ctypes.c_int), ('num_layers2', ctypes.c_int), ('vote_mlp_layers', ctypes.c_int),
                ('vote_limit', ctypes.c_float * 3), ('nms_radius', ctypes.c_float), ('points_in_patch', ctypes.c_int),
                ('num_correspondences', ctypes.c_int), ('dual_normalization', ctypes.c_int),
                ('sinkhorn_iterations', ctypes.c_int), ('acceptance_radius', ctypes.c_float),
                ('correspondence_threshold', ctypes.c_int), ('num_refinement_steps', ctypes.c_int),
                ('use_vote', ctypes.c_int), ('attention_bf16', ctypes.c_int),
                ('arena_bytes', ctypes.c_size_t)]


class EngineResult(ctypes.Structure):
    _fields_ = [('transform', ctypes.c_float * 16), ('n_correspondences', ctypes.c_int32),
                ('n_hypotheses', ctypes.c_int32), ('best_hypothesis', ctypes.c_int32),
                ('n_ref_nodes', ctypes.c_int64), ('n_src_nodes', ctypes.c_int64),
                ('n_node_correspondences', ctypes.c_int64), ('level_sizes', ctypes.c_int64 * 5), ('level_ref_sizes', ctypes.c_int64 * 5),
                ('ref_corr_points', ctypes.c_void_p), ('src_corr_points', ctypes.c_void_p),
                ('corr_scores', ctypes.c_void_p), ('transform_dev', ctypes.c_void_p), ('arena_used', ctypes.c_size_t),
                ('host_ref_corr_points', ctypes.c_void_p), ('host_src_corr_points', ctypes.c_void_p),
                ('host_corr_scores', ctypes.c_void_p), ('n_host_correspondences', ctypes.c_int32)]


class TensorView(ctypes.Structure):
    _fields_ = [('data', ctypes.c_void_p), ('rows', ctypes.c_int64), ('cols', ctypes.c_int64), ('ld', ctypes.c_int64),
                ('dtype', ctypes.c_int)]


class DataDict(ctypes.Structure):
    """rdm_data_dict (include/rdmnet_hip.h): the reference's data_dict as device pointers."""
    _fields_ = [('features', ctypes.c_void_p), ('features_ld', ctypes.c_int64),
                ('points', ctypes.c_void_p * 5), ('lengths', ctypes.c_void_p * 5),
                ('n_points', ctypes.c_int64 * 5), ('n_ref', ctypes.c_int64 * 5),
                ('neighbors', ctypes.c_void_p * 5), ('neighbors_width', ctypes.c_int64 * 5),
                ('neighbors_ld', ctypes.c_int64 * 5), ('neighbors_count', ctypes.c_void_p * 5),
                ('subsampling', ctypes.c_void_p * 4), ('subsampling_width', ctypes.c_int64 * 4),
                ('subsampling_ld', ctypes.c_int64 * 4), ('subsampling_count', ctypes.c_void_p * 4),
                ('upsampling', ctypes.c_void_p * 4), ('upsampling_width', ctypes.c_int64 * 4),
                ('upsampling_ld', ctypes.c_int64 * 4), ('upsampling_count', ctypes.c_void_p * 4),
                ('collate_status', ctypes.c_void_p), ('n_collate_status', ctypes.c_int64)]


class KpconvProfile(ctypes.Structure):
    _fields_ = [('m', ctypes.c_int64), ('h', ctypes.c_int64), ('c_in', ctypes.c_int64), ('c_out', ctypes.c_int64),
                ('pooled_channels', ctypes.c_int64), ('gather_ms', ctypes.c_float), ('total_ms', ctypes.c_float),
                ('fused', ctypes.c_int32), ('reserved', ctypes.c_int32)]


_DTYPES = {0: torch.float32, 1: torch.int64, 2: torch.uint8, 3: torch.int32}
_ESIZE = {0: 4, 1: 8, 2: 1, 3: 4}


def make_config(cfg, arena_bytes=0):
    c = EngineConfig()
    b, t, fm = cfg.backbone, cfg.thdroformer, cfg.fine_matching
    c.num_stages, c.kernel_size, c.group_norm = b.num_stages, b.kernel_size, b.group_norm
    c.init_voxel_size, c.init_radius, c.init_sigma = b.init_voxel_size, b.init_radius, b.init_sigma
    c.neighbor_limits = (ctypes.c_int * 5)(*[int(x) for x in cfg.neighbor_limits])
    c.out_dim, c.num_heads, c.num_layers, c.num_layers2 = t.output_dim, t.num_heads, t.num_layers, t.num_layers2
    c.vote_mlp_layers = len(cfg.Vote.MLPS)
    c.vote_limit = (ctypes.c_float * 3)(*[float(x) for x in cfg.Vote.MAX_TRANSLATE_RANGE])
    c.nms_radius = cfg.Vote.NMS_radius
    c.points_in_patch = cfg.model.num_points_in_patch
    c.num_correspondences = cfg.coarse_matching.num_correspondences
    c.dual_normalization = int(cfg.coarse_matching.dual_normalization)
    c.sinkhorn_iterations = cfg.model.num_sinkhorn_iterations
    c.acceptance_radius, c.correspondence_threshold = fm.acceptance_radius, fm.correspondence_threshold
    c.num_refinement_steps = fm.num_refinement_steps
    c.use_vote = int(bool(cfg.Vote.model_use_vote and cfg.Vote.inference_use_vote))
    c.attention_bf16 = int(bool(getattr(cfg.thdroformer, 'attention_bf16', False)))
    c.arena_bytes = arena_bytes
    return c


class Engine:
    """Owns a native engine bound to the current device.  Not thread-safe: one engine per in-flight pair."""

    def __init__(self, cfg, state, device=None, arena_bytes=0, share_with=None):
        """share_with: another Engine of the same device and configuration whose prepared device parameters this one uses
        (one copy of the weights for all engines in flight; `state` is ignored then).  The parameter set is reference
        counted in the library: no Python reference to `share_with` is kept, so dropping that engine frees its arena
        while this one keeps working."""
        if not torch.cuda.is_available():
            raise RuntimeError('rdmnet_amd.engine needs a GPU (no CPU fallback)')
        self.L = _lib.lib()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.cfg = cfg
        self._h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            c = make_config(cfg, arena_bytes)
            _lib.check(self.L.rdm_engine_create(ctypes.byref(c), ctypes.byref(self._h)), 'rdm_engine_create')
            if share_with is not None:
                _lib.check(self.L.rdm_engine_share_params(self._h, share_with._h), 'rdm_engine_share_params')
            for name, shape in ({} if share_with is not None else weights.schema(cfg)).items():
                v = state[name]
                v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
                if tuple(v.shape) != tuple(shape):
                    raise RuntimeError(f'size mismatch for {name}: {tuple(v.shape)} vs {tuple(shape)}')
                v = np.ascontiguousarray(v, dtype=np.float32)  # (0-d becomes 1-d here, hence the check above)
                shp = (ctypes.c_int64 * max(len(shape), 1))(*shape)
                _lib.check(self.L.rdm_engine_set_param(self._h, name.encode(), v.ctypes.data, ctypes.addressof(shp),
                                                       len(shape)), 'rdm_engine_set_param')
            if share_with is None:
                _lib.check(self.L.rdm_engine_finalize(self._h), 'rdm_engine_finalize')
        self.result = EngineResult()
        self._export_cache = {}
        self._ran, self._prepared = None, []

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            try:
                self.L.rdm_engine_destroy(h)
            except Exception:  # interpreter shutdown: the library or ctypes may already be gone
                pass
            h.value = None

    def set_wait(self, sleep_us=0):
        """0: spin in hipStreamSynchronize at the size read-backs; > 0: poll and sleep (frees the host core)."""
        _lib.check(self.L.rdm_engine_set_wait(self._h, int(sleep_us)), 'rdm_engine_set_wait')

    def set_pairs_in_flight(self, n):
        """Scheduling hint: how many pairs (engines / streams) share this GPU; from 3 the GEMM tiles keep two workgroups per CU."""
        _lib.check(self.L.rdm_engine_set_pairs_in_flight(self._h, int(n)), 'rdm_engine_set_pairs_in_flight')

    def set_overlap(self, mode=1):
        """Latency mode (rdm_engine_set_overlap): 0 = never run parts of a pair on the engine's side stream, 1 = when one pair is
        in flight (default), 2 = always.  Results are bit-identical in every mode; not used on the null stream."""
        _lib.check(self.L.rdm_engine_set_overlap(self._h, int(mode)), 'rdm_engine_set_overlap')

    def reserve(self, arena_bytes):
        """Re-allocates the activation arena at `arena_bytes`, growable (rdm_engine_reserve)."""
        self.clear_pending()
        _lib.check(self.L.rdm_engine_reserve(self._h, int(arena_bytes)), 'rdm_engine_reserve')

    def enable_profile(self, enable=True):
        _lib.check(self.L.rdm_engine_enable_profile(self._h, int(enable)), 'rdm_engine_enable_profile')

    def kpconv_profile(self):
        """Per-KPConv-layer records of the last run: dicts with sizes, SURVEY §8d algorithmic bytes and ms."""
        buf = (KpconvProfile * 16)()
        n = self.L.rdm_engine_get_profile(self._h, ctypes.addressof(buf), 16)
        out = []
        for p in buf[:max(n, 0)]:
            gather = p.m * p.h * (8 + 12 + 4 * p.c_in)
            total = gather + 4 * p.m * p.c_out + (p.m * p.h * (8 + 4 * p.pooled_channels) if p.pooled_channels else 0)
            out.append({'m': p.m, 'h': p.h, 'cin': p.c_in, 'cout': p.c_out, 'bytes': total, 'gather_bytes': gather,
                        'gather_ms': p.gather_ms, 'total_ms': p.total_ms, 'pooled': int(p.pooled_channels), 'fused': int(p.fused)})
        return out

    def keep_taps(self, enable=True):
        self.clear_pending()
        _lib.check(self.L.rdm_engine_keep_taps(self._h, int(enable)), 'rdm_engine_keep_taps')

    @staticmethod
    def _held(ref, src):
        """What a lock-step group / collated batch remembers of a pair's input tensors: the tensors themselves (kept alive, so
        their addresses cannot be handed to another allocation meanwhile) and their version counters."""
        return (ref, src, ref._version, src._version)

    @staticmethod
    def _is_held(held, ref, src):
        """True if (ref, src) are the tensors `held` was made from, unchanged since: same object (or a view of the same
        memory and extent) and the same version counter -- an in-place refill of the buffer bumps it (round 5
        compared addresses and row counts only)."""
        r, s, vr, vs = held
        same = lambda a, b: a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride())
        return same(r, ref) and same(s, src) and ref._version == vr and src._version == vs

    def clear_pending(self):
        """Forgets a lock-step result that was not picked up and the un-run pairs of a collated batch: the next `run` runs."""
        self._ran = None
        self._prepared = []

    def run(self, ref_points, src_points):
        """ref/src: float32 CUDA tensors [n,3] on this engine's device.  Returns the EngineResult (host)."""
        assert ref_points.is_cuda and ref_points.dtype == torch.float32 and ref_points.is_contiguous()
        assert src_points.is_cuda and src_points.dtype == torch.float32 and src_points.is_contiguous()
        ran = getattr(self, '_ran', None)
        if ran is not None:  # this pair has just run in a lock-step group (run_lockstep): its result is in place -- once
            self._ran = None
            if self._is_held(ran, ref_points, src_points):
                return self.result
        prepared = getattr(self, '_prepared', None)
        if prepared:  # the next pair of a collated batch (collate_batch): its forward alone
            k, held = prepared[0]
            if self._is_held(held, ref_points, src_points):
                prepared.pop(0)
                return self.forward_batched(k)
            self._prepared = []  # (another pair: the batch is dropped, rdm_engine_run collates this pair itself)
        _lib.check(self.L.rdm_engine_run(self._h, ref_points.data_ptr(), ref_points.shape[0], src_points.data_ptr(),
                                         src_points.shape[0], ctypes.byref(self.result), _lib.stream_ptr()),
                   'rdm_engine_run')
        return self.result

    # ------------------------------------------------------------------ several pairs per collate
    def collate_batch(self, pairs):
        """The collates of several pairs as ONE sequence of launches (rdm_engine_collate_batch): `pairs` = [(ref, src), ...]
        float32 CUDA tensors [n,3].  The pyramids stay in the engine; `forward_batched(k)` / the next `run(ref, src)` calls on
        exactly these tensors, in this order, run the pairs' forwards on them -- with the bits of `run` on each pair alone."""
        n = len(pairs)
        for r, s in pairs:
            assert r.is_cuda and r.dtype == torch.float32 and r.is_contiguous() and s.is_cuda and s.dtype == torch.float32 and s.is_contiguous()
        P, I = ctypes.c_void_p * n, ctypes.c_int64 * n
        rp, sp = P(*[r.data_ptr() for r, _ in pairs]), P(*[s.data_ptr() for _, s in pairs])
        rn, sn = I(*[r.shape[0] for r, _ in pairs]), I(*[s.shape[0] for _, s in pairs])
        self.clear_pending()
        _lib.check(self.L.rdm_engine_collate_batch(self._h, n, rp, rn, sp, sn, _lib.stream_ptr()), 'rdm_engine_collate_batch')
        # (the tensors are kept alive until their forwards have run; `run` recognises them by identity and version)
        self._prepared = [(k, self._held(r, s)) for k, (r, s) in enumerate(pairs)]
        return n

    @staticmethod
    def run_lockstep(engines, pairs, collate_batched=True):
        """rdm_engine_run of len(pairs) pairs on as many engines (sharing one copy of the weights) on the CURRENT stream, in lock
        step: launches of the same kernel of all pairs go out as one grouped launch (rdm_engine_run_lockstep, experimental).
        Returns the engines' EngineResults; every engine then holds its pair's result as after `run`."""
        n = len(pairs)
        assert 1 <= n <= len(engines)
        for r, s in pairs:
            assert r.is_cuda and r.dtype == torch.float32 and r.is_contiguous() and s.is_cuda and s.dtype == torch.float32 and s.is_contiguous()
        P, I = ctypes.c_void_p * n, ctypes.c_int64 * n
        for e in engines[:n]:
            e.clear_pending()
        hs = P(*[e._h.value if hasattr(e._h, 'value') else e._h for e in engines[:n]])
        rp, sp = P(*[r.data_ptr() for r, _ in pairs]), P(*[s.data_ptr() for _, s in pairs])
        rn, sn = I(*[r.shape[0] for r, _ in pairs]), I(*[s.shape[0] for _, s in pairs])
        res = P(*[ctypes.addressof(e.result) for e in engines[:n]])
        _lib.check(engines[0].L.rdm_engine_run_lockstep(hs, n, rp, rn, sp, sn, res, int(bool(collate_batched)), _lib.stream_ptr()), 'rdm_engine_run_lockstep')
        for e, (r, s) in zip(engines, pairs):  # (`e.run(r, s)` right after this returns the result without running again -- once)
            e._ran = Engine._held(r, s)
        return [e.result for e in engines[:n]]

    @staticmethod
    def lockstep_stats(reset=False):
        """DIAGNOSTIC, process-global counters of the lock-step scheduler (rdm_lockstep_stats): dict with ns in runs / waits,
        waits, grouped launches, records carried, runs since the last reset."""
        out = (ctypes.c_longlong * 8)()
        _lib.lib().rdm_lockstep_stats(out, int(bool(reset)))
        return {'run_ns': out[0], 'wait_ns': out[1], 'waits': out[2], 'launches': out[3], 'records': out[4], 'runs': out[5]}

    def forward_batched(self, k):
        """RDMNet.forward of pair k of the collated batch; returns the EngineResult (as `run`)."""
        _lib.check(self.L.rdm_engine_forward_batched(self._h, int(k), ctypes.byref(self.result), _lib.stream_ptr()),
                   'rdm_engine_forward_batched')
        return self.result

    _COLLATE_NAMES = ([f'points{i}' for i in range(5)] + [f'lengths{i}' for i in range(5)] + [f'neighbors{i}' for i in range(5)] +
                      [f'subsampling{i}' for i in range(4)] + [f'upsampling{i}' for i in range(4)] + ['search_flags'])

    @staticmethod
    def collate_lockstep(engines, pairs):
        """`collate` of len(pairs) pairs on as many engines on the CURRENT stream in lock step (rdm_engine_collate_lockstep):
        -> the list of data_dicts, each equal to `collate` on the pair alone (fresh tensors, independent of the arenas)."""
        n = len(pairs)
        assert 1 <= n <= len(engines)
        for r, s in pairs:
            assert r.is_cuda and r.dtype == torch.float32 and r.is_contiguous() and s.is_cuda and s.dtype == torch.float32 and s.is_contiguous()
        P, I = ctypes.c_void_p * n, ctypes.c_int64 * n
        hs = P(*[e._h.value if hasattr(e._h, 'value') else e._h for e in engines[:n]])
        rp, sp = P(*[r.data_ptr() for r, _ in pairs]), P(*[s.data_ptr() for _, s in pairs])
        rn, sn = I(*[r.shape[0] for r, _ in pairs]), I(*[s.shape[0] for _, s in pairs])
        res = P(*[ctypes.addressof(e.result) for e in engines[:n]])
        for e in engines[:n]:
            e.clear_pending()
        _lib.check(engines[0].L.rdm_engine_collate_lockstep(hs, n, rp, rn, sp, sn, res, _lib.stream_ptr()), 'rdm_engine_collate_lockstep')
        return [e._collated_dict(r.shape[0] + s.shape[0]) for e, (r, s) in zip(engines, pairs)]

    def collate(self, ref_points, src_points):
        """The reference's collate (registration_collate_fn_stack_mode, geotransformer/utils/data.py:139-192) for one pair as
        ONE native call + one batched copy: float32 CUDA clouds [n,3] -> the data_dict (fresh tensors, independent of the
        engine's arena) with 'points', 'lengths', 'neighbors', 'subsampling', 'upsampling', 'features', 'batch_size', plus
        `_widths` / `_flags` (device-resident effective table widths, as rdmnet_amd.collate returns them)."""
        assert ref_points.is_cuda and ref_points.dtype == torch.float32 and ref_points.is_contiguous()
        assert src_points.is_cuda and src_points.dtype == torch.float32 and src_points.is_contiguous()
        self.clear_pending()
        _lib.check(self.L.rdm_engine_collate(self._h, ref_points.data_ptr(), ref_points.shape[0], src_points.data_ptr(),
                                             src_points.shape[0], ctypes.byref(self.result), _lib.stream_ptr()), 'rdm_engine_collate')
        return self._collated_dict(ref_points.shape[0] + src_points.shape[0])

    def _collated_dict(self, n_points):
        """The data_dict of the collate this engine has just run (its stage tensors exported with one batched copy)."""
        t = self.tensors(self._COLLATE_NAMES)
        flags = t['search_flags'][:13]
        d = {'points': [t[f'points{i}'] for i in range(5)], 'lengths': [t[f'lengths{i}'][0] for i in range(5)],
             'neighbors': [t[f'neighbors{i}'] for i in range(5)], 'subsampling': [t[f'subsampling{i}'] for i in range(4)],
             'upsampling': [t[f'upsampling{i}'] for i in range(4)], '_flags': flags, '_widths': {},
             'features': torch.ones((n_points, 1), dtype=torch.float32, device=self.device),
             'batch_size': 1,
             # host copies of the ref/src split of every level (the collate already read them back): forward() then
             # needs no synchronisation of its own before the native call
             '_level_ref_sizes': [int(self.result.level_ref_sizes[i]) for i in range(5)]}
        # the engine searches every level's grid three times in a row -- self(i), sub(i), up(i-1) -- so that one grid serves
        # them: flag rows 0 self0, 1 sub0, 2 self1, 3 sub1, 4 up0, 5 self2, 6 sub2, 7 up1, 8 self3, 9 sub3, 10 up2, 11 self4, 12 up3
        for i in range(5):
            row = 0 if i == 0 else 3 * i - 1
            d['_widths'][('neighbors', i)] = flags[row]
            if i < 4:
                d['_widths'][('subsampling', i)] = flags[row + 1]
                d['_widths'][('upsampling', i)] = flags[(4, 7, 10, 12)[i]]
        return d

    def _pack_data_dict(self, data_dict):
        """The reference's data_dict as an rdm_data_dict of device pointers: -> (DataDict, tensors that must outlive the call).
        One host synchronisation (the ref/src split of the five levels) unless the dict carries `_level_ref_sizes`."""
        dev = self.device
        keep = []  # tensors that must outlive the call

        def dev_t(t, dtype):
            if not isinstance(t, torch.Tensor):
                t = torch.as_tensor(t)
            if t.device != dev or t.dtype != dtype:
                t = t.to(device=dev, dtype=dtype)
            keep.append(t)
            return t

        def table(t):
            t = dev_t(t, torch.int64)
            if t.dim() != 2 or t.stride(1) != 1 or t.shape[1] == 0:
                t = t.contiguous()
                keep.append(t)
            return t

        d = DataDict()
        lengths = [dev_t(x, torch.int64).contiguous() for x in data_dict['lengths']]
        keep.extend(lengths)
        n_ref = data_dict.get('_level_ref_sizes')  # host copies left by Engine.collate: no synchronisation needed here
        if n_ref is None:
            cflags = data_dict.get('_flags')  # status words of this library's collate (13 radius searches), if it built the dict
            head = torch.stack([x[0] for x in lengths])
            if cflags is not None:
                head = torch.cat([head, cflags[:, 1].to(torch.int64)])
            head = head.cpu().tolist()  # the one synchronisation of this wrapper
            n_ref = head[:5]
            if any(head[5:]):
                raise RuntimeError('radius search: a search of the collate reported an internal error (status word set)')
        cflags = data_dict.get('_flags')
        if cflags is not None and cflags.is_cuda and cflags.dtype == torch.int32 and cflags.dim() == 2 and cflags.is_contiguous():
            keep.append(cflags)  # checked by the native call at its first read-back (no synchronisation here)
            d.collate_status, d.n_collate_status = cflags.data_ptr(), cflags.shape[0]
        widths = data_dict.get('_widths', {})
        feats = dev_t(data_dict['features'], torch.float32)
        if feats.dim() != 2 or feats.stride(1) != 1:
            feats = feats.reshape(feats.shape[0], -1).contiguous()
            keep.append(feats)
        d.features, d.features_ld = feats.data_ptr(), feats.stride(0) if feats.shape[0] > 1 else feats.shape[1]
        for i in range(5):
            p = dev_t(data_dict['points'][i], torch.float32).contiguous()
            keep.append(p)
            d.points[i], d.lengths[i], d.n_points[i], d.n_ref[i] = p.data_ptr(), lengths[i].data_ptr(), p.shape[0], int(n_ref[i])
            for key, arr_i in (('neighbors', i), ('subsampling', i), ('upsampling', i)):
                if key != 'neighbors' and i == 4:
                    continue
                t = table(data_dict[key][i])
                getattr(d, key)[i] = t.data_ptr()
                getattr(d, key + '_width')[i] = t.shape[1]
                getattr(d, key + '_ld')[i] = t.stride(0) if t.shape[0] > 1 else t.shape[1]
                w = widths.get((key, i))
                getattr(d, key + '_count')[i] = w.data_ptr() if w is not None else None
        return d, keep

    def forward(self, data_dict):
        """RDMNet.forward(data_dict) as ONE native call (rdm_engine_forward).  data_dict: the collate's dictionary
        (rdmnet_amd.collate or the reference's registration_collate_fn_stack_mode moved to this engine's device).
        Returns the EngineResult; stage tensors through tensor() when keep_taps is on.  One host synchronisation
        here (the ref/src split of the five levels) besides the engine's own."""
        d, keep = self._pack_data_dict(data_dict)
        self.clear_pending()
        _lib.check(self.L.rdm_engine_forward(self._h, ctypes.byref(d), ctypes.byref(self.result), _lib.stream_ptr()),
                   'rdm_engine_forward')
        del keep
        return self.result

    @staticmethod
    def forward_lockstep(engines, data_dicts):
        """RDMNet.forward of len(data_dicts) callers' data_dicts on as many engines (sharing one copy of the weights) on the
        CURRENT stream, in lock step (rdm_engine_forward_lockstep): the same kernel of all pairs goes out as one grouped
        launch.  Every engine then holds its pair's result and -- with keep_taps -- its stage tensors, exactly as after
        `forward` on that pair alone.  Returns the engines' EngineResults."""
        n = len(data_dicts)
        assert 1 <= n <= len(engines)
        packed = [e._pack_data_dict(dd) for e, dd in zip(engines, data_dicts)]
        P = ctypes.c_void_p * n
        hs = P(*[e._h.value if hasattr(e._h, 'value') else e._h for e in engines[:n]])
        dds = P(*[ctypes.addressof(d) for d, _ in packed])
        res = P(*[ctypes.addressof(e.result) for e in engines[:n]])
        for e in engines[:n]:
            e.clear_pending()
        _lib.check(engines[0].L.rdm_engine_forward_lockstep(hs, n, dds, res, _lib.stream_ptr()), 'rdm_engine_forward_lockstep')
        del packed
        return [e.result for e in engines[:n]]

    def transform(self):
        return np.ctypeslib.as_array(self.result.transform).reshape(4, 4).copy()

    def tensor(self, name):
        """Copy of a stage tensor of the last run (requires keep_taps before the run)."""
        v = TensorView()
        _lib.check(self.L.rdm_engine_get_tensor(self._h, name.encode(), ctypes.byref(v)), 'rdm_engine_get_tensor')
        dt = _DTYPES[v.dtype]
        out = torch.empty((v.rows, v.ld), dtype=dt, device=self.device)
        if v.rows > 0:
            _lib.check(self.L.rdm_copy_device(out.data_ptr(), v.data, out.numel() * out.element_size(), _lib.stream_ptr()),
                       'rdm_copy_device')
        return out[:, :v.cols]

    def tensors(self, names):
        """Copies of several stage tensors of the last run with one batched launch (rdm_engine_export): {name: tensor}.
        All of them live in ONE allocation (strided views), so the call costs one torch.empty and one kernel."""
        n = len(names)
        key = tuple(names)
        cached = self._export_cache.get(key)
        if cached is None:
            cached = self._export_cache[key] = ((ctypes.c_char_p * n)(*[x.encode() for x in names]), (TensorView * n)(),
                                                (ctypes.c_void_p * n)())
        arr_n, arr_v, arr_d = cached
        _lib.check(self.L.rdm_engine_describe(self._h, n, arr_n, arr_v), 'rdm_engine_describe')
        offs, total = [], 0
        for v in arr_v:
            offs.append(total)
            total += (v.rows * v.ld * _ESIZE[v.dtype] + 15) // 16 * 16
        buf = torch.empty((max(total, 16),), dtype=torch.uint8, device=self.device)
        base = buf.data_ptr()
        for i in range(n):
            arr_d[i] = base + offs[i]
        _lib.check(self.L.rdm_engine_export(self._h, n, arr_n, arr_d, _lib.stream_ptr()), 'rdm_engine_export')
        typed = {0: buf.view(torch.float32), 1: buf.view(torch.int64), 2: buf, 3: buf.view(torch.int32)}
        return {names[i]: torch.as_strided(typed[v.dtype], (v.rows, v.cols), (v.ld, 1), offs[i] // _ESIZE[v.dtype])
                for i, v in enumerate(arr_v)}

    def host_corr(self):
        """(ref_corr_points [n,3], src_corr_points [n,3], corr_scores [n]) of the last run as numpy copies of the engine's
        pinned host buffer -- already on the host when run() / forward() return (no device copy, no synchronisation)."""
        n = int(self.result.n_host_correspondences)

        def view(ptr, count):
            if count == 0:
                return np.zeros((0,), np.float32)
            return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), shape=(count,)).copy()
        return (view(self.result.host_ref_corr_points, 3 * n).reshape(n, 3), view(self.result.host_src_corr_points, 3 * n).reshape(n, 3),
                view(self.result.host_corr_scores, n))

    def corr(self):
        """(ref_corr_points, src_corr_points, corr_scores) of the last run as fresh tensors."""
        n = self.result.n_correspondences
        outs = []
        for ptr, cols in ((self.result.ref_corr_points, 3), (self.result.src_corr_points, 3), (self.result.corr_scores, 1)):
            t = torch.empty((n, cols), dtype=torch.float32, device=self.device)
            if n > 0:
                _lib.check(self.L.rdm_copy_device(t.data_ptr(), ptr, t.numel() * 4, _lib.stream_ptr()), 'rdm_copy_device')
            outs.append(t)
        return outs[0], outs[1], outs[2][:, 0]
