"""Golden-vector tooling (runs ONLY in the build container, never on the GPU box).

Imports the read-only reference at /root/reference on CPU so that gen_golden.py can capture
golden inputs/outputs from the reference's own Python (`experiments/model_infer.py`,
`geotransformer/utils/data.py`).  Nothing from the reference is copied: this module only
installs the import shims SURVEY.md §8c lists (missing third-party modules, hard-coded
`.cuda()` calls) and backs `rdmnet.ext` with oracle/_ref/libref_ext.so, i.e. the reference's
own C++ compiled where it lies.
"""
import os
import struct
import sys
import types

import numpy as np
import torch

REF = os.environ.get('RDM_REFERENCE', '/root/reference')
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _read_ply_points(path):
    """Minimal binary-little-endian PLY vertex reader (x, y, z doubles) for the kernel disposition."""
    with open(path, 'rb') as f:
        n, props = 0, []
        while True:
            line = f.readline().decode('ascii').strip()
            if line.startswith('element vertex'):
                n = int(line.split()[-1])
            elif line.startswith('property'):
                props.append(line.split()[1])
            elif line == 'end_header':
                break
        fmt = {'double': 'd', 'float64': 'd', 'float': 'f', 'float32': 'f'}
        rec = '<' + ''.join(fmt[p] for p in props)
        size = struct.calcsize(rec)
        pts = [struct.unpack(rec, f.read(size))[:3] for _ in range(n)]
    return np.asarray(pts, dtype=np.float64)


def install():
    if getattr(install, 'done', False):
        return
    install.done = True
    sys.path.insert(0, REPO)
    from oracle import native
    ref_native = native.reference()
    assert ref_native is not None, 'run `make -C oracle` first (needs /root/reference)'

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

    _stub('easydict', EasyDict=EasyDict)

    class _PCD:
        def __init__(self, pts=None):
            self.points = pts

    o3d = _stub('open3d')
    o3d.io = types.SimpleNamespace(read_point_cloud=lambda p: _PCD(_read_ply_points(p)))
    o3d.geometry = types.SimpleNamespace(PointCloud=_PCD)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda x: x)
    _stub('ipdb')
    _stub('IPython', embed=lambda *a, **k: None)
    _stub('zmq', device=None)
    _stub('coloredlogs', install=lambda *a, **k: None)
    np.int = int  # rdmnet/thdroformer/thdroformer.py:71 uses the removed alias

    # hard-coded .cuda() calls -> CPU no-ops (contiguous, as a device copy would be)
    torch.Tensor.cuda = lambda self, *a, **k: self.contiguous()
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _orig_device = torch.device

    # rdmnet.ext -> the reference's own native code (oracle/_ref)
    pkg = _stub('rdmnet')
    pkg.__path__ = [os.path.join(REF, 'rdmnet')]

    def grid_subsampling(points, lengths, voxel_size):
        assert points.dtype == torch.float32 and lengths.dtype == torch.int64
        p, l = ref_native.grid_subsampling(points.numpy(), lengths.numpy(), float(voxel_size))
        return [torch.from_numpy(p), torch.from_numpy(l)]

    def radius_neighbors(q, s, ql, sl, radius):
        idx = ref_native.radius_neighbors(q.contiguous().numpy(), s.contiguous().numpy(), ql.numpy(),
                                          sl.numpy(), float(radius))
        return torch.from_numpy(idx)

    ext = _stub('rdmnet.ext', grid_subsampling=grid_subsampling, radius_neighbors=radius_neighbors)
    pkg.ext = ext
    # rdmnet.utils.visualization imports modules that do not exist in the reference tree
    _stub('rdmnet.utils', __path__=[])
    _stub('rdmnet.utils.visualization', vis_shifte_node=None, visualization=None, vis_node_grouping=None)

    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, 'experiments'))
    import geotransformer.utils.common as common
    common.ensure_dir = lambda p: None  # config.py would mkdir next to the read-only tree
    os.chdir(REF)  # the 'infer' dataset reads ./assets/pc


def make_cfg():
    install()
    import config
    cfg = config.make_cfg()
    cfg.test.vis = False
    return cfg
