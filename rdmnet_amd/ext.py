"""Drop-in for the reference's native module `rdmnet.ext`
(geotransformer/extensions/pybind.cpp:6-17): same function names, argument meaning and
`RuntimeError` behaviour for wrong dtype / non-contiguous inputs
(geotransformer/extensions/common/torch_helper.h:6-35).

Difference, by design: the work runs on the GPU.  CPU tensors (what the reference's callers pass,
geotransformer/utils/data.py:25-67, rdmnet/vote/vote.py:24-31) are staged to `cuda:0` and the
results come back on the CPU; CUDA tensors stay on the device.  There is no CPU fallback.
"""
import torch

from . import _lib


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _check(t, name, dtype, dtype_name):
    _require(isinstance(t, torch.Tensor), f'{name} must be a tensor')
    _require(t.dtype == dtype, f'{name} must be a {dtype_name} tensor')
    _require(t.is_contiguous(), f'{name} must be contiguous')


def _device_of(*tensors):
    devs = {t.device for t in tensors}
    _require(len(devs) == 1, 'all tensors must live on the same device')
    dev = devs.pop()
    if dev.type == 'cpu':
        _require(torch.cuda.is_available(), 'rdmnet_amd.ext needs a GPU (no CPU fallback)')
        return torch.device('cuda', torch.cuda.current_device()), True
    return dev, False


_ws_cache = {}


def _workspace(dev, nbytes):
    # one workspace per (device, stream): concurrent callers on different streams must not share scratch, and a
    # buffer is only ever replaced by work queued on its own stream (torch's allocator is stream-ordered)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


def grid_subsampling(points, lengths, voxel_size):
    """[s_points, s_lengths] = grid_subsampling(points f32[N,3], lengths i64[B], voxel_size)."""
    _check(points, 'points', torch.float32, 'float')
    _check(lengths, 'lengths', torch.int64, 'long')
    _require(points.dim() == 2 and points.shape[1] == 3, 'points must be (N, 3)')
    dev, from_cpu = _device_of(points, lengths)
    L = _lib.lib()
    with torch.cuda.device(dev):
        d_points = points.to(dev)
        d_lengths = lengths.to(dev)
        n, batch = d_points.shape[0], d_lengths.shape[0]
        out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=dev)
        out_len = torch.empty((batch,), dtype=torch.int64, device=dev)
        ws = _workspace(dev, L.rdm_grid_subsample_workspace_bytes(n, batch))
        _lib.check(L.rdm_grid_subsample(d_points.data_ptr(), n, d_lengths.data_ptr(), batch,
                                        float(voxel_size), out.data_ptr(), out_len.data_ptr(),
                                        ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
                   'rdm_grid_subsample')
        total = int(out_len.sum().item())  # output shape is data dependent: one sync
        s_points = out[:total].clone()
    if from_cpu:
        return [s_points.cpu(), out_len.cpu()]
    return [s_points, out_len]


def radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius, *, width=None):
    """neighbor_indices i64[Nq, max_count] = radius_neighbors(q, s, q_lengths, s_lengths, radius).

    `width` (keyword only, not in the reference) skips the counting pass and returns exactly that
    many columns, i.e. what `radius_search(..., neighbor_limit)` keeps when limit <= max_count.
    """
    _check(q_points, 'q_points', torch.float32, 'float')
    _check(s_points, 's_points', torch.float32, 'float')
    _check(q_lengths, 'q_lengths', torch.int64, 'long')
    _check(s_lengths, 's_lengths', torch.int64, 'long')
    _require(q_lengths.shape[0] == s_lengths.shape[0], 'q_lengths and s_lengths must have the same size')
    dev, from_cpu = _device_of(q_points, s_points, q_lengths, s_lengths)
    L = _lib.lib()
    with torch.cuda.device(dev):
        q, s = q_points.to(dev), s_points.to(dev)
        ql, sl = q_lengths.to(dev), s_lengths.to(dev)
        nq, ns, batch = q.shape[0], s.shape[0], ql.shape[0]
        ws = _workspace(dev, L.rdm_radius_neighbors_workspace_bytes(nq, ns, batch))
        flags = torch.zeros(2, dtype=torch.int32, device=dev)  # [max_count, status]
        st = _lib.stream_ptr()

        def run(w, out):
            _lib.check(L.rdm_radius_neighbors(q.data_ptr(), nq, s.data_ptr(), ns, ql.data_ptr(),
                                              sl.data_ptr(), batch, float(radius), w, _lib.ptr(out), 0,
                                              flags.data_ptr(), flags[1:].data_ptr(), ws.data_ptr(),
                                              ws.numel(), st), 'rdm_radius_neighbors')

        if width is None:
            run(0, None)
            width = int(flags[0].item())
        out = torch.empty((nq, width), dtype=torch.int64, device=dev)
        if width > 0 and nq > 0:
            run(width, out)
        if int(flags[1].item()) != 0:  # (no neighbour-count limit: dense rows are produced in rounds; this is an internal error)
            raise RuntimeError(f'rdm_radius_neighbors: status {int(flags[1].item())}')
    return out.cpu() if from_cpu else out
