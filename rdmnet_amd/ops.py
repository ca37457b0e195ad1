"""Functional front-end of the C-ABI kernels on torch CUDA tensors (device memory + stream only).

Every function enqueues on the current torch stream and returns freshly allocated outputs.  Feature
matrices are row-major with a row stride that is a multiple of 4 floats (`.stride(0)`); logical
widths are passed explicitly where they differ.  No function here falls back to torch math.
"""
import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def pad4(n):
    return (int(n) + 3) // 4 * 4


class Scratch:
    """Grow-only device scratch for kernels that take a caller workspace."""

    def __init__(self, device):
        self.device = device
        self.buf = torch.empty(1 << 24, dtype=torch.uint8, device=device)

    def get(self, nbytes):
        if self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return self.buf


_scratch = {}


def scratch(device, nbytes):
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    s = _scratch.get(key)
    if s is None:
        s = _scratch[key] = Scratch(device)
    return s.get(nbytes)


def feat_empty(n, c, device):
    """[n, c] view of a buffer whose row stride is padded to a multiple of 4 floats."""
    ld = pad4(c)
    buf = torch.empty((max(int(n), 1), ld), dtype=torch.float32, device=device)
    return buf[:n, :c]


def _ld(t):
    return t.stride(0) if t.dim() == 2 else t.shape[-1]


def gemm(a, b, k, n, *, trans_b=False, bias=None, rowdiv=None, act=ACT_NONE, out=None):
    """out[m, :n] = act(a[m, :k] @ op(b) / rowdiv + bias).  a: [m, >=k] view, b: [k(pad), n(pad)]
    (trans_b False) or [n, >=k] view (trans_b True); k must be a multiple of 4."""
    L = _lib.lib()
    m = a.shape[0]
    if out is None:
        out = feat_empty(m, n, a.device)
    ws_bytes = L.rdm_gemm_workspace_bytes(m, n, 1)
    ws = scratch(a.device, ws_bytes)
    _lib.check(L.rdm_gemm(a.data_ptr(), _ld(a), 0, b.data_ptr(), _ld(b), 0, int(trans_b), out.data_ptr(), _ld(out),
                          0, m, n, k, 1, _lib.ptr(bias), _lib.ptr(rowdiv), act, ws.data_ptr(), ws.numel(),
                          _lib.stream_ptr()), 'rdm_gemm')
    return out


def gemm_batched(a, b, k, *, trans_b=True, out=None):
    """a [B, m, >=k], b [B, n, >=k] contiguous batches -> out [B, m, n]."""
    L = _lib.lib()
    B, m, n = a.shape[0], a.shape[1], b.shape[1]
    if out is None:
        out = torch.empty((B, m, n), dtype=torch.float32, device=a.device)
    _lib.check(L.rdm_gemm(a.data_ptr(), a.stride(1), a.stride(0), b.data_ptr(), b.stride(1), b.stride(0),
                          int(trans_b), out.data_ptr(), out.stride(1), out.stride(0), m, n, k, B, 0, 0, ACT_NONE, 0, 0,
                          _lib.stream_ptr()), 'rdm_gemm(batched)')
    return out


def row_positive(x):
    L = _lib.lib()
    out = torch.empty((max(x.shape[0], 1),), dtype=torch.uint8, device=x.device)
    _lib.check(L.rdm_row_positive(x.data_ptr(), x.shape[0], x.shape[1], _ld(x), out.data_ptr(), _lib.stream_ptr()),
               'rdm_row_positive')
    return out


def kpconv_gather(q_points, s_points, s_feats, s_positive, idx, kernel_points, sigma, width=None):
    """-> (wf [m, pad4(15*c)] view, nn [m])."""
    L = _lib.lib()
    m, c = q_points.shape[0], s_feats.shape[1]
    kdim = 16 if c == 1 else 15 * c
    wf = feat_empty(m, kdim, q_points.device)
    nn = torch.empty((max(m, 1),), dtype=torch.float32, device=q_points.device)
    _lib.check(L.rdm_kpconv_gather(q_points.data_ptr(), m, s_points.data_ptr(), s_points.shape[0], s_feats.data_ptr(),
                                   c, _ld(s_feats), s_positive.data_ptr(), idx.data_ptr(), idx.shape[1], idx.stride(0),
                                   _lib.ptr(width), kernel_points.data_ptr(), float(sigma), wf.data_ptr(), _ld(wf),
                                   nn.data_ptr(), _lib.stream_ptr()), 'rdm_kpconv_gather')
    return wf, nn


def group_norm(x, gamma, beta, groups, *, act=ACT_NONE, residual=None, want_positive=False, eps=1e-5):
    L = _lib.lib()
    n, c = x.shape
    y = feat_empty(n, c, x.device)
    pos = torch.empty((max(n, 1),), dtype=torch.uint8, device=x.device) if want_positive else None
    ws_bytes = L.rdm_group_norm_workspace_bytes(n, c)
    ws = scratch(x.device, ws_bytes)
    _lib.check(L.rdm_group_norm(x.data_ptr(), n, c, _ld(x), groups, gamma.data_ptr(), beta.data_ptr(), eps,
                                _lib.ptr(residual), _ld(residual) if residual is not None else 0, act, y.data_ptr(),
                                _ld(y), _lib.ptr(pos), ws.data_ptr(), ws.numel(), _lib.stream_ptr()), 'rdm_group_norm')
    return (y, pos) if want_positive else y


def layer_norm(x, gamma, beta, *, residual=None, act=ACT_NONE, eps=1e-5, out=None):
    L = _lib.lib()
    n, c = x.shape
    y = out if out is not None else feat_empty(n, c, x.device)
    _lib.check(L.rdm_layer_norm(x.data_ptr(), n, c, _ld(x), _lib.ptr(residual),
                                _ld(residual) if residual is not None else 0, gamma.data_ptr(), beta.data_ptr(), eps,
                                act, y.data_ptr(), _ld(y), _lib.stream_ptr()), 'rdm_layer_norm')
    return y


def gather_max(x, idx, width=None):
    L = _lib.lib()
    m, c = idx.shape[0], x.shape[1]
    y = feat_empty(m, c, x.device)
    _lib.check(L.rdm_gather_max(x.data_ptr(), x.shape[0], c, _ld(x), idx.data_ptr(), m, idx.shape[1], idx.stride(0),
                                _lib.ptr(width), y.data_ptr(), _ld(y), _lib.stream_ptr()), 'rdm_gather_max')
    return y


def upsample_concat(coarse, idx, skip):
    L = _lib.lib()
    m, c1, c2 = skip.shape[0], coarse.shape[1], skip.shape[1]
    y = feat_empty(m, c1 + c2, skip.device)
    _lib.check(L.rdm_upsample_concat(coarse.data_ptr(), coarse.shape[0], c1, _ld(coarse), idx.data_ptr(), idx.stride(0),
                                     skip.data_ptr(), c2, _ld(skip), m, y.data_ptr(), _ld(y), _lib.stream_ptr()),
               'rdm_upsample_concat')
    return y
