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


def gemm(a, b, k, n, *, trans_b=False, bias=None, rowdiv=None, act=ACT_NONE, out=None, form=0):
    """out[m, :n] = act(a[m, :k] @ op(b) / rowdiv + bias).  a: [m, >=k] view, b: [k(pad), n(pad)]
    (trans_b False) or [n, >=k] view (trans_b True); k must be a multiple of 4.  form: rdm_gemm_form (0 = the library's choice,
    1 / 2 = the wide tile forms; same bits)."""
    L = _lib.lib()
    m = a.shape[0]
    if a.stride(-1) != 1 or b.stride(-1) != 1:
        raise ValueError('gemm operands must be row-major (unit column stride)')
    if out is None:
        out = feat_empty(m, n, a.device)
    ws_bytes = L.rdm_gemm_workspace_bytes(m, n, 1)
    ws = scratch(a.device, ws_bytes)
    _lib.check(L.rdm_gemm_form(a.data_ptr(), _ld(a), 0, b.data_ptr(), _ld(b), 0, int(trans_b), out.data_ptr(), _ld(out),
                               0, m, n, k, 1, _lib.ptr(bias), _lib.ptr(rowdiv), act, ws.data_ptr(), ws.numel(), int(form),
                               _lib.stream_ptr()), 'rdm_gemm')
    return out


def gemm_batched(a, b, k, *, trans_b=True, out=None, rowdiv=None):
    """a [B, m, >=k], b [B, n, >=k] contiguous batches -> out [B, m, n] (/ rowdiv[row])."""
    L = _lib.lib()
    B, m, n = a.shape[0], a.shape[1], b.shape[1]
    if out is None:
        out = torch.empty((B, m, n), dtype=torch.float32, device=a.device)
    _lib.check(L.rdm_gemm(a.data_ptr(), a.stride(1), a.stride(0), b.data_ptr(), b.stride(1), b.stride(0),
                          int(trans_b), out.data_ptr(), out.stride(1), out.stride(0), m, n, k, B, 0, _lib.ptr(rowdiv), ACT_NONE, 0, 0,
                          _lib.stream_ptr()), 'rdm_gemm(batched)')
    return out


def patch_scores(ref_feats, ref_idx, src_feats, src_idx, rowdiv=None):
    """scores[b, i, j] = <ref_feats[ref_idx[b, i]], src_feats[src_idx[b, j]]> / rowdiv[i] (model_infer.py:291-311); an index
    outside the tensor selects a zero row.  ref_idx / src_idx: [B, k] int64 -> [B, k, k]."""
    L = _lib.lib()
    B, k = ref_idx.shape
    out = torch.empty((B, k, k), dtype=torch.float32, device=ref_feats.device)
    _lib.check(L.rdm_patch_scores(ref_feats.data_ptr(), _ld(ref_feats), ref_feats.shape[0], ref_idx.data_ptr(), src_feats.data_ptr(),
                                  _ld(src_feats), src_feats.shape[0], src_idx.data_ptr(), B, k, ref_feats.shape[1], _lib.ptr(rowdiv),
                                  out.data_ptr(), _lib.stream_ptr()), 'rdm_patch_scores')
    return out


def row_positive(x):
    L = _lib.lib()
    out = torch.empty((max(x.shape[0], 1),), dtype=torch.uint8, device=x.device)
    _lib.check(L.rdm_row_positive(x.data_ptr(), x.shape[0], x.shape[1], _ld(x), out.data_ptr(), _lib.stream_ptr()),
               'rdm_row_positive')
    return out


def kpconv_gather(q_points, s_points, s_feats, s_positive, idx, kernel_points, sigma, width=None, order=None, form=0):
    """-> (wf [m, pad4(15*c)] view, nn [m]).  order: the query level's cell-sorted records (radius_grid_records) = the order the
    queries are visited in; form: 0 = the library's choice, 1 = one wavefront per (query, slice), 2 = the LDS-tile form (needs
    `order`, c a multiple of 64 >= 128, at most 128 slots).  The same bits in every form and order."""
    L = _lib.lib()
    m, c = q_points.shape[0], s_feats.shape[1]
    kdim = 16 if c == 1 else 15 * c
    wf = feat_empty(m, kdim, q_points.device)
    nn = torch.empty((max(m, 1),), dtype=torch.float32, device=q_points.device)
    _lib.check(L.rdm_kpconv_gather_form(q_points.data_ptr(), m, s_points.data_ptr(), s_points.shape[0], s_feats.data_ptr(),
                                        c, _ld(s_feats), s_positive.data_ptr(), idx.data_ptr(), idx.shape[1], idx.stride(0),
                                        _lib.ptr(width), kernel_points.data_ptr(), float(sigma), wf.data_ptr(), _ld(wf),
                                        nn.data_ptr(), _lib.ptr(order), int(form), _lib.stream_ptr()), 'rdm_kpconv_gather')
    return wf, nn


def kpconv_fused_enabled():
    return bool(_lib.lib().rdm_kpconv_fused_enabled())


def kpconv_fused_supported(c_in, c_out):
    return bool(_lib.lib().rdm_kpconv_fused_supported(int(c_in), int(c_out)))


def kpconv_pack_weights(w):
    """KPConv weights [15, c_in, c_out] (numpy, checkpoint layout) -> float32 numpy array in the fused kernel's operand order."""
    import numpy as np
    L = _lib.lib()
    _, c_in, c_out = w.shape
    w = np.ascontiguousarray(w, dtype=np.float32)
    out = np.empty((L.rdm_kpconv_packed_floats(c_in, c_out),), dtype=np.float32)
    _lib.check(L.rdm_kpconv_pack_weights(w.ctypes.data, c_in, c_out, out.ctypes.data), 'rdm_kpconv_pack_weights')
    return out


def kpconv_fused(q_points, s_points, s_feats, s_positive, idx, kernel_points, sigma, w_packed, bias, c_out, width=None,
                 want_partials=False, order=None, form=0):
    """The whole KPConv.forward (kpconv.py:79-122) in one kernel (c_in = 1, 32, 64) -> out [m, c_out]
    (, fp64 GroupNorm partials [blocks, 2, c_out])."""
    L = _lib.lib()
    m, c = q_points.shape[0], s_feats.shape[1]
    out = feat_empty(m, c_out, q_points.device)
    part = None
    if want_partials:
        nblk = max(int(L.rdm_kpconv_fused_partial_rows(m, c)), 1)
        part = torch.empty((nblk, 2, c_out), dtype=torch.float64, device=q_points.device)
    _lib.check(L.rdm_kpconv_fused_form(q_points.data_ptr(), m, s_points.data_ptr(), s_points.shape[0], s_feats.data_ptr(), c,
                                  _ld(s_feats), s_positive.data_ptr(), idx.data_ptr(), idx.shape[1], idx.stride(0), _lib.ptr(width),
                                  kernel_points.data_ptr(), float(sigma), w_packed.data_ptr(), bias.data_ptr(), c_out,
                                       out.data_ptr(), _ld(out), _lib.ptr(part), _lib.ptr(order), int(form), _lib.stream_ptr()),
               'rdm_kpconv_fused')
    return (out, part) if want_partials else out


def kpconv_fused_group_norm(q_points, s_points, s_feats, s_positive, idx, kernel_points, sigma, w_packed, bias, c_out, gamma,
                            beta, groups, *, width=None, act=ACT_LEAKY, eps=1e-5, order=None):
    """act(GroupNorm(KPConv(...))) -- the fused convolution followed by the normalisation every backbone block applies."""
    L = _lib.lib()
    m, c = q_points.shape[0], s_feats.shape[1]
    conv, y = feat_empty(m, c_out, q_points.device), feat_empty(m, c_out, q_points.device)
    ws = scratch(q_points.device, L.rdm_kpconv_fused_workspace_bytes(m, c, c_out))
    _lib.check(L.rdm_kpconv_fused_group_norm(q_points.data_ptr(), m, s_points.data_ptr(), s_points.shape[0], s_feats.data_ptr(), c,
                                             _ld(s_feats), s_positive.data_ptr(), idx.data_ptr(), idx.shape[1], idx.stride(0),
                                             _lib.ptr(width), kernel_points.data_ptr(), float(sigma), w_packed.data_ptr(),
                                             bias.data_ptr(), c_out, groups, gamma.data_ptr(), beta.data_ptr(), eps, act,
                                             conv.data_ptr(), _ld(conv), y.data_ptr(), _ld(y), ws.data_ptr(), ws.numel(),
                                             _lib.ptr(order), _lib.stream_ptr()), 'rdm_kpconv_fused_group_norm')
    return y


def group_norm(x, gamma, beta, groups, *, act=ACT_NONE, residual=None, want_positive=False, eps=1e-5, form=0):
    """form: 0 = the library's launch structure (finalize + apply as one launch on the coarse levels), 1 = three launches."""
    L = _lib.lib()
    n, c = x.shape
    y = feat_empty(n, c, x.device)
    pos = torch.empty((max(n, 1),), dtype=torch.uint8, device=x.device) if want_positive else None
    ws_bytes = L.rdm_group_norm_workspace_bytes(n, c)
    ws = scratch(x.device, ws_bytes)
    _lib.check(L.rdm_group_norm_form(x.data_ptr(), n, c, _ld(x), groups, gamma.data_ptr(), beta.data_ptr(), eps,
                                     _lib.ptr(residual), _ld(residual) if residual is not None else 0, act, y.data_ptr(),
                                     _ld(y), _lib.ptr(pos), ws.data_ptr(), ws.numel(), int(form), _lib.stream_ptr()), 'rdm_group_norm')
    return (y, pos) if want_positive else y


def linear_group_norm(x, b, k, n, bias, gamma, beta, groups, *, rowdiv=None, act=ACT_NONE, residual=None,
                      want_positive=False, eps=1e-5, form=0):
    """act(GroupNorm(x[:, :k] @ b + bias [/ rowdiv]) [+ residual]); statistics from the GEMM epilogue."""
    L = _lib.lib()
    m = x.shape[0]
    lin = feat_empty(m, n, x.device)
    y = feat_empty(m, n, x.device)
    pos = torch.empty((max(m, 1),), dtype=torch.uint8, device=x.device) if want_positive else None
    ws = scratch(x.device, L.rdm_linear_group_norm_workspace_bytes(m, n))
    _lib.check(L.rdm_linear_group_norm_form(x.data_ptr(), _ld(x), b.data_ptr(), _ld(b), _lib.ptr(bias), _lib.ptr(rowdiv), m, n, k,
                                            groups, gamma.data_ptr(), beta.data_ptr(), eps, _lib.ptr(residual),
                                            _ld(residual) if residual is not None else 0, act, lin.data_ptr(), _ld(lin),
                                            y.data_ptr(), _ld(y), _lib.ptr(pos), ws.data_ptr(), ws.numel(), int(form), _lib.stream_ptr()),
               'rdm_linear_group_norm')
    return (y, pos) if want_positive else y


def decoder_stage(coarse, idx, skip, b, n, bias, gamma=None, beta=None, groups=32, *, act=ACT_NONE, eps=1e-5, form=0):
    """One decoder stage (backbone.py:118-151): act(GroupNorm([coarse[idx[:, 0]] | skip] @ b + bias)), or the plain Linear when
    gamma is None.  b: [pad4(c1 + c2), pad4(n)] as for gemm."""
    L = _lib.lib()
    m, c1, c2 = skip.shape[0], coarse.shape[1], skip.shape[1]
    lin = feat_empty(m, n, skip.device)
    y = feat_empty(m, n, skip.device) if gamma is not None else None
    ws = scratch(skip.device, L.rdm_decoder_stage_workspace_bytes(m, n, c1 + c2))
    _lib.check(L.rdm_decoder_stage_form(coarse.data_ptr(), coarse.shape[0], c1, _ld(coarse), idx.data_ptr(), _ld(idx), skip.data_ptr(),
                                        c2, _ld(skip), m, b.data_ptr(), _ld(b), _lib.ptr(bias), n, groups, _lib.ptr(gamma),
                                        _lib.ptr(beta), eps, act, lin.data_ptr(), _ld(lin), _lib.ptr(y), _ld(y) if y is not None else 0,
                                        ws.data_ptr(), ws.numel(), int(form), _lib.stream_ptr()), 'rdm_decoder_stage')
    return y if gamma is not None else lin


def layer_norm(x, gamma, beta, *, residual=None, act=ACT_NONE, eps=1e-5, out=None):
    L = _lib.lib()
    n, c = x.shape
    y = out if out is not None else feat_empty(n, c, x.device)
    _lib.check(L.rdm_layer_norm(x.data_ptr(), n, c, _ld(x), _lib.ptr(residual),
                                _ld(residual) if residual is not None else 0, gamma.data_ptr(), beta.data_ptr(), eps,
                                act, y.data_ptr(), _ld(y), _lib.stream_ptr()), 'rdm_layer_norm')
    return y


def attention_tail(hidden, x, wo, bo, gamma1, beta1, w1, b1, w2, b2, gamma2, beta2, *, eps=1e-5, out=None):
    """The tail of an attention layer in one launch: y = LN(hidden @ wo.T + bo + x); out = LN(relu(y @ w1.T + b1) @ w2.T
    + b2 + y).  Width 128, FFN 256; weights as nn.Linear stores them (wo [128,128], w1 [256,128], w2 [128,256])."""
    L = _lib.lib()
    m = hidden.shape[0]
    y = out if out is not None else feat_empty(m, 128, hidden.device)
    _lib.check(L.rdm_attention_tail(hidden.data_ptr(), _ld(hidden), x.data_ptr(), _ld(x), m, wo.shape[0], wo.data_ptr(), _ld(wo),
                                    _lib.ptr(bo), gamma1.data_ptr(), beta1.data_ptr(), w1.data_ptr(), _ld(w1), _lib.ptr(b1),
                                    w2.data_ptr(), _ld(w2), _lib.ptr(b2), gamma2.data_ptr(), beta2.data_ptr(), eps,
                                    y.data_ptr(), _ld(y), _lib.stream_ptr()), 'rdm_attention_tail')
    return y


def attention_tail_pack_weights(wo, w1, w2):
    """wo [128,128], w1 [256,128], w2 [128,256] (nn.Linear layout) -> the tail kernel's operand order (one 81 920-float tensor)."""
    L = _lib.lib()
    packed = torch.empty((L.rdm_attention_tail_packed_floats(),), dtype=torch.float32, device=wo.device)
    _lib.check(L.rdm_attention_tail_pack_weights(wo.data_ptr(), _ld(wo), w1.data_ptr(), _ld(w1), w2.data_ptr(), _ld(w2), packed.data_ptr(),
                                                 _lib.stream_ptr()), 'rdm_attention_tail_pack_weights')
    return packed


def attention_tail_packed(hidden, x, packed, bo, gamma1, beta1, b1, b2, gamma2, beta2, *, eps=1e-5, out=None):
    """attention_tail on weights packed by attention_tail_pack_weights: the same bits, contiguous weight loads."""
    L = _lib.lib()
    m = hidden.shape[0]
    y = out if out is not None else feat_empty(m, 128, hidden.device)
    _lib.check(L.rdm_attention_tail_packed(hidden.data_ptr(), _ld(hidden), x.data_ptr(), _ld(x), m, 128, packed.data_ptr(), _lib.ptr(bo),
                                           gamma1.data_ptr(), beta1.data_ptr(), _lib.ptr(b1), _lib.ptr(b2), gamma2.data_ptr(), beta2.data_ptr(),
                                           eps, y.data_ptr(), _ld(y), _lib.stream_ptr()), 'rdm_attention_tail_packed')
    return y


def linear_layer_norm(x, w, k, n, bias, gamma, beta, *, residual=None, act=ACT_NONE, eps=1e-5, out=None):
    """act(LayerNorm(x[:, :k] @ w.T + bias [+ residual])) in one launch; w = nn.Linear weight [128, k] (n must be 128,
    k a multiple of 16)."""
    L = _lib.lib()
    m = x.shape[0]
    y = out if out is not None else feat_empty(m, n, x.device)
    _lib.check(L.rdm_linear_layer_norm(x.data_ptr(), _ld(x), w.data_ptr(), _ld(w), _lib.ptr(bias), m, n, k, _lib.ptr(residual),
                                       _ld(residual) if residual is not None else 0, gamma.data_ptr(), beta.data_ptr(), eps,
                                       act, y.data_ptr(), _ld(y), _lib.stream_ptr()), 'rdm_linear_layer_norm')
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


def gather_rows(x, idx, out=None):
    """out[i] = x[idx[i]] bitwise (any 4-byte-multiple row type); out-of-range index -> zero row.
    x: [n, ...] with contiguous trailing dims (row stride may be padded), idx int64 [m]."""
    L = _lib.lib()
    n, m = x.shape[0], idx.shape[0]
    row_bytes = x[0].numel() * x.element_size() if n > 0 else 0
    tail = tuple(x.shape[1:])
    assert row_bytes % 4 == 0 and (x.stride(0) * x.element_size()) % 4 == 0
    if out is None:
        out = torch.empty((m,) + tail, dtype=x.dtype, device=x.device)
    _lib.check(L.rdm_gather_rows(x.data_ptr(), n, row_bytes // 4, x.stride(0) * x.element_size() // 4, idx.data_ptr(), m,
                                 out.data_ptr(), out.stride(0) * out.element_size() // 4, _lib.stream_ptr()),
               'rdm_gather_rows')
    return out


def rope(q, k, emb):
    L = _lib.lib()
    _lib.check(L.rdm_rope(q.data_ptr(), _ld(q), _lib.ptr(k), _ld(k) if k is not None else 0, emb.data_ptr(), _ld(emb),
                          q.shape[0], q.shape[1], _lib.stream_ptr()), 'rdm_rope')


def attention(q, k, v, heads, out=None, bf16=False):
    L = _lib.lib()
    nq, d = q.shape
    if out is None:
        out = feat_empty(nq, d, q.device)
    fn = L.rdm_attention_bf16 if bf16 else L.rdm_attention
    _lib.check(fn(q.data_ptr(), _ld(q), k.data_ptr(), _ld(k), v.data_ptr(), _ld(v), out.data_ptr(), _ld(out),
                               nq, k.shape[0], heads, d // heads, _lib.stream_ptr()), 'rdm_attention')
    return out


def attention_self_pair(q, k, v, n0, heads, out=None, bf16=False):
    """Self-attention of two stacked clouds (rows [0, n0) and [n0, n)) in one launch; same results as two attention calls."""
    L = _lib.lib()
    n, d = q.shape
    if out is None:
        out = feat_empty(n, d, q.device)
    _lib.check(L.rdm_attention_self_pair(q.data_ptr(), _ld(q), k.data_ptr(), _ld(k), v.data_ptr(), _ld(v), out.data_ptr(), _ld(out),
                                         n0, n - n0, heads, d // heads, int(bf16), _lib.stream_ptr()), 'rdm_attention_self_pair')
    return out


def vote_shift(xyz, offsets, limits):
    L = _lib.lib()
    out = torch.empty_like(xyz)
    _lib.check(L.rdm_vote_shift(xyz.data_ptr(), offsets.data_ptr(), _ld(offsets), xyz.shape[0], float(limits[0]),
                                float(limits[1]), float(limits[2]), out.data_ptr(), _lib.stream_ptr()), 'rdm_vote_shift')
    return out


def sigmoid_column(x_col):
    """clamp(sigmoid(x), 0, 1) of a (possibly strided) single column view [n, 1] or [n]."""
    L = _lib.lib()
    n = x_col.shape[0]
    out = torch.empty((max(n, 1),), dtype=torch.float32, device=x_col.device)
    _lib.check(L.rdm_sigmoid_column(x_col.data_ptr(), x_col.stride(0), n, out.data_ptr(), _lib.stream_ptr()),
               'rdm_sigmoid_column')
    return out[:n]


def l2_normalize(x):
    L = _lib.lib()
    y = feat_empty(x.shape[0], x.shape[1], x.device)
    _lib.check(L.rdm_l2_normalize(x.data_ptr(), _ld(x), x.shape[0], x.shape[1], y.data_ptr(), _ld(y), _lib.stream_ptr()),
               'rdm_l2_normalize')
    return y


def radius_search_device(q, s, q_lengths, s_lengths, radius, width, flags):
    """Truncated radius search entirely on the device.  flags: int32[2] = [max_count, status] (zeroed by
    the caller).  Returns idx [nq, width]; the effective width is min(width, flags[0]) (device value)."""
    L = _lib.lib()
    nq, ns, batch = q.shape[0], s.shape[0], q_lengths.shape[0]
    ws = scratch(q.device, L.rdm_radius_neighbors_workspace_bytes(nq, ns, batch))
    out = torch.empty((max(nq, 1), width), dtype=torch.int64, device=q.device)
    _lib.check(L.rdm_radius_neighbors(q.data_ptr(), nq, s.data_ptr(), ns, q_lengths.data_ptr(), s_lengths.data_ptr(), batch,
                                      float(radius), width, out.data_ptr(), 0, flags.data_ptr(), flags[1:].data_ptr(),
                                      ws.data_ptr(), ws.numel(), _lib.stream_ptr()), 'rdm_radius_neighbors')
    return out[:nq]


def radius_grid_records(points, lengths, radius):
    """The cell-sorted records {x, y, z, row (int bits)} of the search grid of `points` (stacked clouds, `lengths` int64 on
    the device) for `radius`: float32 [n, 4], in (cloud, cell, row) order -- a function of the points alone.  Used as the
    spatial query order of the KPConv kernels (rdm_kpconv_fused / rdm_kpconv_gather_ordered)."""
    L = _lib.lib()
    n = points.shape[0]
    nbytes = L.rdm_radius_grid_workspace_bytes(n)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)  # (its own allocation: the records are a view of it)
    lengths = lengths.to(device=points.device, dtype=torch.int64).contiguous()
    _lib.check(L.rdm_radius_grid_build(points.data_ptr(), n, lengths.data_ptr(), lengths.shape[0], float(radius), ws.data_ptr(),
                                       nbytes, _lib.stream_ptr()), 'rdm_radius_grid_build')
    rec = L.rdm_radius_grid_records(ws.data_ptr(), nbytes, n)
    off = rec - ws.data_ptr()
    return ws[off:off + 16 * max(n, 1)].view(torch.float32).reshape(-1, 4)[:n]


def grid_subsample_device(points, lengths, voxel, form=0):
    """-> (out_points [n,3] capacity buffer, out_lengths int64[batch]) without synchronising.  form: 0 = kernel form by size,
    1 = one workgroup per cloud, 2 = multi-launch (rdm_grid_subsample_form; same output)."""
    L = _lib.lib()
    n, batch = points.shape[0], lengths.shape[0]
    out = torch.empty((max(n, 1), 3), dtype=torch.float32, device=points.device)
    out_len = torch.empty((batch,), dtype=torch.int64, device=points.device)
    ws = scratch(points.device, L.rdm_grid_subsample_workspace_bytes(n, batch))
    _lib.check(L.rdm_grid_subsample_form(points.data_ptr(), n, lengths.data_ptr(), batch, float(voxel), out.data_ptr(),
                                         out_len.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(), int(form)),
               'rdm_grid_subsample')
    return out, out_len


def nms(idx, width_dev):
    L = _lib.lib()
    n = idx.shape[0]
    keep = torch.empty((max(n, 1),), dtype=torch.uint8, device=idx.device)
    _lib.check(L.rdm_nms(idx.data_ptr(), n, idx.shape[1], idx.stride(0), _lib.ptr(width_dev), keep.data_ptr(),
                         _lib.stream_ptr()), 'rdm_nms')
    return keep[:n]


def compact_indices(keep, begin, end, order, count):
    L = _lib.lib()
    _lib.check(L.rdm_compact_indices(keep.data_ptr(), begin, end, order.data_ptr(), count.data_ptr(), _lib.stream_ptr()),
               'rdm_compact_indices')


def point_to_node(points, nodes, k, status):
    L = _lib.lib()
    n, m = points.shape[0], nodes.shape[0]
    knn = torch.empty((m, k), dtype=torch.int64, device=points.device)
    kmask = torch.empty((m, k), dtype=torch.uint8, device=points.device)
    nmask = torch.empty((m,), dtype=torch.uint8, device=points.device)
    ws = scratch(points.device, L.rdm_point_to_node_workspace_bytes(n, m))
    _lib.check(L.rdm_point_to_node(points.data_ptr(), n, nodes.data_ptr(), m, k, knn.data_ptr(), kmask.data_ptr(),
                                   nmask.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
               'rdm_point_to_node')
    return nmask, knn, kmask


def point_to_node_pair(points_a, nodes_a, points_b, nodes_b, k, status):
    """point_to_node for the ref (a) and src (b) cloud with one set of launches -> ((nmask, knn, kmask) for a, for b)."""
    L = _lib.lib()
    dev = points_a.device
    outs = []
    for pts, nodes in ((points_a, nodes_a), (points_b, nodes_b)):
        m = nodes.shape[0]
        outs.append((torch.empty((m,), dtype=torch.uint8, device=dev), torch.empty((m, k), dtype=torch.int64, device=dev),
                     torch.empty((m, k), dtype=torch.uint8, device=dev)))
    na, ma, nb, mb = points_a.shape[0], nodes_a.shape[0], points_b.shape[0], nodes_b.shape[0]
    ws = scratch(dev, L.rdm_point_to_node_workspace_bytes(na, ma) + L.rdm_point_to_node_workspace_bytes(nb, mb))
    (nma, knna, kma), (nmb, knnb, kmb) = outs
    _lib.check(L.rdm_point_to_node_pair(points_a.data_ptr(), na, nodes_a.data_ptr(), ma, points_b.data_ptr(), nb, nodes_b.data_ptr(),
                                        mb, k, knna.data_ptr(), kma.data_ptr(), nma.data_ptr(), knnb.data_ptr(), kmb.data_ptr(),
                                        nmb.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
               'rdm_point_to_node_pair')
    return outs


def coarse_matching(scores, ref_mask, src_mask, k, dual=True):
    """scores [m, n] view (overwritten) -> (ref_idx i64[k], src_idx i64[k], scores f32[k], count i32[1])."""
    L = _lib.lib()
    m, n = scores.shape
    dev = scores.device
    ri = torch.empty((k,), dtype=torch.int64, device=dev)
    si = torch.empty((k,), dtype=torch.int64, device=dev)
    sc = torch.empty((k,), dtype=torch.float32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = scratch(dev, L.rdm_coarse_matching_workspace_bytes(m, n))
    _lib.check(L.rdm_coarse_matching(scores.data_ptr(), m, n, _ld(scores), ref_mask.data_ptr(), src_mask.data_ptr(),
                                     int(dual), k, ri.data_ptr(), si.data_ptr(), sc.data_ptr(), cnt.data_ptr(),
                                     ws.data_ptr(), ws.numel(), _lib.stream_ptr()), 'rdm_coarse_matching')
    return ri, si, sc, cnt


def coarse_matching_features(ref_feats, src_feats, ref_mask, src_mask, k, dual=True):
    """L2-normalised superpoint features [m, d], [n, d] (views) -> (ref_idx i64[k], src_idx i64[k], scores f32[k],
    count i32[1]); the stage is evaluated in fp64 (rdm_coarse_matching_features)."""
    L = _lib.lib()
    (m, d), n = ref_feats.shape, src_feats.shape[0]
    dev = ref_feats.device
    ri = torch.empty((k,), dtype=torch.int64, device=dev)
    si = torch.empty((k,), dtype=torch.int64, device=dev)
    sc = torch.empty((k,), dtype=torch.float32, device=dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = scratch(dev, L.rdm_coarse_matching_features_workspace_bytes(m, n))
    _lib.check(L.rdm_coarse_matching_features(ref_feats.data_ptr(), _ld(ref_feats), m, src_feats.data_ptr(), _ld(src_feats), n, d,
                                              ref_mask.data_ptr(), src_mask.data_ptr(), int(dual), k, ri.data_ptr(),
                                              si.data_ptr(), sc.data_ptr(), cnt.data_ptr(), ws.data_ptr(), ws.numel(),
                                              _lib.stream_ptr()), 'rdm_coarse_matching_features')
    return ri, si, sc, cnt


def sinkhorn(scores, row_mask, col_mask, alpha, iters):
    L = _lib.lib()
    b, m, n = scores.shape
    out = torch.empty((b, m + 1, n + 1), dtype=torch.float32, device=scores.device)
    _lib.check(L.rdm_sinkhorn(scores.data_ptr(), b, m, n, row_mask.data_ptr(), col_mask.data_ptr(), alpha.data_ptr(), iters,
                              out.data_ptr(), _lib.stream_ptr()), 'rdm_sinkhorn')
    return out


def lgr(log_scores, ref_pts, src_pts, ref_mask, src_mask, radius, min_corr, steps):
    """-> (ref_corr [cap,3], src_corr [cap,3], scores [cap], T [4,4], counts i32[3]) -- counts[0] rows are valid."""
    L = _lib.lib()
    b, side = ref_mask.shape
    dev = log_scores.device
    cap = b * 2 * side
    rc = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    sc = torch.empty((cap, 3), dtype=torch.float32, device=dev)
    cs = torch.empty((cap,), dtype=torch.float32, device=dev)
    T = torch.empty((4, 4), dtype=torch.float32, device=dev)
    counts = torch.empty((3,), dtype=torch.int32, device=dev)
    ws = scratch(dev, L.rdm_lgr_workspace_bytes(b))
    _lib.check(L.rdm_lgr(log_scores.data_ptr(), ref_pts.data_ptr(), src_pts.data_ptr(), ref_mask.data_ptr(),
                         src_mask.data_ptr(), b, side, float(radius), int(min_corr), int(steps), rc.data_ptr(), sc.data_ptr(),
                         cs.data_ptr(), T.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
               'rdm_lgr')
    return rc, sc, cs, T, counts


def voxel_downsample(points, voxel):
    """Raw-scan preprocessing (preporcess/downsample_pcd_kitti.py:21-36): points f32 [N, C>=3] on the GPU ->
    per-voxel means [M, C] (all columns averaged, e.g. xyz + intensity), voxels in first-occurrence order."""
    L = _lib.lib()
    assert points.is_cuda and points.dtype == torch.float32 and points.dim() == 2
    n, c = points.shape
    assert n == 0 or points.stride(1) == 1
    ld = points.stride(0) if n > 1 else c
    out = torch.empty((max(n, 1), c), dtype=torch.float32, device=points.device)
    flags = torch.zeros((2,), dtype=torch.int32, device=points.device)  # [count, status]
    ws = scratch(points.device, L.rdm_voxel_downsample_workspace_bytes(n))
    _lib.check(L.rdm_voxel_downsample(points.data_ptr(), n, ld, c, float(voxel), out.data_ptr(), c,
                                      flags.data_ptr(), flags[1:].data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
               'rdm_voxel_downsample')
    m, status = (int(x) for x in flags.cpu())  # the output size is data dependent: one read-back
    if status != 0:
        raise RuntimeError('rdm_voxel_downsample: non-finite point or extent beyond 2^21 voxels')
    return out[:m]


def ransac_correspondences(src_corr, ref_corr, distance_threshold=0.3, ransac_n=4, num_iterations=50000, seed=0,
                           return_hypotheses=False):
    """geotransformer/utils/open3d.py:173-203 on the GPU: -> (transform [4,4] f32 device, stats int32[2] device =
    {winning iteration, inliers}, inlier rmse f32[1] device[, per-iteration inlier counts])."""
    L = _lib.lib()
    assert src_corr.is_cuda and src_corr.dtype == torch.float32 and src_corr.is_contiguous() and ref_corr.is_contiguous()
    dev = src_corr.device
    T = torch.empty((4, 4), dtype=torch.float32, device=dev)
    stats = torch.empty((2,), dtype=torch.int32, device=dev)
    rmse = torch.empty((1,), dtype=torch.float32, device=dev)
    hyp = torch.empty((num_iterations,), dtype=torch.int32, device=dev) if return_hypotheses else None
    ws = scratch(dev, L.rdm_ransac_workspace_bytes(num_iterations))
    _lib.check(L.rdm_ransac_correspondences(src_corr.data_ptr(), ref_corr.data_ptr(), src_corr.shape[0], float(distance_threshold),
                                            ransac_n, num_iterations, int(seed), T.data_ptr(), stats.data_ptr(), rmse.data_ptr(),
                                            _lib.ptr(hyp), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
               'rdm_ransac_correspondences')
    return (T, stats, rmse, hyp) if return_hypotheses else (T, stats, rmse)
