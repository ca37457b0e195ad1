"""Stack-mode collate on the GPU: the 5-level point pyramid and its 13 neighbour searches.

Mirrors `precompute_data_stack_mode` / `registration_collate_fn_stack_mode`
(geotransformer/utils/data.py:13-77, 139-192): same arguments, same dictionary keys, [ref, src]
stacking.  The reference runs this on the CPU inside DataLoader workers; here the two native
operators are the HIP kernels (rdmnet_amd.ext semantics) and everything stays on the device.
One host synchronisation reads the four subsampled sizes.
"""
import numpy as np
import torch

from . import ops


def precompute_data_stack_mode(points, lengths, num_stages, voxel_size, radius, neighbor_limits, exact_shapes=False):
    """points f32[N,3] (cuda), lengths i64[B] (cuda).  Returns the reference's dict plus `_widths`:
    device int32 scalars holding max neighbour counts, so kernels can apply the reference's
    `[:, :min(limit, max_count)]` without a host round trip.  exact_shapes=True slices the index
    tensors to that width (one more synchronisation), giving the reference's exact shapes."""
    assert num_stages == len(neighbor_limits)
    dev = points.device
    pts_list, len_list = [points], [lengths]
    caps = []
    for i in range(1, num_stages):
        voxel_size *= 2  # data.py:23-28: doubled before its first use
        p, l = ops.grid_subsample_device(pts_list[-1], len_list[-1], voxel_size)
        pts_list.append(p)
        len_list.append(l)
        caps.append(p)
    sizes = torch.stack(len_list[1:]).sum(1).cpu().tolist()  # the one sync of the collate
    for i in range(1, num_stages):
        pts_list[i] = caps[i - 1][:int(sizes[i - 1])]

    n_calls = 3 * num_stages - 2
    flags = torch.zeros((n_calls, 2), dtype=torch.int32, device=dev)  # per call: [max_count, status]
    neighbors, subsampling, upsampling, widths = [], [], [], {}
    call = 0
    for i in range(num_stages):
        cur_p, cur_l = pts_list[i], len_list[i]
        neighbors.append(ops.radius_search_device(cur_p, cur_p, cur_l, cur_l, radius, neighbor_limits[i], flags[call]))
        widths[('neighbors', i)] = flags[call]
        call += 1
        if i < num_stages - 1:
            sub_p, sub_l = pts_list[i + 1], len_list[i + 1]
            subsampling.append(ops.radius_search_device(sub_p, cur_p, sub_l, cur_l, radius, neighbor_limits[i], flags[call]))
            widths[('subsampling', i)] = flags[call]
            call += 1
            upsampling.append(ops.radius_search_device(cur_p, sub_p, cur_l, sub_l, radius * 2, neighbor_limits[i + 1],
                                                       flags[call]))
            widths[('upsampling', i)] = flags[call]
            call += 1
        radius *= 2
    out = {'points': pts_list, 'lengths': len_list, 'neighbors': neighbors, 'subsampling': subsampling,
           'upsampling': upsampling, '_widths': widths, '_flags': flags}
    if exact_shapes:
        fl = flags.cpu()
        if int(fl[:, 1].max()) != 0:
            raise RuntimeError('radius search: internal error (status word set)')
        call = 0
        for i in range(num_stages):
            neighbors[i] = neighbors[i][:, :min(neighbor_limits[i], int(fl[call, 0]))]
            call += 1
            if i < num_stages - 1:
                subsampling[i] = subsampling[i][:, :min(neighbor_limits[i], int(fl[call, 0]))]
                upsampling[i] = upsampling[i][:, :min(neighbor_limits[i + 1], int(fl[call + 1, 0]))]
                call += 2
    return out


def registration_collate_fn_stack_mode(data_dicts, num_stages, voxel_size, search_radius, neighbor_limits,
                                       precompute_data=True, device=None, exact_shapes=False, engine=None):
    """geotransformer/utils/data.py:139-192.  data_dicts: list (batch size 1 for this model) of dicts with
    'ref_points', 'src_points', 'ref_feats', 'src_feats' (numpy or tensors) + passthrough keys.

    engine (keyword only in spirit, not in the reference): an rdmnet_amd.engine.Engine built for the same configuration
    (e.g. `model.engine()`).  The pyramid and its searches then run as ONE native call (rdm_engine_collate) instead of 17
    launches issued from Python, with bit-identical tables; voxel size, radius and limits are the engine's (they must equal
    the arguments)."""
    device = device or torch.device('cuda', torch.cuda.current_device())
    batch_size = len(data_dicts)
    if engine is not None and precompute_data and batch_size == 1:
        c = engine.cfg
        if (num_stages != c.backbone.num_stages or abs(voxel_size - c.backbone.init_voxel_size) > 1e-9 or
                abs(search_radius - c.backbone.init_radius) > 1e-9 or list(neighbor_limits) != list(c.neighbor_limits)):
            raise ValueError('the engine was built for another pyramid configuration')
        d = data_dicts[0]
        ref = torch.as_tensor(d['ref_points']).to(device=device, dtype=torch.float32).contiguous()
        src = torch.as_tensor(d['src_points']).to(device=device, dtype=torch.float32).contiguous()
        collated = {k: v for k, v in d.items() if k not in ('ref_points', 'src_points', 'ref_feats', 'src_feats')}
        collated.update(engine.collate(ref, src))
        collated['features'] = torch.cat([torch.as_tensor(d['ref_feats']), torch.as_tensor(d['src_feats'])], 0).to(
            device=device, dtype=torch.float32)
        if exact_shapes:
            fl = collated['_flags'].cpu()
            if int(fl[:, 1].max()) != 0:
                raise RuntimeError('radius search: internal error (status word set)')
            for key in ('neighbors', 'subsampling', 'upsampling'):
                collated[key] = [t[:, :min(t.shape[1], int(collated['_widths'][(key, i)][0]))] for i, t in enumerate(collated[key])]
        return collated
    collated = {}
    for d in data_dicts:
        for k, v in d.items():
            if isinstance(v, np.ndarray):
                v = torch.from_numpy(v)
            collated.setdefault(k, []).append(v)
    ref_f, src_f = collated.pop('ref_feats'), collated.pop('src_feats')
    feats = torch.cat([torch.as_tensor(x) for x in ref_f + src_f], 0).float().to(device)
    pts_list = [torch.as_tensor(x).float() for x in collated.pop('ref_points') + collated.pop('src_points')]
    lengths = torch.tensor([p.shape[0] for p in pts_list], dtype=torch.int64, device=device)
    points = torch.cat(pts_list, 0).to(device).contiguous()
    if batch_size == 1:
        collated = {k: v[0] for k, v in collated.items()}
    collated['features'] = feats
    if precompute_data:
        collated.update(precompute_data_stack_mode(points, lengths, num_stages, voxel_size, search_radius,
                                                   neighbor_limits, exact_shapes=exact_shapes))
    else:
        collated['points'], collated['lengths'] = points, lengths
    collated['batch_size'] = batch_size
    return collated


def registration_collate_lockstep(items, num_stages, voxel_size, search_radius, neighbor_limits, engines, device=None):
    """`registration_collate_fn_stack_mode([item], ..., engine=e)` for every item of `items` (one pair each) as ONE lock-step
    group on `engines` (e.g. `model.engine_group(len(items))`; round 6): -> the list of collated data_dicts, each equal to the
    one-pair call's.  What the reference's DataLoader workers do pair by pair on the CPU; `model(list_of_dicts)` runs the forwards
    the same way."""
    device = device or torch.device('cuda', torch.cuda.current_device())
    c = engines[0].cfg
    if (num_stages != c.backbone.num_stages or abs(voxel_size - c.backbone.init_voxel_size) > 1e-9 or
            abs(search_radius - c.backbone.init_radius) > 1e-9 or list(neighbor_limits) != list(c.neighbor_limits)):
        raise ValueError('the engines were built for another pyramid configuration')
    pairs = [(torch.as_tensor(d['ref_points']).to(device=device, dtype=torch.float32).contiguous(),
              torch.as_tensor(d['src_points']).to(device=device, dtype=torch.float32).contiguous()) for d in items]
    out = []
    for d, tables in zip(items, type(engines[0]).collate_lockstep(engines, pairs)):
        collated = {k: v for k, v in d.items() if k not in ('ref_points', 'src_points', 'ref_feats', 'src_feats')}
        collated.update(tables)
        collated['features'] = torch.cat([torch.as_tensor(d['ref_feats']), torch.as_tensor(d['src_feats'])], 0).to(device=device, dtype=torch.float32)
        out.append(collated)
    return out


def collate_pair(ref_points, src_points, cfg, device=None, exact_shapes=False):
    """Convenience: two f32 [N,3] clouds -> data_dict ready for RDMNet.forward."""
    item = {'ref_points': ref_points, 'src_points': src_points,
            'ref_feats': np.ones((ref_points.shape[0], 1), np.float32),
            'src_feats': np.ones((src_points.shape[0], 1), np.float32)}
    d = registration_collate_fn_stack_mode([item], cfg.backbone.num_stages, cfg.backbone.init_voxel_size,
                                           cfg.backbone.init_radius, cfg.neighbor_limits, device=device,
                                           exact_shapes=exact_shapes)
    d['testing'] = True
    return d
