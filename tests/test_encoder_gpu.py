"""GPU parity of the KPConv encoder/decoder (HIP) against the oracle on the golden 'small' pair."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def setup(oracle_native, golden_dir):
    from oracle import forward as ofw
    from rdmnet_amd import config, model, weights
    cfg = config.make_cfg()
    state = weights.synthetic_state_dict(cfg, seed=0)
    W = ofw.to_torch(state)
    net = model.create_model(cfg).cuda()
    net.load_state_dict(state)
    net._prepare()
    g = np.load(os.path.join(golden_dir, 'forward_small.npz'))
    rp, sp = g['ref_points_in'], g['src_points_in']
    data = ofw.pyramid(np.concatenate([rp, sp]), np.array([len(rp), len(sp)], np.int64), cfg)
    return ofw, cfg, W, net, data


def to_cuda(data):
    out = {}
    for k, v in data.items():
        if isinstance(v, list):
            out[k] = [t.cuda().contiguous() for t in v]
        elif isinstance(v, torch.Tensor):
            out[k] = v.cuda()
        else:
            out[k] = v
    return out


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-30)).item()


def test_encoder_blocks_match_oracle(setup):
    ofw, cfg, W, net, data = setup
    otaps, gtaps = {}, {}
    ofeats = ofw.encoder(W, cfg, data, otaps)
    gfeats = net.run_encoder(to_cuda(data), gtaps)
    for name in otaps:
        err = rel_err(gtaps[name].cpu(), otaps[name])
        assert err <= 5e-5, (name, err)  # fp32, different summation orders; grows slowly with depth
    # decoder: the forward feeds a 257-channel coarse tensor (256 features + n2p logit); use a zero logit here
    from rdmnet_amd import ops
    of = list(ofeats)
    of[4] = torch.cat([ofeats[4][:, :256], torch.zeros(ofeats[4].shape[0], 1)], 1)
    gf = list(gfeats)
    buf = ops.feat_empty(gfeats[4].shape[0], 257, 'cuda')
    buf.copy_(torch.cat([gfeats[4][:, :256], torch.zeros(gfeats[4].shape[0], 1, device='cuda')], 1))
    gf[4] = buf
    dec_o = ofw.decoder(W, cfg, of, data)
    dec_g = net.run_decoder(gf, to_cuda(data))
    assert rel_err(dec_g.cpu(), dec_o) <= 5e-5
