"""Configuration of the inference path.  Mirrors the constants the reference reads from its
EasyDict (experiments/config.py:84-161); only keys used by inference are kept."""


class Cfg(dict):
    """Attribute-style dict (the reference uses easydict.EasyDict the same way)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def make_cfg():
    c = Cfg()
    c.backbone = Cfg(num_stages=5, init_voxel_size=0.3, kernel_size=15, base_radius=4.25, base_sigma=2.0,
                     group_norm=32, input_dim=1, init_dim=64, output_dim=256)
    c.backbone.init_radius = c.backbone.base_radius * c.backbone.init_voxel_size  # config.py:91
    c.backbone.init_sigma = c.backbone.base_sigma * c.backbone.init_voxel_size    # config.py:92
    c.model = Cfg(num_points_in_patch=128, num_sinkhorn_iterations=100, n2p_score_threshold=0.1,
                  p2p_score_threshold=0.1)
    c.coarse_matching = Cfg(num_correspondences=256, dual_normalization=True)
    c.thdroformer = Cfg(input_dim=2048, hidden_dim=128, output_dim=256, num_heads=4, num_layers=4,
                        input_dim2=256, num_layers2=4, k2=None,
                        attention_bf16=False)  # True: BASELINE.json configs[3] (bf16 QK^T / PV, fp32 softmax)
    c.Vote = Cfg(model_use_vote=True, inference_use_vote=True, MAX_TRANSLATE_RANGE=[3.0, 3.0, 3.0],
                 MLPS=[512, 256], NMS_radius=2.4)
    c.fine_matching = Cfg(acceptance_radius=0.6, mutual=False, topk=1, confidence_threshold=0,
                          use_dustbin=True, use_global_score=False, correspondence_threshold=3,
                          correspondence_limit=None, num_refinement_steps=5)
    c.test = Cfg(vis=False)
    c.neighbor_limits = [65, 63, 69, 70, 81]  # calibrated on the bundled pairs (utils/data.py:195-220)
    return c
