"""ColorJitterPoints (reference pyrl/utils/augmentations/pcd_aug.py:269-303 = torchvision ColorJitter on uint8 rgb): the
stand-alone kernels and the jitter fused into the encoder's rgb load against oracle/color_jitter_ref.py, the CPU restatement
of torchvision 0.14.1's arithmetic (torchvision itself is not installable here: parity against it is unpinned, see the
oracle's header).  uint8 in, uint8 out: the comparison is bit-exact."""
import itertools

import numpy as np
import pytest
import torch

from helpers import make_encoder_weights, make_obs

pytestmark = pytest.mark.gpu

DRAWS = [
    ([0, 1, 2, 3], [1.23, 0.71, 1.37, 0.21]),
    ([3, 2, 1, 0], [0.64, 1.39, 0.62, -0.43]),
    ([1, 3, 0, 2], [1.0, 1.0, 1.0, 0.0]),                 # identity factors
    ([2, 0, 3, 1], [1.4, None, 0.6, 0.5]),                # contrast disabled (empty range)
    ([1, 0, 2, 3], [None, 0.9, None, None]),              # contrast only
    ([3, 0, 1, 2], [None, None, None, -0.5]),             # hue only, extreme shift
]


def rgb_batch(B, N, seed):
    g = np.random.RandomState(seed)
    rgb = g.randint(0, 256, (B, 3, N)).astype(np.uint8)
    rgb[0, :, :8] = np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [7, 7, 7], [255, 255, 0], [1, 0, 255]]).T
    return rgb


@pytest.mark.parametrize("order,factors", DRAWS)
def test_color_jitter_kernels_match_the_torchvision_restatement(cuda, order, factors):
    from oracle import color_jitter_ref
    from pointcloud_rl_amd import hip
    rgb = rgb_batch(5, 777, seed=sum(order) + len(order))
    want = color_jitter_ref.color_jitter(torch.from_numpy(rgb), order, factors).numpy()
    dev = torch.from_numpy(rgb).to(cuda)
    color = dict(order=order, factors=factors)
    color["mean"] = hip.color_contrast_mean(dev, color)
    got = hip.color_jitter_u8(dev, color).cpu().numpy()
    assert got.dtype == np.uint8 and np.array_equal(got, want), (np.abs(got.astype(int) - want.astype(int)).max(), (got != want).mean())


def test_every_order_of_the_four_steps(cuda):
    from oracle import color_jitter_ref
    from pointcloud_rl_amd import hip
    rgb = rgb_batch(2, 300, seed=3)
    dev = torch.from_numpy(rgb).to(cuda)
    fac = [0.8, 1.3, 0.7, 0.17]
    for order in itertools.permutations(range(4)):
        color = dict(order=list(order), factors=fac)
        color["mean"] = hip.color_contrast_mean(dev, color)
        got = hip.color_jitter_u8(dev, color).cpu().numpy()
        want = color_jitter_ref.color_jitter(torch.from_numpy(rgb), list(order), fac).numpy()
        assert np.array_equal(got, want), order


def test_jitter_fused_into_the_encoder_load_equals_the_materialised_tensor(cuda):
    """The encoder reading rgb through the fused colour jitter gives the bits of the encoder reading the jittered tensor,
    forward and backward, also through a virtual repeat (DrQ)."""
    from pointcloud_rl_amd import hip
    from pointcloud_rl_amd.networks.pointnet import AugmentedObs
    B, N = 4, 200
    obs_np = make_obs(B, N, seed=8)
    w = {k: torch.from_numpy(v).to(cuda) for k, v in make_encoder_weights(6, 64, 128, 256, seed=2).items()}
    ew, _ = hip.make_encoder_weights(w["w0"], w["b0"], w["w1"], w["g1"], w["be1"], w["w2"], w["g2"], w["be2"], 1e-6)
    packed = torch.empty(hip.encoder_packed_bytes(ew.c_in, ew.c1, ew.c2, ew.c3) // 4, dtype=torch.float32, device=cuda)
    hip.encoder_pack_weights(ew, packed)
    obs = {k: torch.from_numpy(v).to(cuda) for k, v in obs_np.items()}
    color = dict(order=[2, 1, 3, 0], factors=[1.2, 0.75, 1.3, -0.3])
    color["mean"] = hip.color_contrast_mean(obs["rgb"], color)
    gp = torch.randn(2 * B, 256, device=cuda)

    def run(o, aug):
        desc, keep = hip.make_cloud_desc(o)
        pooled, argmax = hip.encoder_fwd(desc, ew, packed, aug=aug)
        return pooled, argmax, hip.encoder_bwd(desc, ew, packed, argmax, gp, aug=aug, pooled=pooled)

    fused_obs = AugmentedObs(obs)
    fused_obs.repeat = 2
    fused = run(fused_obs, hip.make_aug_desc(color=color))
    mat = dict(obs, rgb=hip.color_jitter_u8(obs["rgb"], color))
    mat = AugmentedObs(mat)
    mat.repeat = 2
    plain = run(mat, None)
    for a, b in zip(fused, plain):
        assert torch.equal(a, b)
    unjittered = run(fused_obs, None)
    assert not torch.equal(unjittered[0], fused[0])


def test_drq_agent_with_color_jitter_points(cuda):
    """configs/mfrl/drq/dm_control/pn_colorjitter.py: the agent draws like torchvision (same torch calls, same order), trains,
    and declines hipGraph replay (the draw is a host value)."""
    from pointcloud_rl_amd import configs
    from pointcloud_rl_amd.methods import build_agent
    from pointcloud_rl_amd.synthetic import SyntheticReplay
    from oracle import color_jitter_ref
    B, N, A = 8, 96, 6
    aug = dict(type="ColorJitterPoints", main_key="rgb", req_keys=["rgb"], brightness=0.4, contrast=0.4, saturation=0.4, hue=0.5)
    cfg = configs.drq_dmc(6, A, B, head_hidden=64, obs_aug=aug)
    cfg["env_params"] = configs.env_params({"xyz": [3, N], "rgb": [3, N]}, A)
    torch.manual_seed(0)
    agent = build_agent(cfg).to(cuda)
    cj = agent.obs_aug[0]
    torch.manual_seed(42)
    mine = cj.draw()
    torch.manual_seed(42)
    theirs = color_jitter_ref.draw_params(0.4, 0.4, 0.4, 0.5)
    assert mine == theirs
    mem = SyntheticReplay(B, N, A, seed=2, device=cuda)
    with pytest.warns(UserWarning):
        agent.enable_graphs()
    rets = [agent.update_parameters(mem, u) for u in range(1, 5)]
    assert all(np.isfinite(list(r.values())).all() for r in rets) and not getattr(agent, "_graphs", None)
