"""Generate tests/golden/ref_sac_dmc_small.ckpt by RUNNING THE REFERENCE ITSELF (build container only): a reference SAC agent
(configs/mfrl/sac/dm_control/pn.py with 64-wide heads) after two reference update steps, written by the reference's own
`save_checkpoint` (pyrl/utils/torch/checkpoint_utils.py:240-269).  The file is data (weights, Adam moments, meta); the parity
test loads it with pointcloud_rl_amd.utils.checkpoint and compares tensors.

    python tools/gen_golden_checkpoint.py
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402  (installs the import stubs and puts /root/reference on sys.path)
from gen_golden import REF, make_obs  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "ref_sac_dmc_small.ckpt")

if __name__ == "__main__":
    from pyrl.utils.data import DictArray
    from pyrl.utils.torch import save_checkpoint
    torch.set_num_threads(4)
    B, N, A = 8, 64, 6
    overrides = {"agent_cfg.actor_cfg.nn_cfg.mlp_cfg.mlp_spec": [50, 64, 64, "action_shape * 2"],
                 "agent_cfg.critic_cfg.nn_cfg.mlp_cfg.mlp_spec": ["50 + action_shape", 64, 64, 1]}
    agent, cfg = ref_stubs.build_reference_agent(f"{REF}/configs/mfrl/sac/dm_control/pn.py", {"xyz": [3, N], "rgb": [3, N]}, A, overrides, seed=5)
    agent.batch_size = B
    g = np.random.RandomState(77)
    for u in (1, 2):
        batch = dict(obs=make_obs(g, B, N), next_obs=make_obs(g, B, N), actions=g.uniform(-1, 1, (B, A)).astype(np.float32),
                     prev_actions=g.uniform(-1, 1, (B, A)).astype(np.float32), rewards=g.randn(B, 1).astype(np.float32),
                     dones=(g.rand(B, 1) < 0.25), episode_dones=(g.rand(B, 1) < 0.25))

        class Mem:
            def sample(self, bs):
                return DictArray(copy.deepcopy(batch))
        agent.update_parameters(Mem(), u)
    save_checkpoint(agent, OUT, meta=dict(updates=2))
    # acting path of the same agent (BaseAgent.forward, module_utils.py:147-159): deterministic modes on a small observation
    agent.eval()
    obs = make_obs(g, 3, N)
    acting = {f"obs/{k}": v for k, v in obs.items()}
    with torch.no_grad():
        acting["eval"] = agent(copy.deepcopy(obs), mode="eval").numpy()
        acting["mean"] = agent(copy.deepcopy(obs), mode="mean").numpy()
        acts, states = agent(copy.deepcopy(obs), mode="eval", rnn_mode="with_states")
        assert states is None
    np.savez_compressed(OUT.replace(".ckpt", "_acting.npz"), **acting)
    ck = torch.load(OUT, weights_only=False)
    print(OUT, f"{os.path.getsize(OUT) / 1e6:.2f} MB", sorted(k for k in ck["state_dict"] if not torch.is_tensor(ck["state_dict"][k])))
