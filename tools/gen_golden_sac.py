#!/usr/bin/env python3
"""Golden vectors for the SAC update, produced by running the reference's own
`SAC.update_critic / update_actor / update_alpha` + `soft_update` (train_rpc.py:1016-1094,
algorithms_rpc/utils.py) on the reference's `GaussianPolicy` / `QNetwork` modules (build container only).

`SAC.__init__` needs a live torch-RPC worker, so the object is created with `object.__new__(SAC)` and
given exactly the attributes `initialise_critic / initialise_policy / initialise_alpha` would set;
every update line that runs is the reference's.  The standard-normal draws of `Normal.rsample` are
recorded so the batched implementation can be fed the same noise.

Writes tests/golden/host_sac_update.pt: per agent {before, batch, eps per update, after each update,
losses} for two agents of different sizes (the 10x10 layout: 320 -> 80 and 8 -> 2) and 3 updates.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import _ref_shims  # noqa: E402
_ref_shims.install()

import torch  # noqa: E402
from torch.optim import Adam  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    from src.reinforcement_learning.rpc_training.train_rpc import SAC
    from src.reinforcement_learning.rpc_training.algorithms_rpc.model_rpc import GaussianPolicy, QNetwork
    from src.reinforcement_learning.rpc_training.algorithms_rpc.utils import soft_update, hard_update
    import torch.distributions.normal as tdn

    recorded = []
    orig = tdn._standard_normal

    def rec(shape, dtype, device):
        e = orig(shape, dtype, device)
        recorded.append(e.clone())
        return e

    tdn._standard_normal = rec
    hidden, batch, lr, gamma, tau = 48, 24, 3e-4, 0.1, 0.005
    out = {"hyper": dict(hidden=hidden, batch=batch, lr=lr, gamma=gamma, tau=tau, updates=3),
           "agents": []}
    for k, (nin, nact) in enumerate([(320, 80), (8, 2)]):
        torch.manual_seed(100 + k)
        sac = object.__new__(SAC)
        sac.device = torch.device("cpu")
        sac.gamma, sac.tau, sac.alpha = gamma, tau, 0.2
        sac.automatic_entropy_tuning = True
        sac.target_update_interval = 1
        # initialise_critic / initialise_policy / initialise_alpha (train_rpc.py:856-912)
        sac.critic = QNetwork(nin, nact, [hidden], 2)
        sac.critic_target = QNetwork(nin, nact, [hidden], 2)
        sac.critic_optim = Adam(sac.critic.parameters(), lr=lr)
        hard_update(sac.critic_target, sac.critic)
        sac.policy = GaussianPolicy(num_inputs=nin, num_actions=nact, hidden_dim=hidden,
                                    action_scale=1.0, action_bias=0.0, num_layers=2,
                                    initialize_last_layer_zero=False,
                                    initialize_last_layer_near_zero=False, activation="relu",
                                    LOG_SIG_MAX=2.0)
        sac.policy_optim = Adam(sac.policy.parameters(), lr=lr)
        sac.target_entropy = -float(nact)
        sac.log_alpha = torch.zeros(1, requires_grad=True)
        sac.alpha_optim = Adam([sac.log_alpha], lr=lr)
        # give the biases something to do
        with torch.no_grad():
            for p in list(sac.policy.parameters()) + list(sac.critic.parameters()):
                if p.dim() == 1:
                    p.normal_(0, 0.05)
        hard_update(sac.critic_target, sac.critic)
        ag = {"nin": nin, "nact": nact,
              "policy0": {n: p.detach().clone() for n, p in sac.policy.state_dict().items()},
              "critic0": {n: p.detach().clone() for n, p in sac.critic.state_dict().items()},
              "updates": []}
        for u in range(3):
            s = torch.randn(batch, nin)
            a = torch.rand(batch, nact) * 2 - 1
            r = -torch.rand(batch, 1)
            s2 = torch.randn(batch, nin)
            mask = torch.ones(batch, 1)
            recorded.clear()
            q1l, q2l = SAC.update_critic(sac, s, a, r, s2, mask)           # train_rpc.py:1016-1038
            log_pi, pl = SAC.update_actor(sac, state_batch=s)              # :1052-1064
            al, alpha_t = SAC.update_alpha(sac, log_pi)                    # :1070-1084
            soft_update(sac.critic_target, sac.critic, sac.tau)            # :1128-1129
            assert len(recorded) == 2
            ag["updates"].append({
                "s": s, "a": a, "r": r, "s2": s2, "mask": mask,
                "eps_next": recorded[0].clone(), "eps_pi": recorded[1].clone(),
                "q1_loss": q1l, "q2_loss": q2l, "policy_loss": pl, "alpha_loss": al,
                "alpha": float(sac.alpha.item()), "log_alpha": sac.log_alpha.detach().clone(),
                "policy": {n: p.detach().clone() for n, p in sac.policy.state_dict().items()},
                "critic": {n: p.detach().clone() for n, p in sac.critic.state_dict().items()},
                "critic_target": {n: p.detach().clone() for n, p in sac.critic_target.state_dict().items()}})
        out["agents"].append(ag)
        print("agent", k, "losses", [(round(x["q1_loss"], 5), round(x["policy_loss"], 5)) for x in ag["updates"]])
    tdn._standard_normal = orig
    torch.save(out, os.path.join(OUT, "host_sac_update.pt"))
    print("wrote", os.path.join(OUT, "host_sac_update.pt"), os.path.getsize(os.path.join(OUT, "host_sac_update.pt")))


if __name__ == "__main__":
    main()
