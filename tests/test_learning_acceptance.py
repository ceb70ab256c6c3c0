"""Learning acceptance: agents trained HERE -- VecAoEnv.step through
aomarl_env_step, actors through aomarl_actor_forward, 2000 aomarl_sac_update calls per episode -- beat the
integrator on fresh atmosphere seeds, in the summed per-agent reward AND in long-exposure Strehl, which is what the
reference's training loop reports (train_rpc.py:452-501, 555-631).  BASELINE configs[1]: production_sh_10x10_2m,
64 environments, 2 agents (80 Btt modes + tip-tilt).  The full 61-episode log: profiles/r03_learning_acceptance.txt
(reward -607 vs -967, LE Strehl 0.900 vs 0.864); here 21 episodes (~25 s)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_trained_agents_beat_the_integrator():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import learning_acceptance as LA
    rc = LA.main(["--episodes", "21", "--test-every", "10", "--updates", "2000"])
    ev = LA.main.evals
    assert [e["episode"] for e in ev] == [0, 10, 20]
    last = ev[-1]
    # measured at episode 20: reward -630 vs -943, LE Strehl 0.893 vs 0.864
    assert last["test_r_rl"] > last["test_r_integrator"] + 100, last
    assert last["test_sr_le_rl"] > last["test_sr_le_integrator"] + 0.01, last
    assert all(rl > it for rl, it in zip(last["test_r_agents_rl"], last["test_r_agents_integrator"])), last   # both agents
    assert rc == 0
