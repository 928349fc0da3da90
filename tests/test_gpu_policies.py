"""GPU parity of the on-device rule-based policies (row f1): GreedyCameraAgent vs GreedyTargetAgent run
closed-loop on the device with the reference agents' recorded draws must reproduce the reference's joint
actions and the resulting environment trace (fixtures greedy_*.npz, recorded by tests/golden/make_golden.py agents)."""
import numpy as np
import pytest
import torch

import golden_util as G
import gpu_util as U

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['greedy_4v8-9_s5', 'greedy_8v8-9_s6', 'greedy_4v2-9_s7'])
def test_greedy_policies_closed_loop(name):
    fx = G.load(name + '.npz')
    N = 2
    eng = U.engine_from_fixture(fx, N, obs_dtype=torch.float64)
    eng.enable_policies()
    Nc, Nt = eng.num_cameras, eng.num_targets
    dev = eng.device
    # the agents first act on the reset observation: reproduce the reference's reset view masks
    # (see-through draws of reset() are not on tape; re-derive the masks from the recorded ones)
    tape0 = torch.from_numpy(np.where(fx['reset/camera_target_view_mask'], 1.0, 0.0)[None].repeat(N, 0)).to(dev)
    eng.observe(tape_ct=tape0)     # u = 1 sees through, u = 0 goes to the occlusion test
    m0 = eng.unpack_masks()
    assert np.array_equal(m0['camera_target_view_mask'][0], fx['reset/camera_target_view_mask'])

    def bc(a, dtype=np.float64):
        a = np.asarray(a)
        return torch.from_numpy(np.broadcast_to(a, (N,) + a.shape).astype(dtype).copy()).to(dev)

    T = len(fx['step/done'])
    for s in range(T):
        tape = {
            'camera_resample_u': bc(np.nan_to_num(fx['step/agent_cam_binom_u'][s], nan=0.0)),
            'camera_sample_u': bc(np.nan_to_num(fx['step/agent_cam_sample_u'][s], nan=0.0)),
            'camera_delay': bc(fx['step/agent_cam_delay'][s], np.int32),
            'target_choice_u': bc(np.nan_to_num(fx['step/agent_tgt_choice_u'][s], nan=0.0)),
            'target_resample_u': bc(np.nan_to_num(fx['step/agent_tgt_binom_u'][s], nan=0.0)),
            'target_sample_u': bc(np.nan_to_num(fx['step/agent_tgt_sample_u'][s], nan=0.0)),
            'target_reset_sample_u': bc(fx['agent/tgt_reset_sample_u']),
        }
        env_tape = bc(np.nan_to_num(fx['step/tape_ct'][s], nan=0.0))
        goal_tape = bc(np.nan_to_num(fx['step/goal_u'][s], nan=0.0))
        eng.step_greedy(policy_tape=tape, tape_ct=env_tape, tape_goal=goal_tape, auto_reset=False)
        cam_act, tgt_act = eng.policy_actions()
        for e in range(N):
            assert np.abs(tgt_act[e].cpu().numpy() - fx['step/tgt_act'][s]).max() < 1e-8, ('target action', s)
            if Nc:
                assert np.abs(cam_act[e].cpu().numpy() - fx['step/cam_act'][s]).max() < 1e-8, ('camera action', s)
        masks = eng.unpack_masks()
        assert np.array_equal(masks['camera_target_view_mask'][0], fx['step/camera_target_view_mask'][s]), s
        sd = eng.state_dict()
        assert np.abs(sd['tgt_x'][0] - fx['step/tgt_xy'][s][:, 0]).max() < 1e-8
        assert np.array_equal(sd['tgt_goals'][0], fx['step/tgt_goals'][s].astype(np.float64)), s
        assert np.array_equal(sd['bounties'][1], fx['step/bounties'][s].astype(np.float64)), s
        assert sd['episode_reward'][0] == fx['step/episode_reward'][s]


def test_greedy_rollout_on_philox_streams():
    """C3-style workload: 8v8 Greedy vs Greedy stays on the device; episodes finish and auto-reset."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    eng = Engine(read_config('MATE-8v8-9.yaml', max_episode_steps=400), 64, seed=4)
    eng.enable_policies()
    eng.reset()
    delivered = 0.0
    for _ in range(450):
        eng.step_greedy(auto_reset=True)
        delivered = max(delivered, float(eng.scalars[:, 6].max()))
    sd = eng.state_dict()
    assert delivered >= 4                      # greedy targets do deliver cargo
    assert (sd['episode'] >= 2).all()          # every environment finished an episode (time limit) and restarted
    assert float(eng.scalars[:, 3].mean()) > 0.2   # greedy cameras keep a sizeable coverage
