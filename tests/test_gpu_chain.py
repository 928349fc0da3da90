"""The wrapper chain every example trainer builds for a camera learner, end to end against the reference's own wrappers:

    mate.make -> DiscreteCamera(5) -> MultiCamera(GreedyTargetAgent(seed=0)) -> RelativeCoordinates -> RescaledObservation ->
    RepeatedRewardIndividualDone -> AuxiliaryCameraRewards({'coverage_rate': 1.0}, 'mean') -> FrameSkip(5)
    (examples/ippo/camera/config.py:19-51; FrameSkip: examples/utils/wrappers.py:301-323)

tests/golden/chain_4v8-9_s15.npz (make_golden.py chain) holds, per frame, what that chain consumed (the learner's grid indices, every
random draw of the environment and of the greedy opponents) and produced (the camera team's observations, the shaped rewards, dones,
masks, state).  Each piece has a fixture of its own (xform_*, discrete_*, softcov_*, greedy_*); this is the composite: the device
flow -- set_action_grids + set_obs_transform + step_versus_greedy on grid indices + auxiliary_camera_rewards, frames summed as
FrameSkip sums them -- must reproduce the chain: observations 1e-5, rewards 1e-6, masks and integers exact."""
import numpy as np
import pytest
import torch

import golden_util as G
import gpu_util as U

pytestmark = pytest.mark.gpu


def _tapes(fx, s, N, dev, team):
    def bc(a, dtype=np.float64):
        a = np.asarray(a)
        return torch.from_numpy(np.broadcast_to(a, (N,) + a.shape).astype(dtype).copy()).to(dev)

    def draws(key, shape):             # the opponents' recorded draws; the learner's own team has no agents: zeros
        return bc(np.nan_to_num(fx['step/' + key][s], nan=0.0)) if 'step/' + key in fx else bc(np.zeros(shape))
    Nc, Nt = int(fx['num_cameras']), int(fx['num_targets'])
    policy = {
        'camera_resample_u': draws('agent_cam_binom_u', Nc), 'camera_sample_u': draws('agent_cam_sample_u', (Nc, 2)),
        'camera_delay': bc(fx['step/agent_cam_delay'][s] if 'step/agent_cam_delay' in fx else np.full((Nc, Nc), -1), np.int32),
        'target_choice_u': draws('agent_tgt_choice_u', Nt), 'target_resample_u': draws('agent_tgt_binom_u', Nt),
        'target_sample_u': draws('agent_tgt_sample_u', (Nt, 2)), 'target_reset_sample_u': bc(fx['agent/tgt_reset_sample_u']),
    }
    idx = fx['step/cam_idx' if team == 'camera' else 'step/tgt_idx'][s]
    return policy, bc(np.nan_to_num(fx['step/tape_ct'][s], nan=0.0)), bc(np.nan_to_num(fx['step/goal_u'][s], nan=0.0)), bc(idx, np.int32)


@pytest.mark.parametrize('obs_dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('name', ['chain_4v8-9_s15', 'chain_target_2v4-0_s16'])
def test_training_chain_of_the_example_trainers(name, obs_dtype):
    """chain_4v8-9_s15: the camera learner's chain (module docstring).  chain_target_2v4-0_s16: the target learner's
    (examples/ippo/target/config.py:20-51: DiscreteTarget(5) -> MultiTarget(GreedyCameraAgent(seed=0)) -> RelativeCoordinates ->
    RescaledObservation -> RepeatedRewardIndividualDone -> AuxiliaryTargetRewards(five terms, 'none') -> FrameSkip(10) on
    MATE-2v4-0, a learner that mostly heads for its goals: seven deliveries in 240 frames)."""
    from mate_amd.environment import BatchedMultiAgentTracking
    fx = G.load(name + '.npz')
    team = str(fx['learner_team'])
    me, opp = ('cam', 'tgt') if team == 'camera' else ('tgt', 'cam')
    N = 2
    levels = int(fx['discrete_levels'])
    env = BatchedMultiAgentTracking(U.config_of_fixture(fx), num_envs=N, obs_dtype=obs_dtype, auto_reset=False, relative_coordinates=True,
                                    rescaled_observation=True, **{f'discrete_{team}_levels': levels})
    eng = U.load_fixture_state(env.engine, fx)
    env.enable_greedy_policies()
    # DiscreteCamera's / DiscreteTarget's grid, by the reference's formula
    np.testing.assert_allclose(getattr(eng, team + '_action_grid'), fx[team + '_action_grid'], rtol=0, atol=1e-15)
    dev = eng.device
    mine = (lambda: eng.camera_obs) if team == 'camera' else (lambda: eng.target_obs)
    # the agents first act on the reset observation: the reference's reset view (its see-through draws are not on tape)
    tape0 = torch.from_numpy(np.where(fx['reset/camera_target_view_mask'], 1.0, 0.0)[None].repeat(N, 0)).to(dev)
    first = eng.observe(tape_ct=tape0)[0 if team == 'camera' else 1]
    assert np.array_equal(eng.unpack_masks()['camera_target_view_mask'][0], fx['reset/camera_target_view_mask'])
    tol = 1e-5 if obs_dtype == torch.float32 else 1e-9
    assert np.abs(first[1].double().cpu().numpy() - fx[f'reset/chain_{me}_obs']).max() <= tol
    keys, coef, reduction = [str(k) for k in fx['aux_keys']], fx['aux_coefficients'], str(fx['aux_reduction'])
    coefficients = dict(zip(keys, (float(c) for c in coef)))
    shaper = env.auxiliary_camera_rewards if team == 'camera' else env.auxiliary_target_rewards
    T = len(fx['step/done'])
    shaped_frames = []
    for s in range(T):
        policy, tape_ct, tape_goal, idx = _tapes(fx, s, N, dev, team)
        eng.step_versus_greedy(team, idx, policy_tape=policy, tape_ct=tape_ct, tape_goal=tape_goal, auto_reset=False)
        theirs = eng.policy_actions()[1 if team == 'camera' else 0]
        # the opponents' joint action is the chain's, and the decoded learner action moved its team as the Discrete wrapper's did
        assert np.abs(theirs[0].cpu().numpy() - fx[f'step/{opp}_act'][s]).max() < 1e-8, s
        got = mine()[1].double().cpu().numpy()
        assert np.abs(got - fx[f'step/chain_{me}_obs'][s]).max() <= tol, (s, np.abs(got - fx[f'step/chain_{me}_obs'][s]).max())
        masks = eng.unpack_masks()
        for mask in ('camera_target_view_mask', 'target_camera_view_mask', 'target_obstacle_view_mask', 'target_target_view_mask'):
            assert np.array_equal(masks[mask][0], fx['step/' + mask][s]), (mask, s)
        sd = eng.state_dict()
        assert np.abs(sd['cam_phi'][1] - fx['step/cam_phi'][s]).max() < 1e-9 and np.abs(sd['cam_theta'][0] - fx['step/cam_theta'][s]).max() < 1e-9, s
        assert np.abs(sd['tgt_x'][1] - fx['step/tgt_xy'][s][:, 0]).max() < 1e-9 and np.abs(sd['tgt_y'][0] - fx['step/tgt_xy'][s][:, 1]).max() < 1e-9, s
        assert np.array_equal(sd['tgt_goals'][0], fx['step/tgt_goals'][s].astype(np.float64)), s
        assert np.array_equal(sd['bounties'][1], fx['step/bounties'][s].astype(np.float64)), s
        assert sd['episode_reward'][0] == fx['step/episode_reward'][s]
        shaped = shaper(coefficients, reduction)[0].cpu().numpy()
        if team == 'camera':
            np.testing.assert_allclose(shaped, fx['step/chain_reward_cam'][s], rtol=0, atol=1e-6, err_msg=str(s))
            assert float(eng.scalars[0, 0]) == np.float32(fx['step/reward_cam'][s])
        else:                          # the raw reward (up to 233 on a delivery) comes from the f32 step record: 1e-6 relative
            np.testing.assert_allclose(shaped, fx['step/chain_reward_tgt'][s], rtol=1e-6, atol=1e-5, err_msg=str(s))
            assert float(eng.scalars[0, 1]) == np.float32(fx['step/reward_tgt'][s])
        assert bool(eng.scalars[0, 2] > 0) == bool(fx['step/done'][s])
        shaped_frames.append(shaped)
    # FrameSkip: the same indices for `skip` frames (recorded so), rewards summed, the last frame's observation
    shaped_frames = np.stack(shaped_frames)
    learner = fx['step/learner_step']
    for ls in range(len(fx['skip/done'])):
        frames = np.nonzero(learner == ls)[0]
        assert len(frames) == int(fx['skip/frames'][ls]) and (fx[f'step/{me}_idx'][frames] == fx[f'skip/{me}_idx'][ls]).all()
        np.testing.assert_allclose(shaped_frames[frames].sum(axis=0), fx[f'skip/reward_{me}'][ls], rtol=1e-6, atol=5e-6 if team == 'camera' else 1e-4)
        assert np.array_equal(fx[f'step/chain_{me}_obs'][frames[-1]], fx[f'skip/chain_{me}_obs'][ls])
    assert len(fx['skip/done']) >= 10 and int(np.isfinite(fx['step/tape_ct']).sum()) > 50
    if team == 'target':
        assert int(fx['step/num_delivered_cargoes'][-1]) >= 5 and fx['step/aux_sparse_delivery'].any()


@pytest.mark.parametrize('team,config,skip', [('camera', 'MATE-4v8-9.yaml', 5), ('target', 'MATE-2v4-0.yaml', 10)])
def test_frame_skip_launch_is_the_sum_of_the_chains_frames(team, config, skip):
    """... and FrameSkip as ONE launch (rollout_versus_greedy on grid indices) is the same `skip` per-step calls, on the engine's own
    draws: rows bit for bit, so the wrapper's summed reward is the column sum of the scalar rows (auxiliary reward terms included:
    coverage_rate is column 3 of every row)."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(config)
    n = 128
    a, b = (Engine(cfg, n, seed=31) for _ in range(2))
    for e in (a, b):
        e.set_action_grids(**{team + '_levels': 5})
        e.enable_policies()
        e.reset()
    agents = a.num_cameras if team == 'camera' else a.num_targets
    gen = torch.Generator(device='cuda')
    gen.manual_seed(3)
    for ls in range(6):
        idx = torch.randint(0, 25, (n, agents), device='cuda', generator=gen, dtype=torch.int32)
        cam_r, tgt_r, sc_r = a.rollout_versus_greedy(team, idx, skip, auto_reset=False)
        for f in range(skip):
            b.step_versus_greedy(team, idx, auto_reset=False)
            assert torch.equal(sc_r[f], b.scalars) and torch.equal(cam_r[f], b.camera_obs) and torch.equal(tgt_r[f], b.target_obs), (ls, f)
    assert torch.equal(a.export_state(), b.export_state())
