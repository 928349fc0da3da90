"""EngineGroups: one batch as G groups of environments on G streams (a learner that interleaves its groups) is the single engine's
batch, bit for bit -- every environment's random streams are keyed by its GLOBAL index."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('flow', ['actions', 'versus'])
def test_groups_on_streams_step_the_same_episodes_as_one_engine(flow):
    from mate_amd.config import read_config
    from mate_amd.engine import Engine, EngineGroups
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=30)
    n, G, steps = 192, 2, 70
    gen = torch.Generator(device='cuda')
    gen.manual_seed(5)
    cam = (torch.rand((steps, n, 4, 2), device='cuda', generator=gen) * 2 - 1) * torch.tensor([5.0, 2.5], device='cuda')
    tgt = (torch.rand((steps, n, 8, 2), device='cuda', generator=gen) * 2 - 1) * 20.0
    one = Engine(cfg, n, seed=3, first_env_index=1000)
    groups = EngineGroups(cfg, n, groups=G, seed=3, first_env_index=1000, policies=flow == 'versus')
    if flow == 'versus':
        one.enable_policies()
    one.reset()
    groups.reset()
    groups.synchronize()
    h = n // G
    assert torch.equal(torch.cat([e.target_obs for e in groups.engines]), one.target_obs)
    for s in range(steps):
        if flow == 'versus':
            one.step_versus_greedy('camera', cam[s], auto_reset=4)
            groups.each(lambda g, eng: eng.step_versus_greedy('camera', cam[s, g * h:(g + 1) * h].contiguous(), auto_reset=4))
        else:
            one.step(cam[s], tgt[s], auto_reset=True)
            groups.each(lambda g, eng: eng.step(cam[s, g * h:(g + 1) * h].contiguous(), tgt[s, g * h:(g + 1) * h].contiguous(), auto_reset=True))
        groups.synchronize()
        for name in ('camera_obs', 'target_obs', 'scalars', 'masks'):
            got = torch.cat([getattr(e, name) for e in groups.engines])
            assert torch.equal(got.view(torch.uint8), getattr(one, name).view(torch.uint8)), (s, name)
    assert torch.equal(torch.cat([e.export_state() for e in groups.engines]), one.export_state())
    assert (one.state_dict()['episode'] >= 2).all()
    # the side streams by trial: any choice steps the same episodes (the trial itself advances the groups: compare them with each other only)
    times = groups.pick_streams(lambda g, eng: eng.step_random(auto_reset=True), candidates=2, warm=1, timed=2)
    assert len(times) == 2 and all(t > 0 for t in times)
    groups.close()
