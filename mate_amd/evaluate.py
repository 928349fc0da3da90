"""Episode harness of the N=1 API: the counterpart of the reference's evaluation loop (mate/evaluate.py:85-167).

The reference wraps the environment so that one team is played by built-in agents (`mate.MultiCamera`), runs one episode
with the other team's agents and reports a status row: episode step, delivered cargoes, step reward, target-team episode
reward, steps per cargo, mean transport rate, mean coverage rate, normalised episode reward, FPS
(mate/evaluate.py:52-73, 129-143).  `evaluate()` below runs the same loop on `mate_amd.MultiAgentTracking` with the same
status keys and the same update rule (the row is kept from the first delivery on, or at `done`).  Both teams act through
`joint_policy`; by default they are the reference's GreedyCameraAgent / GreedyTargetAgent computed ON THE DEVICE
(`Engine.step_greedy`), so the whole loop is one policy launch + one step launch per step.

    python -m mate_amd.evaluate [--config MATE-4v2-9.yaml] [--seed 0] [--episodes 1] [--policy greedy|random]
"""
import argparse
import time
from collections import OrderedDict

import numpy as np

__all__ = ['COLUMNS', 'evaluate', 'random_policy', 'format_row']

# name -> format of the reference's table (mate/evaluate.py:52-71)
COLUMNS = OrderedDict([
    ('Step', '{:d}'.format), ('Cargo', '{:d}'.format), ('Reward', '{:+.2f}'.format), ('Target Episode Reward', '{:+.2f}'.format),
    ('Step / Cargo', '{:.1f}'.format), ('Mean Transport Rate', lambda x: f'{100.0 * x:.3f}%'),
    ('Mean Coverage Rate', lambda x: f'{100.0 * x:.3f}%'), ('Normalized Target Episode Reward', '{:+.5f}'.format), ('FPS', '{:.1f}'.format),
])


def random_policy(seed=0):
    """Uniform samples of the two joint action spaces (examples/random.py)."""
    rng = np.random.RandomState(seed)

    def act(env, observations, infos):
        cam = rng.uniform(-1.0, 1.0, (env.num_cameras, 2)) * np.array([env.camera_rotation_step, env.camera_zooming_step])
        tgt = np.stack([rng.uniform(-1.0, 1.0, 2) * env.targets[t].step_size for t in range(env.num_targets)])
        return cam, tgt
    return act


def format_row(values):
    return '|'.join([''] + [f' {fmt(v)} ' for fmt, v in zip(COLUMNS.values(), values)] + [''])


class _EpisodeLog:
    """Running quantities behind one status row (mate/evaluate.py:129-139)."""

    def __init__(self, env):
        self.env = env
        self.episode_reward = 0.0
        self.coverage_sum = 0.0
        self.steps = 0
        self.started = time.perf_counter()

    def row(self, step_reward):
        env = self.env
        self.steps += 1
        self.episode_reward += step_reward
        self.coverage_sum += env.coverage_rate
        delivered = env.num_delivered_cargoes
        return OrderedDict(zip(COLUMNS, (
            env.episode_step, delivered, step_reward, self.episode_reward,
            env.episode_step / delivered if delivered > 0 else np.nan,
            env.mean_transport_rate, self.coverage_sum / self.steps,
            self.episode_reward / env.max_target_team_episode_reward,
            env.episode_step / (time.perf_counter() - self.started))))


def evaluate(env, joint_policy=None, verbose=False, history=None):
    """One episode (mate/evaluate.py:85-167).  `joint_policy(env, (camera_obs, target_obs), (camera_infos, target_infos))`
    returns the joint action of both teams; None = the on-device Greedy agents of both teams
    (`env.enable_greedy_policies()` must then precede this call).  Returns the reference's status dict -- the row of
    the last step once a cargo has been delivered or the episode is done, else empty; `history` (a list) receives
    every step's row.  The loop ends at `max_episode_steps` like the reference's, one call before the environment's own
    time-limit `done`."""
    observations, infos, status = env.reset(), None, {}
    log = _EpisodeLog(env)
    if verbose:
        print('|'.join([''] + [f' {name} ' for name in COLUMNS] + ['']))
    done = False
    while not done and env.episode_step < env.max_episode_steps:
        if joint_policy is None:
            observations, rewards, done, infos = env.step_greedy()
        else:
            observations, rewards, done, infos = env.step(joint_policy(env, observations, infos))
        row = log.row(rewards[1])                 # the target team's reward, as in the reference's single-team loop
        if row['Cargo'] > 0 or done:
            status = dict(row)
        if history is not None:
            history.append(dict(row))
        if verbose:
            print(format_row(list(row.values())))
    return status


def main():
    ap = argparse.ArgumentParser(prog='python -m mate_amd.evaluate', description=__doc__.split('\n')[0])
    ap.add_argument('--config', '--cfg', default='MATE-4v2-9.yaml')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--episodes', type=int, default=1)
    ap.add_argument('--policy', choices=['greedy', 'random'], default='greedy')
    ap.add_argument('--max-episode-steps', type=int, default=None)
    ap.add_argument('--verbose', action='store_true')
    args = ap.parse_args()
    import mate_amd
    overrides = {} if args.max_episode_steps is None else {'max_episode_steps': args.max_episode_steps}
    env = mate_amd.MultiAgentTracking(args.config, **overrides)
    if args.policy == 'greedy':
        env.enable_greedy_policies()
    env.seed(args.seed)
    rows = []
    for episode in range(args.episodes):
        status = evaluate(env, None if args.policy == 'greedy' else random_policy(args.seed + episode), verbose=args.verbose)
        rows.append(status)
        print(f'episode {episode}: ' + ', '.join(f'{k}: {fmt(status[k])}' for k, fmt in COLUMNS.items() if k in status))
    return rows


if __name__ == '__main__':
    main()
