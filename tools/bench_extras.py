"""Side measurements of bench.py's default N = 1 run (imported by bench.py unless --no-extras / --no-other-configs /
--no-side-measurements): the flows a learner calls one launch per step (`measure_learner_flows`), the other BASELINE.json
configurations that fit one GPU (`measure_other_config`), the N = 1 NumPy API under the evaluate-style harness (`measure_n1_api`) and
the whole-batch reset amortised over an episode (`measure_reset_amortised`).  Their results go to bench.py's details file
(`bench_details.json`) and, as a handful of numbers, into its `side` summary; the headline line does not depend on them."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)
LEARNER_BATCHES = (4096, 16384, 65536)     # `learner_flows`: the per-step flows at the batches a learner on one MI355X runs


def algorithmic_bytes(Nc, Nt, No):
    """B_alg per env-step."""
    Dc = 13 + 9 + 5 * Nt + 4 * No + 7 * Nc
    Dt = 13 + 14 + 7 * Nc + 4 * No + 5 * Nt
    return 4 * (Nc * Dc + Nt * Dt) + 8 * (Nc + Nt) + 2 * (16 * Nc + 35 * Nt + 72) + (24 * Nc + 24 * No + Nt) + 48


class ExternalActions:
    """The learner-in-the-loop flow: `step((camera_actions, target_actions))` with the joint actions in caller-owned
    device buffers that a policy rewrites between steps.  Here the "policy" is one elementwise torch kernel per team
    that refreshes the buffers in place (a stand-in for a network's output layer), so the environment really consumes
    new externally produced actions at every step."""

    def __init__(self, torch, eng, graph_steps, reset_interval=1):
        self.torch, self.eng, self.graph_steps = torch, eng, int(graph_steps)
        self.graph_steps -= self.graph_steps % max(1, reset_interval)      # a graph holds whole reset intervals
        N, Nc, Nt = eng.num_envs, eng.num_cameras, eng.num_targets
        gen = torch.Generator(device=eng.device)
        gen.manual_seed(1234)
        self.flat = torch.rand(N * (Nc + Nt) * 2, device=eng.device, generator=gen) * 2 - 1      # one buffer, two views
        self.cam = self.flat[:N * Nc * 2].view(N, Nc, 2)
        self.tgt = self.flat[N * Nc * 2:].view(N, Nt, 2)
        self.cam.mul_(torch.tensor([5.0, 2.5], device=eng.device))
        self.tgt.mul_(20.0)
        self.stepper = eng.make_stepper(self.cam, self.tgt, auto_reset=max(1, reset_interval), graph_steps=self.graph_steps, between=self.policy)

    def policy(self):
        # a new joint action every step, produced on the device by "someone else's" kernel
        self.flat.mul_(-1.0)

    def step(self):
        self.stepper.run(1)

    def run(self, steps):
        self.stepper.run(steps)


def measure_other_config(torch, device_index, spec, seconds, buffer_gib):
    """One entry of `other_configs`: a BASELINE.json configuration other than the headline, in its default flow (fused
    launches; Greedy vs Greedy restarts finished episodes after every 2nd launch, the random policy about every 128
    steps), timed for about `seconds` of back-to-back launches after an untimed pass and 0.25 s of clock warm-up."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    workload, batch, policy, label = spec[:4]
    eng = Engine(read_config(workload), batch, device=device_index, seed=0, first_env_index=0)
    b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
    b_obs = 4 * (eng.num_cameras * eng.camera_obs_dim + eng.num_targets * eng.target_obs_dim) + 48
    cap = int(buffer_gib * (1 << 30)) // (batch * b_obs)
    if len(spec) > 4:
        cap = min(cap, spec[4])
    if policy == 'greedy':
        eng.enable_policies()
        R, resets, fn, kernel = min(48, cap), 2, eng.rollout_greedy, 'rollout_greedy_kernel'
    else:
        R = next((r for r in (256, 128, 64, 32) if r <= cap), max(1, cap))
        resets, fn, kernel = max(1, 128 // R), eng.rollout_random, 'rollout_kernel'
    eng.reset()
    eng.reserve_rollout(R, search='deep')
    for _ in range(2 * resets):
        fn(R, auto_reset=resets)
    torch.cuda.synchronize()
    t0, n_warm = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.25:
        for _ in range(resets):
            fn(R, auto_reset=resets)
        n_warm += resets
        torch.cuda.synchronize()
    per_launch = (time.perf_counter() - t0) / n_warm
    launches = max(resets, int(seconds / per_launch) // resets * resets)
    eng.kernel_time(enable=1)
    torch.cuda.synchronize()
    idle0, t0 = eng.idle_steps(), time.perf_counter()
    for _ in range(launches):
        fn(R, auto_reset=resets)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    executed = batch * R * launches - (eng.idle_steps() - idle0)
    kernel_ms, timed = eng.kernel_time(enable=False)
    value = executed / elapsed
    out = {'config': label, 'workload': f'{workload} batch={batch} envs, {policy} policy, fused {R}-step launches, restarts every {resets} launch(es)',
           'value': value, 'unit': 'env-steps/s (executed: idle slots of finished episodes excluded)', 'seconds': elapsed, 'launches': launches,
           'kernel': kernel, 'kernel_avg_us': kernel_ms * 1e3, 'launches_timed': timed,
           'algorithmic_bytes_per_env_step': b_alg,
           'frac': (b_alg * batch * R / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kernel_ms > 0 else 0.0,
           'end_to_end_frac': b_alg * value / 1e9 / HBM_PEAK_GBS}
    eng.close()
    del eng
    torch.cuda.empty_cache()
    return out


def measure_reset_amortised(torch, eng, cfg, value, seconds_per_step):
    """The cost the timed region never contains.  Under the random policy cargo never runs out, so every environment of the
    batch hits the time limit on the same step, every max_episode_steps + 1 steps, and the batch restarts in ONE whole-batch
    reset (placement, Nc occlusion tables per environment, first view).  Its time, measured here (median of 7, each
    bracketed by synchronisations), spread over an episode: value_with_resets = N / (t_step + t_reset / (max_episode_steps + 1))."""
    times = []
    for _ in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.reset()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t_reset = sorted(times[1:])[len(times[1:]) // 2]
    episode = int(cfg['max_episode_steps']) + 1
    with_resets = eng.num_envs / (seconds_per_step + t_reset / episode)
    return {'whole_batch_reset_ms': t_reset * 1e3, 'episode_steps': episode, 'value_with_resets': with_resets, 'unit': 'env-steps/s',
            'cost_frac': 1.0 - with_resets / value if value > 0 else None,
            'note': 'value = N / t_step as timed; value_with_resets = N / (t_step + t_reset / episode_steps): one whole-batch reset per episode of the random-policy batch'}


def measure_n1_api(torch, steps=1500):
    """BASELINE config 1 (MATE-4v2-9.yaml, one environment, random actions, the reference's evaluate loop, mate/evaluate.py:85-167
    -> mate_amd/evaluate.py) on the N = 1 NumPy API: every step is a launch, a synchronisation and a PCIe copy of the observations
    and the state -- the compatibility path the reference's own wrappers use, not a throughput path."""
    import mate_amd
    from mate_amd.evaluate import evaluate, random_policy
    env = mate_amd.MultiAgentTracking('MATE-4v2-9.yaml', max_episode_steps=steps)
    env.seed(0)
    evaluate(env, random_policy(0))                  # untimed: code objects, allocations
    history = []
    t0 = time.perf_counter()
    evaluate(env, random_policy(1), history=history)
    elapsed = time.perf_counter() - t0
    out = {'workload': 'MATE-4v2-9.yaml, 1 environment, uniform random actions from NumPy, mate_amd.evaluate (reset + one episode)',
           'value': len(history) / elapsed, 'unit': 'env-steps/s', 'steps': len(history), 'seconds': elapsed,
           'reference_numpy': 1101.0, 'reference_note': 'mate/evaluate.py FPS of the reference on one core of the build container (BASELINE.md section 2)'}
    env.close()
    return out


def measure_learner_flows(torch, device_index, workload, batch, graph_steps, reset_interval, versus_reset_interval, world=1):
    """The flows a learner calls, one launch (or one launch + the learner's own kernel) per step, at `batch` environments:
      per_step_launch   step_random: the engine's step kernel back to back, no caller kernel in between
      external_actions  step(actions): the joint actions in a caller-owned f32 buffer that a stand-in policy kernel rewrites before
                        every step (mate/environment.py:590-676 behind a learner), `graph_steps` (policy, step) pairs per HIP graph
      versus_greedy     MultiCamera(GreedyTargetAgent) (mate/wrappers/single_team.py:245-264; every examples/*/camera/config.py): the
                        learner's stand-in policy kernel writes the camera team's joint action, the on-device greedy targets act and
                        the environment steps in ONE launch (step_greedy_kernel)
      versus_greedy_frameskip5     ... with FrameSkip(5) on top (one fused launch per learner action): the camera trainers' whole flow
      target_learner_frameskip10   the TARGET trainers' flow (examples/*/target/config.py): MATE-2v4-0, MultiTarget(GreedyCameraAgent),
                        FrameSkip(10) -- a small scenario (rows a fifth of MATE-4v8-9's: a wave's lanes are mostly idle)
      external_actions_two_groups   the same batch as two half-batch engines on two streams, graphs replayed alternately -- a learner
                        that alternates between two groups of environments (double-buffered sampling): one group's step runs under
                        the other group's policy and launch ramp
    each timed over whole graphs (median of three passes, all three listed), with one reset launch per `reset_interval` steps
    (`versus_reset_interval` against the greedy opponents, whose episodes end after ~1.2 k steps: a hundred of 4096 environments per
    32 steps, and their restart -- placement, occlusion tables, first view: four latency-bound launches -- costs 3.5 us per step at
    32, half of it at 64); idle slots of finished environments are excluded from `value`.  `kernel_avg_us` / `roofline_frac`: a separate pass of direct launches with a
    dispatch-event pair on every launch."""
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(workload)
    G = max(int(graph_steps), reset_interval)
    G -= G % reset_interval
    steps = max(G, (1024 if batch <= 16384 else 256) // G * G)
    out = {'batch': batch, 'workload': workload, 'steps': steps, 'reset_interval': reset_interval, 'graph_steps': G}

    def timed(run, idle, n_envs):
        run(2 * G)
        torch.cuda.synchronize()
        times = []
        for _ in range(3):
            i0, t0 = idle(), time.perf_counter()
            run(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            times.append((dt, n_envs * steps - (idle() - i0)))
        passes[:] = [t[0] / steps * 1e6 for t in times]
        return sorted(times)[1]

    passes = []

    def entry(eng, dt, executed, kernel=None, km=0.0, flow=None):
        b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
        e = {'value': executed * world / dt, 'unit': 'env-steps/s', 'us_per_step': dt / steps * 1e6,
             'passes_us_per_step': [round(v, 3) for v in passes],
             'end_to_end_frac': b_alg * executed / dt / 1e9 / HBM_PEAK_GBS}
        if kernel:
            e.update({'kernel': kernel, 'kernel_avg_us': km * 1e3, 'roofline_frac': b_alg * batch / (km * 1e-3) / 1e9 / HBM_PEAK_GBS if km > 0 else 0.0})
        if flow:
            e['flow'] = flow
        return e

    with torch.cuda.device(device_index):
        # ---- step_random, direct launches (the host enqueues ahead of the GPU from 4096 environments on)
        eng = Engine(cfg, batch, device=device_index, seed=0)
        eng.reset()
        dt, ex = timed(lambda n: [eng.step_random(auto_reset=reset_interval) for _ in range(n)], eng.idle_steps, batch)
        eng.kernel_time(enable=1)
        for _ in range(256):
            eng.step_random(auto_reset=reset_interval)
        torch.cuda.synchronize()
        km, _ = eng.kernel_time(enable=False)
        out['per_step_launch'] = entry(eng, dt, ex, 'step_kernel', km)
        # ---- step(actions) from a HIP graph
        ext = ExternalActions(torch, eng, G, reset_interval)
        dt, ex = timed(ext.run, eng.idle_steps, batch)
        out['external_actions'] = entry(eng, dt, ex, flow=f'step(actions): f32 joint actions rewritten by a policy kernel in a caller-owned device buffer before every step; '
                                                              f'{ext.graph_steps} (policy kernel, step) pairs + one reset launch per {reset_interval} steps per HIP graph replay')
        ext.stepper.close()
        eng.close()
        del ext, eng
        torch.cuda.empty_cache()
        # ---- learner versus the on-device greedy opponents
        eng = Engine(cfg, batch, device=device_index, seed=0)
        if eng.num_cameras:
            eng.enable_policies()
            eng.reset()
            mine = (torch.rand((batch, eng.num_cameras, 2), device=eng.device) * 2 - 1) * torch.tensor([5.0, 2.5], device=eng.device)
            Gv = max(G, versus_reset_interval) // versus_reset_interval * versus_reset_interval
            st = eng.make_stepper(mine, None, auto_reset=versus_reset_interval, graph_steps=Gv, between=lambda: mine.mul_(-1.0), versus='camera')
            dt, ex = timed(st.run, eng.idle_steps, batch)
            st.close()
            eng.kernel_time(enable=1)
            for _ in range(256):
                eng.step_versus_greedy('camera', mine, auto_reset=versus_reset_interval)
            torch.cuda.synchronize()
            km, _ = eng.kernel_time(enable=False)
            out['versus_greedy'] = entry(eng, dt, ex, 'step_greedy_kernel' if eng.last_flow == 4 else 'rollout_greedy_kernel (one step)', km,
                                         flow='MultiCamera(GreedyTargetAgent): the learner\'s stand-in policy (one elementwise kernel) writes the camera team\'s joint action, '
                                              'the greedy targets act and the environment steps in one launch; executed env-steps (idle slots of finished episodes excluded); '
                                              f'one restart of the finished environments per {versus_reset_interval} steps')
            out['versus_greedy']['reset_interval'] = versus_reset_interval
            del st
            # ---- ... and with FrameSkip(5) on top, what every example trainer's make_env ends with (examples/ippo/camera/config.py:
            # frame_skip = 5; examples/utils/wrappers.py:301-323): ONE launch per learner action (rollout_versus_greedy), replayed from
            # a HIP graph like the per-step flows (Stepper(frame_skip=K): the device-resident step counter advances by K per launch)
            K = 5
            per = max(1, versus_reset_interval // K)               # launches per reset interval
            Gs = max(per, (G // K) // per * per)                   # launches per graph
            launches = max(Gs, steps // K // Gs * Gs)
            st = eng.make_stepper(mine, None, auto_reset=per, graph_steps=Gs, between=lambda: mine.mul_(-1.0), versus='camera', frame_skip=K)
            st.run(2 * Gs)
            torch.cuda.synchronize()
            times = []
            for _ in range(3):
                i0, t0 = eng.idle_steps(), time.perf_counter()
                st.run(launches)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                times.append((dt, batch * launches * K - (eng.idle_steps() - i0)))
            dt, ex = sorted(times)[1]
            st.close()
            del st
            eng.kernel_time(enable=1)
            for _ in range(4 * per):
                eng.rollout_versus_greedy('camera', mine, K, auto_reset=per)
            torch.cuda.synchronize()
            km, _ = eng.kernel_time(enable=False)
            b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
            out['versus_greedy_frameskip5'] = {
                'value': ex * world / dt, 'unit': 'env-steps/s', 'us_per_step': dt / (launches * K) * 1e6, 'us_per_launch': dt / launches * 1e6,
                'passes_us_per_step': [round(t[0] / (launches * K) * 1e6, 3) for t in times],
                'end_to_end_frac': b_alg * ex / dt / 1e9 / HBM_PEAK_GBS, 'kernel': 'rollout_greedy_kernel', 'kernel_avg_us': km * 1e3,
                'roofline_frac': b_alg * batch * K / (km * 1e-3) / 1e9 / HBM_PEAK_GBS if km > 0 else 0.0,
                'flow': f'FrameSkip({K}) over MultiCamera(GreedyTargetAgent): one policy kernel and ONE fused launch per learner action ({K} frames, the greedy targets '
                        f'act anew on every frame), {Gs} (policy kernel, launch) pairs per HIP graph replay, one restart of the finished environments per {per} launches; '
                        'executed env-steps'}
            del mine
        eng.close()
        del eng
        torch.cuda.empty_cache()
        # ---- the TARGET learner's flow of the example trainers (examples/ippo/target/config.py:20-67 and its siblings): MATE-2v4-0,
        # MultiTarget(GreedyCameraAgent), FrameSkip(10) -- one policy kernel and one ten-frame launch per learner action, from a HIP graph
        if workload == 'MATE-4v8-9.yaml':
            cfg_t = read_config('MATE-2v4-0.yaml')
            eng = Engine(cfg_t, batch, device=device_index, seed=0)
            eng.enable_policies()
            eng.reset()
            K = 10
            per = max(1, versus_reset_interval // K)
            Gs = max(per, (G // K) // per * per)
            launches = max(Gs, steps // K // Gs * Gs)
            eng.reserve_rollout(K, search='none')
            mine = (torch.rand((batch, eng.num_targets, 2), device=eng.device) * 2 - 1) * 10.0
            st = eng.make_stepper(None, mine, auto_reset=per, graph_steps=Gs, between=lambda: mine.mul_(-1.0), versus='target', frame_skip=K)
            st.run(2 * Gs)
            torch.cuda.synchronize()
            times = []
            for _ in range(3):
                i0, t0 = eng.idle_steps(), time.perf_counter()
                st.run(launches)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                times.append((dt, batch * launches * K - (eng.idle_steps() - i0)))
            dt, ex = sorted(times)[1]
            st.close()
            del st
            eng.kernel_time(enable=1)
            for _ in range(4 * per):
                eng.rollout_versus_greedy('target', mine, K, auto_reset=per)
            torch.cuda.synchronize()
            km, _ = eng.kernel_time(enable=False)
            b_alg = algorithmic_bytes(eng.num_cameras, eng.num_targets, eng.num_obstacles)
            out['target_learner_frameskip10'] = {
                'workload': 'MATE-2v4-0.yaml', 'value': ex * world / dt, 'unit': 'env-steps/s', 'us_per_step': dt / (launches * K) * 1e6, 'us_per_launch': dt / launches * 1e6,
                'passes_us_per_step': [round(t[0] / (launches * K) * 1e6, 3) for t in times], 'algorithmic_bytes_per_env_step': b_alg,
                'end_to_end_frac': b_alg * ex / dt / 1e9 / HBM_PEAK_GBS, 'kernel': 'rollout_greedy_kernel', 'kernel_avg_us': km * 1e3,
                'roofline_frac': b_alg * batch * K / (km * 1e-3) / 1e9 / HBM_PEAK_GBS if km > 0 else 0.0,
                'flow': f'FrameSkip({K}) over MultiTarget(GreedyCameraAgent) on MATE-2v4-0 (the target trainers\' scenario): one policy kernel and ONE fused launch per learner '
                        f'action ({K} frames, the greedy cameras act anew on every frame), {Gs} (policy kernel, launch) pairs per HIP graph replay, one restart of the finished '
                        f'environments per {per} launches; executed env-steps; rows of this scenario are a fifth of MATE-4v8-9\'s'}
            del mine
            eng.close()
            del eng
            torch.cuda.empty_cache()
        # ---- two half-batch groups on two streams (mate_amd.engine.EngineGroups): step(actions), and the learner versus the greedy opponents
        if batch % 2 == 0 and batch >= 2048:
            from mate_amd.engine import EngineGroups
            half = batch // 2
            for key, versus in (('external_actions_two_groups', False), ('versus_greedy_two_groups', True)):
                if versus and 'versus_greedy' not in out:      # (a scenario without cameras)
                    continue
                interval = versus_reset_interval if versus else reset_interval
                Gk = max(G, interval) // interval * interval
                groups = EngineGroups(cfg, batch, groups=2, device=device_index, seed=0, policies=versus)
                groups.reset()
                keep = []

                def make(g, e):
                    if versus:
                        mine = (torch.rand((half, e.num_cameras, 2), device=e.device) * 2 - 1) * torch.tensor([5.0, 2.5], device=e.device)
                        keep.append(mine)
                        return e.make_stepper(mine, None, auto_reset=interval, graph_steps=Gk, between=(lambda m=mine: m.mul_(-1.0)), versus='camera')
                    ext = ExternalActions(torch, e, Gk, interval)
                    keep.append(ext)
                    return ext.stepper

                steppers = groups.each(make)
                torch.cuda.synchronize()
                # (HIP maps streams onto a handful of hardware queues, and two streams that share one run their graphs one after the
                # other -- 23 instead of 13 us per step at 4096 when that happens: the second group's stream is chosen by a short trial)
                trials = groups.pick_streams(lambda g, e: steppers[g].run(Gk), candidates=3, warm=2, timed=4)
                dt, ex = timed(lambda n: [groups.each(lambda g, e: steppers[g].run(Gk)) for _ in range(n // Gk)], groups.idle_steps, batch)
                out[key] = entry(groups.engines[0], dt, ex, flow=f'{"MultiCamera(GreedyTargetAgent)" if versus else "step(actions)"} as two engines of {half} environments (global indices 0.. and {half}..) on two streams '
                                                                   '(mate_amd.engine.EngineGroups), their HIP graphs replayed alternately: us_per_step = per step of the WHOLE batch')
                out[key]['stream_trials_us_per_step'] = [round(t / (4 * Gk) * 1e6, 2) for t in trials]
                out[key]['reset_interval'] = interval
                for st in steppers:
                    st.close()
                groups.close()
                del steppers, groups, keep
                torch.cuda.empty_cache()
    return out


OTHER_CONFIGS = (
    ('MATE-8v8-9.yaml', 8192, 'greedy', 'BASELINE config 3'),
    ('MATE-4v8-0.yaml', 8192, 'random', 'BASELINE config 4, the shard of one of its 8 GPUs'),
    ('MATE-Navigation.yaml', 4096, 'random', 'BASELINE config 5, the shard of one of its 8 GPUs'),
    # the same two configurations WHOLE on this one GPU (sixteen / eight generations of resident waves; launches of 64 steps: 19 / 17 GB of rows)
    ('MATE-4v8-0.yaml', 65536, 'random', 'BASELINE config 4, its whole batch on ONE GPU', 64),
    ('MATE-Navigation.yaml', 32768, 'random', 'BASELINE config 5, its whole batch on ONE GPU', 64),
)


