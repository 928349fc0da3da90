"""ctypes front-end of ``oracle/libmate_oracle.so`` (see ``mate_oracle.h``).

TEST INFRASTRUCTURE ONLY: parity checker + CPU baseline, never the product path.
"""
import ctypes
import os
import subprocess

import numpy as np

__all__ = ['lib', 'build', 'OracleEnv', 'OracleBatch', 'obstruct', 'clamp_step', 'normalize_angle',
           'camera_simulate', 'build_lut', 'camera_perceive', 'interp', 'random_actions', 'philox', 'GreedyPolicies']

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, 'libmate_oracle.so')

c_double_p = ctypes.POINTER(ctypes.c_double)
c_float_p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile the C restatement with gcc (a few seconds)."""
    src = os.path.join(HERE, 'mate_oracle.c')
    if force or not os.path.exists(SO_PATH) or os.path.getmtime(SO_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', HERE, '-B', 'libmate_oracle.so'], stdout=subprocess.DEVNULL)
    return SO_PATH


class PolicyTape(ctypes.Structure):
    _fields_ = [('cam_binom_u', c_double_p), ('cam_sample_u', c_double_p), ('cam_delay', ctypes.POINTER(ctypes.c_int)),
                ('tgt_choice_u', c_double_p), ('tgt_binom_u', c_double_p), ('tgt_sample_u', c_double_p),
                ('tgt_reset_sample_u', c_double_p)]


def _load():
    if not os.path.exists(SO_PATH):
        build()
    lib = ctypes.CDLL(SO_PATH)
    D, I, P = ctypes.c_double, ctypes.c_int, ctypes.c_void_p
    lib.mo_normalize_angle.restype = D
    lib.mo_normalize_angle.argtypes = [D]
    lib.mo_clamp_step.argtypes = [D, D, D, c_double_p]
    lib.mo_obstruct.argtypes = [D] * 7 + [I, I, c_double_p]
    lib.mo_camera_simulate.argtypes = [D] * 8 + [c_double_p]
    lib.mo_interp.restype = D
    lib.mo_interp.argtypes = [c_double_p, c_double_p, I, D]
    lib.mo_build_lut_raw.restype = I
    lib.mo_build_lut_raw.argtypes = [D, D, D, c_double_p, I, D, I, c_double_p, c_double_p, I]
    lib.mo_camera_perceive.restype = I
    lib.mo_camera_perceive.argtypes = [D] * 9 + [c_double_p, c_double_p, I]
    lib.mo_create.restype = P
    lib.mo_create.argtypes = [I, I, I]
    lib.mo_destroy.argtypes = [P]
    lib.mo_set.restype = I
    lib.mo_set.argtypes = [P, ctypes.c_char_p, c_double_p, I]
    lib.mo_get.restype = I
    lib.mo_get.argtypes = [P, ctypes.c_char_p, c_double_p, I]
    for name in ('mo_camera_obs_dim', 'mo_target_obs_dim', 'mo_state_dim'):
        getattr(lib, name).restype = I
        getattr(lib, name).argtypes = [P]
    lib.mo_build_luts.argtypes = [P]
    lib.mo_get_lut.restype = I
    lib.mo_get_lut.argtypes = [P, I, I, c_double_p, c_double_p, I]
    lib.mo_set_lut.restype = I
    lib.mo_set_lut.argtypes = [P, I, I, c_double_p, c_double_p, I]
    lib.mo_update_view.argtypes = [P, c_double_p]
    lib.mo_step.argtypes = [P, c_double_p, c_double_p, c_double_p, c_double_p]
    lib.mo_observe.argtypes = [P, c_double_p, c_double_p]
    lib.mo_policy_create.restype = P
    lib.mo_policy_create.argtypes = []
    lib.mo_policy_destroy.argtypes = [P]
    lib.mo_policy_act.argtypes = [P, P, ctypes.POINTER(PolicyTape), c_double_p, c_double_p]
    lib.mo_observe_mode.argtypes = [P, I, I, c_double_p, c_double_p]
    lib.mo_decode_discrete.argtypes = [P, ctypes.POINTER(ctypes.c_int), c_double_p, ctypes.POINTER(ctypes.c_int), c_double_p, c_double_p, c_double_p]
    lib.mo_soft_coverage.argtypes = [P, c_double_p, c_double_p]
    lib.mo_state.argtypes = [P, c_double_p]
    lib.mo_reset.argtypes = [P]
    lib.mo_reset_tape.argtypes = [P, c_double_p, ctypes.c_int, c_double_p]
    lib.mo_reset_tape.restype = ctypes.c_int
    lib.mo_update_view_reset.argtypes = [P]
    lib.mo_batch_create.restype = P
    lib.mo_batch_create.argtypes = [P, I, ctypes.c_uint64, ctypes.c_uint64]
    lib.mo_batch_destroy.argtypes = [P]
    lib.mo_batch_env.restype = P
    lib.mo_batch_env.argtypes = [P, I]
    lib.mo_batch_get.restype = I
    lib.mo_batch_get.argtypes = [P, ctypes.c_char_p, c_double_p, I]
    lib.mo_batch_reset.argtypes = [P, I]
    lib.mo_batch_step.argtypes = [P, c_float_p, c_float_p, I, I]
    lib.mo_batch_observe.argtypes = [P, c_float_p, c_float_p, I]
    lib.mo_random_actions.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, I, I, D, D, D, c_float_p, c_float_p]
    lib.mo_philox4x32.argtypes = [ctypes.c_uint32] * 6 + [ctypes.POINTER(ctypes.c_uint32)]
    return lib


lib = _load()


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _fp(a):
    return a.ctypes.data_as(c_float_p)


def normalize_angle(a):
    return lib.mo_normalize_angle(float(a))


def clamp_step(ax, ay, step_size):
    out = np.zeros(2)
    lib.mo_clamp_step(float(ax), float(ay), float(step_size), _dp(out))
    return out


def obstruct(origin, vec, center, radius, keep_tangential=False, outer=False):
    out = np.zeros(2)
    lib.mo_obstruct(float(origin[0]), float(origin[1]), float(vec[0]), float(vec[1]), float(center[0]),
                    float(center[1]), float(radius), int(keep_tangential), int(outer), _dp(out))
    return out


def camera_simulate(phi, theta, dphi, dtheta, theta_min, rmax, rot_step, zoom_step):
    out = np.zeros(3)
    lib.mo_camera_simulate(float(phi), float(theta), float(dphi), float(dtheta), float(theta_min), float(rmax),
                           float(rot_step), float(zoom_step), _dp(out))
    return out


def interp(xp, fp, x):
    xp = np.ascontiguousarray(xp, dtype=np.float64)
    fp = np.ascontiguousarray(fp, dtype=np.float64)
    return lib.mo_interp(_dp(xp), _dp(fp), len(xp), float(x))


def build_lut(cam_xy, rmax, obstacles_xyr, tau=0.0, outer=False, cap=8192):
    obstacles_xyr = np.ascontiguousarray(obstacles_xyr, dtype=np.float64).reshape(-1, 3)
    phis, rhos = np.zeros(cap), np.zeros(cap)
    n = lib.mo_build_lut_raw(float(cam_xy[0]), float(cam_xy[1]), float(rmax), _dp(obstacles_xyr), len(obstacles_xyr),
                             float(tau), int(outer), _dp(phis), _dp(rhos), cap)
    assert n > 0, n
    return phis[:n].copy(), rhos[:n].copy()


def camera_perceive(cam_xy, phi, theta, sight, point, u, tau, lut_phi, lut_rho):
    lut_phi = np.ascontiguousarray(lut_phi, dtype=np.float64)
    lut_rho = np.ascontiguousarray(lut_rho, dtype=np.float64)
    return bool(lib.mo_camera_perceive(float(cam_xy[0]), float(cam_xy[1]), float(phi), float(theta), float(sight),
                                       float(point[0]), float(point[1]), float(u), float(tau), _dp(lut_phi),
                                       _dp(lut_rho), len(lut_phi)))


def philox(k0, k1, c0, c1, c2, c3):
    out = (ctypes.c_uint32 * 4)()
    lib.mo_philox4x32(k0, k1, c0, c1, c2, c3, out)
    return [int(v) for v in out]


def random_actions(seed, env_index, tick, Nc, Nt, rot_step, zoom_step, step_size):
    cam = np.zeros((max(Nc, 1), 2), dtype=np.float32)
    tgt = np.zeros((Nt, 2), dtype=np.float32)
    lib.mo_random_actions(seed, env_index, tick, Nc, Nt, float(rot_step), float(zoom_step), float(step_size), _fp(cam), _fp(tgt))
    return cam[:Nc], tgt


# max extents of the fixed-size arrays inside mo_env (mate_oracle.h)
MAXC, MAXT, MAXO, NW = 16, 16, 64, 4
_PADDED = {  # field -> (row stride in the C struct, logical shape builder)
    'cam_range': (4, lambda e: (e.Nc, 4)), 'tgt_range': (4, lambda e: (e.Nt, 4)), 'obs_range': (4, lambda e: (e.No, 4)),
    'camera_obstacle_view_mask': (MAXO, lambda e: (e.Nc, e.No)),
    'tgt_empty_bits': (NW, lambda e: (e.Nt, NW)), 'tgt_goal_bits': (NW, lambda e: (e.Nt, NW)),
    'remaining_cargoes': (NW, lambda e: (NW, NW)),
    'camera_target_view_mask': (MAXT, lambda e: (e.Nc, e.Nt)), 'target_camera_view_mask': (MAXC, lambda e: (e.Nt, e.Nc)),
    'target_obstacle_view_mask': (MAXO, lambda e: (e.Nt, e.No)), 'target_target_view_mask': (MAXT, lambda e: (e.Nt, e.Nt)),
    'camera_camera_view_mask': (MAXC, lambda e: (e.Nc, e.Nc)), 'target_warehouse_distances': (NW, lambda e: (e.Nt, NW)),
}
_VECTORS = {
    'cam_x': 'Nc', 'cam_y': 'Nc', 'cam_radius': 'Nc', 'cam_min_viewing_angle': 'Nc', 'cam_max_sight_range': 'Nc',
    'cam_rotation_step': 'Nc', 'cam_zooming_step': 'Nc', 'cam_phi': 'Nc', 'cam_theta': 'Nc', 'cam_sight': 'Nc',
    'obs_x': 'No', 'obs_y': 'No', 'obs_radius': 'No',
    'tgt_capacity': 'Nt', 'tgt_step_size': 'Nt', 'tgt_sight_range': 'Nt', 'tgt_x': 'Nt', 'tgt_y': 'Nt',
    'tgt_colliding': 'Nt', 'tgt_goals': 'Nt', 'freights': 'Nt', 'bounties': 'Nt', 'target_steps': 'Nt',
    'tracked_steps': 'Nt', 'tracked_bits': 'Nt', 'target_dones': 'Nt', 'awaiting_cargo_counts': 'NW',
    'obs_radius_range': '2',
}


class GreedyPolicies:
    """The reference's GreedyCameraAgent / GreedyTargetAgent teams of one environment (mate/agents/greedy.py)."""

    def __init__(self):
        self._h = lib.mo_policy_create()

    def __del__(self):
        if getattr(self, '_h', None) and lib is not None:
            lib.mo_policy_destroy(self._h)
            self._h = None

    def act(self, env, cam_binom_u, cam_sample_u, cam_delay, tgt_choice_u, tgt_binom_u, tgt_sample_u, tgt_reset_sample_u):
        keep = [np.ascontiguousarray(np.nan_to_num(np.asarray(a, dtype=np.float64), nan=0.0))
                for a in (cam_binom_u, cam_sample_u, tgt_choice_u, tgt_binom_u, tgt_sample_u, tgt_reset_sample_u)]
        delay = np.ascontiguousarray(cam_delay, dtype=np.int32)
        pad = np.zeros(4)
        dp = lambda a: _dp(a if a.size else pad)  # noqa: E731
        tape = PolicyTape(dp(keep[0]), dp(keep[1]), delay.ctypes.data_as(ctypes.POINTER(ctypes.c_int)) if delay.size else None,
                          dp(keep[2]), dp(keep[3]), dp(keep[4]), dp(keep[5]))
        cam_act, tgt_act = np.zeros((max(env.Nc, 1), 2)), np.zeros((env.Nt, 2))
        lib.mo_policy_act(self._h, env._h, ctypes.byref(tape), _dp(cam_act), _dp(tgt_act))
        return cam_act[:env.Nc], tgt_act


class OracleEnv:
    """One environment of the CPU oracle (f64 state, named fields)."""

    def __init__(self, Nc, Nt, No, handle=None, owner=True):
        self.Nc, self.Nt, self.No, self.NW = int(Nc), int(Nt), int(No), NW
        self._h = handle if handle is not None else lib.mo_create(self.Nc, self.Nt, self.No)
        assert self._h, 'mo_create failed'
        self._owner = owner and handle is None
        self.Dc = lib.mo_camera_obs_dim(self._h)
        self.Dt = lib.mo_target_obs_dim(self._h)
        self.S = lib.mo_state_dim(self._h)

    def __del__(self):
        if getattr(self, '_owner', False) and self._h and lib is not None:   # `lib` is gone at interpreter shutdown
            lib.mo_destroy(self._h)
            self._h = None

    def _count(self, key):
        return {'Nc': self.Nc, 'Nt': self.Nt, 'No': self.No, 'NW': NW, '2': 2}[key]

    def set(self, field, value):
        value = np.asarray(value, dtype=np.float64)
        if field in _PADDED:
            stride, shape = _PADDED[field]
            rows, cols = shape(self)
            value = value.reshape(rows, cols)
            buf = np.zeros((max(rows, 1), stride))
            buf[:rows, :cols] = value
            flat = np.ascontiguousarray(buf[:rows].ravel())
        else:
            flat = np.ascontiguousarray(value.ravel())
        n = lib.mo_set(self._h, field.encode(), _dp(flat), flat.size) if flat.size else 0
        assert n == flat.size, (field, n, flat.size)

    def get(self, field):
        if field in _PADDED:
            stride, shape = _PADDED[field]
            rows, cols = shape(self)
            buf = np.zeros(max(rows * stride, 1))
            if rows:
                n = lib.mo_get(self._h, field.encode(), _dp(buf), rows * stride)
                assert n == rows * stride, (field, n)
            return buf[:rows * stride].reshape(rows, stride)[:, :cols].copy()
        count = self._count(_VECTORS[field]) if field in _VECTORS else 1
        buf = np.zeros(max(count, 1))
        if count:
            n = lib.mo_get(self._h, field.encode(), _dp(buf), count)
            assert n == count, (field, n)
        return buf[:count].copy() if field in _VECTORS else float(buf[0])

    def build_luts(self):
        lib.mo_build_luts(self._h)

    def get_lut(self, camera, outer=False, cap=8192):
        phis, rhos = np.zeros(cap), np.zeros(cap)
        n = lib.mo_get_lut(self._h, camera, int(outer), _dp(phis), _dp(rhos), cap)
        assert n >= 0
        return phis[:n].copy(), rhos[:n].copy()

    def set_lut(self, camera, phis, rhos, outer=False):
        phis = np.ascontiguousarray(phis, dtype=np.float64)
        rhos = np.ascontiguousarray(rhos, dtype=np.float64)
        lib.mo_set_lut(self._h, camera, int(outer), _dp(phis), _dp(rhos), len(phis))

    def update_view(self, tape_ct=None):
        if tape_ct is not None:
            tape_ct = np.ascontiguousarray(tape_ct, dtype=np.float64)
        lib.mo_update_view(self._h, _dp(tape_ct) if tape_ct is not None and tape_ct.size else None)

    def step(self, cam_act, tgt_act, tape_ct=None, goal_u=None):
        cam_act = np.ascontiguousarray(cam_act, dtype=np.float64).reshape(-1)
        tgt_act = np.ascontiguousarray(tgt_act, dtype=np.float64).reshape(-1)
        if cam_act.size == 0:
            cam_act = np.zeros(2)
        tp = gp = None
        if tape_ct is not None:
            tape_ct = np.ascontiguousarray(np.nan_to_num(tape_ct, nan=0.0), dtype=np.float64)
            if tape_ct.size == 0:
                tape_ct = np.zeros(1)
            tp = _dp(tape_ct)
        if goal_u is not None:
            goal_u = np.ascontiguousarray(np.nan_to_num(goal_u, nan=0.0), dtype=np.float64)
            gp = _dp(goal_u)
        lib.mo_step(self._h, _dp(cam_act), _dp(tgt_act), tp, gp)

    def observe(self):
        cam = np.zeros((max(self.Nc, 1), self.Dc))
        tgt = np.zeros((self.Nt, self.Dt))
        lib.mo_observe(self._h, _dp(cam), _dp(tgt))
        return cam[:self.Nc], tgt

    MODES = {'plain': 0, 'enhanced': 1, 'shared': 2}

    def observe_mode(self, camera='plain', target='plain'):
        """joint_observation() followed by the reference's EnhancedObservation / SharedFieldOfView per team."""
        cam = np.zeros((max(self.Nc, 1), self.Dc))
        tgt = np.zeros((self.Nt, self.Dt))
        lib.mo_observe_mode(self._h, self.MODES[camera], self.MODES[target], _dp(cam), _dp(tgt))
        return cam[:self.Nc], tgt

    def decode_discrete(self, cam_idx, cam_grid, tgt_idx, tgt_grid):
        """DiscreteCamera / DiscreteTarget: grid indices -> continuous joint actions."""
        ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))  # noqa: E731
        cam_act, tgt_act = np.zeros((max(self.Nc, 1), 2)), np.zeros((self.Nt, 2))
        ci = np.ascontiguousarray(cam_idx, dtype=np.int32) if cam_idx is not None and self.Nc else None
        ti = np.ascontiguousarray(tgt_idx, dtype=np.int32) if tgt_idx is not None else None
        cg = np.ascontiguousarray(cam_grid, dtype=np.float64) if ci is not None else None
        tg = np.ascontiguousarray(tgt_grid, dtype=np.float64) if ti is not None else None
        lib.mo_decode_discrete(self._h, ip(ci) if ci is not None else None, _dp(cg) if cg is not None else None,
                               ip(ti) if ti is not None else None, _dp(tg) if tg is not None else None, _dp(cam_act), _dp(tgt_act))
        return cam_act[:self.Nc], tgt_act

    def soft_coverage(self):
        """AuxiliaryCameraRewards: (score matrix [Nc, Nt], per-camera soft coverage scores [Nc]) of the current view."""
        matrix, scores = np.zeros((max(self.Nc, 1), self.Nt)), np.zeros(max(self.Nc, 1))
        lib.mo_soft_coverage(self._h, _dp(matrix), _dp(scores))
        return matrix[:self.Nc], scores[:self.Nc]

    def state(self):
        out = np.zeros(self.S)
        lib.mo_state(self._h, _dp(out))
        return out

    def reset(self):
        lib.mo_reset(self._h)

    def reset_tape(self, tape, tape_ct=None):
        """reset() consuming the uniforms recorded from the reference (tests/golden/reset_*.npz); returns the number of
        draws consumed (-1: the tape ran out)."""
        tape = np.ascontiguousarray(tape, dtype=np.float64)
        tp = None
        if tape_ct is not None and self.Nc:
            tape_ct = np.ascontiguousarray(np.nan_to_num(tape_ct, nan=0.0), dtype=np.float64)
            tp = _dp(tape_ct)
        return lib.mo_reset_tape(self._h, _dp(tape if tape.size else np.zeros(1)), tape.size, tp)

    def update_view_reset(self):
        lib.mo_update_view_reset(self._h)


class OracleBatch:
    """N oracle environments stepped with OpenMP (cpu_baseline, GPU parity at scale)."""

    def __init__(self, prototype, n, seed=0, first_env_index=0):
        self.n = int(n)
        self.proto = prototype
        self._h = lib.mo_batch_create(prototype._h, self.n, int(seed), int(first_env_index))

    def __del__(self):
        if getattr(self, '_h', None) and lib is not None:
            lib.mo_batch_destroy(self._h)
            self._h = None

    def env(self, i):
        return OracleEnv(self.proto.Nc, self.proto.Nt, self.proto.No, handle=lib.mo_batch_env(self._h, int(i)), owner=False)

    def reset(self, threads=1):
        lib.mo_batch_reset(self._h, int(threads))

    def step(self, cam_act=None, tgt_act=None, auto_reset=True, threads=1):
        cp = tp = None
        if cam_act is not None:
            cam_act = np.ascontiguousarray(cam_act, dtype=np.float32)
            cp = _fp(cam_act)
        if tgt_act is not None:
            tgt_act = np.ascontiguousarray(tgt_act, dtype=np.float32)
            tp = _fp(tgt_act)
        lib.mo_batch_step(self._h, cp, tp, int(auto_reset), int(threads))

    def observe(self, threads=1):
        p = self.proto
        cam = np.zeros((self.n, max(p.Nc, 1), p.Dc), dtype=np.float32)
        tgt = np.zeros((self.n, p.Nt, p.Dt), dtype=np.float32)
        lib.mo_batch_observe(self._h, _fp(cam), _fp(tgt), int(threads))
        return cam[:, :p.Nc], tgt

    def update_view_reset(self):
        """Recompute every environment's first view of the episode (after its occlusion tables were replaced)."""
        for i in range(self.n):
            lib.mo_update_view_reset(lib.mo_batch_env(self._h, i))

    def gather(self, field):
        """`field` of every environment, stacked: [n, ...] (same values as env(i).get(field))."""
        p = self.proto
        if field in _PADDED:
            stride, shape = _PADDED[field]
            rows, cols = shape(p)
            buf = np.zeros((self.n, max(rows * stride, 1)))
            if rows:
                assert lib.mo_batch_get(self._h, field.encode(), _dp(buf), rows * stride) == self.n, field
            return buf[:, :rows * stride].reshape(self.n, rows, stride)[:, :, :cols].copy()
        count = p._count(_VECTORS[field]) if field in _VECTORS else 1
        buf = np.zeros((self.n, max(count, 1)))
        if count:
            assert lib.mo_batch_get(self._h, field.encode(), _dp(buf), count) == self.n, field
        return buf[:, :count].copy() if field in _VECTORS else buf[:, 0].copy()
