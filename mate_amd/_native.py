"""ctypes binding of ``mate_amd/lib/libmate_engine.so`` (C ABI: ``include/mate_engine.h``).

There is no CPU fallback: if the HIP library is missing or fails to load this
module raises, and every engine call that returns a non-zero status raises
``EngineError`` carrying ``mate_engine_last_error()``.
"""
import ctypes
import os

__all__ = ['lib', 'load', 'EngineError', 'MateConfig', 'MateLayout', 'MateStepIO', 'MatePolicyTape', 'LIB_PATH', 'check', 'EXPORTED_SYMBOLS']

LIB_PATH = os.environ.get('MATE_ENGINE_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libmate_engine.so')

EXPORTED_SYMBOLS = (
    'mate_engine_last_error', 'mate_engine_abi_version', 'mate_engine_create', 'mate_engine_destroy',
    'mate_engine_get_layout', 'mate_engine_set_obs_transform', 'mate_engine_set_obs_mode', 'mate_engine_set_action_grids', 'mate_engine_seed', 'mate_engine_reset', 'mate_engine_reset_tape', 'mate_engine_step', 'mate_engine_device_tick', 'mate_engine_set_episode_stats', 'mate_engine_snapshot_episode_stats', 'mate_engine_step_random',
    'mate_engine_rollout_random', 'mate_engine_policy_enable', 'mate_engine_step_greedy', 'mate_engine_step_versus_greedy', 'mate_engine_rollout_greedy', 'mate_engine_rollout_versus_greedy', 'mate_engine_policy_actions',
    'mate_engine_observe', 'mate_engine_export_state', 'mate_engine_import_state', 'mate_engine_lut_read',
    'mate_engine_block_alloc', 'mate_engine_block_free', 'mate_engine_block_probe', 'mate_engine_set_store_form',
    'mate_engine_memory_hold', 'mate_engine_memory_release', 'mate_engine_hbm_probe', 'mate_engine_set_sub_wave',
    'mate_engine_lut_write', 'mate_engine_enable_outer_boundary', 'mate_engine_lut_read_outer', 'mate_engine_lut_write_outer', 'mate_engine_soft_coverage', 'mate_engine_rebuild_luts', 'mate_engine_idle_steps', 'mate_engine_kernel_time', 'mate_engine_last_flow',
)


class EngineError(RuntimeError):
    """A C-ABI call failed (status code + message from the engine)."""

    def __init__(self, code, message):
        super().__init__(f'mate_engine error {code}: {message}')
        self.code = code


class MateConfig(ctypes.Structure):
    _fields_ = [
        ('num_cameras', ctypes.c_int32), ('num_targets', ctypes.c_int32), ('num_obstacles', ctypes.c_int32),
        ('max_episode_steps', ctypes.c_int32), ('sparse_reward', ctypes.c_int32),
        ('num_cargoes_per_target', ctypes.c_int32), ('shuffle_entities', ctypes.c_int32),
        ('targets_start_with_cargoes', ctypes.c_int32),
        ('high_capacity_target_split', ctypes.c_double), ('bounty_factor', ctypes.c_double),
        ('transmittance', ctypes.c_double),
        ('camera_radius', ctypes.c_double), ('camera_min_viewing_angle', ctypes.c_double),
        ('camera_max_sight_range', ctypes.c_double), ('camera_rotation_step', ctypes.c_double),
        ('camera_zooming_step', ctypes.c_double),
        ('target_step_size', ctypes.c_double), ('target_sight_range', ctypes.c_double),
        ('obstacle_radius_range', ctypes.c_double * 2),
        ('camera_location_ranges', ctypes.POINTER(ctypes.c_double)),
        ('target_location_ranges', ctypes.POINTER(ctypes.c_double)),
        ('obstacle_location_ranges', ctypes.POINTER(ctypes.c_double)),
        ('obs_dtype', ctypes.c_int32),
    ]


class MateLayout(ctypes.Structure):
    _fields_ = [(name, ctypes.c_int32) for name in (
        'camera_obs_dim', 'target_obs_dim', 'state_dim', 'mask_words', 'bit_camera_target', 'bit_camera_camera',
        'bit_target_row', 'bit_camera_obstacle', 'export_width', 'lut_capacity', 'scalars_per_env', 'specialised')]


class MatePolicyTape(ctypes.Structure):
    _fields_ = [(name, ctypes.c_void_p) for name in (
        'camera_resample_u_dev', 'camera_sample_u_dev', 'camera_delay_dev', 'target_choice_u_dev',
        'target_resample_u_dev', 'target_sample_u_dev', 'target_reset_sample_u_dev')]


class MateStepIO(ctypes.Structure):
    _fields_ = [
        ('camera_actions_dev', ctypes.c_void_p), ('target_actions_dev', ctypes.c_void_p), ('act_dtype', ctypes.c_int32),
        ('tape_camera_target_dev', ctypes.c_void_p), ('tape_goal_dev', ctypes.c_void_p),
        ('camera_obs_dev', ctypes.c_void_p), ('target_obs_dev', ctypes.c_void_p),
        ('scalars_dev', ctypes.c_void_p), ('masks_dev', ctypes.c_void_p),
    ]


lib = None


def load():
    """Load the HIP engine; raises if it has not been built (python -m mate_amd.build)."""
    global lib
    if lib is not None:
        return lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f'{LIB_PATH} is missing: build the HIP engine first (python -m mate_amd.build or '
            f'__graft_entry__.build()).  mate_amd has no CPU fallback.')
    # torch first: the engine must bind to the HIP runtime torch already loaded (one runtime per
    # process; loading ROCm's copy before torch's leaves the later one without a device).
    import torch  # noqa: F401
    handle = ctypes.CDLL(LIB_PATH)
    P, I32, I64, U64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
    handle.mate_engine_last_error.restype = ctypes.c_char_p
    handle.mate_engine_last_error.argtypes = []
    handle.mate_engine_abi_version.restype = ctypes.c_int
    handle.mate_engine_create.argtypes = [ctypes.POINTER(MateConfig), I64, I32, U64, U64, ctypes.POINTER(P)]
    handle.mate_engine_destroy.argtypes = [P]
    handle.mate_engine_get_layout.argtypes = [P, ctypes.POINTER(MateLayout)]
    handle.mate_engine_seed.argtypes = [P, U64]
    handle.mate_engine_set_obs_transform.argtypes = [P, I32, P, P, P, P]
    handle.mate_engine_set_obs_mode.argtypes = [P, I32, I32]
    handle.mate_engine_set_action_grids.argtypes = [P, P, I32, P, I32]
    handle.mate_engine_reset.argtypes = [P, P, ctypes.POINTER(MateStepIO), P]
    handle.mate_engine_reset_tape.argtypes = [P, P, ctypes.POINTER(MateStepIO), P, I32, P, P]
    handle.mate_engine_step.argtypes = [P, ctypes.POINTER(MateStepIO), I32, P]
    handle.mate_engine_device_tick.argtypes = [P, I32, P]
    handle.mate_engine_set_episode_stats.argtypes = [P, P]
    handle.mate_engine_step_random.argtypes = [P, ctypes.POINTER(MateStepIO), I32, P]
    handle.mate_engine_rollout_random.argtypes = [P, ctypes.POINTER(MateStepIO), I32, I32, P]
    handle.mate_engine_policy_enable.argtypes = [P]
    handle.mate_engine_step_greedy.argtypes = [P, ctypes.POINTER(MateStepIO), ctypes.POINTER(MatePolicyTape), I32, P]
    handle.mate_engine_step_versus_greedy.argtypes = [P, I32, ctypes.POINTER(MateStepIO), ctypes.POINTER(MatePolicyTape), I32, P]
    handle.mate_engine_policy_actions.argtypes = [P, P, P, P]
    handle.mate_engine_rollout_greedy.argtypes = [P, ctypes.POINTER(MateStepIO), I32, I32, P]
    handle.mate_engine_rollout_versus_greedy.argtypes = [P, I32, ctypes.POINTER(MateStepIO), I32, I32, P]
    handle.mate_engine_observe.argtypes = [P, ctypes.POINTER(MateStepIO), P]
    handle.mate_engine_export_state.argtypes = [P, P, P]
    handle.mate_engine_import_state.argtypes = [P, P, P]
    handle.mate_engine_lut_read.argtypes = [P, I64, I32, P, P, I32, ctypes.POINTER(I32)]
    handle.mate_engine_lut_read_outer.argtypes = [P, I64, I32, P, P, I32, ctypes.POINTER(I32)]
    handle.mate_engine_enable_outer_boundary.argtypes = [P, ctypes.POINTER(I32)]
    handle.mate_engine_lut_write.argtypes = [P, I64, I32, P, P, I32]
    handle.mate_engine_lut_write_outer.argtypes = [P, I64, I32, P, P, I32]
    handle.mate_engine_soft_coverage.argtypes = [P, P, P, P, P]
    handle.mate_engine_rebuild_luts.argtypes = [P, P]
    handle.mate_engine_idle_steps.argtypes = [P, ctypes.POINTER(I64)]
    handle.mate_engine_kernel_time.argtypes = [P, I32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(I64)]
    handle.mate_engine_last_flow.argtypes = [P]
    handle.mate_engine_block_alloc.argtypes = [I32, I64, ctypes.POINTER(P)]
    handle.mate_engine_block_free.argtypes = [P]
    handle.mate_engine_set_store_form.argtypes = [P, I32]
    handle.mate_engine_memory_hold.argtypes = [I32, I64, ctypes.POINTER(P)]
    handle.mate_engine_memory_release.argtypes = [P]
    handle.mate_engine_block_probe.argtypes = [I32, P, I64, I32, I32, P, ctypes.POINTER(ctypes.c_double)]
    handle.mate_engine_snapshot_episode_stats.argtypes = [P, P, P]
    handle.mate_engine_set_sub_wave.argtypes = [P, I32, ctypes.POINTER(I32)]
    handle.mate_engine_hbm_probe.argtypes = [I32, P, P, I64, I32, P, ctypes.POINTER(ctypes.c_double)]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(handle, name)
        if name not in ('mate_engine_last_error',):
            fn.restype = ctypes.c_int
    lib = handle
    return lib


class HeldMemory:
    """Physical device memory taken and held, unmapped (``mate_engine_memory_hold``): a spacer between two block candidates."""

    def __init__(self, device_index, nbytes):
        token = ctypes.c_void_p()
        check(load().mate_engine_memory_hold(int(device_index), int(nbytes), ctypes.byref(token)))
        self.token, self._lib = token.value, lib

    def __del__(self):
        token, self.token = getattr(self, 'token', None), None
        if token:
            try:
                self._lib.mate_engine_memory_release(ctypes.c_void_p(token))
            except Exception:
                pass


class ScatteredBlock:
    """Device memory from ``mate_engine_block_alloc`` (2 MiB physical chunks mapped in a shuffled order: what the fused
    rollouts store fastest into, include/mate_engine.h) as a zero-copy torch tensor: ``tensor(dtype, shape)``.  The memory
    lives as long as any tensor made from it."""

    def __init__(self, device_index, nbytes):
        ptr = ctypes.c_void_p()
        check(load().mate_engine_block_alloc(int(device_index), int(nbytes), ctypes.byref(ptr)))
        self.ptr, self.nbytes, self.device_index, self._lib = ptr.value, int(nbytes), int(device_index), lib
        self.__cuda_array_interface__ = {'shape': (self.nbytes,), 'typestr': '|u1', 'data': (self.ptr, False), 'version': 2}

    def store_rate(self, rows_per_step, row_bytes, stream=None):
        """GB/s this block takes in the fused rollouts' store pattern (mate_engine_block_probe); leaves it zeroed."""
        rate = ctypes.c_double()
        check(self._lib.mate_engine_block_probe(self.device_index, ctypes.c_void_p(self.ptr), self.nbytes, int(rows_per_step), int(row_bytes),
                                                stream, ctypes.byref(rate)))
        return rate.value

    def tensor(self, dtype, shape):
        import torch
        flat = torch.as_tensor(self, device=torch.device('cuda', self.device_index))       # keeps a reference to this object
        return flat.view(dtype).view(shape)

    def __del__(self):
        ptr, self.ptr = getattr(self, 'ptr', None), None
        if ptr:
            try:
                import torch
                torch.cuda.synchronize(self.device_index)      # block_free does not wait for launches in flight
                status = self._lib.mate_engine_block_free(ctypes.c_void_p(ptr))
            except Exception:       # interpreter shutdown: the process's memory goes with it
                return
            if status != 0:         # (a finaliser cannot raise: say that device memory may not have come back)
                import warnings
                warnings.warn(f'mate_engine_block_free failed ({status}): {self._lib.mate_engine_last_error().decode()}', RuntimeWarning)


def hbm_rates(device_index, gib=1.0):
    """GB/s of this GPU under the library's own streaming kernels (``mate_engine_hbm_probe``): ``{'copy': read + write bytes of a
    copy, 'fill': write-only (the faster of non-temporal and plain stores), 'read': read-only}`` over two ``gib``-sized buffers, each the
    median of five launches."""
    import torch
    n = int(gib * (1 << 30)) // 16 * 16
    out = {}
    with torch.cuda.device(device_index):
        a = torch.zeros(n, dtype=torch.uint8, device='cuda')
        b = torch.zeros(n, dtype=torch.uint8, device='cuda')
        torch.cuda.synchronize()
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        for name, mode in (('copy', 0), ('fill', 1), ('read', 2), ('fill', 3)):
            rate = ctypes.c_double()
            check(load().mate_engine_hbm_probe(int(device_index), ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), n, mode, stream, ctypes.byref(rate)))
            out[name] = max(out.get(name, 0.0), rate.value)
        del a, b
        torch.cuda.empty_cache()
    return out


def check(status):
    if status != 0:
        raise EngineError(status, load().mate_engine_last_error().decode())
