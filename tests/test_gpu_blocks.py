"""Observation blocks of the fused rollouts: the scattered allocator behind ``mate_engine_block_alloc`` and the line-aligned
row stores of the row-image kernels (a block may begin at any multiple of 16 bytes)."""
import ctypes
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_scattered_block_is_ordinary_device_memory():
    """block_alloc hands out a contiguous virtual range (2 MiB chunks mapped in a shuffled order): a torch view over it reads
    back what was written, a second block does not alias the first, block_free refuses a foreign pointer."""
    from mate_amd import _native
    nbytes = 5 * (1 << 20) + 4096                   # not a multiple of the chunk size
    a = _native.ScatteredBlock(0, nbytes)
    b = _native.ScatteredBlock(0, nbytes)
    assert a.ptr % (2 << 20) == 0 and a.ptr != b.ptr
    ta, tb = a.tensor(torch.int32, (nbytes // 4,)), b.tensor(torch.int32, (nbytes // 4,))
    ref = torch.arange(nbytes // 4, dtype=torch.int32, device='cuda')
    ta.copy_(ref); tb.copy_(-ref)
    torch.cuda.synchronize()
    assert torch.equal(ta, ref) and torch.equal(tb, -ref)
    assert ta.data_ptr() == a.ptr
    lib = _native.load()
    assert lib.mate_engine_block_free(ctypes.c_void_p(ta.data_ptr() + 4096)) != 0
    assert b'block_alloc' in lib.mate_engine_last_error()
    del a                                            # the tensor keeps the block alive; freeing happens with the last reference
    import gc
    gc.collect()
    ta.add_(1)
    torch.cuda.synchronize()
    assert torch.equal(ta, ref + 1)
    del ta
    gc.collect()
    torch.cuda.synchronize()
    assert torch.equal(tb, -ref)


@pytest.mark.parametrize('workload,n', [('MATE-4v8-9.yaml', 203), ('MATE-4v8-0.yaml', 66)])
def test_rollout_rows_do_not_depend_on_the_blocks(workload, n):
    """The same rollout into scattered blocks (the default of Engine.reserve_rollout for blocks of 64 MiB and more), into
    torch's own memory (MATE_PLAIN_BLOCKS=1) and into a caller's blocks that begin 16 .. 112 bytes into a cache line (any
    16-byte-aligned device pointer is a valid block; the line-aligned form of the row stores, mate_engine_set_store_form,
    derives its lane shifts from the base and the row index): every row bit for bit, and nothing written outside the rows."""
    from mate_amd._native import MateStepIO, check
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    cfg = read_config(workload, max_episode_steps=13)
    steps = 24

    def run(mode):
        os.environ['MATE_PLAIN_BLOCKS'] = '1' if mode == 'plain' else '0'
        try:
            eng = Engine(cfg, n, seed=5)
            eng.reset()
            L = eng.layout
            big = (steps * n * eng.num_cameras * L.camera_obs_dim * 4) >= (64 << 20)
            if isinstance(mode, int):            # a caller's own blocks, `mode` 16-byte chunks into a cache line, guard words around them
                check(eng.lib.mate_engine_set_store_form(eng._h, mode & 1))      # (odd offsets: the line-aligned form of the row stores)
                cam_elems, tgt_elems = n * eng.num_cameras * L.camera_obs_dim, n * eng.num_targets * L.target_obs_dim
                pad = 64
                cam_raw = torch.full((steps * cam_elems + 2 * pad,), -3.0, dtype=torch.float32, device='cuda')
                tgt_raw = torch.full((steps * tgt_elems + 2 * pad,), -3.0, dtype=torch.float32, device='cuda')
                cam = cam_raw[pad + 4 * mode: pad + 4 * mode + steps * cam_elems]
                tgt = tgt_raw[pad + 4 * ((mode + 3) % 8): pad + 4 * ((mode + 3) % 8) + steps * tgt_elems]
                sc = torch.zeros((steps, n, 8), dtype=torch.float32, device='cuda')
                io = MateStepIO()
                io.camera_obs_dev, io.target_obs_dev, io.scalars_dev, io.masks_dev = cam.data_ptr(), tgt.data_ptr(), sc.data_ptr(), None
                out = []
                for _ in range(2):
                    cam.fill_(-3.0); tgt.fill_(-3.0)      # (rows of idle slots are not written: the same fill everywhere)
                    check(eng.lib.mate_engine_rollout_random(eng._h, ctypes.byref(io), steps, 1, eng._stream()))
                    torch.cuda.synchronize()
                    out.append((cam.clone(), tgt.clone(), sc.clone()))
                for raw, view in ((cam_raw, cam), (tgt_raw, tgt)):
                    first = (view.data_ptr() - raw.data_ptr()) // 4
                    assert (raw[:first] == -3.0).all() and (raw[first + view.numel():] == -3.0).all()
                return out, eng.export_state().clone(), big
            out = []
            for _ in range(2):
                buf = eng.reserve_rollout(steps)
                buf['camera_obs'].fill_(-3.0); buf['target_obs'].fill_(-3.0)
                c, t, s = eng.rollout_random(steps, auto_reset=True)
                torch.cuda.synchronize()
                out.append((c.reshape(-1).clone(), t.reshape(-1).clone(), s.clone()))
            return out, eng.export_state().clone(), big
        finally:
            os.environ.pop('MATE_PLAIN_BLOCKS', None)

    ref, ref_state, _ = run('plain')
    for mode in ('scattered', 1, 2, 5, 7):
        got, state, _ = run(mode)
        for (c0, t0, s0), (c1, t1, s1) in zip(ref, got):
            assert torch.equal(c0.view(torch.int32), c1.view(torch.int32)), mode
            assert torch.equal(t0.view(torch.int32), t1.view(torch.int32)), mode
            assert torch.equal(s0, s1), mode
        assert torch.equal(ref_state, state), mode


def test_reserve_rollout_uses_scattered_blocks_for_large_rollouts():
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    eng = Engine(read_config('MATE-4v8-9.yaml'), 2048, seed=0)
    buf = eng.reserve_rollout(32)                   # 2048 x 32 x 4192 B = 262 MiB of target rows
    assert buf['target_obs'].data_ptr() % (2 << 20) == 0
    eng.reset()
    cam, tgt, sc = eng.rollout_random(32, auto_reset=True)
    torch.cuda.synchronize()
    assert torch.isfinite(cam).all() and torch.isfinite(tgt).all() and (sc[..., 2] != 2).all()
    eng._rollout = None                              # frees the blocks (after a device synchronise)
    del buf, cam, tgt
    torch.cuda.synchronize()


def test_block_probe_and_candidate_search():
    """block_probe reports a plausible store rate, leaves zeros behind and refuses rows that are no multiple of 16 bytes;
    reserve_rollout's search over candidates (blocks of a GiB and more) records the rates it saw."""
    from mate_amd import _native
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    rows, row_bytes, steps = 1024, 4192, 64
    blk = _native.ScatteredBlock(0, rows * row_bytes * steps + 4096)
    t = blk.tensor(torch.uint8, (blk.nbytes,))
    t.fill_(7)
    rate = blk.store_rate(rows, row_bytes)
    torch.cuda.synchronize()
    assert 200.0 < rate < 9000.0 and int(t.max()) == 0
    with pytest.raises(_native.EngineError):
        blk.store_rate(rows, row_bytes + 4)
    os.environ['MATE_BLOCK_CANDIDATES'] = '2'
    try:
        eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
        eng.reserve_rollout(5)                       # 4096 x 5 x 4192 B = 82 MiB of target rows: not searched
        assert [len(r) for r in eng.block_rates] == [0, 0] and eng._rollout['target_obs'].data_ptr() % (2 << 20) == 0
        eng.reserve_rollout(64)                      # 1.02 GiB of target rows and 0.5 GiB of camera rows: both searched
    finally:
        os.environ.pop('MATE_BLOCK_CANDIDATES', None)
    # (block_rates = [target block's candidates, camera block's]; the target block's search may end at the first candidate if that one is fast)
    assert eng.reserve_seconds > 0.0
    assert len(eng.block_rates) == 2 and len(eng.block_rates[0]) in (1, 2) and len(eng.block_rates[1]) == 2
    assert all(200.0 < x < 9000.0 for r in eng.block_rates for x in r)
    eng.reset()
    cam, tgt, sc = eng.rollout_random(64, auto_reset=True)
    torch.cuda.synchronize()
    assert torch.isfinite(tgt).all() and float(tgt.abs().sum()) > 0.0


def test_rows_survive_block_churn():
    """Blocks allocated and freed over and over, then a fused rollout into fresh ones: every row arrives (a virtual range handed
    out again after a free lost rows of the last step -- translations of a virtual address outlive hipMemUnmap on this driver,
    tools/va_reuse.hip -- so block_free releases the memory and keeps the range reserved: no range is ever mapped twice)."""
    from mate_amd import _native
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    for i in range(24):
        blk = _native.ScatteredBlock(0, (96 + 16 * (i % 7)) << 20)
        blk.store_rate(4096, 4192)
        del blk
    cfg = read_config('MATE-8v8-9.yaml', max_episode_steps=50)
    outs = []
    for plain in ('1', '0'):
        os.environ['MATE_PLAIN_BLOCKS'] = plain
        try:
            eng = Engine(cfg, 8192, seed=13)
            eng.reset()
            cam, tgt, sc = eng.rollout_random(5, auto_reset=False)      # 8192 x 5 x 5088 B = 199 MiB of target rows: searched
            torch.cuda.synchronize()
            outs.append((cam.clone(), tgt.clone(), sc.clone()))
        finally:
            os.environ.pop('MATE_PLAIN_BLOCKS', None)
        del eng
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert float(outs[0][1][-1, -1].abs().sum()) > 0.0


def test_block_free_returns_the_memory():
    """Blocks allocated and freed: the device's free memory comes back (every chunk unmapped and released; the address range stays
    reserved), also after the candidate search of reserve_rollout, which frees everything but the winners."""
    import gc
    from mate_amd import _native
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    base = torch.cuda.mem_get_info(0)[0]
    lib = _native.load()
    for i in range(6):
        blk = _native.ScatteredBlock(0, (256 + 64 * i) << 20)
        blk.tensor(torch.uint8, (blk.nbytes,)).fill_(3)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] <= base - (200 << 20)
        ptr, blk.ptr = blk.ptr, None                 # free it here, with the status checked (__del__ only warns)
        assert lib.mate_engine_block_free(ctypes.c_void_p(ptr)) == 0, lib.mate_engine_last_error()
        del blk
    assert abs(torch.cuda.mem_get_info(0)[0] - base) <= (64 << 20)
    os.environ['MATE_BLOCK_CANDIDATES'] = '3'
    try:
        eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
    finally:
        os.environ.pop('MATE_BLOCK_CANDIDATES', None)
    torch.cuda.synchronize()
    before = torch.cuda.mem_get_info(0)[0]
    buf = eng.reserve_rollout(64)                    # 1.02 GiB + 0.5 GiB, up to three candidates each
    kept = sum(buf[k].numel() * buf[k].element_size() for k in ('camera_obs', 'target_obs', 'scalars'))
    gc.collect()
    torch.cuda.synchronize()
    assert before - torch.cuda.mem_get_info(0)[0] <= kept + (96 << 20)      # the losers of the search are gone
    eng._rollout = None
    del buf, eng
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    assert abs(torch.cuda.mem_get_info(0)[0] - base) <= (128 << 20)


def test_a_failed_block_alloc_leaves_no_sticky_error_and_the_fallback_runs():
    """mate_engine_block_alloc fails (a block larger than the device): the error comes back through the return code ONLY -- the
    next kernel launch check, the engine's own and torch's, must not see it -- and Engine.reserve_rollout falls back to
    torch.zeros with a warning; the rollout into the fallback buffers equals the one into scattered blocks."""
    from mate_amd import _native
    from mate_amd.config import read_config
    from mate_amd.engine import Engine
    lib = _native.load()
    ptr = ctypes.c_void_p()
    assert lib.mate_engine_block_alloc(0, 1 << 42, ctypes.byref(ptr)) != 0 and not ptr.value      # 4 TiB
    torch.zeros(16, device='cuda').add_(1)           # torch's launch check
    torch.cuda.synchronize()
    cfg = read_config('MATE-4v8-9.yaml', max_episode_steps=30)
    outs = []
    for broken in (False, True):
        eng = Engine(cfg, 1024, seed=3)
        eng.reset()
        if broken:
            real = _native.ScatteredBlock

            def failing(device_index, nbytes):
                return real(device_index, 1 << 42)
            _native.ScatteredBlock = failing
        try:
            with (pytest.warns(UserWarning, match='block_alloc failed') if broken else _nullcontext()):
                cam, tgt, sc = eng.rollout_random(40, auto_reset=True)      # 1024 x 40 x 4192 B = 164 MiB of target rows
        finally:
            if broken:
                _native.ScatteredBlock = real
        torch.cuda.synchronize()
        assert (eng._rollout['target_obs'].data_ptr() % (2 << 20) == 0) or broken
        outs.append((cam.clone(), tgt.clone(), sc.clone(), eng.export_state().clone()))
    for x, y in zip(*outs):
        assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False
