"""Dynamic instruction counts per phase: run under
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d DIR -o pmc -- python3 tools/ablate_pmc.py
with MATE_ENGINE_LIB pointing at the -DMATE_PHASE_CLOCKS build; then `python3 tools/ablate_pmc.py DIR` summarises."""
import ctypes, glob, os, sys, csv, collections
MASKS = [(0, 'nothing'), (1, 'draws'), (2, 'cameras'), (4, 'targets'), (8, 'view'), (32, 'assign'), (64, 'scratch'), (128, 'pack'), (255, 'everything')]
REPS = 10
if len(sys.argv) > 1:
    rows = collections.defaultdict(dict)
    for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'step_kernel' in r['Kernel_Name']:
                rows[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(rows)[-REPS * len(MASKS):]
    base = None
    for k, (mask, name) in enumerate(MASKS):
        sel = ids[k * REPS:(k + 1) * REPS]
        avg = {c: sum(rows[i][c] for i in sel) / len(sel) / 4096 for c in rows[sel[0]]}
        base = base or avg
        print(f'skip {name:10s}', ' '.join(f'{c}={v:8.1f} (-{base[c] - v:7.1f})' for c, v in sorted(avg.items())))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
# ABLATE_FLOW=actions: step(actions) with f32 joint actions (the learner's flow) instead of step_random
flow = os.environ.get('ABLATE_FLOW', 'random')
eng = Engine(read_config('MATE-4v8-9.yaml'), 4096, seed=0)
eng.lib.mate_engine_debug_skip.argtypes = [ctypes.c_void_p, ctypes.c_int32]
eng.reset()
cam = (torch.rand((4096, 4, 2), device='cuda') * 2 - 1) * torch.tensor([5.0, 2.5], device='cuda')
tgt = (torch.rand((4096, 8, 2), device='cuda') * 2 - 1) * 20.0
one = (lambda: eng.step_random(auto_reset=False)) if flow == 'random' else (lambda: eng.step(cam, tgt, auto_reset=False))
for _ in range(50):
    one()
for mask, name in MASKS:
    eng.lib.mate_engine_debug_skip(eng._h, mask)
    for _ in range(REPS):
        one()
torch.cuda.synchronize()
