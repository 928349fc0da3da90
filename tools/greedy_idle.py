import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from mate_amd.config import read_config
from mate_amd.engine import Engine
for n in (4096, 8192):
    eng = Engine(read_config('MATE-8v8-9.yaml'), n, seed=0)
    eng.enable_policies(); eng.reset()
    for k in range(6):
        i0 = eng.idle_steps()
        _, _, sc = eng.rollout_greedy(32)
        torch.cuda.synchronize()
        sd = eng.state_dict()
        print(n, 'launch', k, 'idle slots', eng.idle_steps() - i0, 'rows with done=2', int((sc[:, :, 2] == 2).sum()), 'done rows', int((sc[:, :, 2] == 1).sum()),
              'episodes', int(sd['episode'].min()), int(sd['episode'].max()), 'ep_step range', int(sd['episode_step'].min()), int(sd['episode_step'].max()))
    del eng
