import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zoomearth_amd.config import ModelConfig
from zoomearth_amd.engine import Engine
e = Engine(ModelConfig.zoomearth_3b(), max_seqs=1, max_ctx=2048, max_patches=2048, max_tile_side=1024)
e.fill_synthetic(0)
for rnd in range(3):
    for v in (0,1,2,3):
        e.lib.ze_tune(0, v); us,b = e.profile_decode_kernel(3, 144); print('down cfg', v, round(us,2), round(b/us/1e3,1),'GB/s')
    for v in (0,1):
        e.lib.ze_tune(1, v); us,b = e.profile_decode_kernel(2, 144); print('gate_up cfg', v, round(us,2), round(b/us/1e3,1),'GB/s')
e.close()
