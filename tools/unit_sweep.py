"""Debug: kernel time of configs[1] vs the work-unit schedule (rt_debug_set_work_units). Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (device memory for the accumulator)
from raytracing_amd import api
ctx = api.Context(0)
w = api.World("FinalStatic", seed=1, aspect=1.5)
ctx.upload(w.scene)
acc = torch.empty(800, 1200, 3, dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
p = api.make_params(1200, 800, 1000, seed=w.render_seed)
sched = os.environ.get('SCHED')
scheds = [tuple(int(v) for v in t.split(':')) for t in sched.split(',')] if sched else [(0, 0, -1), (48, 8, 128), (96, 8, 128), (96, 16, 192), (192, 8, 192), (64, 16, 128), (128, 16, 256), (218, 8, 128), (0, 0, -1)]
for a, b, span in scheds:
    ctx.debug_set_work_units(a, b, span)
    ctx.render_accum_device(w.camera, p, acc.data_ptr())
    ms = min(ctx.render_accum_device(w.camera, p, acc.data_ptr()).kernel_ms for _ in range(3))
    print(f"A={a} B={b} span={span}: kernel_ms {ms:.2f}", flush=True)
