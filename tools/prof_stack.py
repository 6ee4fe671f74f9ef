import collections, importlib, os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.getcwd())
S_ = importlib.import_module("robust-dynrf_amd.step")
cfg = S_.scene_config("nvidia", "stage0")
tr = S_.Trainer(cfg, torch.device("cuda", 0), dead_work=True)
for _ in range(3):
    tr.step(); tr.finish_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(); tr.finish_step()
    torch.cuda.synchronize()
by = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CPU and e.name in ("aten::fill_", "aten::cat", "aten::add", "aten::add_", "aten::index", "aten::copy_", "aten::sub", "aten::zero_", "aten::zeros", "aten::zeros_like") and e.kernels:
        st = [f for f in (e.stack or []) if "robust-dynrf" in f or "bench" in f]
        key = (e.name, st[0].split("/")[-1][:60] if st else "(no py frame: autograd engine)", str(e.input_shapes)[:50])
        by[key] += len(e.kernels)
for k, v in by.most_common(60):
    print(v, k)
