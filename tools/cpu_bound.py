import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
S_ = importlib.import_module("robust-dynrf_amd.step")
cfg = S_.balloon1_config("stage0")
tr = S_.Trainer(cfg, torch.device("cuda", 0))
for _ in range(3):
    tr.step(); tr.finish_step()
torch.cuda.synchronize()
for K in (10,):
    t0 = time.perf_counter()
    for _ in range(K):
        tr.step(); tr.finish_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0)/K:.2f} ms/step, total {1e3*(t2-t0)/K:.2f} ms/step, drain {1e3*(t2-t1):.2f} ms")
# CPU-only cost: how long does python take when the GPU is not the limiter? use a tiny problem
cfg2 = S_.balloon1_config("stage0"); cfg2.update(batch_size=64)
tr2 = S_.Trainer(cfg2, torch.device("cuda", 0))
for _ in range(3):
    tr2.step(); tr2.finish_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr2.step(); tr2.finish_step()
torch.cuda.synchronize()
print(f"64-ray step (launch-bound): {1e3*(time.perf_counter()-t0)/10:.2f} ms/step")
