import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
S_ = importlib.import_module("robust-dynrf_amd.step")
cfg = S_.balloon1_config("stage0")
tr = S_.Trainer(cfg, torch.device("cuda", 0), dead_work=True)
t0 = time.perf_counter()
for i in range(400):
    loss = tr.step(); tr.finish_step()
    if i % 50 == 0 or i == 399:
        torch.cuda.synchronize()
        fin = all(bool(torch.isfinite(p).all()) for m in (tr.st, tr.dy) for p in m.parameters())
        print(i, float(loss.detach()), "finite params:", fin, f"{time.perf_counter()-t0:.1f}s", flush=True)
print("peak mem GB", torch.cuda.max_memory_allocated() / 2**30)
