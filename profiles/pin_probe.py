import torch, time
buf = torch.empty((4, 8*40000*4), dtype=torch.float32).share_memory_()
rc = torch.cuda.cudart().cudaHostRegister(buf.data_ptr(), buf.numel()*4, 0)
print("rc", int(rc), "is_pinned", buf.is_pinned(), buf[1].is_pinned())
pin = torch.empty((8*40000*4,), dtype=torch.float32).pin_memory()
page = torch.empty((8*40000*4,), dtype=torch.float32)
for name, src in (("registered shm", buf[1]), ("pinned", pin), ("pageable", page)):
    for _ in range(3): d = src.to("cuda", non_blocking=True)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(20): d = src.to("cuda", non_blocking=True)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print("%-16s host %.3f ms per copy, total %.3f ms per copy (%.1f MB)" % (name, (t1-t0)/20*1e3, (t2-t0)/20*1e3, src.numel()*4/1e6))
