import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(p, open(p).read().strip())
    except Exception as e: print(p, "n/a")
print("loadavg", open("/proc/loadavg").read().strip())
x = torch.randn(32, 256, 32, 32); w = torch.randn(256, 256, 3, 3)
for nt in (1, 4, 8, 16, 32, 64, 128, 256):
    torch.set_num_threads(nt)
    torch.nn.functional.conv2d(x, w, padding=1)
    t0 = time.time()
    for _ in range(2): torch.nn.functional.conv2d(x, w, padding=1)
    dt = (time.time() - t0) / 2
    print("threads %3d: conv 32x256x32x32 3x3 -> %.3f s  (%.1f GFLOP/s)" % (nt, dt, 2 * 32 * 1024 * 2304 * 256 / dt / 1e9))
