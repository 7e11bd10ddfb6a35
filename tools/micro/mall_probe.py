"""Does the Infinity Cache keep freshly WRITTEN data for the next kernel?  write S bytes (fill), then read them (sum);
compare with the same read after 1 GiB of unrelated traffic.  python tools/micro/mall_probe.py"""
import torch
dev = torch.device("cuda")
evict = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev)   # 1 GiB
for mb in (32, 64, 128, 192, 256, 384, 512, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    res = {}
    for mode in ("hot", "cold"):
        ts = []
        for rep in range(5):
            a.fill_(1.0)
            if mode == "cold":
                evict.fill_(0.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            s = a.sum()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        res[mode] = min(ts)
    # write timing: fill after a read of the same buffer (resident) vs after eviction
    tw = {}
    for mode in ("hot", "cold"):
        ts = []
        for rep in range(5):
            a.sum()
            if mode == "cold":
                evict.fill_(0.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            a.fill_(2.0)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        tw[mode] = min(ts)
    print("%5d MB  read hot %.0f GB/s cold %.0f GB/s | write hot %.0f GB/s cold %.0f GB/s" % (
        mb, mb / 1024 / res["hot"] * 1e3, mb / 1024 / res["cold"] * 1e3, mb / 1024 / tw["hot"] * 1e3, mb / 1024 / tw["cold"] * 1e3))
