#!/usr/bin/env python3
"""Is work enqueued on a stream AFTER a hipGraphLaunch ordered behind the graph's last node?  (kernel, async pinned D2H copy + event,
pageable copy; on the null stream and on a created stream.)"""
import torch

dev = torch.device("cuda")


def probe(stream, label):
    with torch.cuda.stream(stream):
        a = torch.zeros(1 << 24, device=dev)                  # 64 MB
        flag = torch.zeros(1, dtype=torch.int64, device=dev)
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(cap):
            g.capture_begin()
            for _ in range(200):                              # ~ a few ms of dependent work
                a.add_(1.0)
            flag.add_(1)                                      # the graph's LAST node
            g.capture_end()
        torch.cuda.synchronize()
        res = []
        for it in range(5):
            host = torch.full((1,), -7, dtype=torch.int64).pin_memory()
            g.replay()
            seen = flag.clone()                               # eager kernel behind the replay, same stream
            host.copy_(seen, non_blocking=True)               # async D2H into pinned memory
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            ev.synchronize()
            got_event = int(host[0])
            got_item = int(flag.item())                       # torch's own blocking read
            torch.cuda.synchronize()
            res.append((it + 1, got_event, got_item, int(flag.item())))
        print(label, "expected / via kernel+pinned copy+event / via .item() / after device sync:", res, flush=True)


probe(torch.cuda.default_stream(), "null stream   ")
probe(torch.cuda.Stream(), "created stream")
