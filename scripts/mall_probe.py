#!/usr/bin/env python3
"""What does the memory-side cache (Infinity Cache, 256 MiB) keep?  Streams a buffer of S MB with scripts/mall_probe.hip's
kernels and reports the READ rate (GB/s) of: a cold buffer (2 GiB of other traffic in front), the buffer just WRITTEN, the buffer
just READ, and -- for a producer / consumer hand-off larger than the cache -- the last-written quarter read first.
Diagnostic for DESIGN.md section 4g (where are the weight-gradient launch's 300 MB of stored operands served from)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "m2_mixer_amd", "libm2mixer_exp_mallprobe.so"))
lib.probe_write_launch.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
lib.probe_read_launch.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
MB = 1 << 20
flush = torch.empty(2048 * MB, dtype=torch.uint8, device=dev)
sink = torch.zeros(16, dtype=torch.int32, device=dev)
NWG = 2048


def st():
    return torch.cuda.current_stream().cuda_stream


def write(buf, nbytes, nt=0, off=0):
    assert lib.probe_write_launch(buf.data_ptr() + off, nbytes, nt, NWG, st()) == 0


def read(buf, nbytes, nt=0, off=0):
    assert lib.probe_read_launch(buf.data_ptr() + off, nbytes, nt, NWG, sink.data_ptr(), st()) == 0


def timed(fn):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3


def rate(nbytes, prep, fn, reps=5):
    ts = []
    for _ in range(reps):
        write(flush, flush.numel())          # 2 GiB of other traffic: whatever the cache held is gone
        prep()
        ts.append(timed(fn))
    ts.sort()
    return nbytes / ts[len(ts) // 2] * 1e-9


def main():
    write(flush, flush.numel()); torch.cuda.synchronize()
    print("size MB | read cold | read after WRITE | read after nt-WRITE | nt-read after WRITE | read after READ | write rate   (GB/s, median of 5)")
    for s in (32, 64, 128, 192, 256, 320, 384, 512):
        n = s * MB
        buf = torch.empty(n, dtype=torch.uint8, device=dev)
        cold = rate(n, lambda: None, lambda: read(buf, n))
        aw = rate(n, lambda: write(buf, n), lambda: read(buf, n))
        antw = rate(n, lambda: write(buf, n, 1), lambda: read(buf, n))
        ntaw = rate(n, lambda: write(buf, n), lambda: read(buf, n, 1))
        ar = rate(n, lambda: read(buf, n), lambda: read(buf, n))
        wr = rate(n, lambda: None, lambda: write(buf, n))
        print(f"{s:7d} | {cold:9.0f} | {aw:16.0f} | {antw:19.0f} | {ntaw:19.0f} | {ar:15.0f} | {wr:10.0f}", flush=True)
        del buf
    # a hand-off larger than the cache: 340 MB written front to back, then read (a) front to back, (b) the LAST 128 MB only,
    # (c) the FIRST 128 MB only
    n = 340 * MB
    buf = torch.empty(n, dtype=torch.uint8, device=dev)
    q = 128 * MB
    print("340 MB written front to back, then:")
    print(f"  all of it, front to back   {rate(n, lambda: write(buf, n), lambda: read(buf, n)):.0f} GB/s")
    print(f"  all of it, nt loads        {rate(n, lambda: write(buf, n), lambda: read(buf, n, 1)):.0f} GB/s")
    print(f"  the LAST 128 MB            {rate(q, lambda: write(buf, n), lambda: read(buf, q, 0, n - q)):.0f} GB/s")
    print(f"  the LAST 128 MB, nt loads  {rate(q, lambda: write(buf, n), lambda: read(buf, q, 1, n - q)):.0f} GB/s")
    print(f"  the FIRST 128 MB           {rate(q, lambda: write(buf, n), lambda: read(buf, q, 0, 0)):.0f} GB/s")


def residency():
    """Does streaming traffic evict a resident buffer?  buf (S MB) is written with plain stores, then X MB of OTHER traffic passes
    (plain or non-temporal, reads + writes as Adam does), then buf is read back with plain loads: ~5 TB/s = it stayed in the cache."""
    other = torch.empty(1024 * MB, dtype=torch.uint8, device=dev)
    print("resident buffer survives other traffic?  (read-back rate of the resident buffer, GB/s)")
    print("resident MB | nothing between | nt-read of itself between | 256 MB plain r+w between | 256 MB nt r+w between | 1 GiB nt r+w between | 1 GiB nt-read only between")
    for s in (64, 128, 192):
        n = s * MB
        buf = torch.empty(n, dtype=torch.uint8, device=dev)
        def prep(kind):
            def f():
                write(buf, n)
                if kind == "self_nt":
                    read(buf, n, 1)
                elif kind == "plain256":
                    read(other, 256 * MB, 0); write(other, 256 * MB, 0)
                elif kind == "nt256":
                    read(other, 256 * MB, 1); write(other, 256 * MB, 1)
                elif kind == "nt1024":
                    read(other, 1024 * MB, 1); write(other, 1024 * MB, 1)
                elif kind == "ntr1024":
                    read(other, 1024 * MB, 1)
            return f
        row = [rate(n, prep(k), lambda: read(buf, n)) for k in ("none", "self_nt", "plain256", "nt256", "nt1024", "ntr1024")]
        print(f"{s:11d} | " + " | ".join(f"{v:8.0f}" for v in row), flush=True)
        del buf
    # steady state of a producer / consumer cycle: [write buf (plain) ; 600 MB of nt streaming ; read buf (plain)] repeated -- the
    # write lands on lines that are still resident from the last cycle (no write-back in between?)
    print("cycle [plain write of S MB ; 600 MB nt streaming r+w ; plain read of S MB] x 6, times of the last cycle (us): write / stream / read")
    for s in (64, 128, 192, 256):
        n = s * MB
        buf = torch.empty(n, dtype=torch.uint8, device=dev)
        write(flush, flush.numel())
        for it in range(6):
            tw = timed(lambda: write(buf, n))
            ts = timed(lambda: (read(other, 300 * MB, 1), write(other, 300 * MB, 1)))
            tr = timed(lambda: read(buf, n))
        print(f"  {s:4d} MB: write {tw * 1e6:7.1f} ({n / tw * 1e-9:5.0f} GB/s)  stream {ts * 1e6:7.1f} ({600 * MB / ts * 1e-9:5.0f} GB/s)  read {tr * 1e6:7.1f} ({n / tr * 1e-9:5.0f} GB/s)", flush=True)
        del buf
    print("same cycle with PLAIN streaming in between:")
    for s in (64, 128):
        n = s * MB
        buf = torch.empty(n, dtype=torch.uint8, device=dev)
        write(flush, flush.numel())
        for it in range(6):
            tw = timed(lambda: write(buf, n))
            ts = timed(lambda: (read(other, 300 * MB, 0), write(other, 300 * MB, 0)))
            tr = timed(lambda: read(buf, n))
        print(f"  {s:4d} MB: write {tw * 1e6:7.1f} ({n / tw * 1e-9:5.0f} GB/s)  stream {ts * 1e6:7.1f} ({600 * MB / ts * 1e-9:5.0f} GB/s)  read {tr * 1e6:7.1f} ({n / tr * 1e-9:5.0f} GB/s)", flush=True)
        del buf


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "residency":
        residency()
        sys.exit(0)
    main()
