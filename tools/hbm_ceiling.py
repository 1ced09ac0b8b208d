#!/usr/bin/env python3
"""tools/hbm_ceiling.py — what one MI355X actually sustains on plain streaming kernels ("a measured copy-kernel
ceiling on the box, report both"): device-to-device copy (read + write), fill (write only) and a sum (read only) over
buffers far larger than the 256 MiB Infinity Cache.  torch elementwise kernels; times from CUDA(HIP) events."""
import json
import subprocess

import torch


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    n = 1 << 30                                   # 4 GiB of fp32 per buffer
    a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    b = torch.empty_like(a)
    out = {"bytes_per_buffer": n * 4}
    out["copy_GBs"] = 2 * n * 4 / timed(lambda: b.copy_(a)) / 1e9
    out["fill_GBs"] = n * 4 / timed(lambda: b.fill_(1.0)) / 1e9
    out["read_sum_GBs"] = n * 4 / timed(lambda: a.sum()) / 1e9
    out["add_GBs"] = 3 * n * 4 / timed(lambda: torch.add(a, b, out=b)) / 1e9
    try:
        smi = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--showclocks"], capture_output=True, text=True, timeout=30).stdout
        out["rocm_smi"] = [ln.strip() for ln in smi.splitlines() if "Total Memory" in ln or "mclk" in ln or "sclk" in ln][:6]
    except Exception as e:  # noqa: BLE001
        out["rocm_smi"] = str(e)
    out["device"] = torch.cuda.get_device_name(0)
    out["spec_peak_GBs"] = 8000
    print(json.dumps(out))


if __name__ == "__main__":
    main()
