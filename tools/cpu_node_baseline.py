#!/usr/bin/env python3
"""tools/cpu_node_baseline.py — SURVEY §8d(b): the CPU restatement (oracle, TEST INFRASTRUCTURE used here only as the reported
baseline) with one process per host core, voices partitioned evenly: voice*samples/s per core and per node for config 2a."""
import ctypes as C, json, multiprocessing as mp, os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker(args):
    idx, budget = args
    ko = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libklang_oracle.so"))
    ko.ko_bank_create.restype = C.c_void_p; ko.ko_bank_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
    ko.ko_bank_note_on.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_long]
    ko.ko_bank_process.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    ko.ko_patch_from_name.argtypes = [C.c_char_p]
    bank = ko.ko_bank_create(ko.ko_patch_from_name(b"sub2a"), 1, 128, C.c_float(48000.0))
    rng = np.random.default_rng(idx)
    for p in rng.integers(36, 97, size=128): ko.ko_bank_note_on(bank, 0, int(p), C.c_float(0.8), 1)
    mix = np.zeros((2, 256), np.float32); mp_ = mix.ctypes.data_as(C.c_void_p)
    for _ in range(30): ko.ko_bank_process(bank, None, mp_, None, 256)          # past attack/decay
    t0 = time.perf_counter(); blocks = 0
    while time.perf_counter() - t0 < budget:
        for _ in range(20): ko.ko_bank_process(bank, None, mp_, None, 256)
        blocks += 20
    return 128 * 256 * blocks / (time.perf_counter() - t0)

if __name__ == "__main__":
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    with mp.Pool(n) as pool: rates = pool.map(worker, [(i, 6.0) for i in range(n)])
    print(json.dumps(dict(workload="sub2a, 128 voices per process, sustain, 256-sample blocks", processes=n, host_cpus=os.cpu_count(),
                          per_core_mean=float(np.mean(rates)), per_core_min=float(np.min(rates)), node_total=float(np.sum(rates)))))
