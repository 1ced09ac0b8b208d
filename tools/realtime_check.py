#!/usr/bin/env python3
"""tools/realtime_check.py — "max real-time voice count" (SURVEY.md §8d): the largest V whose block time stays under
N / 48000 s (5.33 ms at N = 256) over >= 2000 consecutive blocks (p99), each block synchronised like a real-time host
would.  Measures: blocks 0..1999 from a simultaneous note-on of every voice (attack + decay ramps, then sustain) and the
release phase (every voice ramping) separately."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voices", type=int, nargs="+", default=[8 << 20, 16 << 20])
    ap.add_argument("--blocks", type=int, default=2000)
    ap.add_argument("--patch", default="sub2a")
    a = ap.parse_args()
    import torch, klang_amd
    N = 256
    deadline = 1e3 * N / 48000.0
    for V in a.voices:
        notes = 128
        bank = klang_amd.SynthBank(a.patch, synths=V // notes, notes=notes, max_block=N)
        rng = np.random.default_rng(1)
        pitches = rng.integers(36, 97, size=V); owner = np.arange(V) // notes
        bank.note_on_many(owner, pitches, np.full(V, 0.8, np.float32))
        mix = torch.zeros((2, N), dtype=torch.float32, device="cuda")
        torch.cuda.set_stream(torch.cuda.Stream())      # a stream of our own: handle 0 (torch's default) means "the bank's own stream" to the library, which torch's clears are not ordered with
        st = torch.cuda.current_stream().cuda_stream
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st); torch.cuda.synchronize()      # block 0 also carries the event upload
        t = np.empty(a.blocks)
        for b in range(a.blocks):
            t0 = time.perf_counter()
            mix.zero_(); bank.process_device(mix.data_ptr(), N, st); torch.cuda.synchronize()
            t[b] = 1e3 * (time.perf_counter() - t0)
        bank.note_off_many(owner, pitches, np.zeros(V, np.float32))
        mix.zero_(); bank.process_device(mix.data_ptr(), N, st); torch.cuda.synchronize()      # event upload block
        r = np.empty(46)
        for b in range(46):
            t0 = time.perf_counter()
            mix.zero_(); bank.process_device(mix.data_ptr(), N, st); torch.cuda.synchronize()
            r[b] = 1e3 * (time.perf_counter() - t0)
        alive = int((bank.stages() != 3).sum())
        print(json.dumps(dict(patch=a.patch, voices=V, blocks=a.blocks, deadline_ms=deadline,
                              ramp_blocks_ms_max=float(t[:22].max()), sustain_p50_ms=float(np.median(t[30:])), p99_ms=float(np.percentile(t, 99)), max_ms=float(t.max()),
                              release_p99_ms=float(np.percentile(r, 99)), release_max_ms=float(r.max()), voices_alive_during_release=alive,
                              realtime=bool(np.percentile(t, 99) <= deadline and np.percentile(r, 99) <= deadline))), flush=True)
        bank.close()


if __name__ == "__main__":
    main()
