"""tests/klg_driver.py — drive a Scenario (tests/scenario_io.py) through the C-ABI (GPU path) and, for the
checker, through the TEST-ONLY oracle binary.  Both return the structure load_ref_output() gives."""
import os
import subprocess
import tempfile

import numpy as np

from scenario_io import EV_CTL, EV_OFF, EV_ON, Scenario, load_ref_output


def run_scenario_gpu(s: Scenario, per_voice=True):
    import klang_amd
    bank = klang_amd.SynthBank(s.patch, synths=s.synths, notes=s.notes, fs=s.fs, max_block=s.block)
    try:
        for sy in range(s.synths):
            for i, v in s.ctl:
                bank.set_control(sy, i, v)
        V, N, B = s.voices, s.block, s.blocks
        dumps, mixes, stages = [], np.zeros((B, 2, N), np.float32), np.zeros((B, V), np.uint8)
        evi = 0
        for b in range(B):
            while evi < len(s.ev) and s.ev[evi][0] <= b:
                _, t, sy, a, bb, seed = s.ev[evi]
                if t == EV_ON:
                    if seed >= 0:
                        bank.random(seed)
                    bank.note_on(sy, int(a), bb)
                elif t == EV_OFF:
                    bank.note_off(sy, int(a), bb)
                elif t == EV_CTL:
                    bank.set_control(sy, int(a), bb)
                evi += 1
            if per_voice:
                pv, _ = bank.process_voices(N, mixes[b])
                if b in s.dump:
                    dumps.append(pv)
            else:
                bank.process(mixes[b])
            stages[b] = bank.stages()
        out = dict(mix=mixes, stages=stages)
        if per_voice:
            out["per_voice"] = np.stack(dumps) if dumps else np.zeros((0, V, N), np.float32)
        return out
    finally:
        bank.close()


def run_scenario_oracle(s: Scenario, oracle_build):
    """CHECKER ONLY: the C restatement under oracle/ (test infrastructure)."""
    with tempfile.TemporaryDirectory() as d:
        scn, out = os.path.join(d, "s.scn"), os.path.join(d, "o.bin")
        s.save(scn)
        subprocess.run([os.path.join(oracle_build, "ko_run"), scn, out], check=True)
        r = load_ref_output(out)
        return {k: np.array(v) for k, v in r.items()}


def rel_err(got, ref):
    """The parity metric of SURVEY.md §7: |a-b| <= tol * max(|ref|, block_peak) per voice-block.
    Returns the max over everything of |a-b| / max(|ref|, peak of that voice's block)."""
    ref = np.asarray(ref, np.float64)
    got = np.asarray(got, np.float64)
    peak = np.max(np.abs(ref), axis=-1, keepdims=True)
    scale = np.maximum(np.abs(ref), peak)
    scale = np.where(scale == 0, 1.0, scale)
    return float(np.max(np.abs(got - ref) / scale))


def bit_exact_fraction(got, ref):
    a = np.ascontiguousarray(got, np.float32).view(np.uint32)
    b = np.ascontiguousarray(ref, np.float32).view(np.uint32)
    same = (a == b) | ((np.asarray(got) == 0) & (np.asarray(ref) == 0))
    return float(np.mean(same))


def run_fx_scenario_gpu(s: Scenario):
    """Effect scenario through the C-ABI: per-instance hash-noise burst input (scenario_io.fx_input), in place."""
    import klang_amd
    from scenario_io import fx_input
    K, N, B = s.instances, s.block, s.blocks
    bank = klang_amd.FxBank(s.patch, K, fs=s.fs, max_block=N)
    try:
        for k in range(K):
            for i, v in s.ctl:
                bank.set_control(k, i, v)
        dumps = []
        evi = 0
        t = np.arange(N, dtype=np.uint64)
        for b in range(B):
            while evi < len(s.ev) and s.ev[evi][0] <= b:
                _, ty, inst, a, bb, _seed = s.ev[evi]
                if ty == EV_CTL:
                    bank.set_control(inst, int(a), bb)
                evi += 1
            io = np.empty((K, 2, N), np.float32)
            for k in range(K):
                for ch in range(2):
                    io[k, ch] = fx_input(s.seed, k, ch, t + np.uint64(b * N), s.burst)
            bank.process(io)
            if b in s.dump:
                dumps.append(io.copy())
        return dict(per_voice=np.stack(dumps))
    finally:
        bank.close()
