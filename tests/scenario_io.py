"""tests/scenario_io.py — scenario files and result files shared by the oracle harness and the tests.

Formats are documented in oracle/ref/ref_common.h.  A Scenario is a seeded note-event script:
the same text drives the genuine reference (oracle/_ref, build container only), the C
restatement (oracle/_build) and the HIP path (through the C-ABI, see tests/klg_driver.py).
"""
from dataclasses import dataclass, field

import numpy as np

EV_ON, EV_OFF, EV_CTL = 0, 1, 2


def _f32(x):
    return float(np.float32(x))


@dataclass
class Scenario:
    patch: str
    fs: float = 48000.0
    block: int = 256
    blocks: int = 1
    synths: int = 1
    notes: int = 1
    dump: list = field(default_factory=list)
    ctl: list = field(default_factory=list)          # (index, value) applied to every instance at start
    ev: list = field(default_factory=list)           # (block, type, synth, a, b, seed)
    instances: int = 0                               # effect scenarios
    burst: int = 0
    seed: int = 0

    # Event values are quantised to float32 HERE: the text form ("%.9g") round-trips a float32 exactly, whereas a python
    # double printed with 9 digits and re-read can round to the neighbouring float32 (the oracle reads the text, the GPU
    # driver gets the python value: they must see the same float).
    def on(self, block, synth, pitch, velocity, seed=-1):
        self.ev.append((int(block), EV_ON, int(synth), float(pitch), _f32(velocity), int(seed)))

    def off(self, block, synth, pitch, velocity=0.0):
        self.ev.append((int(block), EV_OFF, int(synth), float(pitch), _f32(velocity), -1))

    def control(self, block, synth, index, value):
        self.ev.append((int(block), EV_CTL, int(synth), float(index), _f32(value), -1))

    def sort(self):
        self.ev.sort(key=lambda e: e[0])             # stable: keeps file order inside a block

    @property
    def voices(self):
        return self.synths * self.notes

    def text(self):
        lines = ["klgscn 1", f"patch {self.patch}", f"fs {self.fs:.9g}", f"block {self.block}", f"blocks {self.blocks}"]
        if self.instances:
            lines += [f"instances {self.instances}", f"burst {self.burst}", f"seed {self.seed}"]
        else:
            lines += [f"synths {self.synths}", f"notes {self.notes}"]
        lines.append("dump %d %s" % (len(self.dump), " ".join(str(d) for d in self.dump)))
        for i, v in self.ctl:
            lines.append(f"ctl {i} {_f32(v):.9g}")
        for b, t, s, a, bb, seed in self.ev:
            lines.append(f"ev {b} {t} {s} {a:.9g} {bb:.9g} {seed}")
        lines.append("end")
        return "\n".join(lines) + "\n"

    def save(self, path):
        with open(path, "w") as f:
            f.write(self.text())

    @staticmethod
    def load(path):
        toks = open(path).read().split()
        assert toks[0] == "klgscn"
        s = Scenario(patch="")
        i = 2
        while i < len(toks):
            t = toks[i]
            if t == "end":
                break
            if t == "patch":
                s.patch = toks[i + 1]; i += 2
            elif t == "fs":
                s.fs = float(toks[i + 1]); i += 2
            elif t in ("block", "blocks", "synths", "notes", "instances", "burst", "seed"):
                setattr(s, t, int(toks[i + 1])); i += 2
            elif t == "dump":
                k = int(toks[i + 1]); s.dump = [int(x) for x in toks[i + 2:i + 2 + k]]; i += 2 + k
            elif t == "ctl":
                s.ctl.append((int(toks[i + 1]), float(toks[i + 2]))); i += 3
            elif t == "ev":
                s.ev.append((int(toks[i + 1]), int(toks[i + 2]), int(toks[i + 3]), float(toks[i + 4]), float(toks[i + 5]), int(toks[i + 6]))); i += 7
            else:
                raise ValueError(f"bad token {t}")
        return s


def load_ref_output(path):
    """Parse a result file written by oracle/_ref/ref_* or oracle/_build/ko_run."""
    d = open(path, "rb").read()
    magic, V, N, nd, B = (int(x) for x in np.frombuffer(d, dtype=np.int32, count=5))
    o = 20
    if magic == 0x4F474C4B:      # 'KLGO' synth
        pv = np.frombuffer(d, dtype=np.float32, count=nd * V * N, offset=o).reshape(nd, V, N); o += nd * V * N * 4
        mix = np.frombuffer(d, dtype=np.float32, count=B * 2 * N, offset=o).reshape(B, 2, N); o += B * 2 * N * 4
        st = np.frombuffer(d, dtype=np.uint8, count=B * V, offset=o).reshape(B, V)
        return dict(per_voice=pv, mix=mix, stages=st)
    if magic == 0x53474C4B:      # 'KLGS' synth of stereo notes: per-voice output per channel
        pv = np.frombuffer(d, dtype=np.float32, count=nd * V * 2 * N, offset=o).reshape(nd, V, 2, N); o += nd * V * 2 * N * 4
        mix = np.frombuffer(d, dtype=np.float32, count=B * 2 * N, offset=o).reshape(B, 2, N); o += B * 2 * N * 4
        st = np.frombuffer(d, dtype=np.uint8, count=B * V, offset=o).reshape(B, V)
        return dict(per_voice=pv, mix=mix, stages=st)
    if magic == 0x46474C4B:      # 'KLGF' effect
        pv = np.frombuffer(d, dtype=np.float32, count=nd * V * 2 * N, offset=o).reshape(nd, V, 2, N)
        return dict(per_voice=pv)
    raise ValueError("bad magic")


def load_kat(path):
    """Known-answer vectors: records {u32 name_len, name, u32 count, f32[count]}."""
    import struct
    d = open(path, "rb").read()
    o = 0
    kat = {}
    while o < len(d):
        n, = struct.unpack_from("<I", d, o); o += 4
        name = d[o:o + n].decode(); o += n
        c, = struct.unpack_from("<I", d, o); o += 4
        kat[name] = np.frombuffer(d, dtype=np.float32, count=c, offset=o); o += 4 * c
    return kat


def fx_input(seed, instance, ch, t, burst):
    """numpy restatement of ref_fx_input / ko_fx_input (lowbias32 hash noise burst)."""
    def h32(x):
        x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
        x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
        x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
        x ^= x >> 16
        return x
    t = np.asarray(t, dtype=np.uint64)
    h = h32(np.uint64(seed) ^ h32(np.uint64(instance * 2 + ch)) ^ ((t * 0x9e3779b9) & 0xFFFFFFFF))
    v = ((h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)) - np.float32(0.5)
    return np.where(t < burst, v, np.float32(0)).astype(np.float32)
