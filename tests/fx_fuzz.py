"""tests/fx_fuzz.py — random EFFECT programs in the recorded-graph format (include/klang_mi355_graph.h), for tests that run the two generated kernels of the
same program against each other: klg_fx_graph<P> (one lane per instance, samples in order — the form every example effect is pinned to the genuine header
with) and klg_fx_staged (klg_graph_staged.hpp: samples side by side, levels, the control path a chunk ahead, ring checks, parts of a failed chunk).

What a program is made of (every piece is something a shipped or own patch does): two input channels, four dials (feedback, delay time, LFO depth, mix), a
smoothed dial, an LFO whose rate is set per sample, members written by process() (a one-sample feedback outside any delay line), biquads set in prepare(),
one to three delay lines used as
    near      `in >> d; d(t)`                      a line fed by the effect's own input: taps may sit inside their own chunk (Flanger / Chorus)
    feedback  `d(t) * g + in >> d`                 tap(float) / tap(int) / lagrange before the input(): the ring check decides (Echo / Feedback)
    head      `d.set(t); (in + d * g) >> d`        set() and process() per sample (PingPong)
    walk      prepare(): d.set(t); `(in + d*g) >> d`   the read head placed once per block, walked by every process()
    cross     two lines feeding each other (PingPong)
with taps optionally inside an `if`, and arithmetic sprinkled between.  Feedback gains stay below one: the signals stay finite."""
import struct

import numpy as np


def f32(x):
    return "%08x" % struct.unpack("<I", struct.pack("<f", float(x)))[0]


class Builder:
    def __init__(self, rng):
        self.rng = rng
        self.nodes = []          # (kind, arg or None)
        self.prep = []           # prepare() ops
        self.ops = []            # sample ops
        self.reg = 0

    def node(self, kind, arg=None):
        self.nodes.append((kind, arg))
        return len(self.nodes) - 1

    def new(self):
        self.reg += 1
        return self.reg - 1

    def op(self, code, dst=-1, a=-1, b=-1, node=-1, imm="00000000", prepare=False):
        (self.prep if prepare else self.ops).append(f"op {code} {dst} {a} {b} {node} {imm}")
        return dst

    def const(self, v, prepare=False):
        return self.op("const", self.new(), imm=f32(v), prepare=prepare)

    def bin(self, code, a, b):
        return self.op(code, self.new(), a, b)

    def scaled(self, a, k):
        return self.bin("mul", a, self.const(k))


def program(seed, channels=2):
    """-> (text, description)"""
    rng = np.random.default_rng(seed)
    B = Builder(rng)
    what = []
    # dials: 0 feedback, 1 delay time (samples), 2 LFO depth (samples), 3 mix
    hi_time = float(rng.choice([40.0, 120.0, 600.0]))
    dials = [(0.0, 0.9, 0.5), (1.0, hi_time, min(30.0, hi_time)), (0.0, 12.0, 0.0), (0.0, 1.0, 0.5)]
    # ---- nodes ----
    n_smooth = B.node("smooth")
    n_lfo = B.node("bsine")
    n_mem = B.node("param")
    nd = int(rng.integers(1, 4))
    uses = []
    for i in range(nd):
        kinds = ["near", "feedback", "head", "walk"] + (["cross"] if i + 1 < nd else [])
        uses.append(str(rng.choice(kinds)))
    skip = set()
    for i, u in enumerate(uses):
        if u == "cross" and i not in skip:
            skip.add(i + 1)
    # (a line read by process() gets the sample-parallel form only when it is longer than 1,024 samples per process() call: klg_graph_staged.hpp)
    sizes = [int(rng.choice([4096, 20000])) if (uses[i] in ("head", "walk", "cross") or i in skip) else int(rng.choice([256, 1000, 4096])) for i in range(nd)]
    delays = [B.node("delay", s) for s in sizes]
    filt = [B.node("lpf") for _ in range(int(rng.integers(1, 3)))]
    # ---- prepare(): the biquads; a line whose head is placed once per block ----
    for n in filt:
        B.op("lpfset", -1, B.const(float(rng.uniform(60.0, 6000.0)), True), B.const(float(rng.uniform(0.6, 2.0)), True), n, "%08x" % int(rng.integers(0, 2)), prepare=True)
    for i, u in enumerate(uses):
        if u == "walk" and i not in skip:
            t = B.op("ctl", B.new(), imm="00000001", prepare=True)
            if rng.random() < 0.5:
                t = B.op("mul", B.new(), t, B.const(float(rng.uniform(0.3, 1.0)), True), prepare=True)
            B.op("delayset", -1, t, node=delays[i], prepare=True)
    # ---- process() ----
    ins = [B.op("in", B.new(), imm="%08x" % c) for c in range(channels)]
    g = B.op("ctl", B.new(), imm="00000000")
    depth = B.op("ctl", B.new(), imm="00000002")
    mix = B.op("ctl", B.new(), imm="00000003")
    sm = B.op("smooth", B.new(), node=n_smooth, imm="00000001")
    B.op("oscset", -1, B.bin("add", B.const(float(rng.uniform(0.5, 9.0))), B.scaled(mix, float(rng.uniform(0.0, 20.0)))), node=n_lfo)      # lfo.set(rate) per sample
    lfo = B.op("osc", B.new(), node=n_lfo)
    mem = B.op("param", B.new(), node=n_mem)
    wob = B.bin("mul", lfo, depth)
    pool = list(ins) + [B.scaled(mem, 0.5)]

    def time_reg():
        """a delay time in samples: the smoothed dial, scaled, optionally wobbling with the LFO, never negative"""
        t = B.scaled(sm, float(rng.uniform(0.2, 1.0)))
        if rng.random() < 0.5:
            t = B.bin("add", t, B.scaled(wob, float(rng.uniform(0.2, 1.0))))
        if rng.random() < 0.3:
            t = B.bin("add", t, B.const(float(rng.uniform(0.0, 3.0))))
        return B.op("abs", B.new(), t)

    def tap(node, t):
        kind = int(rng.choice([0, 0, 0, 1, 3]))
        if kind == 1:
            t = B.op("trunc", B.new(), t)
        r = B.op("delaytap", B.new(), t, node=node, imm="%08x" % kind)
        return r

    def maybe_if(x):
        """`if (x > c) y = x * a; else y = x * b;` — with a tap of some line inside one side, sometimes"""
        if rng.random() > 0.35:
            return x
        c = B.op("cmp", B.new(), x, B.const(float(rng.uniform(-0.2, 0.2))), imm="%08x" % int(rng.integers(0, 4)))
        B.op("if", -1, c)
        a = B.scaled(x, float(rng.uniform(0.3, 0.9)))
        if rng.random() < 0.5:
            j = int(rng.integers(0, nd))
            if uses[j] in ("near", "feedback") and j not in skip:
                a = B.bin("add", a, B.scaled(tap(delays[j], time_reg()), 0.25))
        B.op("else", -1)
        b = B.scaled(x, float(rng.uniform(-0.9, -0.3)))
        B.op("endif", -1)
        return B.op("phi", B.new(), a, b)

    outs = []
    for i, u in enumerate(uses):
        if i in skip:
            continue
        x = pool[int(rng.integers(0, len(pool)))] if rng.random() < 0.3 else ins[i % channels]
        d = delays[i]
        if u == "near":
            B.op("delayin", -1, ins[i % channels], node=d)
            y = tap(d, time_reg())
            if rng.random() < 0.5:
                y = B.bin("add", y, B.scaled(tap(d, time_reg()), 0.5))
        elif u == "feedback":
            y = tap(d, time_reg())
            B.op("delayin", -1, B.bin("add", x, B.bin("mul", y, g)), node=d)
            if rng.random() < 0.4:
                y = B.bin("add", y, B.scaled(tap(d, time_reg()), 0.5))       # a tap after the input(): it may see this sample
        elif u == "head":
            B.op("delayset", -1, time_reg(), node=d)
            y = B.op("delayout", B.new(), node=d)
            B.op("delayin", -1, B.bin("add", x, B.bin("mul", y, g)), node=d)
            if rng.random() < 0.4:
                y = B.bin("add", y, B.scaled(B.op("delayout", B.new(), node=d), 0.5))
        elif u == "walk":
            y = B.op("delayout", B.new(), node=d)
            B.op("delayin", -1, B.bin("add", x, B.bin("mul", y, g)), node=d)
        else:                                                                   # cross: PingPong's shape over lines i and i + 1
            e = delays[i + 1]
            B.op("delayset", -1, time_reg(), node=d)
            B.op("delayset", -1, time_reg(), node=e)
            ye = B.op("delayout", B.new(), node=e)
            B.op("delayin", -1, B.bin("add", ins[0], B.bin("mul", ye, g)), node=d)
            yd = B.op("delayout", B.new(), node=d)
            B.op("delayin", -1, B.bin("add", ins[-1], B.bin("mul", yd, g)), node=e)
            y = B.bin("add", yd, B.scaled(ye, 0.5))
        y = maybe_if(y)
        pool.append(y)
        outs.append(y)
        what.append(f"{u}[{sizes[i]}]")
    # members written by process(): next sample reads them
    B.op("setparam", -1, B.bin("add", B.scaled(mem, 0.5), B.scaled(outs[0], 0.25)), node=n_mem)
    res = []
    for c in range(channels):
        wet = outs[c % len(outs)]
        if len(outs) > 1 and rng.random() < 0.5:
            wet = B.bin("add", wet, B.scaled(outs[(c + 1) % len(outs)], 0.5))
        wet = B.op("lpf", B.new(), wet, node=filt[c % len(filt)])
        dry = B.bin("mul", ins[c], B.bin("sub", B.const(1.0), mix))
        res.append(B.bin("add", dry, B.bin("mul", wet, mix)))
    lines = ["klgg 1", f"kind effect {channels}", "ctl 4"]
    lines += [f"dial {i} {np.float32(lo):.9g} {np.float32(hi):.9g} {np.float32(init):.9g}" for i, (lo, hi, init) in enumerate(dials)]
    lines += [f"node {i} {k}" + (f" {a}" if a is not None else "") for i, (k, a) in enumerate(B.nodes)]
    lines += B.prep + B.ops
    lines.append(f"prepare {len(B.prep)}")
    lines.append("ret " + str(res[0]) if channels == 1 else f"ret2 {res[0]} {res[1]}")
    lines.append("end")
    return "\n".join(lines) + "\n", " + ".join(what), dials
