"""Graph patches on the CPU side (no GPU needed): the program format, the generated HIP source, hipRTC compilation for
gfx950 (a cross-compile: no device is touched), and the DSL facade's recording of a .k patch's process()."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SUB2B_LIKE = """klgg 1
ctl 1
dial 0 0.1 20 10
node 0 pulse
node 1 adsr
node 2 env
node 3 lpf
node 4 param
op env 3 -1 -1 2 0
op ctl 4 -1 -1 -1 0
op lpfset -1 3 4 3 0
op osc 5 -1 -1 0 0
op lpf 6 5 -1 3 0
op env 7 -1 -1 1 0
op mul 8 6 7 -1 0
op param 9 -1 -1 4 0
op add 10 8 9 -1 0
op setparam -1 8 -1 4 0      # a member written by process(): next sample reads it
op stopif -1 -1 -1 1 0
ret 10
end
"""


def check(program, want_source=False):
    from klang_amd._lib import lib
    buf = C.create_string_buffer(1 << 22)
    rc = lib().klg_graph_check(program.encode(), 1 if want_source else 0, buf, len(buf))
    return rc, buf.value.decode()


def test_program_compiles_for_gfx950_without_a_device():
    rc, src = check(SUB2B_LIKE, want_source=True)
    assert rc == 0, src
    # the generated patch is built from the same device primitives as the hand-written ones
    for needle in ("struct PatchGen", "osm_pulse(L.n0)", "biquad_lpf_set(L.n3, L.n3s, r3, r4, c.fs.w)", "adsr_process(L.n1, c.fs)",
                   "env_process_rt(L.n2", "ctl_read(c, 0u)", "L.n4 = r8;", "stage_off_if(env_is_off(L.n1.e.stage), L.stage)"):
        assert needle in src, needle
    words = 1 + 6 + 9 + 15 + 9 + 1
    assert f"uint32_t w[{words}]" in src


@pytest.mark.parametrize("program,message", [
    ("klgg 2\nend\n", "unsupported version"),
    ("klgg 1\nctl 0\nnode 0 lpf\nop lpf 1 0 -1 0 0\nret 1\nend\n", "operand a is not defined"),
    ("klgg 1\nctl 0\nnode 0 saw\nop env 0 -1 -1 0 0\nret 0\nend\n", "node is not an envelope"),
    ("klgg 1\nctl 0\nnode 0 saw\nop osc 0 -1 -1 0 0\nop osc 0 -1 -1 0 0\nret 0\nend\n", "register assigned twice"),
    ("klgg 1\nctl 0\nnode 0 saw\nop osc 0 -1 -1 0 0\nret 3\nend\n", "undefined register"),
    ("klgg 1\nctl 0\nnode 1 saw\nend\n", "numbered 0,1,2"),
    ("klgg 1\nctl 1\nnode 0 saw\nop ctl 0 -1 -1 -1 5\nret 0\nend\n", "control index out of range"),
])
def test_malformed_programs_are_rejected_with_a_reason(program, message):
    rc, msg = check(program)
    assert rc < 0 and message in msg, msg


CLIPPER = """klgg 1
kind effect 1
ctl 1
dial 0 1 11 1
op ctl 0 -1 -1 -1 0
op in 1 -1 -1 -1 0
op mul 2 1 0 -1 0
op const 3 -1 -1 -1 3f800000
op const 4 -1 -1 -1 bf800000
op cmp 5 2 3 -1 1            # in * gain > 1
op if -1 5 -1 -1 0
op else -1 -1 -1 -1 0
op cmp 6 2 4 -1 0            # in * gain < -1
op if -1 6 -1 -1 0
op else -1 -1 -1 -1 0
op endif -1 -1 -1 -1 0
op phi 7 4 2 -1 0
op endif -1 -1 -1 -1 0
op phi 8 3 7 -1 0
ret 8
end
"""


def test_structured_branches_compile():
    rc, src = check(CLIPPER, want_source=True)
    assert rc == 0, src
    for needle in ("const float r5 = (r2 > r3) ? 1.f : 0.f;", "float r8;", "if (r5 != 0.f) {", "r8 = r3;", "} else {", "float r7;", "r7 = r4;", "r7 = r2;", "r8 = r7;"):
        assert needle in src, needle


@pytest.mark.parametrize("program,message", [
    (CLIPPER.replace("op phi 7 4 2 -1 0\n", "op add 9 2 2 -1 0\nop phi 7 4 2 -1 0\n"), "a phi must directly follow"),
    (CLIPPER.replace("op endif -1 -1 -1 -1 0\nop phi 8 3 7 -1 0\n", ""), "not closed"),
    (CLIPPER.replace("op else -1 -1 -1 -1 0\nop cmp 6", "op const 9 -1 -1 -1 0\nop else -1 -1 -1 -1 0\nop cmp 6").replace("ret 8", "ret 9"), "undefined register"),   # a side's register is not visible after it
    (CLIPPER.replace("op phi 8 3 7 -1 0", "op phi 8 6 7 -1 0"), "not visible at the end of the `if` side"),   # r6 belongs to the else side
    (CLIPPER.replace("op cmp 5 2 3 -1 1", "op cmp 5 2 3 -1 9"), "unknown relation"),
    ("klgg 1\nctl 0\nop else -1 -1 -1 -1 0\nop const 0 -1 -1 -1 0\nret 0\nend\n", "no open `if`"),
])
def test_malformed_branches_are_rejected(program, message):
    rc, msg = check(program)
    assert rc < 0 and message in msg, msg


SAW_LPF_ADSR = "klgg 1\nctl 0\nnode 0 saw\nnode 1 lpf\nnode 2 adsr\nop osc 0 -1 -1 0 0\nop lpf 1 0 -1 1 0\nop env 2 -1 -1 2 0\nop mul 3 1 2 -1 0\nop stopif -1 -1 -1 2 0\nret 3\nend\n"


def check_mode(program, mode):
    from klang_amd._lib import lib
    buf = C.create_string_buffer(1 << 17)
    rc = lib().klg_graph_check(program.encode(), mode, buf, len(buf))
    return rc, buf.value.decode()


def test_two_voices_per_lane_form_and_the_three_bodies():
    """A program whose nodes all have packed forms also compiles with two voices per lane: the SAME body text over 2-vector
    types, plus the bodies for chunks in which the envelopes hold (adsr_hold) and the saws are in their duty-0 form."""
    rc1, one = check_mode(SAW_LPF_ADSR, 1)
    rc2, two = check_mode(SAW_LPF_ADSR, 2)
    assert rc1 == 0 and rc2 == 0, one + two
    assert "struct Rec { uint32_t w[25]; }" in one and "struct Live { int stage; float tinc; Osm n0;" in one and "klg_render_x2" not in one.split("namespace klg")[1]
    assert "struct Rec { u2 w[25]; }" in two and "struct Live { i2 stage; float tinc; Osm2 n0;" in two and "const BlockCtx2& c" in two
    body = lambda src, name: src[src.index(name):].split("return r3;")[0].split("{\n", 1)[1]
    # one voice per lane: the quiet bodies are EVENT-FREE chunks (every envelope glides: holding at sustain is the step -0.0);
    # two voices per lane: the ADSR holds (adsr_hold)
    for src, F, env in ((one, "float", "env_glide(L.n2.e, L.n2gs, L.n2gt)"), (two, "f2", "adsr_hold(L.n2, c.fs)")):
        assert body(src, F + " sample(") == body(src, F + " sample(")                                            # sanity
        assert "adsr_process(L.n2, c.fs)" in body(src, F + " sample(") and "osm_saw(L.n0)" in body(src, F + " sample(")
        assert env in body(src, F + " sample_quiet(") and "L.n0d0 ? osm_saw_duty0(L.n0)" in body(src, F + " sample_quiet(")
        assert env in body(src, F + " sample_fast(") and "= osm_saw_duty0(L.n0);" in body(src, F + " sample_fast(")
        assert "stage_off_if(env_is_off(L.n2.e.stage), L.stage)" in src.split("void end(")[1]                    # `if (adsr.finished()) stop();` once per block
    assert "env_safe(L.n2.e, L.n2.e.point == 2, L.n2gs, L.n2gt, L.tinc)" in one.split("int quiet(")[1]
    assert body(one, "float sample(").replace("float", "f2") == body(two, "f2 sample(")                          # the same text, other types
    rc, msg = check_mode(SUB2B_LIKE, 2)                                                                          # lpfset has no packed form
    assert rc < 0 and "two-voices-per-lane" in msg


def test_wavetable_and_notedelay_nodes_belong_to_synth_programs():
    ok = "klgg 1\nctl 0\nnode 0 notedelay 4800\nnode 1 wavetable\nop delayout 0 -1 -1 0 0\nop osc 1 -1 -1 1 0\nop add 2 0 1 -1 0\nop delayin -1 2 -1 0 0\nop const 3 -1 -1 -1 42c80000\nop delaytap 4 3 -1 0 0\nop tabread 5 4 -1 -1 1\nop add 6 2 5 -1 0\nret 6\nend\n"
    rc, src = check(ok, want_source=True)
    assert rc == 0, src
    for needle in ("delay_process(Ring{ c.ring + (size_t)0ll, 1, 4800 }, L.n0t)", "wavetable_process(L.n1)", "table_read(c.tables, 1u, r4)", "static constexpr int kWavesPerEu = 1;", "uint32_t w[10]"):
        assert needle in src, needle
    for bad, msg in ((ok.replace("klgg 1\n", "klgg 1\nkind effect 1\n"), "only available to synth notes"),
                     (ok.replace("node 0 notedelay 4800", "node 0 notedelay"), "delay needs its SIZE"),
                     (ok.replace("op delayout 0 -1 -1 0 0", "op delayout 0 -1 -1 1 0"), "node is not a delay"),
                     (ok.replace("tabread 5 4 -1 -1 1", "tabread 5 4 -1 -1 0"), "table id 0 is reserved")):
        rc, m = check(bad)
        assert rc < 0 and msg in m, m
    rc, m = check_mode(ok, 2)
    assert rc < 0 and "two-voices-per-lane" in m


def test_noise_draws_are_indexed_per_sample_and_stay_outside_branches():
    ok = "klgg 1\nkind effect 1\nctl 0\nop noise 0 -1 -1 -1 1\nop noise 1 -1 -1 -1 0\nop add 2 0 1 -1 0\nret 2\nend\n"
    rc, src = check(ok, want_source=True)
    assert rc == 0, src
    assert "fast_noise(c.rand[(size_t)(L.sidx * 2 + 0) * c.rstride])" in src and "basic_noise(c.rand[(size_t)(L.sidx * 2 + 1) * c.rstride])" in src and "L.sidx++;" in src
    # ... and a Note's: its voice's draws of the block are [sample][generator]; the three bodies all count samples; one voice per lane
    rc, src = check(ok.replace("kind effect 1\n", ""), want_source=True)
    assert rc == 0, src
    assert "struct Live { int stage; float tinc; int sidx;" in src and "L.sidx = 0;" in src and src.count("L.sidx++;") == 3
    assert "fast_noise(c.nz[((L.sidx & (KLG_NZ_GROUP - 1)) * 2 + 0) * 64])" in src and "kNoiseDraws = 2" in src and "klg_render_x2<" not in src
    branchy = "klgg 1\nkind effect 1\nctl 0\nop in 0 -1 -1 -1 0\nop cmp 1 0 0 -1 1\nop if -1 1 -1 -1 0\nop noise 2 -1 -1 -1 1\nop endif -1 -1 -1 -1 0\nret 0\nend\n"
    rc, msg = check(branchy)
    assert rc < 0 and "inside an `if`" in msg


def test_facade_records_branches_once_per_outcome_and_merges_them():
    """tests/patches/branches.k: an else-if chain with `!` / `&&`, a nested `if`, an oscillator advanced on one side only, a
    member written on both sides.  process() is run once per branch outcome; the traces are merged into if / else / endif + phi."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    exe = os.path.join(ROOT, "tests", "cpp", "_bin", "facade_graph_own_branches")
    scn = os.path.join(ROOT, "tests", "golden", "own_branches_solo.scn")
    r = subprocess.run([exe, scn, "/dev/null"], env=dict(os.environ, KLANG_MI355_DUMP_GRAPH="1", HIP_VISIBLE_DEVICES="-1"), capture_output=True, text=True)
    err = r.stderr
    text = err[err.index("klgg 1"):err.index("end\n") + 4]
    ops = [ln.split()[1] for ln in text.splitlines() if ln.startswith("op ")]
    assert ops.count("if") == ops.count("endif") == 6 and ops.count("phi") == 6, ops   # `a && b` is two nested ifs
    assert ops.count("osc") == 3 and ops.count("setparam") == 1                          # each oscillator appears once: the sides were merged, not duplicated
    lines = text.splitlines()
    osc_b = next(i for i, ln in enumerate(lines) if ln.startswith("op osc") and ln.split()[5] == "1")
    first_if = next(i for i, ln in enumerate(lines) if ln.startswith("op if"))
    first_else = next(i for i, ln in enumerate(lines) if ln.startswith("op else"))
    assert first_if < osc_b < first_else + 20                                              # Sine b is read inside the first `if` side only
    rc, msg = check(text)
    assert rc == 0, msg


def test_facade_records_our_dsl_patch():
    """tests/patches/sub2a.k compiled against include/klang/klang.h with NO binding: notes.add<T>() records process().
    Without a GPU the run then stops at klg_synth_create_graph (no CPU fallback) — after printing the program."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    exe = os.path.join(ROOT, "tests", "cpp", "_bin", "facade_graph_sub2a_n4")
    scn = os.path.join(ROOT, "tests", "golden", "sub2a_steal.scn")
    r = subprocess.run([exe, scn, "/dev/null"], env=dict(os.environ, KLANG_MI355_DUMP_GRAPH="1", HIP_VISIBLE_DEVICES="-1"), capture_output=True, text=True)
    err = r.stderr
    assert "node 0 saw\nnode 1 lpf\nnode 2 adsr\n" in err, err
    ops = [ln.split()[1] for ln in err.splitlines() if ln.startswith("op ")]
    assert ops == ["osc", "lpf", "env", "mul", "stopif"], ops
    rc, msg = check(err[err.index("klgg 1"):err.index("end\n") + 4])
    assert rc == 0, msg


def test_a_notes_smoothed_control_is_a_record_word_the_bank_sets_per_block():
    """controls[i].smooth() in a Note (klang.h:1715): the control is the Synth's and every sounding note advances it in turn, so the
    generated body repeats the reference's operations on a record word that klg_process fills with the value this voice's block starts
    from (note_prepass, klg_api.hip).  Two calls per sample on one control are two steps; such a program runs one voice per lane."""
    prog = "klgg 1\nctl 2\nnode 0 fsine\nnode 1 smooth\nop osc 0 -1 -1 0 0\nop smooth 1 -1 -1 1 1\nop smooth 2 -1 -1 1 1\nop mul 3 0 1 -1 0\nop mul 4 3 2 -1 0\nret 4\nend\n"
    rc, src = check(prog, want_source=True)
    assert rc == 0, src
    assert src.count("L.n1 = L.n1 * 0.999f + (1.f - 0.999f) * c.ctl[1];") == 2 * 3 and "klg_render_x2<" not in src      # three bodies
    rc, msg = check(prog.replace("op smooth 2 -1 -1 1 1", "op smooth 2 -1 -1 1 5"))
    assert rc < 0 and "not a smoothed control" in msg
    branchy = "klgg 1\nctl 1\nnode 0 fsine\nnode 1 smooth\nop osc 0 -1 -1 0 0\nop cmp 1 0 0 -1 1\nop if -1 1 -1 -1 0\nop smooth 2 -1 -1 1 0\nop endif -1 -1 -1 -1 0\nret 0\nend\n"
    rc, msg = check(branchy)
    assert rc < 0 and "inside an `if`" in msg          # the chain through the sounding notes assumes one step per call and sample


def test_oscillators_can_be_rephased_per_sample():
    """osc.set(f, phase) and osc.reset() inside process() (hard sync, re-phasing): `oscset` with imm 1 (a = f, b = phase) and imm 2."""
    prog = ("klgg 1\nctl 0\nnode 0 saw\nnode 1 fsine\nnode 2 bsine\nop const 0 -1 -1 -1 1135869952\nop const 1 -1 -1 -1 0\n"
            "op oscset -1 0 1 0 1\nop oscset -1 0 1 1 1\nop oscset -1 0 1 2 1\nop oscset -1 -1 -1 1 2\nop oscset -1 -1 -1 2 2\n"
            "op osc 2 -1 -1 0 0\nop osc 3 -1 -1 1 0\nop osc 4 -1 -1 2 0\nop add 5 2 3 -1 0\nop add 6 5 4 -1 0\nret 6\nend\n")
    rc, src = check(prog, want_source=True)
    assert rc == 0, src
    for needle in ("osm_set_fp(L.n0, L.n0f, r0, r1, c.fs.f)", "fsine_set_fp(L.n1, L.n1f, r0, r1, c.fs.f)", "L.n2.position = r1; L.n2f = r0;", "L.n1.pos = 0u;", "L.n2.position = 0.f;"):
        assert needle in src, needle
    rc, msg = check(prog.replace("op oscset -1 -1 -1 1 2", "op oscset -1 -1 -1 0 2"))       # a Fast::OSM oscillator has no reset() of its own
    assert rc < 0 and "no such set() / reset()" in msg


def test_a_duty_can_be_set_per_sample_and_finished_is_a_value():
    """`oscset` imm 3 (duty = a: OSM::setDuty klang.h:5246-5249, Basic::Pulse::duty 4936 — `set(f, phase, duty)` is oscset 1 then oscset 3) and `envoff`
    (Envelope::finished() as a value, klang.h:4094): generated text, and what the validator refuses."""
    prog = ("klgg 1\nctl 0\nnode 0 pulse\nnode 1 bpulse\nnode 2 adsr\nop const 0 -1 -1 -1 1135869952\nop const 1 -1 -1 -1 0\nop const 2 -1 -1 -1 1050253722\n"
            "op oscset -1 0 1 0 1\nop oscset -1 2 -1 0 3\nop oscset -1 0 1 1 1\nop oscset -1 2 -1 1 3\n"
            "op osc 3 -1 -1 0 0\nop osc 4 -1 -1 1 0\nop add 5 3 4 -1 0\nop env 6 -1 -1 2 0\nop mul 7 5 6 -1 0\n"
            "op envoff 8 -1 -1 2 0\nop if -1 8 -1 -1 0\nop stop -1 -1 -1 -1 0\nop else -1 -1 -1 -1 0\nop endif -1 -1 -1 -1 0\nret 7\nend\n")
    rc, src = check(prog, want_source=True)
    assert rc == 0, src
    for needle in ("osm_set_duty(L.n0, r2);", "L.n1d = r2;", "env_is_off(L.n2.e.stage) ? 1.f : 0.f;", "L.stage = stage_all_off(L.stage);"):
        assert needle in src, needle
    assert "f2u(L.n1d)" in src and "L.n0.duty" in src                                          # the duties are written back
    rc, msg = check(prog.replace("node 0 pulse", "node 0 fsine"))                               # a Fast::Sine has no duty
    assert rc < 0 and "no such set() / reset()" in msg
    rc, msg = check(prog.replace("op envoff 8 -1 -1 2 0", "op envoff 8 -1 -1 0 0"))
    assert rc < 0 and "not an envelope" in msg


def test_facade_folds_the_stop_idiom_and_records_finished_as_a_value():
    """tests/patches/finish_body.k: `if (adsr.finished()) { ...; stop(); return; }`, `!shape.finished()`, `shape.finished() && x < 0` record as envoff +
    structured branches; tests/patches/pwm.k: `if (adsr.finished()) stop();` is still ONE stopif, set(f, phase, duty) is oscset 1 + oscset 3, and the
    common tail after the outer `if` is there once (every member is written back on every path, idle write-backs dropped afterwards)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp")], check=True)
    def program(name):
        exe = os.path.join(ROOT, "tests", "cpp", "_bin", "facade_graph_own_" + name)
        r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", f"own_{name}_solo.scn"), "/dev/null"], env=dict(os.environ, KLANG_MI355_DUMP_GRAPH="1", HIP_VISIBLE_DEVICES="-1"), capture_output=True, text=True)
        text = r.stderr[r.stderr.index("klgg 1"):r.stderr.index("end\n") + 4]
        return text, [ln.split() for ln in text.splitlines() if ln.startswith("op ")]
    text, ops = program("finish_body")
    codes = [o[1] for o in ops]
    assert codes.count("envoff") == 3 and codes.count("stop") == 1 and "stopif" not in codes, codes
    assert codes.index("envoff") + 1 == codes.index("if") and codes[codes.index("if") + 1] == "stop"          # stop() is the `if` side, the rest of the body its `else`
    rc, msg = check(text)
    assert rc == 0, msg
    text, ops = program("pwm")
    codes = [o[1] for o in ops]
    assert codes.count("stopif") == 1 and "envoff" not in codes and codes.count("osc") == 5 and codes.count("env") == 1, codes
    sets = [(o[5], int(o[6], 16)) for o in ops if o[1] == "oscset"]
    assert sets == [("2", 1), ("2", 3), ("1", 1), ("1", 3), ("3", 1), ("3", 3)], sets
    rc, msg = check(text)
    assert rc == 0, msg


def test_an_effect_may_write_its_controls_take_abs_and_place_delay_heads_per_sample():
    """What examples/PingPong.k needs of a recorded effect (tests/golden/pingpong_recorded.klgg is the whole program): `setctl` (controls[i].set(x):
    the dial's clamp, the instance's own copy `ctlvar` that every later read and smooth() of the control takes), `abs`, `delayset` + `delayout`
    on an effect's Delay (the read head of set(), advanced by every process())."""
    prog = open(os.path.join(ROOT, "tests", "golden", "pingpong_recorded.klgg")).read()
    rc, src = check(prog, want_source=True)
    assert rc == 0, src
    for needle in ("float n7;", "const float r14 = L.n7;", "__builtin_fabsf(r26)", "L.n7 = r30;", "const float r30h = (u2f(0x3f800000u) < r25) ? u2f(0x3f800000u) : r25;", "(r25 < u2f(0x3a83126fu)) ? u2f(0x3a83126fu) : r30h;",
                   "L.n8 = L.n8 * 0.999f + (1.f - 0.999f) * L.n7;", "L.n6 = L.n6 * 0.999f + (1.f - 0.999f) * L.ctl5;", "L.ctl5 = c.ctl[5];", "L.n0t = delay_set(L.n0pos, 192000, r46);", "Rows2 h1; h1.i = L.n0t.position;", "delay_process_h(Ring{",
                   "osc.set" if False else "L.n2.position = r31;"):
        assert needle in src, needle
    rc, msg = check(prog.replace("kind effect 2\n", "").replace("ret2 67 68", "ret 67"))
    assert rc < 0                                                                      # a Note does not write its Synth's controls (nor has it `in`)


def test_the_recorded_config_4_effects_get_a_staged_form():
    """klang_amd/csrc/klg_graph_staged.hpp on the CPU (hipRTC needs no GPU): the recorded PingPong.k and Reverb.k both get the sample-parallel kernel beside
    klg_fx_graph<P> — PingPong.k with its control path (the dial smoothers, the scratch detector, the LFO's phase, the delay times) a chunk ahead of the audio
    path, the LFO's sine taken per (sample, instance), the DC filters as serial loops; Reverb.k's twenty biquads as serial loops between two parallel levels.
    A parallel level's ring reads come in two halves (rows + check + loads, then the arithmetic) with several under way at a time; a read recorded inside an
    `if` is taken outside it, its check counting under the branch's condition alone.  KLG_FX_STAGED=0 leaves the one-lane kernel alone;
    KLG_FX_STAGED_BATCH=0 writes every read in one piece."""
    for name, needles in (("pingpong_recorded", ("extern \"C\" __global__", "klg_fx_staged", "beside the", "basic_sine_of(", "basic_sine_arg(L.n2)", "staged_process_fetch(RingS{ (const char*)(ring0 + ", "staged_process_finish(tf", "xc_d0t = xn_d0t;", "plain(s0 + OFF, CC, false, OFF)", "a part of the chunk, level 0")),
                          ("reverb_recorded", ("klg_fx_staged", "staged_tap_stereo_fetch(RingS{", "bad |= ((r", "asm volatile(\"\" ::: \"memory\");", "staged_tap_stereo_finish(tf", "biquad_process(L.n", "if (bad && k0 + pg < a.K) *flag = 1;", "plain(s0 + OFF, CC, false, OFF)", "a part of the chunk, level 0"))):
        prog = open(os.path.join(ROOT, "tests", "golden", name + ".klgg")).read()
        rc, src = check(prog, want_source=True)
        assert rc == 0, src
        for needle in needles:
            assert needle in src, (name, needle)
    os.environ["KLG_FX_STAGED"] = "0"
    try:
        rc, src = check(open(os.path.join(ROOT, "tests", "golden", "pingpong_recorded.klgg")).read(), want_source=True)
        assert rc == 0 and "klg_fx_staged" not in src
    finally:
        del os.environ["KLG_FX_STAGED"]
    # Reverb.k's sixteen damping filters are the same code on different nodes: four of them side by side in a wave's lanes (lane = strand x instance), what differs by quarter
    rc, src = check(open(os.path.join(ROOT, "tests", "golden", "reverb_recorded.klgg")).read(), want_source=True)
    assert rc == 0 and "const int qq = ln / G, gi = ln - qq * G;" in src and "ln < 64) {" in src and "(qq == 0 ? 0 : (qq == 1 ? 13 : (qq == 2 ? 26 : 39)))" in src
    os.environ["KLG_FX_STAGED_PACK"] = "0"
    try:
        rc, src = check(open(os.path.join(ROOT, "tests", "golden", "reverb_recorded.klgg")).read(), want_source=True)
        assert rc == 0 and "const int qq = ln / G" not in src
    finally:
        del os.environ["KLG_FX_STAGED_PACK"]
    os.environ["KLG_FX_STAGED_BATCH"] = "0"
    try:
        rc, src = check(open(os.path.join(ROOT, "tests", "golden", "reverb_recorded.klgg")).read(), want_source=True)
        assert rc == 0 and "staged_tap_stereo(RingS{" in src and "TapFetch tf" not in src
    finally:
        del os.environ["KLG_FX_STAGED_BATCH"]
    # a chunk that fails its ring check is tried again in halves and quarters (the audio path alone; the control path's serial loops catch up on the
    # architectural records only in front of a plain walk): KLG_FX_STAGED_RETRY=0 sends it straight to the plain body
    rc, src = check(open(os.path.join(ROOT, "tests", "golden", "pingpong_recorded.klgg")).read(), want_source=True)
    assert rc == 0 and "while (OFF < cl)" in src and "if (OFF > ctl_at) { const int part_at = OFF;" in src and "for (int sb = OFF; sb < OFF + CC; sb += 8)" in src
    os.environ["KLG_FX_STAGED_RETRY"] = "0"
    try:
        rc, src = check(open(os.path.join(ROOT, "tests", "golden", "pingpong_recorded.klgg")).read(), want_source=True)
        assert rc == 0 and "while (OFF < cl)" not in src and "plain(s0, cl, false, 0)" in src
    finally:
        del os.environ["KLG_FX_STAGED_RETRY"]


def test_a_conditional_delay_input_keeps_the_one_lane_kernel():
    """What the staged form refuses: an input() of a Delay inside an `if` (its write cursor would depend on the samples).  The program still compiles — as
    klg_fx_graph<P> only."""
    prog = """klgg 1
kind effect 1
ctl 1
dial 0 0 1 0.5
node 0 delay 4800
op in 0 -1 -1 -1 00000000
op ctl 1 -1 -1 -1 00000000
op const 2 -1 -1 -1 3f000000
op cmp 3 1 2 -1 00000001
op if -1 3 -1 -1 00000000
op delayin -1 0 -1 0 00000000
op endif -1 -1 -1 -1 00000000
op const 4 -1 -1 -1 42c80000
op delaytap 5 4 -1 0 00000000
op add 6 0 5 -1 00000000
ret 6
end
"""
    rc, src = check(prog, want_source=True)
    assert rc == 0, src
    assert "klg_fx_graph" in src or "struct PatchGen" in src
    assert "klg_fx_staged" not in src


def test_code_objects_are_cached_on_disk(tmp_path):
    """A second PROCESS that compiles the same program finds the first one's code object (klg_graph.hpp cache_file: keyed by the generated source, the headers'
    bytes, the compiler's version and options) instead of running hipRTC again; KLG_CACHE=0 writes nothing; a damaged file is ignored and replaced."""
    import sys
    import time
    code = ("import sys, time; sys.path.insert(0, %r); import ctypes as C; from klang_amd._lib import lib; L = lib(); b = C.create_string_buffer(1 << 16); "
            "t = time.perf_counter(); rc = L.klg_graph_check(%r.encode(), 0, b, len(b)); print(rc, time.perf_counter() - t)") % (ROOT, SUB2B_LIKE.replace("0.1 20 10", "0.1 20 9.5"))
    def run(env_extra):
        env = dict(os.environ, KLG_CACHE_DIR=str(tmp_path / "cache"), **env_extra)
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, check=True).stdout.split()
        return int(out[0]), float(out[1])
    rc, _ = run({"KLG_CACHE": "0"})
    assert rc == 0 and not (tmp_path / "cache").exists() or not list((tmp_path / "cache").glob("*.klgco"))
    rc, cold = run({})
    files = list((tmp_path / "cache").glob("*.klgco"))
    assert rc == 0 and len(files) == 1
    rc, warm = run({})
    assert rc == 0 and warm < 0.25 * cold and warm < 0.5, (cold, warm)
    blob = bytearray(files[0].read_bytes()); blob[len(blob) // 2] ^= 0xFF; files[0].write_bytes(bytes(blob))      # a damaged file: compiled again, rewritten
    rc, again = run({})
    assert rc == 0 and again > 2 * warm
    rc, warm2 = run({})
    assert rc == 0 and warm2 < 0.25 * cold


def test_random_effect_programs_compile_for_gfx950_with_a_staged_form():
    """tests/fx_fuzz.py: the random effect programs test_gpu_fx_fuzz.py runs through both generated kernels — here that they parse, get the sample-parallel form
    (with the parts of a failed chunk) and compile for gfx950, without a device."""
    from fx_fuzz import program                      # (tests/ is on sys.path: conftest.py / rootdir)
    for seed, ch in ((0, 2), (5, 2), (14, 1)):
        prog, what, _ = program(seed, channels=ch)
        rc, src = check(prog, want_source=True)
        assert rc == 0, (seed, what, src[:2000])
        assert "klg_fx_staged" in src and "while (OFF < cl)" in src and "no sample-parallel form" not in src, (seed, what)
    # a line read by process() that is no longer than 1,024 samples keeps the one-lane kernel, and the source says why
    prog, _, _ = program(0)
    rc, src = check(prog.replace("delay 4096", "delay 512"), want_source=True)
    assert rc == 0 and "// no sample-parallel form (klg_graph_staged.hpp): a Delay shorter than a chunk's inputs" in src and "klg_fx_staged" not in src
