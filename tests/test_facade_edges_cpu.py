"""The edge of the path (VERDICT r5 "what's missing" 6): pieces of klang.h's interface no shipped patch uses.  Host-side checks against the facade header — no GPU:
the programs stop (or finish) before any device work."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "klang_amd")
HEAD = '#include <klang.h>\nusing namespace klang::optimised;\n'


def build_and_run(tmp_path, name, text):
    src = tmp_path / (name + ".cpp")
    src.write_text(HEAD + text)
    exe = tmp_path / name
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + INC, "-I" + os.path.join(INC, "klang"), str(src), "-L" + LIBDIR, "-lklang_mi355", "-Wl,-rpath," + LIBDIR, "-o", str(exe)], check=True)
    return subprocess.run([str(exe)], capture_output=True, text=True)


def test_a_user_ramp_stops_with_a_message(tmp_path):
    """Envelope::set(Ramp*) (klang.h:4057-4060): `new Envelope::Linear()` is accepted (the fixture own_leftovers does it in on(), tests/test_gpu_facade.py); a user subclass of
    Envelope::Ramp — its own operator++ — has no device form and must stop the program with a message, not render a line."""
    r = build_and_run(tmp_path, "user_ramp",
                      'struct Expo : Envelope::Ramp { signal operator++(int) override { const signal o = out; out += (target - out) * rate; return o; } };\n'
                      'int main() { Envelope e; e.set(new Envelope::Linear()); std::puts("linear ok"); std::fflush(stdout); e.set(new Expo()); std::puts("not reached"); return 0; }\n')
    assert "linear ok" in r.stdout and "not reached" not in r.stdout
    assert r.returncode != 0 and "user-defined Ramp" in r.stderr


def test_the_linear_ramp_steps_as_in_the_reference(tmp_path):
    """Envelope::Linear on the host (klang.h:3781-3807): out, then one step of `rate` towards the target, clamped where it arrives."""
    r = build_and_run(tmp_path, "linear_ramp",
                      'int main() { klang::fs = 48000; Envelope::Linear up(0.f, 1.f, 0.0001f), down(1.f, 0.25f, 0.0002f);\n'
                      '  for (int i = 0; i < 10; i++) { const signal a = up++; const signal b = down++; std::printf("%.9g %.9g %d %d\\n", (float)a, (float)b, (int)up.isActive(), (int)down.isActive()); }\n'
                      '  return 0; }\n')
    assert r.returncode == 0, r.stderr
    import numpy as np
    rate_up, rate_dn = np.float32(1.0) / (np.float32(0.0001) * np.float32(48000)), np.float32(1.0) / (np.float32(0.0002) * np.float32(48000))
    u, d = np.float32(0), np.float32(1)
    rows = [ln.split() for ln in r.stdout.strip().splitlines()]
    for row in rows:
        assert np.float32(row[0]) == u and np.float32(row[1]) == d
        u = min(np.float32(u + rate_up), np.float32(1))
        d = max(np.float32(d - rate_dn), np.float32(0.25))
    assert rows[-1][2] == "0" and rows[-1][3] == "0"                     # both have arrived


def test_wavetables_sine_and_saw_hold_a_cycle_of_the_basic_oscillators(tmp_path):
    """Generators::Wavetables::{Sine,Saw} (klang.h:5369-5380): 2,048 samples, one cycle of Basic::Sine / Basic::Saw rendered by the constructor at fs / 2048 Hz."""
    r = build_and_run(tmp_path, "wavetables",
                      'int main() { klang::fs = 48000; Generators::Wavetables::Sine s; Generators::Wavetables::Saw w;\n'
                      '  const int at[5] = { 0, 1, 512, 1024, 2047 };\n'
                      '  for (int i : at) std::printf("%.9g %.9g\\n", (float)s[i], (float)w[i]);\n'
                      '  return 0; }\n')
    assert r.returncode == 0, r.stderr
    import numpy as np
    rows = [[float(x) for x in ln.split()] for ln in r.stdout.strip().splitlines()]
    assert rows[0][0] == 0.0 and abs(rows[2][0] - 1.0) < 1e-6 and abs(rows[3][0]) < 1e-5        # sin at 0, a quarter, a half of the cycle
    assert rows[0][1] == -1.0 and abs(rows[3][1]) < 1e-5 and rows[4][1] > 0.99                  # the saw rises from -1 through 0 to just below 1
