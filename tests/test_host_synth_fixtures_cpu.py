"""tests/golden/host_synth_{modular,inheritance}.npz (the mono Synth's own block entry, Synth::process(float*, int), klang.h:4450-4457, through tests/hosts/synth_host.cpp
against the GENUINE header) against the per-voice fixtures of the same scenarios (oracle/ref/ref_examples through the Stereo::Synth entry: another program of the reference
build): the note stages after every block are the same whichever entry rendered it, the host copies the mono block to both channels, and a block in which no note sounds is
left as it was handed over (zeros)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OFF = 3


@pytest.mark.parametrize("name", ["modular", "inheritance"])
def test_mono_entry_fixture_agrees_with_the_per_voice_fixture_of_the_same_scenario(name):
    ex = np.load(os.path.join(GOLDEN, f"ex_{name}.npz"))
    host = np.load(os.path.join(GOLDEN, f"host_synth_{name}.npz"))
    assert np.array_equal(ex["stages"], host["stages"])
    mix, st = host["mix"], host["stages"]
    assert np.array_equal(mix[:, 0].view(np.uint32), mix[:, 1].view(np.uint32))
    assert np.isfinite(mix).all() and np.abs(mix).max() > 0.05
    sounding = [(st[b] != OFF).any() or (b > 0 and (st[b - 1] != OFF).any()) for b in range(st.shape[0])]
    for b, s in enumerate(sounding):
        if not s:
            assert not mix[b].any(), f"block {b}: nobody sounds, the block stays as handed over"
    # (that the block is the LAST sounding note's alone is shown on the GPU: tests/test_gpu_hosts.py compares the façade's mono entry with this fixture bit for bit)
