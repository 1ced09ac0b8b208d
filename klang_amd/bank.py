"""SynthBank / FxBank: Python mirrors of the reference's host-facing interface
(Synth::noteOn / noteOff / onControl / process(float**, int) — klang.h:4399-4466, 4789-4858),
implemented by forwarding to the C-ABI."""
import ctypes as C

import numpy as np

from ._lib import KlangError, check, lib

PATCH_IDS = {"sine": 0, "bsine": 1, "sub2a": 2, "sub2b": 3, "supersaw": 4, "fm3": 5, "fm4": 6, "pingpong": 7, "reverb": 8}
F32P = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(F32P)


def init(devices):
    """klg_init: the GPUs of this process.  One id = every bank on that GPU; several ids = synth banks created afterwards are sharded over
    them inside the library (contiguous ranges of synth instances, one RCCL all-reduce of the stereo block per klg_process)."""
    ids = (C.c_int * len(devices))(*[int(d) for d in devices])
    check(lib().klg_init(ids, len(devices)), "klg_init")


class SynthBank:
    """`synths` instances of one patch, `notes` Note slots each; one GPU lane per voice."""

    def __init__(self, patch, synths=1, notes=32, fs=48000.0, max_block=256, device=None):
        self._L = lib()
        self._h = None
        if device is not None:
            ids = (C.c_int * 1)(int(device))
            check(self._L.klg_init(ids, 1), "klg_init")
        if isinstance(patch, str) and patch.lstrip().startswith("klgg"):       # a graph program (include/klang_mi355_graph.h)
            h = self._L.klg_synth_create_graph(patch.encode(), int(synths), int(notes), float(fs), int(max_block))
            patch = "graph"
        else:
            pid = PATCH_IDS[patch] if isinstance(patch, str) else int(patch)
            h = self._L.klg_synth_create(pid, int(synths), int(notes), float(fs), int(max_block))
        if not h:
            raise KlangError("klg_synth_create failed: " + self._L.klg_last_error().decode())
        self._h = h
        self.patch, self.synths, self.notes, self.fs, self.max_block = patch, synths, notes, fs, max_block
        self.voices = self._L.klg_synth_voices(h)
        self.state_bytes = self._L.klg_synth_state_bytes(h)
        self.n_controls = self._L.klg_synth_controls(h)

    def close(self):
        if self._h:
            self._L.klg_synth_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- events (host side of the reference API) ---
    def random(self, seed):
        self._L.klg_random_seed(int(seed) & 0xFFFFFFFF)

    def note_on(self, synth, pitch, velocity=1.0):
        return check(self._L.klg_note_on(self._h, int(synth), int(pitch), float(velocity)), "klg_note_on")

    def note_off(self, synth, pitch, velocity=0.0):
        return check(self._L.klg_note_off(self._h, int(synth), int(pitch), float(velocity)), "klg_note_off")

    def _many(self, fn, name, synths, pitches, velocities):
        sy = np.ascontiguousarray(synths, dtype=np.int32); pi = np.ascontiguousarray(pitches, dtype=np.int32)
        ve = np.ascontiguousarray(velocities, dtype=np.float32)
        assert sy.shape == pi.shape == ve.shape
        ip = C.POINTER(C.c_int)
        return check(fn(self._h, len(sy), sy.ctypes.data_as(ip), pi.ctypes.data_as(ip), _fp(ve)), name)

    def note_on_many(self, synths, pitches, velocities):
        return self._many(self._L.klg_note_on_many, "klg_note_on_many", synths, pitches, velocities)

    def note_off_many(self, synths, pitches, velocities):
        return self._many(self._L.klg_note_off_many, "klg_note_off_many", synths, pitches, velocities)

    def set_control(self, synth, index, value):
        return check(self._L.klg_set_control(self._h, int(synth), int(index), float(value)), "klg_set_control")

    def get_control(self, synth, index):
        v = C.c_float()
        check(self._L.klg_get_control(self._h, int(synth), int(index), C.byref(v)), "klg_get_control")
        return v.value

    def set_control_smoothed(self, synth, index, value):
        """Control::smoothed (klang.h:1707): what `controls[index].smooth()` in a recorded Note advances — the Synth's, shared by its notes."""
        return check(self._L.klg_set_control_smoothed(self._h, int(synth), int(index), float(value)), "klg_set_control_smoothed")

    def get_control_smoothed(self, synth, index):
        v = C.c_float()
        check(self._L.klg_get_control_smoothed(self._h, int(synth), int(index), C.byref(v)), "klg_get_control_smoothed")
        return v.value

    # --- blocks ---
    def process(self, out, parameters=None):
        """out: float32 [channels][n] (accumulated into, like Stereo::Synth::process)."""
        assert out.dtype == np.float32 and out.ndim == 2 and out.flags.c_contiguous
        ch, n = out.shape
        ptrs = (F32P * ch)(*[_fp(out[c]) for c in range(ch)])
        par = _fp(parameters) if parameters is not None else None
        check(self._L.klg_process(self._h, ptrs, ch, n, par), "klg_process")
        return out

    @property
    def note_channels(self):
        """1: a voice's `out` is mono; 2: the bank renders Stereo::Notes with a stereo `out` (klg_synth_note_channels)"""
        return int(self._L.klg_synth_note_channels(self._h))

    def process_voices(self, n, out=None):
        pv = np.empty((self.voices, n) if self.note_channels == 1 else (self.voices, 2, n), dtype=np.float32)
        if out is None:
            out = np.zeros((2, n), dtype=np.float32)
        ch = out.shape[0]
        ptrs = (F32P * ch)(*[_fp(out[c]) for c in range(ch)])
        check(self._L.klg_process_voices(self._h, _fp(pv), ptrs, ch, n), "klg_process_voices")
        return pv, out

    def note_records(self, synths, pitches, velocities):
        """the patch's on() for each (synth, pitch, velocity): uint32 [n][words], nothing queued (klg_note_records)"""
        sy = np.ascontiguousarray(synths, dtype=np.int32); pi = np.ascontiguousarray(pitches, dtype=np.int32); ve = np.ascontiguousarray(velocities, dtype=np.float32)
        assert sy.shape == pi.shape == ve.shape
        out = np.empty((len(sy), self.state_bytes // 4), dtype=np.uint32)
        ip = C.POINTER(C.c_int)
        check(self._L.klg_note_records(self._h, len(sy), sy.ctypes.data_as(ip), pi.ctypes.data_as(ip), _fp(ve), out.ctypes.data_as(C.c_void_p)), "klg_note_records")
        return out

    def set_mix_mode(self, mode):
        check(self._L.klg_synth_set_mix_mode(self._h, int(mode)), "klg_synth_set_mix_mode")

    def stages(self):
        st = np.empty(self.voices, dtype=np.uint8)
        check(self._L.klg_voice_stages(self._h, st.ctypes.data_as(C.POINTER(C.c_uint8)), self.voices), "klg_voice_stages")
        return st

    def process_device(self, d_mix_ptr, n, stream=None):
        """stream: a hipStream_t handle; None / 0 = the bank's OWN stream, which is non-blocking — work queued on torch's default stream (a clear of the
        mix, say) is not ordered with it: pass `torch.cuda.Stream().cuda_stream` and make that stream current, or synchronise first."""
        check(self._L.klg_process_device(self._h, C.c_void_p(int(d_mix_ptr)), int(n), C.c_void_p(int(stream)) if stream else None), "klg_process_device")

    def sync(self):
        check(self._L.klg_sync(self._h), "klg_sync")

    def voice_download(self, voice):
        buf = np.empty(self.state_bytes // 4, dtype=np.uint32)
        check(self._L.klg_voice_download(self._h, int(voice), buf.ctypes.data_as(C.c_void_p), self.state_bytes), "klg_voice_download")
        return buf

    def voice_upload(self, voice, words):
        words = np.ascontiguousarray(words, dtype=np.uint32)
        check(self._L.klg_voice_upload(self._h, int(voice), words.ctypes.data_as(C.c_void_p), words.nbytes), "klg_voice_upload")

    def voices_upload(self, voices, words):
        """words[i] (one record each) replaces the state of voices[i]; applied at the start of the next block."""
        voices = np.ascontiguousarray(voices, dtype=np.int32)
        words = np.ascontiguousarray(words, dtype=np.uint32).reshape(len(voices), self.state_bytes // 4)
        check(self._L.klg_voices_upload(self._h, len(voices), voices.ctypes.data_as(C.POINTER(C.c_int)), words.ctypes.data_as(C.c_void_p)), "klg_voices_upload")

    @property
    def voices_per_lane(self):
        return self._L.klg_synth_voices_per_lane(self._h)

    def table_upload(self, samples, dedup=True):
        """Copies a sample table to HBM (graph banks); returns its id: a wavetable node's `table` word / a tabread op's imm."""
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        tid = self._L.klg_table_upload(self._h, samples.ctypes.data_as(C.POINTER(C.c_float)), samples.size, 1 if dedup else 0)
        if tid < 0:
            check(tid, "klg_table_upload")
        return tid

    def timing_begin(self):
        check(self._L.klg_timing_begin(self._h), "klg_timing_begin")

    def timing_end_aux(self):
        """(launches, ms) of the blocks' event kernels and reduces since timing_begin (klg_timing_end_aux); before timing_end"""
        n, ms = C.c_int(), C.c_float()
        check(self._L.klg_timing_end_aux(self._h, C.byref(n), C.byref(ms)), "klg_timing_end_aux")
        return n.value, ms.value

    def timing_end(self):
        n, ms = C.c_int(), C.c_float()
        check(self._L.klg_timing_end(self._h, C.byref(n), C.byref(ms)), "klg_timing_end")
        return n.value, ms.value

    def multi_info(self, n=256, probe_reps=0):
        """klg_synth_multi_info: {shards, rccl_ranks (what the communicator reports), distinct_devices, per_shard_kernel_ms (since timing_begin; before timing_end),
        allreduce_us (mean of `probe_reps` all-reduces of a [2][n] block, measured by the library)}"""
        sh, rk, dd, us = C.c_int(), C.c_int(), C.c_int(), C.c_float()
        per = (C.c_float * 64)()
        check(self._L.klg_synth_multi_info(self._h, n, probe_reps, C.byref(sh), C.byref(rk), C.byref(dd), per, 64, C.byref(us)), "klg_synth_multi_info")
        return {"shards": sh.value, "rccl_ranks": rk.value, "distinct_devices": dd.value, "per_shard_kernel_ms": [per[i] for i in range(min(sh.value, 64))], "allreduce_us": us.value}


class EventScript:
    """An event stream known in advance, resident in HBM (klg_script_*): every on() runs on the host once, up front; the blocks then
    play with no host work between them.  Voices are addressed explicitly."""

    def __init__(self, bank, blocks):
        self._L = lib()
        self.bank, self.blocks = bank, int(blocks)
        self._k = self._L.klg_script_create(bank._h, self.blocks)
        if not self._k:
            raise KlangError("klg_script_create failed: " + self._L.klg_last_error().decode())

    def close(self):
        if self._k:
            self._L.klg_script_destroy(self._k)
            self._k = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_records(self, records):
        """records: uint32 [n][words]; returns the pool index of the first"""
        r = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1, self.bank.state_bytes // 4)
        return check(self._L.klg_script_add_records(self._k, len(r), r.ctypes.data_as(C.c_void_p)), "klg_script_add_records")

    def note_on(self, blocks, voices, record_indices):
        ip = C.POINTER(C.c_int)
        b = np.ascontiguousarray(blocks, dtype=np.int32); v = np.ascontiguousarray(voices, dtype=np.int32); r = np.ascontiguousarray(record_indices, dtype=np.int32)
        assert b.shape == v.shape == r.shape
        check(self._L.klg_script_note_on_many(self._k, len(b), b.ctypes.data_as(ip), v.ctypes.data_as(ip), r.ctypes.data_as(ip)), "klg_script_note_on_many")

    def note_off(self, blocks, voices):
        ip = C.POINTER(C.c_int)
        b = np.ascontiguousarray(blocks, dtype=np.int32); v = np.ascontiguousarray(voices, dtype=np.int32)
        assert b.shape == v.shape
        check(self._L.klg_script_note_off_many(self._k, len(b), b.ctypes.data_as(ip), v.ctypes.data_as(ip)), "klg_script_note_off_many")

    def commit(self):
        check(self._L.klg_script_commit(self._k), "klg_script_commit")

    def render_device(self, first_block, blocks, d_out_ptr, n, stream=None):
        """blocks first_block .. + blocks - 1 into d_out[blocks][2][n] (overwritten), one call (klg_script_render_device)"""
        check(self._L.klg_script_render_device(self._k, int(first_block), int(blocks), C.c_void_p(int(d_out_ptr)), int(n), C.c_void_p(int(stream)) if stream else None), "klg_script_render_device")

    def capture_span(self, first_block, blocks, d_out_ptr, n, stream=None):
        """capture the span as a hipGraph without running it; False when it cannot be captured now (klg_script_capture_span)"""
        return self._L.klg_script_capture_span(self._k, int(first_block), int(blocks), C.c_void_p(int(d_out_ptr)), int(n), C.c_void_p(int(stream)) if stream else None) == 0

    def play_device(self, block, d_mix_ptr, n, stream=None):
        check(self._L.klg_script_play_device(self._k, int(block), C.c_void_p(int(d_mix_ptr)), int(n), C.c_void_p(int(stream)) if stream else None), "klg_script_play_device")


class FxBank:
    """`instances` independent Stereo::Effect objects of one patch (PingPong.k / Reverb.k)."""

    def __init__(self, patch, instances, fs=48000.0, max_block=256, initial_record=None, channels=2, device=None):
        self._L = lib()
        self._h = None
        self.channels = 2
        # device: THIS bank's GPU (klg_fx_create_on) — the process-wide selection of klg_init(), and every other bank's device, stay as they are
        if isinstance(patch, str) and patch.lstrip().startswith("klgg"):       # a recorded Effect::process() body (`kind effect`)
            rec = None if initial_record is None else np.ascontiguousarray(initial_record, dtype=np.uint32)
            recp = rec.ctypes.data_as(C.c_void_p) if rec is not None else None
            if device is not None:
                h = self._L.klg_fx_create_on(int(device), -1, patch.encode(), int(instances), float(fs), int(max_block), recp)
            else:
                h = self._L.klg_fx_create_graph(patch.encode(), int(instances), float(fs), int(max_block), recp)
            self.channels = int(channels)
        else:
            pid = PATCH_IDS[patch] if isinstance(patch, str) else int(patch)
            if device is not None:
                h = self._L.klg_fx_create_on(int(device), pid, None, int(instances), float(fs), int(max_block), None)
            else:
                h = self._L.klg_fx_create(pid, int(instances), float(fs), int(max_block))
        if not h:
            raise KlangError("klg_fx_create failed: " + self._L.klg_last_error().decode())
        self._h = h
        self.instances = instances
        self.state_bytes = self._L.klg_fx_state_bytes(h)

    def close(self):
        if self._h:
            self._L.klg_fx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_control(self, instance, index, value):
        return check(self._L.klg_fx_set_control(self._h, int(instance), int(index), float(value)), "klg_fx_set_control")

    def get_control(self, instance, index):
        """The control as the effect left it (klg_fx_get_control): a control its process() writes comes back from the instance's state."""
        v = C.c_float()
        check(self._L.klg_fx_get_control(self._h, int(instance), int(index), C.byref(v)), "klg_fx_get_control")
        return v.value

    def graph_form(self):
        """How a graph effect bank runs its body: dict(staged, instances_per_workgroup, samples_per_chunk, levels, lds_values, why) (klg_fx_graph_form)."""
        g, c, lv, sl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        why = C.create_string_buffer(512)
        rc = check(self._L.klg_fx_graph_form(self._h, C.byref(g), C.byref(c), C.byref(lv), C.byref(sl), why, len(why)), "klg_fx_graph_form")
        return dict(staged=bool(rc), instances_per_workgroup=g.value, samples_per_chunk=c.value, levels=lv.value, lds_values=sl.value, why=why.value.decode())

    def record_words(self):
        return check(self._L.klg_fx_record_words(self._h), "klg_fx_record_words")

    def download_record(self, instance):
        """The instance's record as the device last left it (uint32 words): what a host-run prepare() starts from (klg_fx_download_record)."""
        w = np.empty(self.record_words(), dtype=np.uint32)
        check(self._L.klg_fx_download_record(self._h, int(instance), w.ctypes.data_as(C.c_void_p), w.nbytes), "klg_fx_download_record")
        return w

    def upload_words(self, instance, first, words):
        """Overwrite words [first, first + len(words)) of the instance's record before the next block (klg_fx_upload_words)."""
        w = np.ascontiguousarray(words, dtype=np.uint32)
        check(self._L.klg_fx_upload_words(self._h, int(instance), int(first), int(w.size), w.ctypes.data_as(C.c_void_p)), "klg_fx_upload_words")

    def process(self, io):
        """io: float32 [instances][channels][n], processed in place."""
        assert io.dtype == np.float32 and io.flags.c_contiguous and io.shape[:2] == (self.instances, self.channels)
        check(self._L.klg_fx_process(self._h, _fp(io), io.shape[2]), "klg_fx_process")
        return io

    def process_device(self, d_io_ptr, n, stream=None):
        check(self._L.klg_fx_process_device(self._h, C.c_void_p(int(d_io_ptr)), int(n), C.c_void_p(int(stream)) if stream else None), "klg_fx_process_device")

    def render_device(self, d_io_ptr, blocks, n, stream=None):
        """a span of `blocks` blocks: d_io [blocks][instances][channels][n] on the device, in place (klg_fx_render_device)"""
        check(self._L.klg_fx_render_device(self._h, C.c_void_p(int(d_io_ptr)), int(blocks), int(n), C.c_void_p(int(stream)) if stream else None), "klg_fx_render_device")

    def sync(self):
        check(self._L.klg_fx_sync(self._h), "klg_fx_sync")

    def timing_begin(self):
        check(self._L.klg_fx_timing_begin(self._h), "klg_fx_timing_begin")

    def timing_end(self):
        n, ms = C.c_int(), C.c_float()
        check(self._L.klg_fx_timing_end(self._h, C.byref(n), C.byref(ms)), "klg_fx_timing_end")
        return n.value, ms.value
