"""ctypes loader for libklang_mi355.so (the C-ABI of include/klang_mi355.h).

There is no fallback of any kind: if the shared library is missing or a symbol is absent, importing
fails loudly; if no gfx950 device is visible, every create/process call raises KlangError.
"""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KLANG_MI355_LIB") or os.path.join(HERE, "libklang_mi355.so")   # (KLANG_MI355_LIB: another build of the same library, for A/B measurements: tools/noise_bench.py)
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "klang_mi355.h")


class KlangError(RuntimeError):
    pass


def declared_symbols(header=HEADER_PATH):
    """Every function name include/klang_mi355.h declares."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(klg_[a-z0-9_]+)\s*\(", text)))


def load():
    if not os.path.exists(LIB_PATH):
        raise KlangError(f"{LIB_PATH} is missing: build it with klang_amd/csrc/build.sh (python -c 'import __graft_entry__ as g; g.build()'). "
                         "klang_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  If torch is going to be
    # used in this process (bench.py, torch.distributed) it must be loaded FIRST so that libklang_mi355.so's
    # DT_NEEDED libamdhip64 resolves to the already-loaded copy; two runtimes in one process fight over the
    # device (observed: hipGetDeviceCount() == 0 in the second one).  C/C++ hosts have no torch and no issue.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    f32p, u8p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.c_void_p
    sig = {
        "klg_last_error": (C.c_char_p, []),
        "klg_version": (C.c_int, []),
        "klg_init": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
        "klg_random_seed": (None, [C.c_uint]),
        "klg_rand_sync": (C.c_int, []),
        "klg_rand_fill_device": (C.c_int, [vp, C.c_size_t, C.c_uint, C.c_int, vp]),
        "klg_synth_create": (vp, [C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]),
        "klg_synth_destroy": (None, [vp]),
        "klg_synth_voices": (C.c_int, [vp]),
        "klg_synth_controls": (C.c_int, [vp]),
        "klg_synth_note_channels": (C.c_int, [vp]),
        "klg_synth_state_bytes": (C.c_size_t, [vp]),
        "klg_note_on": (C.c_int, [vp, C.c_int, C.c_int, C.c_float]),
        "klg_note_off": (C.c_int, [vp, C.c_int, C.c_int, C.c_float]),
        "klg_note_on_many": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), f32p]),
        "klg_note_off_many": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), f32p]),
        "klg_set_control": (C.c_int, [vp, C.c_int, C.c_int, C.c_float]),
        "klg_get_control": (C.c_int, [vp, C.c_int, C.c_int, f32p]),
        "klg_fx_get_control": (C.c_int, [vp, C.c_int, C.c_int, f32p]),
        "klg_fx_record_words": (C.c_int, [vp]),
        "klg_fx_download_record": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
        "klg_fx_upload_words": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
        "klg_set_control_smoothed": (C.c_int, [vp, C.c_int, C.c_int, C.c_float]),
        "klg_get_control_smoothed": (C.c_int, [vp, C.c_int, C.c_int, f32p]),
        "klg_process": (C.c_int, [vp, C.POINTER(f32p), C.c_int, C.c_int, f32p]),
        "klg_process_voices": (C.c_int, [vp, f32p, C.POINTER(f32p), C.c_int, C.c_int]),
        "klg_synth_set_mix_mode": (C.c_int, [vp, C.c_int]),
        "klg_voice_stages": (C.c_int, [vp, u8p, C.c_int]),
        "klg_process_device": (C.c_int, [vp, vp, C.c_int, vp]),
        "klg_sync": (C.c_int, [vp]),
        "klg_note_record": (C.c_int, [vp, C.c_int, C.c_int, C.c_float, vp, C.c_size_t]),
        "klg_script_create": (vp, [vp, C.c_int]),
        "klg_script_destroy": (None, [vp]),
        "klg_script_add_record": (C.c_int, [vp, vp, C.c_size_t]),
        "klg_script_note_on": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "klg_script_note_off": (C.c_int, [vp, C.c_int, C.c_int]),
        "klg_note_records": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), f32p, vp]),
        "klg_script_add_records": (C.c_int, [vp, C.c_int, vp]),
        "klg_script_note_on_many": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "klg_script_note_off_many": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "klg_script_commit": (C.c_int, [vp]),
        "klg_script_play_device": (C.c_int, [vp, C.c_int, vp, C.c_int, vp]),
        "klg_script_render_device": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp]),
        "klg_script_capture_span": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, vp]),
        "klg_voice_download": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
        "klg_voice_upload": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
        "klg_voices_upload": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int), vp]),
        "klg_table_upload": (C.c_int, [vp, C.POINTER(C.c_float), C.c_int, C.c_int]),
        "klg_voice_delay_clear": (C.c_int, [vp, C.c_int, C.c_int]),
        "klg_synth_voices_per_lane": (C.c_int, [vp]),
        "klg_synth_create_graph": (vp, [C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_int]),
        "klg_graph_check": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
        "klg_fx_create_graph": (vp, [C.c_char_p, C.c_int, C.c_float, C.c_int, vp]),
        "klg_fx_graph_form": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]),
        "klg_timing_begin": (C.c_int, [vp]),
        "klg_timing_end": (C.c_int, [vp, C.POINTER(C.c_int), f32p]),
        "klg_timing_end_aux": (C.c_int, [vp, C.POINTER(C.c_int), f32p]),
        "klg_synth_multi_info": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), f32p, C.c_int, f32p]),
        "klg_selftest": (C.c_int, [C.c_int, f32p, C.c_int, f32p, C.c_int, f32p, C.c_int, C.c_int]),
        "klg_selftest_host": (C.c_int, [C.c_int, f32p, C.c_int, C.c_float, f32p, C.c_int]),
        "klg_fx_create": (vp, [C.c_int, C.c_int, C.c_float, C.c_int]),
        "klg_fx_create_on": (vp, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_float, C.c_int, vp]),
        "klg_fx_destroy": (None, [vp]),
        "klg_fx_set_control": (C.c_int, [vp, C.c_int, C.c_int, C.c_float]),
        "klg_fx_process": (C.c_int, [vp, f32p, C.c_int]),
        "klg_fx_process_device": (C.c_int, [vp, vp, C.c_int, vp]),
        "klg_fx_render_device": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
        "klg_fx_sync": (C.c_int, [vp]),
        "klg_fx_state_bytes": (C.c_size_t, [vp]),
        "klg_fx_timing_begin": (C.c_int, [vp]),
        "klg_fx_timing_end": (C.c_int, [vp, C.POINTER(C.c_int), f32p]),
    }
    for name in declared_symbols():
        if not hasattr(lib, name):
            if os.environ.get("KLANG_MI355_LIB"):      # an older build under A/B measurement: it simply lacks the newer entries
                continue
            raise KlangError(f"libklang_mi355.so does not export {name} declared in include/klang_mi355.h")
        if name not in sig:
            raise KlangError(f"klang_amd/_lib.py has no signature for {name}")
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = sig[name]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(rc, what):
    if rc is None or (isinstance(rc, int) and rc < 0):
        msg = lib().klg_last_error()
        raise KlangError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
    return rc
