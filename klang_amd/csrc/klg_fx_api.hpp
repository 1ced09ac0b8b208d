// klang_amd/csrc/klg_fx_api.hpp — C-ABI entry points of the effect banks (included by klg_api.hip).
#pragma once
extern "C" klg_fx* klg_fx_create(int patch_id, int, float, int) { fail(KLG_ERR_INVALID, "klg_fx_create: effect patch %d is not built into this library yet", patch_id); return nullptr; }
extern "C" void klg_fx_destroy(klg_fx*) {}
extern "C" int klg_fx_set_control(klg_fx*, int, int, float) { return fail(KLG_ERR_INVALID, "effects not built"); }
extern "C" int klg_fx_process(klg_fx*, float*, int) { return fail(KLG_ERR_INVALID, "effects not built"); }
extern "C" int klg_fx_process_device(klg_fx*, float*, int, void*) { return fail(KLG_ERR_INVALID, "effects not built"); }
extern "C" int klg_fx_sync(klg_fx*) { return fail(KLG_ERR_INVALID, "effects not built"); }
extern "C" size_t klg_fx_state_bytes(const klg_fx*) { return 0; }
extern "C" int klg_fx_timing_begin(klg_fx*) { return fail(KLG_ERR_INVALID, "effects not built"); }
extern "C" int klg_fx_timing_end(klg_fx*, int*, float*) { return fail(KLG_ERR_INVALID, "effects not built"); }
