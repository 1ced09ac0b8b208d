/* include/klang_mi355.h — C-ABI of libklang_mi355.so, the MI355X (gfx950) replacement for klang's
 * per-block signal-graph evaluation path.
 *
 * The reference (nashaudio/klang, single header klang.h v0.7.8) has no FFI: a host calls C++
 * virtual methods on a user-derived Synth/Effect object once per audio block.  Each entry point
 * below names the reference interface it replaces (file:line into the reference's klang.h); the
 * binding a maintainer would add to klang.h is shown in INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 on success
 * or a negative klg_status, and klg_last_error() describes the last failure of the calling thread;
 * no exceptions cross this boundary; one calling thread per handle (the reference is single audio
 * thread per plugin instance, klang.h:4440-4466); sample buffers are caller-owned, non-interleaved
 * float32, processed in place (Synth ACCUMULATES into a pre-cleared buffer like klang.h:4751-4752,
 * Effect overwrites like klang.h:4708-4716).
 *
 * There is NO CPU fallback: if no gfx950 device is usable every create/process call fails with
 * KLG_ERR_NO_DEVICE.
 */
#ifndef KLANG_MI355_H
#define KLANG_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	KLG_OK = 0,
	KLG_ERR_NO_DEVICE = -1,   /* no HIP device / hipSetDevice failed */
	KLG_ERR_INVALID = -2,     /* bad argument (index out of range, n > max_block, NULL, ...) */
	KLG_ERR_HIP = -3,         /* a HIP runtime call failed; see klg_last_error() */
	KLG_ERR_NOMEM = -4
} klg_status;

/* Patch ids: the hand-written kernels shipped in this library (SURVEY.md §2 "Config patches"). */
typedef enum {
	KLG_PATCH_SINE = 0,       /* 1 x Generators::Fast::Sine note         (BASELINE config 1) */
	KLG_PATCH_BSINE = 1,      /* 1 x Generators::Basic::Sine note        (BASELINE config 1) */
	KLG_PATCH_SUB2A = 2,      /* Saw >> Biquad LPF (static) >> ADSR      (BASELINE config 2, north_star) */
	KLG_PATCH_SUB2B = 3,      /* shipped subtractive.k: Square >> swept LPF >> ADSR (config 2b) */
	KLG_PATCH_SUPERSAW = 4,   /* shipped SuperSaw.k: 7 x OSM saw / 7 * ADSR         (config 3) */
	KLG_PATCH_FM3 = 5,        /* shipped FM.k: 3 x Operator<Sine> * ADSR * 0.1 */
	KLG_PATCH_FM4 = 6,        /* 4-operator FM chain                               (config 5) */
	KLG_PATCH_PINGPONG = 7,   /* shipped PingPong.k (Stereo::Effect)               (config 4) */
	KLG_PATCH_REVERB = 8,     /* shipped Reverb.k   (Stereo::Effect)               (config 4) */
	KLG_PATCH_COUNT
} klg_patch;

/* NoteBase::Stage, klang.h:4286 */
enum { KLG_STAGE_ONSET = 0, KLG_STAGE_SUSTAIN = 1, KLG_STAGE_RELEASE = 2, KLG_STAGE_OFF = 3 };

const char* klg_last_error(void);
int klg_version(void);

/* Select the GPU(s) this process renders on.  n_devices == 1: every bank created afterwards lives on
 * device_ids[0] (the one-process-per-GPU arrangement; without this call device 0 is used).
 * n_devices in 2..64: every synth bank created afterwards SPANS the listed devices — its instances are
 * dealt to them in contiguous ranges, note and control calls are routed by instance, and
 * klg_process / klg_process_device render every share and combine the [channels][n] blocks with ONE
 * ncclAllReduce (RCCL, loaded on first use) before returning the global mix; a device listed more than
 * once gets as many shares, combined by a plain add.  Returns KLG_ERR_INVALID for n < 1 or n > 64 or an
 * id the runtime does not know.  (No reference equivalent: the reference is CPU;
 * this is what lets ONE Stereo::Synth::process(float**, int, float*) of klang.h:4830 cover a node.) */
int klg_init(const int* device_ids, int n_devices);

/* klang::random(seed) (klang.h:239): seeds the libc rand() stream the host-side on() code of
 * SuperSaw-style patches draws detune from (SuperSaw.k:17) — and that every Noise generator draws
 * from once per sample (klang.h:4949, 5363). */
void klg_random_seed(unsigned seed);

/* The reference has ONE rand() sequence per process, shared by host code (klang::random in on() /
 * prepare()) and by Generators::{Basic,Fast}::Noise::process() (klang.h:4947-4951, 5357-5366).  This
 * library produces the Noise draws ON THE DEVICE (glibc's generator restated with jump-ahead:
 * klang_amd/csrc/klg_rand.hpp): while banks with Noise generators are being processed the sequence
 * lives in device memory and no block waits for the host.  Host code that is about to call rand() /
 * klang::random(a, b) itself calls klg_rand_sync() first: the C library's generator is set to where
 * the device has got to (waits for the last enqueued Noise block; a no-op when no Noise bank has run
 * since the last call).  The library's own host-side draws and include/klang/klang.h's random() do. */
int klg_rand_sync(void);

/* `ranks` x `per` consecutive values of that sequence into device memory, d_out[i * rstride + r] =
 * the (r * per + i)-th rand() from where the sequence stands (rstride >= ranks) — what `ranks` Noise
 * objects drawing `per` values one after the other (klang.h:4842-4848) would get; the sequence moves
 * on by ranks * per.  Enqueued on `hip_stream`; what the banks' own Noise path uses. */
int klg_rand_fill_device(int* d_out, size_t rstride, unsigned ranks, int per, void* hip_stream);

/* ------------------------------------------------------------------------------------------------
 * Synth banks.  One klg_synth = `synths` independent instances of klang::Synth / Stereo::Synth
 * (klang.h:4375-4466, 4761-4860) of one patch, each with `notes_per_synth` <= 128 Note slots
 * (Array<NOTE*,128>, klang.h:4311), rendered by ONE kernel launch per block: voice v = instance
 * v / notes_per_synth, slot v % notes_per_synth, one GPU lane per voice.
 * ------------------------------------------------------------------------------------------------ */
typedef struct klg_synth klg_synth;
typedef struct klg_fx klg_fx;

/* replaces: constructing the user's Synth subclass + notes.add<T>(n) (klang.h:4324-4331); klang::fs (1604) */
klg_synth* klg_synth_create(int patch_id, int synths, int notes_per_synth, float sample_rate, int max_block);
void klg_synth_destroy(klg_synth* s);
int klg_synth_voices(const klg_synth* s);
int klg_synth_controls(const klg_synth* s);              /* controls.size() of the patch */
size_t klg_synth_state_bytes(const klg_synth* s);        /* resident HBM bytes per voice */
/* replaces: the choice of Note base class (klang.h:4721-4757).  1: a voice's `out` is one signal — klang::Note, or Stereo::Mono::Note whose
 * process(Stereo::buffer) adds it to BOTH channels (4747-4756); 2: Stereo::Note, `out` is a stereo signal and `buffer++ += out` (4731) adds
 * out.l to the left and out.r to the right channel (graph banks whose program ends in `ret2`). */
int klg_synth_note_channels(const klg_synth* s);

/* replaces: Synth::noteOn(int pitch, float velocity) klang.h:4423-4427 / 4813-4817 — Notes::assign()
 * voice allocation + NoteBase::start() + the patch's on() run on the HOST; the resulting lane state is
 * queued and applied on the GPU at the start of the next block.  Returns the slot (>= 0) or an error. */
int klg_note_on(klg_synth* s, int synth, int pitch, float velocity);
/* replaces: Synth::noteOff(int pitch, float velocity) klang.h:4430-4434 / 4820-4824 (NoteBase::release + off()) */
int klg_note_off(klg_synth* s, int synth, int pitch, float velocity);
/* Batched forms for banks of many instances: event i is (synth[i], pitch[i], velocity[i]); applied in order, exactly
 * as n successive klg_note_on / klg_note_off calls (one host->device transfer at the next block either way). */
int klg_note_on_many(klg_synth* s, int n, const int* synth, const int* pitch, const float* velocity);
int klg_note_off_many(klg_synth* s, int n, const int* synth, const int* pitch, const float* velocity);
/* replaces: controls[index].set(value) (clamped, klang.h:1725-1728) as done by the parameter sync of
 * Synth::process (klang.h:4444-4447, 4836-4839) followed by Synth::onControl (4399-4404). */
int klg_set_control(klg_synth* s, int synth, int index, float value);
int klg_get_control(klg_synth* s, int synth, int index, float* value);
/* replaces: Control::smoothed (klang.h:1707), the state `controls[index].smooth()` (1715) advances.  In a Synth the control belongs to
 * the Synth and EVERY sounding note's process() advances it — note by note, each through the whole block (4842-4848) — so a bank whose
 * recorded Note calls smooth() keeps the value here, per synth instance, and hands every sounding voice the value its block starts
 * from.  set: what a host's Control starts with (or a preset load); get: the value after the last block. */
int klg_set_control_smoothed(klg_synth* s, int synth, int index, float smoothed);
int klg_get_control_smoothed(klg_synth* s, int synth, int index, float* smoothed);

/* replaces: Stereo::Synth::process(float** buffers, int length, float* parameters) klang.h:4830-4858
 * (and mono Synth::process(float*, int, float*) 4440-4466 with channels == 1): pending events are applied,
 * every voice whose stage != Off renders n samples, voices are summed, the sum is ADDED to out[c][0..n).
 * `parameters`, if not NULL, holds klg_synth_controls() floats per instance [synths][controls] and is
 * copied in (clamped) before and out after the block.  Synchronous on return. */
int klg_process(klg_synth* s, float* const* out, int channels, int n, float* parameters);
/* Same block, additionally returning every voice's own n samples (per_voice[v*n + i], zeros for Off voices):
 * the quantity Note::process(buffer) writes (klang.h:4295-4303).  A bank of stereo notes (klg_synth_note_channels() == 2) returns
 * per_voice[(v*2 + c)*n + i]: what Stereo::Note::process(Stereo::buffer) adds to channel c (klang.h:4727-4733).  Parity/debug path. */
int klg_process_voices(klg_synth* s, float* per_voice, float* const* out, int channels, int n);
/* How the voices of one synth instance combine.  KLG_MIX_SUM (default): the block is the SUM of the sounding voices and is ADDED to the
 * caller's samples — Stereo::Synth::process (klang.h:4842-4848; Stereo::Note::process `buffer++ += out`, 4731).
 * KLG_MIX_LAST_ACTIVE: the mono klang::Synth::process(float*, int, float*) (klang.h:4450-4457) lets every sounding note OVERWRITE the
 * block in note order (Note::process: `buffer++ = out`, 4299), so the block is the output of the LAST sounding note of the instance
 * alone and REPLACES the caller's samples (they stay as they are while no note sounds).  Per-voice outputs are the same in both modes. */
enum { KLG_MIX_SUM = 0, KLG_MIX_LAST_ACTIVE = 1 };
int klg_synth_set_mix_mode(klg_synth* s, int mode);
/* replaces: reading note->stage after the block (klang.h:4455-4456: `if (!note->process(..)) note->stop()`). */
int klg_voice_stages(klg_synth* s, uint8_t* stages, int n_voices);

/* Throughput path: d_mix is a DEVICE pointer to [2][n] floats that the block is accumulated into on
 * `hip_stream` (a hipStream_t, or NULL for the handle's own stream); no host copies, no synchronisation.
 * (NULL is the handle's OWN stream, which is created non-blocking: work the caller has queued on the legacy default stream — a framework's clear of d_mix,
 * say — is not ordered with it.  A caller whose buffers are produced on the default stream passes a stream of its own, or synchronises.)
 * klg_sync() waits for everything queued on the handle.
 * A graph bank whose Note::process() draws Noise or calls controls[i].smooth() shares state between its notes in the order Synth::process walks
 * them (klang.h:4842-4848: libc rand(), Control::smoothed): that order is settled per block by kernels on the same stream (the sounding voices'
 * ranks, the rand() values from the sequence's state on the device, the smoothing chain: klang_amd/csrc/klg_rand_dev.hpp) — no copy to the host, no
 * wait.  (The host waits only when the buffer of draws must grow: a block with more sounding voices than any before it.) */
int klg_process_device(klg_synth* s, float* d_mix, int n, void* hip_stream);
int klg_sync(klg_synth* s);

/* ------------------------------------------------------------------------------------------------
 * Event scripts resident in HBM — offline / throughput rendering of an event stream that is known in advance (a MIDI file, a benchmark
 * script).  replaces: the host's per-block loop of noteOn / noteOff calls followed by process() (templates/juce/synth/Source/
 * PluginProcessor.cpp:170-177; klang.h:4423-4434) — here every on() runs on the host ONCE, up front, and the blocks then play with no host
 * work and no transfer between them.  Voices are addressed explicitly (the caller does the voice allocation a script implies).
 *   klg_note_record      the patch's on() for (pitch, velocity) on instance `synth` -> the lane record it would upload; nothing is queued
 *   klg_script_add_record / klg_script_note_on / klg_script_note_off   build the script: block b starts `voice` from pool record r /
 *                        releases `voice` (the patch's off(), applied to the voice's state at that block, only if it is at Sustain)
 *   klg_script_commit    sorts each block's events per voice (call order kept) and uploads records + index arrays to HBM
 *   klg_script_play_device  block `block` of the script: its events are applied by the event kernel from HBM, then the bank renders n
 *                        samples into d_mix ([2][n], accumulated) on hip_stream; asynchronous like klg_process_device
 * ------------------------------------------------------------------------------------------------ */
typedef struct klg_script klg_script;
int klg_note_record(klg_synth* s, int synth, int pitch, float velocity, void* record, size_t bytes);
klg_script* klg_script_create(klg_synth* s, int blocks);
void klg_script_destroy(klg_script* k);
int klg_script_add_record(klg_script* k, const void* record, size_t bytes);
int klg_script_note_on(klg_script* k, int block, int voice, int record_index);
int klg_script_note_off(klg_script* k, int block, int voice);
/* bulk forms: event i is (block[i], voice[i], record_index[i]); records are AoS, klg_synth_state_bytes() bytes each; klg_script_add_records
 * returns the pool index of the first record */
int klg_note_records(klg_synth* s, int n, const int* synth, const int* pitch, const float* velocity, void* records);
int klg_script_add_records(klg_script* k, int n, const void* records);
int klg_script_note_on_many(klg_script* k, int n, const int* block, const int* voice, const int* record_index);
int klg_script_note_off_many(klg_script* k, int n, const int* block, const int* voice);
int klg_script_commit(klg_script* k);
int klg_script_play_device(klg_script* k, int block, float* d_mix, int n, void* hip_stream);
/* replaces: the host's whole block loop for a stream known in advance (offline rendering: PluginProcessor.cpp:170-177 called `blocks` times with
 * cleared buffers): blocks first_block .. first_block + blocks - 1 of the script are rendered into d_out[b][2][n] (OVERWRITTEN: the library clears
 * the whole span once, then every block adds its mix) on hip_stream, asynchronously.  One call = one clear + one launch per block for banks of a
 * few workgroups (events and the voice sum inside the render launch), no host work between the blocks. */
int klg_script_render_device(klg_script* k, int first_block, int blocks, float* d_out, int n, void* hip_stream);
/* The span's launches are captured as a hipGraph the first time klg_script_render_device sees (span, n, d_out, stream) and replayed afterwards.
 * klg_script_capture_span does the capture WITHOUT running anything (returns 0 when the span is — now or already — captured; an error when it
 * cannot be: queued events or uploads pending, kernel timing on, KLG_GRAPH=0): the first real call is then a replay. */
int klg_script_capture_span(klg_script* k, int first_block, int blocks, float* d_out, int n, void* hip_stream);

/* Voice state transfer (checkpoint / debugging).  `state` is the patch's packed per-voice record of
 * klg_synth_state_bytes() bytes.  Replaces nothing in the reference (it has no checkpointing). */
int klg_voice_download(klg_synth* s, int voice, void* state, size_t bytes);
int klg_voice_upload(klg_synth* s, int voice, const void* state, size_t bytes);

/* Bulk form of klg_voice_upload: record i (AoS, klg_synth_state_bytes() bytes each) replaces the state of voices[i];
 * queued and applied by one kernel at the start of the next block (how a host starts many notes at once). */
int klg_voices_upload(klg_synth* s, int n, const int* voices, const void* states);

/* ------------------------------------------------------------------------------------------------
 * Graph patches (format and record layout: include/klang_mi355_graph.h).
 * replaces: constructing a user Synth whose Note::process() body (klang.h:4295-4303 calls it per sample) is not one of
 * the klg_patch ids above.  `program` is the body recorded by the DSL facade (include/klang/klang.h runs the user's
 * process() once in recording mode); it is compiled for gfx950 with hipRTC on top of the same device primitives and
 * render kernel as the shipped patches.  on()/off() stay with the caller: note events arrive as voice records
 * (klg_voice_download / klg_voice_upload / klg_voices_upload); klg_note_on / klg_note_off are rejected for such a bank.
 * klg_graph_check compiles a program WITHOUT a device (0, or KLG_ERR_INVALID with the message in `out`; with
 * want_source != 0 a successful check returns the generated HIP source instead; want_source == 2 asks for the
 * two-voices-per-lane form, which klg_synth_create_graph picks by itself when every node / op of the program has a packed
 * primitive and the kernel keeps both voices in registers — KLG_GRAPH_X1=1 in the environment forces one voice per lane).
 * ------------------------------------------------------------------------------------------------ */
klg_synth* klg_synth_create_graph(const char* program, int synths, int notes_per_synth, float sample_rate, int max_block);
/* Sample tables of a graph bank.  replaces: the `buffer` a Wavetable / Sample owns (klang.h:3626-3720:
 * `Wavetable(osc, 2048)` renders one cycle into it, `wavetable[i] = x` writes it, process() reads it with linear
 * interpolation) and a Table<float, SIZE> read with a fractional index (3365-3377).  Copies n floats to HBM and returns the
 * table's id (>= 1; < 0 = error): the value of a wavetable node's `table` record word, or the imm of a `tabread` op.  With
 * dedup != 0 a table with exactly the same samples as an earlier one returns that one's id (a bank of notes built from the
 * same oscillator shares ONE table).  Tables live as long as the bank. */
int klg_table_upload(klg_synth* s, const float* samples, int n, int dedup);
/* Note delays of a graph bank: a Delay<SIZE> member of the Note (klang.h:3381-3512; Karplus-Strong strings, waveguides) is a
 * `notedelay` node — its SIZE samples per voice live in HBM, zero-filled at creation like a fresh Delay, its cursors in the
 * voice record.  replaces: Delay::clear() (klang.h:3392-3394) of voice `voice`'s delay number `delay_index` (the program's
 * delay nodes in node order); queued in order with the blocks. */
int klg_voice_delay_clear(klg_synth* s, int voice, int delay_index);
int klg_graph_check(const char* program, int want_source, char* out, size_t out_cap);
/* How many voices one GPU lane renders in this bank: 2 for the packed Subtractive kernel and for graph patches that run packed,
 * else 1.  (Diagnostics / tests; no reference counterpart.) */
int klg_synth_voices_per_lane(const klg_synth* s);
/* The same for a recorded Effect::process() body (`kind effect 1|2` programs; replaces constructing `instances` copies of a
 * user klang::Effect / Stereo::Effect, klang.h:4190-4216, 4703-4717).  `initial_record`: the record words of one freshly
 * constructed instance (Program::words() x 4 bytes; NULL = zeros), every instance starts from it; Delay<SIZE> members become
 * zero-filled rings in HBM.  The handle is used with klg_fx_set_control / klg_fx_process[_device] / klg_fx_destroy; io is
 * [instances][channels][n]. */
klg_fx* klg_fx_create_graph(const char* program, int instances, float sample_rate, int max_block, const void* initial_record);
/* How a graph effect bank runs its recorded body (diagnostics / tests; no reference counterpart).  Returns 1 for the SAMPLE-PARALLEL form
 * (klang_amd/csrc/klg_graph_staged.hpp: a workgroup takes *instances_per_workgroup instances x *samples_per_chunk samples of the block at a time —
 * the parts of Effect::process() (klang.h:4208-4216 runs it sample after sample) that do not depend on the previous sample with a lane per
 * (sample, instance), the recurrences in sample order on a lane per instance, *levels barrier-separated steps per chunk, *lds_values values handed
 * between them through LDS), 0 for one lane per instance walking the samples in order — `why` then says what the body has that the staged form
 * does not handle —, < 0 on a bad handle.  Any output pointer may be NULL.  KLG_FX_STAGED=0 in the environment keeps every bank on the second form. */
int klg_fx_graph_form(const klg_fx* f, int* instances_per_workgroup, int* samples_per_chunk, int* levels, int* lds_values, char* why, size_t why_cap);

/* Measurement hooks used by bench.py: timing of the render kernel with a pair of HIP events per launch, on the
 * stream the kernel is launched on, ATTACHED TO THE DISPATCH (hipExtLaunchKernelGGL / hipExtModuleLaunchKernel:
 * the kernel's own start and end, what rocprofv3's kernel trace reports — events recorded around the launch add
 * 2 - 3 us of dispatch latency, which is nothing on a 0.38 ms launch and 15 % of a 17 us one).  klg_timing_begin()
 * arms it, klg_timing_end() returns the number of render launches since begin and their summed duration in
 * milliseconds.  klg_fx_timing_*: the same for an effect bank's kernel; a Reverb block is two kernels (the early
 * sums of the block, klg_fx_reverb_q) and is timed by events recorded around the pair. */
int klg_timing_begin(klg_synth* s);
int klg_timing_end(klg_synth* s, int* launches, float* total_ms);
/* ... and, while timing is armed, the same for a block's OTHER kernels — the event kernel and the voice-mix reduce (none for small banks, whose render launch
 * does both): launches and summed duration since klg_timing_begin.  Call it before klg_timing_end.  (What lets bench.py check that the kernels of a step fit the step.)
 * Banks with Noise generators: the figure covers the rank pass's first launch and a Note's smooth() pass, NOT the draw kernels (klg_rand_fill / klg_rand_advance)
 * nor the later launches of a multi-group rank pass — for those take the stream's own time around the block. */
int klg_timing_end_aux(klg_synth* s, int* launches, float* total_ms);
/* What a multi-device bank (klg_init with several ids; the voice sum of Stereo::Synth::process, klang.h:4830-4858, taken across GPUs) really runs through —
 * so that a scaling measurement proves itself instead of being believed (bench.py --in-library):
 *   *shards            how many shards the bank has (1: a one-device bank);
 *   *rccl_ranks        the ranks the RCCL communicator ITSELF reports (ncclCommCount of shard 0's communicator), 0 if there is none — one device, or shards that share a GPU
 *                      and are summed by a device-side add;
 *   *distinct_devices  how many different HIP devices the shards sit on;
 *   per_shard_kernel_ms[0 .. min(shards, cap))   every shard's render-kernel milliseconds since klg_timing_begin (call before klg_timing_end; may be NULL);
 *   *allreduce_us      measured here: the mean duration of ONE all-reduce of a [2][n] block over the shards (`probe_reps` of them, each between a pair of events on shard 0's
 *                      stream, outside any block); 0 without a communicator or with probe_reps <= 0. */
int klg_synth_multi_info(klg_synth* s, int n, int probe_reps, int* shards, int* rccl_ranks, int* distinct_devices, float* per_shard_kernel_ms, int cap, float* allreduce_us);

/* ------------------------------------------------------------------------------------------------
 * Effect banks: `instances` independent Stereo::Effect objects (klang.h:4703-4717) of one patch.
 * ------------------------------------------------------------------------------------------------ */
klg_fx* klg_fx_create(int patch_id, int instances, float sample_rate, int max_block);
/* The same on a NAMED device, whatever klg_init() selected and without changing it: one bank on one GPU — a rank's share of a sharded effect bank
 * (klang_amd/shard.py), several banks of one process on different GPUs.  `program` != NULL: a recorded effect (klg_fx_create_graph's program and
 * initial_record), else patch_id as for klg_fx_create.  (No reference counterpart: the reference constructs its effects on the CPU.) */
klg_fx* klg_fx_create_on(int device, int patch_id, const char* program, int instances, float sample_rate, int max_block, const void* initial_record);
void klg_fx_destroy(klg_fx* f);
int klg_fx_set_control(klg_fx* f, int instance, int index, float value);
/* replaces: `parameters[c] = controls[c].value` after the block (Effect::process(float*, int, float*) klang.h:4213-4215): the value as the
 * effect left it — a control its process() writes (examples/PingPong.k:48,60) comes back from the instance's state (as of the blocks
 * that have completed on the bank's own stream: after klg_fx_process, or klg_fx_sync following klg_fx_process_device). */
int klg_fx_get_control(klg_fx* f, int instance, int index, float* value);
/* An instance's record (graph effects: klg_fx_create_graph).  An effect whose prepare() is host code in the reference — Controls::changed(),
 * libc rand() tables, loops over a count: examples/Reverb.k:238-241 with 23-56, 127-131 — keeps it host code: the facade (include/klang/klang.h,
 * gpu::EffectBank) runs prepare() on its own mirror of the instance, starting from the record as the device last left it, and uploads the words
 * prepare() changed; they are applied before the next block.  klg_fx_record_words: 32-bit words per record. */
int klg_fx_record_words(const klg_fx* f);
int klg_fx_download_record(klg_fx* f, int instance, void* words, size_t bytes);
int klg_fx_upload_words(klg_fx* f, int instance, int first, int count, const void* values);
/* replaces: Stereo::Effect::process(Stereo::buffer) klang.h:4708-4716 for every instance:
 * io[(k*2 + c)*n + i] is channel c of instance k, processed in place.  Host buffers, synchronous. */
int klg_fx_process(klg_fx* f, float* io, int n);
/* Device-resident variant (d_io device pointer, asynchronous on hip_stream). */
int klg_fx_process_device(klg_fx* f, float* d_io, int n, void* hip_stream);
/* replaces: the host's block loop around an effect for a stream known in advance — Stereo::Effect::process(Stereo::buffer) (klang.h:4708-4716) called
 * `blocks` times by templates/juce/effect/Source/PluginProcessor.cpp:153-178 (offline rendering, a benchmark).  d_io is a DEVICE buffer
 * [blocks][instances][channels][n], processed in place, asynchronously on hip_stream; dials keep the values they have at the call (a host that moves a
 * dial ends the span there).  Same bits as `blocks` calls of klg_fx_process_device.  One launch per span where the kernel can walk the blocks itself
 * (PingPong.k's pipeline across the block boundaries — blocks of a multiple of 32 samples —, the staged form of a recorded effect with prepare() at the head
 * of every block); otherwise the blocks' launches back to back without host work in between (Reverb.k: its kernel stages a whole block in LDS). */
int klg_fx_render_device(klg_fx* f, float* d_io, int blocks, int n, void* hip_stream);
int klg_fx_sync(klg_fx* f);
size_t klg_fx_state_bytes(const klg_fx* f);
int klg_fx_timing_begin(klg_fx* f);
int klg_fx_timing_end(klg_fx* f, int* launches, float* total_ms);

/* ------------------------------------------------------------------------------------------------
 * Diagnostics (no reference equivalent): run ONE device primitive (oscillator, filter, envelope, delay tap, ...)
 * for n samples in a single GPU lane, so each row of the hot-path table can be checked against the reference's
 * known-answer vectors on its own.  Primitive ids and parameter packs: klang_amd/csrc/klg_selftest.hpp.
 * klg_selftest_host exposes the host halves (set(): increments, coefficient design, breakpoint state).
 * ------------------------------------------------------------------------------------------------ */
int klg_selftest(int primitive, const float* params, int n_params, const float* in, int n_in, float* out, int n_out, int n);
int klg_selftest_host(int kind, const float* args, int n_args, float sample_rate, float* out, int n_out);

#ifdef __cplusplus
}
#endif
#endif /* KLANG_MI355_H */
