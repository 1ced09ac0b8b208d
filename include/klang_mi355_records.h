/* include/klang_mi355_records.h — the packed per-voice records of klg_voice_download / klg_voice_upload
 * (include/klang_mi355.h): what one GPU lane keeps in HBM between blocks for each shipped patch.
 *
 * Shared by the kernels (klang_amd/csrc/klg_patches.hpp), the library's host side and the source-compatible DSL
 * header (include/klang/klang.h), so that a host-side Note object can be packed into / unpacked from a lane.
 * Plain C++ PODs: no HIP, no torch.  Word 0 of every record is `flags`:
 *   bits 0-1   NoteBase::stage (klang.h:4286)            KLG_STAGE_*
 *   then per patch 6-bit envelope fields { stage:2 | point:3 | active:1 } and 2-bit OSM states, see below.
 */
#ifndef KLANG_MI355_RECORDS_H
#define KLANG_MI355_RECORDS_H
#ifndef __HIPCC_RTC__
#include <stdint.h>
#else            /* hiprtc (generated patches): no libc headers, the fixed-width types are spelled out */
typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;
typedef int int32_t; typedef unsigned int uint32_t; typedef long long int64_t; typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#endif

namespace klg {

enum { ST_ONSET = 0, ST_SUSTAIN = 1, ST_RELEASE = 2, ST_OFF = 3 };   /* NoteBase::Stage */
enum { ENV_SUSTAIN = 0, ENV_RELEASE = 1, ENV_OFF = 2 };              /* Envelope::Stage klang.h:3809 */
enum { KLG_MAX_CTL = 32 };                                            /* controls per synth / effect instance (Vocoder.k: 5 dials + 22 meters) */

struct OsmRec { int32_t inc; uint32_t offset, duty; float delta; };                 /* Fast::OSM klang.h:5196-5200 */
struct AdsrRec { float r_out, r_target, r_rate, time, A, AD, S, R; };               /* ADSR: points (0,0) (A,1) (A+D,S); R for release() */
struct BiquadRec { float b0, b1, b2, a1, a2, z0, z1; };                             /* Biquad::Filter klang.h:5556-5563 */
struct Env3Rec { float r_out, r_target, r_rate, time, px[3], py[3]; };              /* Envelope with <= 3 breakpoints */
struct SweepRec { float f, Q; BiquadRec c; };                                       /* Biquad whose set(f, Q) runs per sample */
struct OpRec { int32_t inc; uint32_t pos; float r_out, r_target, r_rate, time, px[2], py[2]; };   /* Operator<Fast::Sine> */

namespace rec {
struct Sine { uint32_t flags; int32_t inc; uint32_t pos; };
struct BSine { uint32_t flags; float increment, position, offset; };
/* flags: [0:2) note | [2:8) adsr | [8:10) osm state */
struct Sub2a { uint32_t flags; OsmRec osc; BiquadRec lpf; AdsrRec adsr; };
/* flags: [0:2) note | [2:8) adsr | [8:14) env | [14:16) osm state */
struct Sub2b { uint32_t flags; OsmRec osc; AdsrRec adsr; Env3Rec env; SweepRec filter; };
/* flags: [0:2) note | [2:8) adsr | [8:22) 7 x osm state */
struct SuperSaw { uint32_t flags; OsmRec osc[7]; AdsrRec adsr; };
/* flags: [0:2) note | [2:8) adsr | [8+6k:14+6k) operator k envelope ; meta: 2 bits npoints per operator */
template<int NOPS> struct FM { uint32_t flags, meta; OpRec op[NOPS]; AdsrRec adsr; };
}

/* envelope state <-> 6 flag bits */
inline uint32_t env_bits(int stage, int point, bool active) { return (uint32_t)stage | ((uint32_t)point << 2) | ((uint32_t)active << 5); }

}  /* namespace klg */
#endif
