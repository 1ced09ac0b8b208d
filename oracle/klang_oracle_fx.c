/* oracle/klang_oracle_fx.c — TEST INFRASTRUCTURE ONLY.  Effect patches (PingPong.k, Reverb.k). */
#include "klang_oracle.h"
#include <stdlib.h>

struct ko_fxbank { int patch, K; };
ko_fxbank* ko_fxbank_create(int patch, int instances, float fs) { (void)patch; (void)instances; (void)fs; return NULL; }
void ko_fxbank_destroy(ko_fxbank* b) { free(b); }
void ko_fxbank_control(ko_fxbank* b, int instance, int index, float value) { (void)b; (void)instance; (void)index; (void)value; }
void ko_fxbank_process(ko_fxbank* b, float* io, int n) { (void)b; (void)io; (void)n; }
