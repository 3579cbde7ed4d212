// rm_sweep32_n3_s2.hip -- specialisation 2 of the fp32 sweep family "n4" (see the .inc)
#define RM_SPEC 2
#include "rm_sweep32_n4_body.inc"
