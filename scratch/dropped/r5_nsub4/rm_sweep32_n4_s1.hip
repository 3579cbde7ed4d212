// rm_sweep32_n3_s1.hip -- specialisation 1 of the fp32 sweep family "n4" (see the .inc)
#define RM_SPEC 1
#include "rm_sweep32_n4_body.inc"
