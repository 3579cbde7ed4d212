// rm_sweep32_n3.hip -- specialisation 0 of the fp32 sweep family "n4" (see the .inc)
#define RM_SPEC 0
#include "rm_sweep32_n4_body.inc"
