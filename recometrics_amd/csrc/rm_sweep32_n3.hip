// rm_sweep32_n3.hip -- specialisation 0 of the fp32 sweep family "n3" (see the .inc)
#define RM_SPEC 0
#include "rm_sweep32_n3_body.inc"
