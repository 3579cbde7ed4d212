// rm_sweep32_lds_s1.hip -- specialisation 1 of the fp32 sweep family "lds" (see the .inc)
#define RM_SPEC 1
#include "rm_sweep32_lds_body.inc"
