// rm_sweep32_lds_s2.hip -- specialisation 2 of the fp32 sweep family "lds" (see the .inc)
#define RM_SPEC 2
#include "rm_sweep32_lds_body.inc"
