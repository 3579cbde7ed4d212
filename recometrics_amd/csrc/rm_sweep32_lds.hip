// rm_sweep32_lds.hip -- specialisation 0 of the fp32 sweep family "lds" (see the .inc)
#define RM_SPEC 0
#include "rm_sweep32_lds_body.inc"
