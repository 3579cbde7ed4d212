// rm_sweep32_hbm_s2.hip -- specialisation 2 of the fp32 sweep family "hbm" (see the .inc)
#define RM_SPEC 2
#include "rm_sweep32_hbm_body.inc"
