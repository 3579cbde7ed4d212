// rm_sweep32_hbm_s1.hip -- specialisation 1 of the fp32 sweep family "hbm" (see the .inc)
#define RM_SPEC 1
#include "rm_sweep32_hbm_body.inc"
