// rm_sweep32_hbm.hip -- specialisation 0 of the fp32 sweep family "hbm" (see the .inc)
#define RM_SPEC 0
#include "rm_sweep32_hbm_body.inc"
