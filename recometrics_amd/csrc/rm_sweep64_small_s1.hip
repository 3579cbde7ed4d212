// rm_sweep64_small_s1.hip -- specialisation 1 of the fp64 sweep, small factor counts (see the .inc)
#define RM_SPEC 1
#include "rm_sweep64_small_body.inc"
