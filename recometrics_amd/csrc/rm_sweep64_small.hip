// rm_sweep64_small.hip -- specialisation 0 of the fp64 sweep, small factor counts (see the .inc)
#define RM_SPEC 0
#include "rm_sweep64_small_body.inc"
