// rm_sweep64_large_s1.hip -- specialisation 1 of the fp64 sweep, large factor counts (see the .inc)
#define RM_SPEC 1
#include "rm_sweep64_large_body.inc"
