// rm_sweep64_large.hip -- specialisation 0 of the fp64 sweep, large factor counts (see the .inc)
#define RM_SPEC 0
#include "rm_sweep64_large_body.inc"
