// rm_sweep32_n3.hip -- specialisation 0 of the fp32 sweep family "n3" (see the .inc)
#define RM_SPEC 0
#include "rm_sweep32_n3_body.inc"

#ifdef RM_STATS
extern "C" int rm_debug_stats_n3(unsigned long long *out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(rm::g_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(rm::g_stats), z, sizeof(z)); }
    return 0;
}
#endif
