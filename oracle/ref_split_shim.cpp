/* ref_split_shim.cpp -- TEST INFRASTRUCTURE ONLY.  C-callable doorway to the REAL reference's split functions
 * (split_data_*_double/float, declared in /root/reference/src/recometrics_signatures.hpp:100-220 and defined in
 * recometrics_instantiated.cpp), which take std::vector& outputs and therefore cannot be reached through ctypes.
 * Compiled by oracle/Makefile together with the reference's own source file, where it lies, into oracle/_ref/.
 * Used only by tests/golden/make_golden_split.py to capture fixtures. */
#include <cstring>
#include <stdexcept>
#include <vector>
#include "recometrics_signatures.hpp"

namespace {
std::vector<int32_t> g_p[3], g_i[3], g_users;      /* 0 train, 1 test, 2 rem */
std::vector<double> g_v[3];
}

/* mode 0 = selected (all rows), 1 = separated, 2 = joined; returns 0 or 1 (the reference threw) */
extern "C" int ref_split_f64(const int32_t *p, const int32_t *i, const double *v, int32_t m, int32_t n, int mode,
                             int32_t n_users_test, double frac, int cold, int32_t min_items_pool, int32_t min_pos_test,
                             uint64_t seed, long long *sizes /* [10] */)
{
    for (int q = 0; q < 3; q++) { g_p[q].clear(); g_i[q].clear(); g_v[q].clear(); }
    g_users.clear();
    try {
        if (mode == 0)
            split_data_selected_users_double(p, i, v, m, n, g_p[0], g_i[0], g_v[0], g_p[1], g_i[1], g_v[1], frac, seed);
        else if (mode == 1)
            split_data_separate_users_double(p, i, v, m, n, g_users, g_p[2], g_i[2], g_v[2], g_p[0], g_i[0], g_v[0],
                                             g_p[1], g_i[1], g_v[1], n_users_test, frac, cold != 0, min_items_pool, min_pos_test, seed);
        else
            split_data_joined_users_double(p, i, v, m, n, g_users, g_p[0], g_i[0], g_v[0], g_p[1], g_i[1], g_v[1],
                                           n_users_test, frac, cold != 0, min_items_pool, min_pos_test, seed);
    } catch (const std::exception &) { return 1; }
    for (int q = 0; q < 3; q++) { sizes[3 * q] = (long long)g_p[q].size(); sizes[3 * q + 1] = (long long)g_i[q].size(); sizes[3 * q + 2] = (long long)g_v[q].size(); }
    sizes[9] = (long long)g_users.size();
    return 0;
}

extern "C" void ref_split_copy(int which, void *dst)
{
    if (which == 9) { if (!g_users.empty()) std::memcpy(dst, g_users.data(), g_users.size() * 4); return; }
    const int q = which / 3;
    if (which % 3 == 0) { if (!g_p[q].empty()) std::memcpy(dst, g_p[q].data(), g_p[q].size() * 4); }
    else if (which % 3 == 1) { if (!g_i[q].empty()) std::memcpy(dst, g_i[q].data(), g_i[q].size() * 4); }
    else { if (!g_v[q].empty()) std::memcpy(dst, g_v[q].data(), g_v[q].size() * 8); }
}
