// CPU harness around the PRODUCT's strip-assignment function (ray_tracer_s8_amd/csrc/rt_assign.h), for tests/test_assign.py.
#include <cstdint>
#include <vector>

#include "rt_assign.h"

extern "C" {
// mode: 0 static k % n, 1 snake, 2 longest-first by cost.  owner_out: divisions entries.  Returns max / mean entry load under
// `cost` (which may be NULL for modes 0 and 1: then 0).
double assign_strips(uint32_t divisions, uint32_t n_entries, const double* cost, uint32_t mode, uint32_t* owner_out) {
    std::vector<uint32_t> owner;
    rtassign::assign(divisions, n_entries, cost, (rtassign::Mode)mode, owner);
    for (uint32_t k = 0; k < divisions; k++) owner_out[k] = owner[k];
    return cost ? rtassign::max_over_mean(divisions, n_entries, cost, owner) : 0.0;
}
}
