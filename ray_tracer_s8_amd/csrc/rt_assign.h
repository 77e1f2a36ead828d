// rt_assign.h — which device entry of a frame context renders which strip (host code, no HIP: shared with the CPU harness
// tests/host/assign_host.cpp).
//
// The reference fires all `divisions` strip requests at once and lets Docker's DNS spread them over whatever slaves exist
// (controller main.rs:47-75); a frame is done when its slowest slave is.  Strips are not equally expensive — the rows near the top of a
// frame are mostly sky, one segment per sample, the rows at the bottom bounce — so the assignment decides how long the slowest
// entry works:
//   * no cost known yet (the first frame of a job): SNAKE — strip k of "row" k / n goes to entry k % n in even rows and to
//     n - 1 - k % n in odd ones.  Every entry gets one strip of each row, and the sum over an entry's strips is the same for every
//     entry whenever the cost is a linear function of the strip's position (and nearly so for any smooth profile): plain k % n leaves
//     the last entry 7 % above the mean on c4 at 8 entries, where the bottom strips cost about twice the top ones.
//   * costs known (ray segments per strip of the job's previous frame, counted by the kernels): LONGEST PROCESSING TIME FIRST —
//     strips in order of decreasing cost, each to the entry with the least load so far (ties: the lower entry, the lower strip).
//     Graham's bound is 4/3 - 1/(3n) of the optimum; with a few strips per entry of comparable cost it lands within a per cent or two.
// Either way an entry renders its strips in ONE batched launch, in increasing strip order.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace rtassign {

enum Mode : uint32_t { STATIC_MOD = 0, SNAKE = 1, BY_COST = 2, QUEUE = 3 };

// owner[k] = entry of strip k.  cost: nullptr, or one non-negative value per strip.
inline void assign(uint32_t divisions, uint32_t n_entries, const double* cost, Mode mode, std::vector<uint32_t>& owner) {
    owner.assign(divisions, 0u);
    if (n_entries <= 1u) return;
    if (mode == STATIC_MOD || (mode == BY_COST && !cost)) {
        for (uint32_t k = 0; k < divisions; k++) owner[k] = k % n_entries;
        return;
    }
    if (mode == SNAKE) {
        for (uint32_t k = 0; k < divisions; k++) {
            const uint32_t row = k / n_entries, col = k % n_entries;
            owner[k] = (row & 1u) ? n_entries - 1u - col : col;
        }
        return;
    }
    std::vector<uint32_t> order(divisions);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    std::vector<double> load(n_entries, 0.0);
    for (uint32_t k : order) {
        uint32_t best = 0;
        for (uint32_t e = 1; e < n_entries; e++)
            if (load[e] < load[best]) best = e;
        owner[k] = best;
        load[best] += cost[k];
    }
}

// largest entry load / mean entry load under `owner` (1 = perfectly even; 0 when nothing costs anything)
inline double max_over_mean(uint32_t divisions, uint32_t n_entries, const double* cost, const std::vector<uint32_t>& owner) {
    std::vector<double> load(n_entries, 0.0);
    double tot = 0.0;
    for (uint32_t k = 0; k < divisions; k++) {
        load[owner[k]] += cost[k];
        tot += cost[k];
    }
    if (!(tot > 0.0)) return 0.0;
    return *std::max_element(load.begin(), load.end()) * n_entries / tot;
}

}  // namespace rtassign
