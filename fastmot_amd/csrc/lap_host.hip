// Host-side rectangular linear sum assignment for SMALL problems (nr <= nc, nr * nc <= a few
// thousand): the same shortest-augmenting-path algorithm, scan order and tie-breaking as lap_kernel /
// lap64_kernel in assoc.hip, i.e. as scipy.optimize.linear_sum_assignment (the reference's solver,
// utils/matching.py:27; Crouse, "On implementing 2D rectangular assignment algorithms", 2016).
//
// Why a host solver inside a GPU library: the algorithm is ~nr dependent Dijkstra steps of a few
// hundred scalar operations each.  One 2.4 GHz in-order wavefront needs ~1 us per step (50 us for a
// 50 x 50 frame, measured), a host core ~0.1 us, and the assignment has to reach the host anyway (it
// drives the Python-side track bookkeeping).  The stage-cost kernel therefore writes the small cost
// matrix straight into pinned host memory and this function consumes it; larger problems stay on the
// device kernels.  Both paths are tested against SciPy for identical (rows, cols).
#include "common.h"

#include <cmath>
#include <limits>
#include <vector>

// cost element (i, j) = cost[i * rs + j * cs]; requires nr <= nc.  Returns false if infeasible.
bool fm_lap_host(const double* cost, int nr, int nc, long rs, long cs, int32_t* col4row_out) {
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> u(nr, 0.), v(nc, 0.), spc(nc);
    std::vector<int32_t> path(nc, -1), col4row(nr, -1), row4col(nc, -1), remaining(nc);
    std::vector<uint8_t> SR(nr), SC(nc);
    for (int cur = 0; cur < nr; ++cur) {
        std::fill(SR.begin(), SR.end(), 0);
        std::fill(SC.begin(), SC.end(), 0);
        std::fill(spc.begin(), spc.end(), INF);
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        int num_remaining = nc, sink = -1, i = cur;
        double minVal = 0.;
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            SR[i] = 1;
            const double ui = u[i];
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = ((minVal + cost[(long)i * rs + (long)j * cs]) - ui) - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                // strictly lower, or equally low and unassigned (a later one overrides an earlier one)
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            minVal = lowest;
            if (index < 0 || minVal == INF) return false;
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += minVal;
        for (int r = 0; r < nr; ++r)
            if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; ++j)
            if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        while (true) {
            const int r = path[j];
            row4col[j] = r;
            const int tmp = col4row[r];
            col4row[r] = j;
            j = tmp;
            if (r == cur) break;
        }
    }
    for (int r = 0; r < nr; ++r) col4row_out[r] = col4row[r];
    return true;
}
