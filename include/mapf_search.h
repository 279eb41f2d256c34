/*
 * mapf_search.h -- C ABI of the host-side expert planner used to label scenario fixtures.
 *
 * Replaces the reference's search.find_path = CBSSolver + space-time A* (reference search.py:58-442), which
 * test.create_test calls to record `opt_steps` (reference test.py:50-58).  Host C++, no GPU work: this is
 * fixture tooling next to the hot path, not part of it.
 *
 * Low level: space-time A* with negative vertex / edge constraints, the reference's goal test ("at the goal
 * and no constraint of this agent lies in the future", search.py:173-177) and its 256-step horizon
 * (config.max_steps, search.py:179).  High level: conflict-based search, best-first on (sum of costs, number
 * of colliding pairs, generation order) like search.py:303-306, standard two-way negative splitting on the
 * first conflict (the reference splits disjointly on a RANDOM conflict and gives up after 5 s wall clock, so
 * its output is not reproducible; parity is "valid, collision-free, sum of costs <= the reference's").
 */
#ifndef MAPF_SEARCH_H
#define MAPF_SEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAPF_ERR_TIMEOUT (-8) /* no solution within the time / node budget (the reference returns None) */

/*
 * map int8 [L][L]; agents/goals int16 [N][2] (row, col).  On success writes the joint plan as per-step action
 * rows actions_out[t][i] (0 stay 1 up 2 down 3 left 4 right, reference search.py:416-433) for t < *num_steps
 * (= makespan = the reference's len(actions) = `opt_steps`) and the sum of individual path costs.
 * actions_out must hold max_steps * N entries.
 */
int mapf_find_path(int map_len, int num_agents, const int8_t *map, const int16_t *agents, const int16_t *goals,
                   double time_limit_s, int max_steps, int8_t *actions_out, int *num_steps, int *sum_of_costs);

/* search.compute_heuristics (search.py:24-55): BFS distance from `goal` over free cells; -1 = unreachable. */
int mapf_distance_field(int map_len, const int8_t *map, int goal_row, int goal_col, int32_t *dist_out);

#ifdef __cplusplus
}
#endif
#endif /* MAPF_SEARCH_H */
