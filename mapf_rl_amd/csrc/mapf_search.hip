// mapf_search.hip -- host-only C++ (no kernels): conflict-based search + space-time A*, see include/mapf_search.h.
// Built from scratch; follows the BEHAVIOUR of reference search.py:58-442 (constraint semantics, goal test,
// horizon, collision definition, plan -> action conversion), not its code.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <queue>
#include <unordered_set>
#include <vector>

#include "mapf_env.h"
#include "mapf_search.h"

namespace {

struct Constraint {
    int agent, t;
    int from, to;  // vertex constraint: from == -1, cell `to` forbidden at time t; edge: move from->to arriving at t
};

struct Grid {
    int L;
    const int8_t *map;
    bool free_cell(int c) const { return map[c] == 0; }
};

const int kDX[5] = {0, 1, 0, -1, 0};  // reference search.move: directions (0,-1),(1,0),(0,1),(-1,0) + stay
const int kDY[5] = {-1, 0, 1, 0, 0};

void bfs(const Grid &g, int goal, std::vector<int32_t> &dist) {
    dist.assign((size_t)g.L * g.L, -1);
    std::vector<int> q;
    q.push_back(goal);
    dist[goal] = 0;  // search.py:33-34 (the goal is seeded even if it is an obstacle)
    for (size_t h = 0; h < q.size(); ++h) {
        int c = q[h], x = c / g.L, y = c % g.L;
        for (int d = 0; d < 4; ++d) {
            int nx = x + kDX[d], ny = y + kDY[d];
            if (nx < 0 || ny < 0 || nx >= g.L || ny >= g.L) continue;
            int n = nx * g.L + ny;
            if (!g.free_cell(n) || dist[n] >= 0) continue;
            dist[n] = dist[c] + 1;
            q.push_back(n);
        }
    }
}

inline uint64_t vkey(int cell, int t) { return ((uint64_t)t << 32) | (uint32_t)cell; }
inline uint64_t ekey(int from, int to, int t) { return ((uint64_t)t << 48) | ((uint64_t)(uint32_t)from << 24) | (uint32_t)to; }

// space-time A* (search.py:145-216). Returns the path as cells (index = timestep) or empty on failure.
std::vector<int> low_level(const Grid &g, int start, int goal, const std::vector<int32_t> &h, int agent,
                           const std::vector<Constraint> &cons, int max_steps) {
    std::unordered_set<uint64_t> vcon, econ;
    int max_t = 0;
    for (const auto &c : cons)
        if (c.agent == agent) {
            if (c.from < 0) vcon.insert(vkey(c.to, c.t));
            else econ.insert(ekey(c.from, c.to, c.t));
            max_t = std::max(max_t, c.t);  // search.py:156-159
        }
    if (h[start] < 0) return {};
    struct Node {
        int cell, t, parent;
    };
    std::vector<Node> nodes;
    using QE = std::tuple<int, int, int, int>;  // (f, h, cell, node id) like search.py:129
    std::priority_queue<QE, std::vector<QE>, std::greater<QE>> open;
    std::unordered_set<uint64_t> seen;
    nodes.push_back({start, 0, -1});
    open.emplace(h[start], h[start], start, 0);
    seen.insert(vkey(start, 0));
    while (!open.empty()) {
        auto [f, hv, cell, id] = open.top();
        open.pop();
        (void)f;
        (void)hv;
        const Node cur = nodes[id];
        if (cur.cell == goal && cur.t >= max_t) {  // search.py:175
            std::vector<int> path;
            for (int n = id; n >= 0; n = nodes[n].parent) path.push_back(nodes[n].cell);
            std::reverse(path.begin(), path.end());
            return path;
        }
        if (cur.t >= max_steps) continue;  // search.py:179
        const int x = cur.cell / g.L, y = cur.cell % g.L;
        for (int d = 0; d < 5; ++d) {
            const int nx = x + kDX[d], ny = y + kDY[d];
            if (nx < 0 || ny < 0 || nx >= g.L || ny >= g.L) continue;
            const int n = nx * g.L + ny;
            if (!g.free_cell(n) || h[n] < 0) continue;
            const int nt = cur.t + 1;
            if (vcon.count(vkey(n, nt)) || econ.count(ekey(cur.cell, n, nt))) continue;
            if (!seen.insert(vkey(n, nt)).second) continue;  // g == t for every state: first visit is optimal
            nodes.push_back({n, nt, id});
            open.emplace(nt + h[n], h[n], n, (int)nodes.size() - 1);
        }
    }
    return {};
}

struct Collision {
    int a1, a2, t;
    int c1, c2;  // vertex: c2 == -1 (cell c1 at time t); edge: a1 moves c1->c2, a2 moves c2->c1, arriving at t
};

inline int at(const std::vector<int> &p, int t) { return t < (int)p.size() ? p[t] : p.back(); }  // search.py:80-86

bool first_collision(const std::vector<int> &p1, const std::vector<int> &p2, Collision &out) {  // search.py:219-239
    const int len = (int)std::max(p1.size(), p2.size());
    for (int i = 0; i < len; ++i) {
        const int l1 = at(p1, i), l2 = at(p2, i);
        if (l1 == l2) {
            out.t = i, out.c1 = l1, out.c2 = -1;
            return true;
        }
        const int n1 = at(p1, i + 1), n2 = at(p2, i + 1);
        if (l1 == n2 && l2 == n1) {
            out.t = i + 1, out.c1 = l1, out.c2 = l2;
            return true;
        }
    }
    return false;
}

struct HLNode {
    int cost = 0;
    std::vector<Constraint> cons;
    std::vector<std::vector<int>> paths;
    std::vector<Collision> cols;
};

void detect(HLNode &n) {
    n.cols.clear();
    for (size_t i = 0; i < n.paths.size(); ++i)
        for (size_t j = i + 1; j < n.paths.size(); ++j) {
            Collision c;
            if (first_collision(n.paths[i], n.paths[j], c)) {
                c.a1 = (int)i, c.a2 = (int)j;
                n.cols.push_back(c);
            }
        }
    n.cost = 0;
    for (auto &p : n.paths) n.cost += (int)p.size() - 1;  // search.py:17-21
}

}  // namespace

extern "C" {

int mapf_distance_field(int map_len, const int8_t *map, int goal_row, int goal_col, int32_t *dist_out) {
    if (map_len < 1 || !map || !dist_out || goal_row < 0 || goal_col < 0 || goal_row >= map_len || goal_col >= map_len)
        return MAPF_ERR_INVALID_ARG;
    Grid g{map_len, map};
    std::vector<int32_t> d;
    bfs(g, goal_row * map_len + goal_col, d);
    std::memcpy(dist_out, d.data(), d.size() * sizeof(int32_t));
    return MAPF_OK;
}

int mapf_find_path(int map_len, int num_agents, const int8_t *map, const int16_t *agents, const int16_t *goals,
                   double time_limit_s, int max_steps, int8_t *actions_out, int *num_steps, int *sum_of_costs) {
    if (map_len < 1 || num_agents < 1 || !map || !agents || !goals || !actions_out || !num_steps || max_steps < 1)
        return MAPF_ERR_INVALID_ARG;
    const int L = map_len, N = num_agents;
    Grid g{L, map};
    std::vector<int> start(N), goal(N);
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < 2; ++k)
            if (agents[2 * i + k] < 0 || agents[2 * i + k] >= L || goals[2 * i + k] < 0 || goals[2 * i + k] >= L)
                return MAPF_ERR_INVALID_ARG;
        start[i] = agents[2 * i] * L + agents[2 * i + 1];
        goal[i] = goals[2 * i] * L + goals[2 * i + 1];
    }
    std::vector<std::vector<int32_t>> h(N);
    for (int i = 0; i < N; ++i) bfs(g, goal[i], h[i]);  // search.py:300-302
    const auto t0 = std::chrono::steady_clock::now();
    auto timed_out = [&]() {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > time_limit_s;
    };

    auto root = std::make_shared<HLNode>();
    root->paths.resize(N);
    for (int i = 0; i < N; ++i) {
        root->paths[i] = low_level(g, start[i], goal[i], h[i], i, root->cons, max_steps);
        if (root->paths[i].empty()) return MAPF_ERR_NO_SPACE;  // reference: assert 'no solution for A-star search'
    }
    detect(*root);
    using QE = std::tuple<int, int, long long, std::shared_ptr<HLNode>>;
    auto cmp = [](const QE &a, const QE &b) {
        return std::tie(std::get<0>(a), std::get<1>(a), std::get<2>(a)) > std::tie(std::get<0>(b), std::get<1>(b), std::get<2>(b));
    };
    std::priority_queue<QE, std::vector<QE>, decltype(cmp)> open(cmp);
    long long generated = 0;
    open.emplace(root->cost, (int)root->cols.size(), generated++, root);
    std::shared_ptr<HLNode> sol;
    while (!open.empty()) {
        auto P = std::get<3>(open.top());
        open.pop();
        if (P->cols.empty()) {
            sol = P;
            break;
        }
        if (timed_out()) return MAPF_ERR_TIMEOUT;  // search.py:355
        const Collision c = P->cols.front();
        for (int side = 0; side < 2; ++side) {
            Constraint k;
            k.agent = side == 0 ? c.a1 : c.a2;
            k.t = c.t;
            if (c.c2 < 0) {
                k.from = -1, k.to = c.c1;
            } else if (side == 0) {
                k.from = c.c1, k.to = c.c2;
            } else {
                k.from = c.c2, k.to = c.c1;
            }
            auto Q = std::make_shared<HLNode>();
            Q->cons = P->cons;
            Q->cons.push_back(k);
            Q->paths = P->paths;
            auto path = low_level(g, start[k.agent], goal[k.agent], h[k.agent], k.agent, Q->cons, max_steps);
            if (path.empty()) continue;
            Q->paths[k.agent] = std::move(path);
            detect(*Q);
            open.emplace(Q->cost, (int)Q->cols.size(), generated++, Q);
        }
    }
    if (!sol) return MAPF_ERR_TIMEOUT;
    // plan -> per-step actions (search.py:408-440)
    size_t max_len = 0;
    for (auto &p : sol->paths) max_len = std::max(max_len, p.size());
    const int steps = (int)max_len - 1;
    if (steps > max_steps) return MAPF_ERR_TIMEOUT;
    for (int t = 1; t <= steps; ++t)
        for (int i = 0; i < N; ++i) {
            const int a = at(sol->paths[i], t - 1), b = at(sol->paths[i], t);
            const int dx = b / L - a / L, dy = b % L - a % L;
            int8_t act = 0;  // action_list: stay, (-1,0) up, (1,0) down, (0,-1) left, (0,1) right
            if (dx == -1) act = 1;
            else if (dx == 1) act = 2;
            else if (dy == -1) act = 3;
            else if (dy == 1) act = 4;
            actions_out[(size_t)(t - 1) * N + i] = act;
        }
    *num_steps = steps;
    if (sum_of_costs) *sum_of_costs = sol->cost;
    return MAPF_OK;
}

}  // extern "C"
