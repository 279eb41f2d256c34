// host_driver.cc -- TEST INFRASTRUCTURE: every host-only piece of native code of this repository in one executable built with
// g++/gcc -fsanitize=address,undefined (tests/test_sanitize_cpu.py; never on the GPU):
//   * the CBS + space-time A* planner of csrc/mapf_search.hip (compiled with -x c++), on the golden cases of tests/golden/search.npz,
//   * the scenario generator of csrc/mapf_generate_host.inc (what mapf_generate runs), over a sweep of shapes and densities,
//   * the CPU oracle oracle/mapf_oracle.c (compiled as C), stepping generated scenarios under random action tapes.
// Input file (argv[1]): int32 count, then per search case {int32 L, int32 N, int8 map[L*L], int16 agents[N*2], int16 goals[N*2]}.
// Output file (argv[2]): per search case {int32 status, int32 steps, int32 cost, int8 actions[steps*N], int32 dist0[L*L]}; then per
// generator sweep point {int32 status, int32 redraws, maps, agents, goals}; then one uint64 FNV hash of the oracle rollouts.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mapf_env.h"
#include "mapf_search.h"

namespace {
#include "mapf_generate_host.inc"
}

extern "C" {
void oracle_navi(int L, int N, const int8_t *map, const int16_t *goals, uint8_t *navi);
int oracle_step(int L, int N, const int8_t *map, int16_t *agents, const int16_t *goals, const int8_t *actions, int8_t *rclass, uint8_t *done);
void oracle_observe(int L, int N, int r, const int8_t *map, const int16_t *agents, const uint8_t *navi, uint8_t *obs);
}

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *in = std::fopen(argv[1], "rb"), *out = std::fopen(argv[2], "wb");
    if (!in || !out) return 2;
    int32_t count = 0;
    if (std::fread(&count, 4, 1, in) != 1) return 2;
    const double limit = argc > 3 ? std::atof(argv[3]) : 120.0;
    for (int k = 0; k < count; ++k) {
        int32_t L = 0, N = 0;
        if (std::fread(&L, 4, 1, in) != 1 || std::fread(&N, 4, 1, in) != 1) return 2;
        std::vector<int8_t> map((size_t)L * L);
        std::vector<int16_t> ag((size_t)N * 2), gl((size_t)N * 2);
        if (std::fread(map.data(), 1, map.size(), in) != map.size() || std::fread(ag.data(), 2, ag.size(), in) != ag.size() ||
            std::fread(gl.data(), 2, gl.size(), in) != gl.size())
            return 2;
        const int max_steps = 256;
        std::vector<int8_t> acts((size_t)max_steps * N);
        int steps = 0, cost = 0;
        int32_t st = mapf_find_path(L, N, map.data(), ag.data(), gl.data(), limit, max_steps, acts.data(), &steps, &cost);
        int32_t s32 = steps, c32 = cost;
        std::fwrite(&st, 4, 1, out);
        std::fwrite(&s32, 4, 1, out);
        std::fwrite(&c32, 4, 1, out);
        std::fwrite(acts.data(), 1, (size_t)steps * N, out);
        std::vector<int32_t> dist((size_t)L * L);
        if (mapf_distance_field(L, map.data(), gl[0], gl[1], dist.data()) != MAPF_OK) return 3;
        std::fwrite(dist.data(), 4, dist.size(), out);
    }
    // generator sweep (the same points tests/test_sanitize_cpu.py asks the product library for) + oracle rollouts on what it made
    const struct { int E, L, N; float rho; uint64_t seed; } sweep[] = {
        {16, 10, 1, -1.f, 1}, {16, 20, 6, -1.f, 2}, {8, 32, 40, 0.3f, 3}, {4, 40, 16, 0.3f, 4}, {2, 64, 128, 0.3f, 5}, {8, 16, 40, 0.3f, 6},
        {32, 8, 4, 0.45f, 7}, {4, 12, 60, 0.1f, 8}, {3, 5, 3, 0.0f, 9}};
    uint64_t h = 1469598103934665603ull;
    for (const auto &p : sweep) {
        std::vector<int8_t> maps((size_t)p.E * p.L * p.L);
        std::vector<int16_t> ag((size_t)p.E * p.N * 2), gl((size_t)p.E * p.N * 2);
        int32_t redraws = 0;
        int32_t st = generate_scenarios(p.E, p.L, p.N, p.rho, p.seed, maps.data(), ag.data(), gl.data(), &redraws);
        std::fwrite(&st, 4, 1, out);
        std::fwrite(&redraws, 4, 1, out);
        std::fwrite(maps.data(), 1, maps.size(), out);
        std::fwrite(ag.data(), 2, ag.size(), out);
        std::fwrite(gl.data(), 2, gl.size(), out);
        if (st != MAPF_OK) continue;
        Rng rng(p.seed * 77 + 5);
        std::vector<uint8_t> navi((size_t)p.N * 4 * p.L * p.L), obs((size_t)p.N * 6 * 81);
        std::vector<int8_t> act(p.N), rc(p.N);
        for (int e = 0; e < p.E; ++e) {
            const int8_t *m = maps.data() + (size_t)e * p.L * p.L;
            int16_t *a = ag.data() + (size_t)e * p.N * 2;
            const int16_t *g = gl.data() + (size_t)e * p.N * 2;
            oracle_navi(p.L, p.N, m, g, navi.data());
            for (int t = 0; t < 24; ++t) {
                for (int i = 0; i < p.N; ++i) act[i] = (int8_t)rng.below(5);
                uint8_t done = 0;
                if (oracle_step(p.L, p.N, m, a, g, act.data(), rc.data(), &done) != 0) return 4;
                oracle_observe(p.L, p.N, 4, m, a, navi.data(), obs.data());
                h = fnv(h, a, (size_t)p.N * 4);
                h = fnv(h, rc.data(), p.N);
                h = fnv(h, obs.data(), obs.size());
            }
        }
    }
    std::fwrite(&h, 8, 1, out);
    std::fclose(in);
    std::fclose(out);
    std::printf("host_driver ok: %d search cases, %zu generator points, rollout hash %016llx\n", count, sizeof(sweep) / sizeof(sweep[0]),
                (unsigned long long)h);
    return 0;
}
