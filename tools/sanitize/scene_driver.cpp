// tools/sanitize/scene_driver.cpp — drives libptmi_scene's entry points (scene_prep.cpp linked in directly) under the CPU
// sanitizers: the partial sort (arr.ts vectors + random arrays), the BVH builder with 1 and with 8 threads on the same triangle soup
// (outputs must be byte-identical), the emissive-light list. tools/sanitize/run.sh builds it with -fsanitize=address,undefined
// and with -fsanitize=thread and keeps the logs under profiles/.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "ptmi_scene.h"

static std::vector<ptmi_triangle> soup(uint32_t n, uint32_t seed) {
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f), s(0.001f, 0.05f);
    std::vector<ptmi_triangle> t(n);
    std::memset(t.data(), 0, n * sizeof(ptmi_triangle));
    for (uint32_t i = 0; i < n; i++) {
        const float c[3] = {u(g), u(g), u(g)};
        for (int k = 0; k < 3; k++) {
            t[i].v0[k] = c[k] + s(g); t[i].v1[k] = c[k] - s(g); t[i].v2[k] = c[k] + s(g) * (k == 1 ? -1.0f : 1.0f);
            t[i].n0[k] = t[i].n1[k] = t[i].n2[k] = k == 1 ? 1.0f : 0.0f;
        }
        if (i % 97 == 0) std::memcpy(t[i].v1, t[i].v0, sizeof t[i].v0);      // degenerate
        if (i % 53 == 0 && i) std::memcpy(&t[i], &t[i - 1], sizeof t[i]);     // equal centroids: the unstable sort's swap order
        t[i].material_index = i % 3;
    }
    return t;
}

int main() {
    int bad = 0;
    {   // arr.ts: sortArrayPartially on random arrays and sub-ranges, both orders
        std::mt19937 g(7);
        for (int rep = 0; rep < 200; rep++) {
            const int64_t n = 1 + g() % 300;
            std::vector<double> a(n);
            for (auto &x : a) x = (double)(g() % 50) - 25.0;
            const int64_t s = g() % n, e = s + 1 + g() % (n - s);
            const int desc = rep & 1;
            if (ptmi_scene_sort_partially_f64(a.data(), n, s, e, desc) != 0) { bad++; continue; }
            for (int64_t i = s + 1; i < e; i++) if (desc ? a[i - 1] < a[i] : a[i - 1] > a[i]) { bad++; break; }
        }
        double z[4] = {3, 1, 2, 0};
        if (ptmi_scene_sort_partially_f64(z, 4, 2, 2, 0) != -1 || ptmi_scene_sort_partially_f64(z, 4, 0, 5, 0) != -1) bad++;   // "Invalid indices"
    }
    for (uint32_t n : {1u, 5u, 4096u, 150000u}) {          // the last one is above the builder's 32 768-triangle task size
        std::vector<ptmi_triangle> a = soup(n, 11), b = a;
        const uint32_t cap = ptmi_scene_bvh_node_bound(n);
        std::vector<ptmi_bvh_node> na(cap), nb(cap);
        uint32_t ca = 0, cb = 0, da = 0, db = 0;
        ptmi_scene_set_threads(1);
        if (ptmi_scene_build_bvh(a.data(), n, 4, 12, na.data(), cap, &ca, &da) != 0) { std::fprintf(stderr, "%s\n", ptmi_scene_last_error()); bad++; }
        ptmi_scene_set_threads(8);
        if (ptmi_scene_build_bvh(b.data(), n, 4, 12, nb.data(), cap, &cb, &db) != 0) { std::fprintf(stderr, "%s\n", ptmi_scene_last_error()); bad++; }
        if (ca != cb || da != db || std::memcmp(na.data(), nb.data(), ca * sizeof(ptmi_bvh_node)) || std::memcmp(a.data(), b.data(), n * sizeof(ptmi_triangle))) {
            std::fprintf(stderr, "threaded build differs at %u triangles\n", n); bad++;
        }
        std::vector<ptmi_material> mats(3);
        std::memset(mats.data(), 0, mats.size() * sizeof(ptmi_material));
        mats[1].emission[0] = 1.0f;
        std::vector<ptmi_light> lights(n + 2);
        std::memset(lights.data(), 0, lights.size() * sizeof(ptmi_light));
        uint32_t nl = 1;
        if (ptmi_scene_emissive_lights(a.data(), n, mats.data(), 3, lights.data(), (uint32_t)lights.size(), &nl) != 0) bad++;
        uint32_t tiny = 1;
        if (n > 5 && ptmi_scene_emissive_lights(a.data(), n, mats.data(), 3, lights.data(), 2, &tiny) == 0) bad++;      // capacity too small must fail cleanly
        std::printf("%u triangles: %u nodes, depth %u, %u lights\n", n, ca, da, nl);
    }
    std::printf(bad ? "FAILED: %d\n" : "ok\n", bad);
    return bad != 0;
}
