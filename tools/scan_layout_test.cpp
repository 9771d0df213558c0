// scan_layout_test.cpp — CPU test of raytracing_amd/csrc/scan_layout.h (the device-free layout of an rt_scene for the
// scan), built with AddressSanitizer + UBSan by `make -C raytracing_amd/csrc layout_test` and run by
// tests/test_scan_layout.py.  Checks, on random and constructed scenes, the invariants the conservative scan rests on:
//   1. big list + scan slots hold every sphere exactly once; pads are -1 / +inf and never nominated
//   2. every member sphere, at time 0 and at time 1 (hence for every time between: a sphere is convex), lies inside its
//      bound at EVERY level, evaluated from the rounded fp32 numbers the kernel reads, with the scan-side (widened) radius
//   3. the planar list: every member and every bound has the shared coordinate to the bit
//   4. w = |c|^2 - r^2 of the expanded test is the f64 value rounded once; the scan-side -r^2 is wider than r^2 by the
//      stated slack and no more; cmax and reach bound what they say they bound
//   5. the big / small split, the list membership and the order are the same for the scene scaled by 0.01 and by 100
//   6. limits: 0, 1, kDirectMax, kDirectMax + 1 and 16384 spheres, all-moving, all-big, coincident duplicates keep world order
//   7. upper-level masks only name levels that exist; sparse masks contain the dense ones
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../raytracing_amd/csrc/scan_layout.h"

using namespace rtd;

static int g_fail = 0;
#define CHECK(cond, ...)                                                       \
    do {                                                                       \
        if (!(cond)) {                                                         \
            if (++g_fail <= 20) { std::printf("FAIL %s:%d %s: ", __FILE__, __LINE__, #cond); std::printf(__VA_ARGS__); std::printf("\n"); } \
        }                                                                      \
    } while (0)

struct Scene {
    std::vector<float> c0x, c0y, c0z, dcx, dcy, dcz, rad;
    std::vector<uint32_t> mat;
    rt_material m{};
    rt_scene s{};
    const rt_scene* view() {
        s = rt_scene{};
        s.n_spheres = (uint32_t)rad.size();
        s.c0x = c0x.data(); s.c0y = c0y.data(); s.c0z = c0z.data();
        s.dcx = dcx.data(); s.dcy = dcy.data(); s.dcz = dcz.data();
        s.radius = rad.data(); s.material_id = mat.data();
        m.type = RT_MAT_METAL;
        s.n_materials = 1; s.materials = &m;
        return &s;
    }
    void add(double x, double y, double z, double r, double dx = 0, double dy = 0, double dz = 0) {
        c0x.push_back((float)x); c0y.push_back((float)y); c0z.push_back((float)z);
        dcx.push_back((float)dx); dcy.push_back((float)dy); dcz.push_back((float)dz);
        rad.push_back((float)r); mat.push_back(0);
    }
    Scene scaled(double k) const {
        Scene o;
        for (size_t i = 0; i < rad.size(); ++i)
            o.add((double)c0x[i] * k, (double)c0y[i] * k, (double)c0z[i] * k, (double)rad[i] * k, (double)dcx[i] * k, (double)dcy[i] * k, (double)dcz[i] * k);
        return o;
    }
};

static void check_list(const char* what, const Scene& sc, const ScanLayout& L, const std::vector<F4>& a, const std::vector<F4>* dc,
                       const std::vector<float>* nr2, const std::vector<int32_t>& ids, const std::vector<F4>& bound, int levels,
                       int levels_sparse) {
    const float inf = std::numeric_limits<float>::infinity();
    CHECK(a.size() % kChunk == 0 && ids.size() == a.size(), "%s size %zu", what, a.size());
    if (nr2) CHECK(nr2->size() == a.size(), "%s nr2 size", what);
    if (dc) CHECK(dc->size() == a.size(), "%s dc size", what);
    const int n_chunks = (int)(a.size() / kChunk);
    CHECK(bound.size() == (size_t)bound_level_offset(n_chunks, kLevels), "%s bound array size %zu", what, bound.size());
    bool seen_pad = false;
    for (size_t i = 0; i < a.size(); ++i) {
        const float nr = nr2 ? (*nr2)[i] : a[i].w;
        if (ids[i] < 0) {
            seen_pad = true;
            CHECK(nr == inf, "%s pad %zu has -r^2 = %g", what, i, nr);
            if (nr2) CHECK(a[i].w == inf, "%s pad %zu has w = %g", what, i, a[i].w);
            continue;
        }
        CHECK(!seen_pad, "%s: a sphere after a pad at %zu", what, i);
        const int id = ids[i];
        const double r = sc.rad[id];
        // the rounded centre is the world centre minus the frame origin, rounded once
        const double ex[3] = {(double)sc.c0x[id] - L.org[0], (double)sc.c0y[id] - L.org[1], (double)sc.c0z[id] - L.org[2]};
        CHECK(a[i].x == (float)ex[0] && a[i].y == (float)ex[1] && a[i].z == (float)ex[2], "%s slot %zu centre", what, i);
        // scan-side radius: wider than r by the slack, and by no more than 1.01 x the slack (+ an fp32 rounding)
        const double slack = kDirectSlack * (r * r + r * L.reach);
        CHECK(-(double)nr >= r * r && -(double)nr <= r * r + 1.01 * slack + 1.2e-7 * r * r, "%s slot %zu -r^2 %g vs r^2 %g slack %g", what, i, -(double)nr, r * r, slack);
        if (nr2) {  // the expanded test's w, from the ROUNDED centre, rounded once
            const double w = (double)a[i].x * a[i].x + (double)a[i].y * a[i].y + (double)a[i].z * a[i].z - r * r;
            CHECK(a[i].w == (float)w, "%s slot %zu w %g vs %g", what, i, a[i].w, w);
            CHECK(std::sqrt((double)a[i].x * a[i].x + (double)a[i].y * a[i].y + (double)a[i].z * a[i].z) <= (double)L.cmax, "%s slot %zu beyond cmax", what, i);
        }
        if (dc) CHECK((*dc)[i].x == sc.dcx[id] && (*dc)[i].y == sc.dcy[id] && (*dc)[i].z == sc.dcz[id] && (*dc)[i].w == 0.0f, "%s slot %zu dc", what, i);
        // reach: |c| + r over the motion
        for (int e = 0; e < 2; ++e) {
            const double px = a[i].x + (dc ? e * (double)(*dc)[i].x : 0.0), py = a[i].y + (dc ? e * (double)(*dc)[i].y : 0.0), pz = a[i].z + (dc ? e * (double)(*dc)[i].z : 0.0);
            CHECK(std::sqrt(px * px + py * py + pz * pz) + r <= L.reach * (1 + 1e-6) + 1e-30, "%s slot %zu beyond reach", what, i);
        }
    }
    // containment at every level
    for (int lev = 0; lev < kLevels; ++lev) {
        const size_t span = lev == 0 ? (size_t)kGroup : (size_t)kChunk << (2 * (lev - 1));
        const int off = bound_level_offset(n_chunks, lev), cnt = bound_level_count(n_chunks, lev);
        for (int j = 0; j < cnt; ++j) {
            const F4 b = bound[(size_t)off + j];
            const size_t lo = (size_t)j * span, hi = std::min(a.size(), lo + span);
            bool any = false;
            for (size_t i = lo; i < hi; ++i) any |= ids[i] >= 0;
            if (!any) { CHECK(b.w == inf, "%s level %d node %d covers only pads but w = %g", what, lev, j, b.w); continue; }
            const double R2 = (double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z - (double)b.w;  // as the kernel's test sees it
            CHECK(R2 > 0, "%s level %d node %d R^2 %g", what, lev, j, R2);
            const double R = std::sqrt(std::max(R2, 0.0));
            CHECK(std::sqrt((double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z) <= (double)L.cmax, "%s bound centre beyond cmax", what);
            for (size_t i = lo; i < hi; ++i) {
                if (ids[i] < 0) continue;
                const double rr = std::sqrt(-(double)(nr2 ? (*nr2)[i] : a[i].w));
                for (int e = 0; e < (dc ? 2 : 1); ++e) {
                    const double dx = a[i].x + (dc ? e * (double)(*dc)[i].x : 0.0) - b.x, dy = a[i].y + (dc ? e * (double)(*dc)[i].y : 0.0) - b.y,
                                 dz = a[i].z + (dc ? e * (double)(*dc)[i].z : 0.0) - b.z;
                    const double far = std::sqrt(dx * dx + dy * dy + dz * dz) + rr;
                    // w = |b|^2 - R^2 is rounded to fp32 once: half an ulp of |w|, i.e. <= 2^-24 max(|b|^2, R^2), which
                    // the kernel's test covers 32 times over (kExpandErr = 32 ulp of (|o'| + cmax)^2 >= 32 ulp of |b|^2)
                    const double w_round = std::ldexp(std::max((double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z, R2), -24);
                    CHECK(far * far <= R2 + w_round, "%s level %d node %d: member %zu at time %d reaches %.9g beyond R %.9g", what, lev, j, i, e, far, R);
                    CHECK(R <= far * (1 + 1e-4) + 2 * w_round / std::max(far, 1e-30) || hi - lo > 1, "%s level %d node %d: a lone member's bound is loose: R %.9g vs %.9g", what, lev, j, R, far);
                }
            }
        }
        if (lev >= 1) {
            if ((levels >> lev) & 1) CHECK(cnt >= 2, "%s level %d enabled with %d nodes", what, lev, cnt);
            if ((levels_sparse >> lev) & 1) CHECK(cnt >= 2, "%s sparse level %d enabled with %d nodes", what, lev, cnt);
        }
    }
    CHECK((levels & ~levels_sparse) == 0 && (levels & 1) == 0 && (levels_sparse & 1) == 0, "%s level masks %x %x", what, levels, levels_sparse);
}

static void check_layout(const char* name, Scene& sc, const LayoutOptions& opt = LayoutOptions()) {
    ScanLayout L;
    std::string err;
    const int rc = L.build(sc.view(), opt, err);
    CHECK(rc == RT_OK, "%s: build failed: %s", name, err.c_str());
    if (rc != RT_OK) return;
    const size_t n = sc.rad.size();
    // 1. every sphere exactly once
    std::vector<int> count(n, 0);
    for (int id : L.big_id) { CHECK(id >= 0 && (size_t)id < n && L.is_big[id], "%s big id %d", name, id); if (id >= 0 && (size_t)id < n) count[id]++; }
    for (size_t k = 1; k < L.big_id.size(); ++k) CHECK(L.big_id[k - 1] < L.big_id[k], "%s big list not in world order", name);
    CHECK(L.big_id.size() <= kMaxBig, "%s %zu big spheres", name, L.big_id.size());
    CHECK(L.scan_id.size() == L.stat.size() + L.plan.size() + L.mov_a.size(), "%s scan_id size", name);
    for (size_t k = 0; k < L.scan_id.size(); ++k) {
        const int id = L.scan_id[k];
        const int expect = k < L.stat.size() ? L.stat_id[k] : k < L.stat.size() + L.plan.size() ? L.plan_id[k - L.stat.size()] : L.mov_id[k - L.stat.size() - L.plan.size()];
        CHECK(id == expect, "%s scan_id[%zu]", name, k);
        if (id >= 0) { CHECK((size_t)id < n && !L.is_big[id], "%s scan id %d", name, id); if ((size_t)id < n) count[id]++; }
    }
    for (size_t i = 0; i < n; ++i) CHECK(count[i] == 1, "%s sphere %zu appears %d times", name, i, count[i]);
    // membership: moving <=> dc != 0; planar members static
    for (size_t k = 0; k < L.mov_id.size(); ++k)
        if (L.mov_id[k] >= 0) CHECK(sc.dcx[L.mov_id[k]] != 0 || sc.dcy[L.mov_id[k]] != 0 || sc.dcz[L.mov_id[k]] != 0, "%s static sphere on the moving list", name);
    for (const std::vector<int32_t>* ids : {&L.stat_id, &L.plan_id})
        for (int id : *ids)
            if (id >= 0) CHECK(sc.dcx[id] == 0 && sc.dcy[id] == 0 && sc.dcz[id] == 0, "%s moving sphere on a static list", name);
    // 3. the planar list
    if (!L.plan.empty()) {
        size_t members = 0;
        for (size_t k = 0; k < L.plan.size(); ++k) {
            if (L.plan_id[k] < 0) continue;
            ++members;
            const float c3[3] = {L.plan[k].x, L.plan[k].y, L.plan[k].z};
            CHECK(c3[L.plan_axis] == L.plan_val, "%s planar member %zu off the plane", name, k);
        }
        CHECK(members >= kPlanarMin && !opt.no_planar, "%s planar list of %zu", name, members);
        for (const F4& b : L.plan_bound) {
            const float b3[3] = {b.x, b.y, b.z};
            if (b.w != std::numeric_limits<float>::infinity()) CHECK(b3[L.plan_axis] == L.plan_val, "%s planar bound off the plane", name);
        }
        // no static-list sphere shares the plane (they would all have been taken)
        for (size_t k = 0; k < L.stat.size(); ++k) {
            const float c3[3] = {L.stat[k].x, L.stat[k].y, L.stat[k].z};
            if (L.stat_id[k] >= 0) CHECK(c3[L.plan_axis] != L.plan_val, "%s static-list sphere in the plane", name);
        }
    }
    // 2, 4, 7
    check_list("static", sc, L, L.stat, nullptr, &L.stat_nr2, L.stat_id, L.stat_bound, L.stat_levels, L.stat_levels_sparse);
    check_list("planar", sc, L, L.plan, nullptr, &L.plan_nr2, L.plan_id, L.plan_bound, L.plan_levels, L.plan_levels_sparse);
    check_list("moving", sc, L, L.mov_a, &L.mov_b, nullptr, L.mov_id, L.mov_bound, L.mov_levels, L.mov_levels_sparse);
    CHECK(L.key_eps0 >= (float)L.reach * kKeyEps0 && L.key_eps0 <= (float)L.reach * kKeyEps0 * 1.0001f, "%s key_eps0", name);
    int axes = 0, moving = 0;
    for (size_t i = 0; i < n; ++i) {
        const bool mv = sc.dcx[i] != 0 || sc.dcy[i] != 0 || sc.dcz[i] != 0;
        moving |= mv;
        if (mv && !L.is_big[i]) axes |= (sc.dcx[i] != 0 ? 1 : 0) | (sc.dcy[i] != 0 ? 2 : 0) | (sc.dcz[i] != 0 ? 4 : 0);
    }
    CHECK(L.mov_axes == axes && L.any_moving == moving, "%s mov_axes %d vs %d, any_moving %d vs %d", name, L.mov_axes, axes, L.any_moving, moving);
    for (size_t i = 0; i < n; ++i) CHECK(L.inv_rad[i] == (float)(1.0 / (double)sc.rad[i]), "%s inv_rad", name);
    // 5. scale invariance of classification, membership and order (powers of two keep every rounding: exact invariance;
    // x0.01 / x100 change roundings, so only the classification and the list sizes are compared there)
    for (double k : {0.25, 64.0, 0.01, 100.0}) {
        Scene s2 = sc.scaled(k);
        ScanLayout L2;
        std::string e2;
        CHECK(L2.build(s2.view(), opt, e2) == RT_OK, "%s x%g: %s", name, k, e2.c_str());
        CHECK(L2.is_big == L.is_big, "%s x%g: big / small split differs", name, k);
        const bool pow2 = k == 0.25 || k == 64.0;
        if (pow2) {
            CHECK(L2.scan_id == L.scan_id && L2.big_id == L.big_id, "%s x%g: list order differs", name, k);
            CHECK(L2.stat_levels == L.stat_levels && L2.plan_levels == L.plan_levels && L2.mov_levels == L.mov_levels, "%s x%g: level masks differ", name, k);
            CHECK(L2.plan_axis == L.plan_axis || L.plan.empty(), "%s x%g: planar axis", name, k);
        } else {
            const long a = (long)(L.stat.size() + L.plan.size()), b = (long)(L2.stat.size() + L2.plan.size());
            CHECK(L2.mov_a.size() == L.mov_a.size() && std::labs(a - b) <= kChunk, "%s x%g: list sizes differ", name, k);
        }
    }
}

int main(int argc, char** argv) {
    const int n_random = argc > 1 ? std::atoi(argv[1]) : 60;
    std::mt19937_64 rng(12345);
    auto U = [&](double a, double b) { return std::uniform_real_distribution<double>(a, b)(rng); };
    // constructed scenes
    {
        Scene s;
        check_layout("empty", s);
    }
    {
        Scene s;
        s.add(0, -1000, 0, 1000);
        check_layout("lone ground", s);
    }
    for (int n_small : {kDirectMax, kDirectMax + 1}) {
        Scene s;
        s.add(0, -1000, 0, 1000);
        for (int i = 0; i < n_small; ++i) s.add(i * 2.5, 1, 0, 1);
        ScanLayout L;
        std::string err;
        CHECK(L.build(s.view(), LayoutOptions(), err) == RT_OK, "build");
        CHECK((L.scan_id.empty()) == (n_small <= kDirectMax), "direct rule: %d small spheres, %zu scan slots", n_small, L.scan_id.size());
        LayoutOptions nd;
        nd.no_direct = true;
        CHECK(L.build(s.view(), nd, err) == RT_OK && L.big_id.size() == 1 && !L.scan_id.empty(), "no_direct keeps the lists");
        check_layout("few spheres", s);
        check_layout("few spheres, no_direct", s, nd);
    }
    {  // the reference's final scene shape: ground + 22 x 22 grid at y = 0.2 + three r = 1 (planar list)
        Scene s;
        s.add(0, -1000, 0, 1000);
        for (int a = -11; a < 11; ++a)
            for (int b = -11; b < 11; ++b) s.add(a + 0.9 * U(0, 1), 0.2, b + 0.9 * U(0, 1), 0.2);
        s.add(0, 1, 0, 1); s.add(-4, 1, 0, 1); s.add(4, 1, 0, 1);
        check_layout("final scene", s);
        ScanLayout L;
        std::string err;
        L.build(s.view(), LayoutOptions(), err);
        CHECK(L.big_id.size() == 1 && L.plan_axis == 1 && L.plan.size() >= 484 && L.stat.size() == kChunk, "final scene lists: big %zu planar %zu (axis %d) static %zu",
              L.big_id.size(), L.plan.size(), L.plan_axis, L.stat.size());
        LayoutOptions np;
        np.no_planar = true;
        check_layout("final scene, no planar list", s, np);
        LayoutOptions wo;
        wo.world_order = true;
        check_layout("final scene, world order", s, wo);
        L.build(s.view(), wo, err);
        for (size_t k = 1; k < L.plan_id.size(); ++k)
            if (L.plan_id[k] >= 0) CHECK(L.plan_id[k - 1] < L.plan_id[k], "world order kept");
    }
    {  // all moving, on every axis; times 0 and 1 both inside every bound
        Scene s;
        for (int i = 0; i < 700; ++i) s.add(U(-20, 20), U(0, 5), U(-20, 20), U(0.1, 0.5), U(-1, 1), U(-1, 1), U(-1, 1));
        check_layout("all moving", s);
    }
    {  // all the same size, huge: nothing is "big" relative to the median
        Scene s;
        for (int i = 0; i < 40; ++i) s.add(U(-1e4, 1e4), U(-1e4, 1e4), U(-1e4, 1e4), 900);
        check_layout("all large", s);
    }
    {  // more big candidates than kMaxBig: the largest stay, the rest ride the scan
        Scene s;
        for (int i = 0; i < 400; ++i) s.add(U(-10, 10), 0.2, U(-10, 10), 0.2);
        for (int i = 0; i < 80; ++i) s.add(U(-500, 500), U(-500, 500), U(-500, 500), 10 + i);
        check_layout("too many big", s);
    }
    {  // coincident duplicates keep world order inside a list (exact ties: first wins, hit.rs:100-110)
        Scene s;
        for (int i = 0; i < 40; ++i) s.add(U(-5, 5), U(0, 2), U(-5, 5), 0.3);
        for (int k = 0; k < 6; ++k) s.add(1.5, 0.75, -2.0, 0.3);
        ScanLayout L;
        std::string err;
        L.build(s.view(), LayoutOptions(), err);
        int last = -1;
        for (int id : L.stat_id)
            if (id >= 40) { CHECK(id > last, "duplicates out of world order: %d after %d", id, last); last = id; }
        check_layout("duplicates", s);
    }
    {  // the limit
        Scene s;
        s.add(0, -1000, 0, 1000);
        for (int i = 1; i < RT_MAX_SPHERES; ++i) s.add(U(-40, 40), U(0.2, 6), U(-40, 40), U(0.15, 0.45), 0, i % 3 == 0 ? U(0.05, 0.5) : 0, 0);
        check_layout("16384 spheres", s);
    }
    {  // validation
        Scene s;
        s.add(0, 0, 0, 1);
        s.rad[0] = 0.0f;
        ScanLayout L;
        std::string err;
        CHECK(L.build(s.view(), LayoutOptions(), err) == RT_ERR_BAD_ARG && !err.empty(), "zero radius accepted");
        s.rad[0] = 1.0f; s.c0x[0] = std::numeric_limits<float>::quiet_NaN();
        CHECK(L.build(s.view(), LayoutOptions(), err) == RT_ERR_BAD_ARG, "NaN centre accepted");
        s.c0x[0] = 0.0f; s.mat[0] = 3;
        CHECK(L.build(s.view(), LayoutOptions(), err) == RT_ERR_BAD_ARG, "bad material id accepted");
    }
    // random scenes: sizes, radii over two decades, motion on random axes, big spheres, planes, three length scales
    for (int c = 0; c < n_random; ++c) {
        Scene s;
        const int sizes[] = {1, 3, 5, 6, 15, 17, 63, 200, 700, 3000};
        const int n = sizes[rng() % 10];
        const double ext = (const double[]){1.0, 10.0, 50.0, 1e-3, 1e4}[rng() % 5];
        const int plane = rng() % 2 ? (int)(rng() % 3) : -1;
        const double plane_val = U(-ext, ext), plane_frac = U(0.3, 1.0);
        const int mode = (int)(rng() % 4);
        for (int i = 0; i < n; ++i) {
            double c3[3] = {U(-ext, ext), U(-ext, ext), U(-ext, ext)}, d3[3] = {0, 0, 0};
            if (plane >= 0 && U(0, 1) < plane_frac) c3[plane] = plane_val;
            if (mode && U(0, 1) < 0.5)
                for (int a = 0; a < mode; ++a) d3[(a + c) % 3] = U(-0.05 * ext, 0.05 * ext);
            s.add(c3[0], c3[1], c3[2], ext * std::pow(10.0, U(-2.5, -0.7)), d3[0], d3[1], d3[2]);
        }
        const int n_big = (int)(rng() % 3);
        for (int k = 0; k < std::min(n_big, n); ++k) s.rad[k] = (float)(ext * U(20, 200));
        char name[64];
        std::snprintf(name, sizeof name, "random %d (n %d, ext %g)", c, n, ext);
        check_layout(name, s);
    }
    if (g_fail) { std::printf("%d check(s) FAILED\n", g_fail); return 1; }
    std::printf("scan_layout_test: ok (%d random scenes)\n", n_random);
    return 0;
}
