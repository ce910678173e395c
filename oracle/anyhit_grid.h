/* Any-hit shadow query of the CHECKER at config size -- TEST INFRASTRUCTURE, never linked into the product.
 *
 * What it answers: the reference's shadow ray (render/optixutils/c_src/envsampling/kernel.cu:101-117: optixTrace with tmin 0,
 * tmax 1e16, TERMINATE_ON_FIRST_HIT, closest-hit disabled; the miss program :543-546 sets the payload) = "does ANY triangle of the
 * mesh intersect the ray at t in (0, 1e16)".  OptiX's hardware predicate is not specified anywhere (PARITY UNPINNED for the predicate
 * itself); the checker's predicate is Moeller-Trumbore in float32 without contraction, `ah_tri_hit` below -- the SAME sequence of
 * float operations as oracle/shade_oracle.py::any_hit_bruteforce and as the leaves of the product's traversal.
 *
 * Two evaluations of that one predicate:
 *   ah_brute  -- every triangle, in order: the definition.
 *   ah_grid   -- a uniform grid over the mesh as a CANDIDATE FILTER: only triangles registered in a cell the ray passes are tested
 *                (with the same predicate on the same full ray).  The filter is conservative twice over -- triangle boxes are padded
 *                by AH_PAD x the scene's extent before they are registered, and the ray's footprint in every slab of its major axis is
 *                padded by the same amount, both in double precision -- so a predicate-true triangle whose rounding-error-free
 *                intersection lies up to ~1e-4 x extent outside its own box is still a candidate.  "By construction" is not a
 *                proof for a float predicate; tests/test_oracle_anyhit_cpu.py asserts grid == brute on every golden scene, on
 *                grazing / degenerate / axis-parallel rays and on surface-start rays, and the config-size GPU tests re-assert it on a
 *                sample of the rays they check.
 * Written for this repository (no reference code involved): the reference has no software any-hit, OptiX does it in hardware.
 */
#ifndef GS_ORACLE_ANYHIT_GRID_H
#define GS_ORACLE_ANYHIT_GRID_H
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define AH_PAD 1e-4

/* Moeller-Trumbore, r = (v0, e1, e2), t in (0, 1e16).  float32, compile with -ffp-contract=off. */
static inline int ah_tri_hit(const float* r, float ox, float oy, float oz, float dx, float dy, float dz) {
    const float v0x = r[0], v0y = r[1], v0z = r[2], e1x = r[3], e1y = r[4], e1z = r[5], e2x = r[6], e2y = r[7], e2z = r[8];
    float px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
    float det = e1x * px + e1y * py + e1z * pz;
    if (!(fabsf(det) > 1e-20f)) return 0;
    float inv = 1.0f / det;
    float tx = ox - v0x, ty = oy - v0y, tz = oz - v0z;
    float u = (tx * px + ty * py + tz * pz) * inv;
    if (!(u >= 0.0f && u <= 1.0f)) return 0;
    float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
    float v = (dx * qx + dy * qy + dz * qz) * inv;
    if (!(v >= 0.0f && u + v <= 1.0f)) return 0;
    float t = (e2x * qx + e2y * qy + e2z * qz) * inv;
    return t > 0.0f && t < 1e16f;
}

/* a direction the query accepts: finite and not the zero vector (a zero / NaN direction never hits) */
static inline int ah_dir_valid(float dx, float dy, float dz) {
    return (dx == dx && dy == dy && dz == dz) && !(dx == 0.f && dy == 0.f && dz == 0.f);
}

typedef struct {
    const float* rec;   /* [T][9] = v0, e1, e2 (not owned) */
    long long T;
    int G[3];           /* cells per axis */
    double lo[3], h[3], pad;
    int64_t* start;     /* [G0*G1*G2 + 1] CSR offsets */
    int32_t* items;     /* triangle ids per cell */
    int built;
} AhGrid;

static inline int ah_brute(const float* rec, long long T, float ox, float oy, float oz, float dx, float dy, float dz) {
    if (!ah_dir_valid(dx, dy, dz)) return 0;
    for (long long t = 0; t < T; ++t)
        if (ah_tri_hit(rec + 9 * t, ox, oy, oz, dx, dy, dz)) return 1;
    return 0;
}

static inline void ah_grid_free(AhGrid* g) {
    free(g->start);
    free(g->items);
    memset(g, 0, sizeof(*g));
}

static inline int ah_clampi(long long v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : (int)v); }

/* cell range [c0, c1] of the interval [a, b] along axis k (already padded by the caller) */
static inline void ah_cells(const AhGrid* g, int k, double a, double b, int* c0, int* c1) {
    double fa = floor((a - g->lo[k]) / g->h[k]), fb = floor((b - g->lo[k]) / g->h[k]);
    if (!(fa == fa)) fa = 0;
    if (!(fb == fb)) fb = g->G[k] - 1;
    if (fa < -1) fa = -1;
    if (fb > g->G[k]) fb = g->G[k];
    *c0 = ah_clampi((long long)fa, 0, g->G[k] - 1);
    *c1 = ah_clampi((long long)fb, 0, g->G[k] - 1);
}

static inline void ah_tri_box(const float* r, double pad, double* lo, double* hi) {
    for (int k = 0; k < 3; ++k) {
        double a = r[k], b = (double)r[k] + (double)r[3 + k], c = (double)r[k] + (double)r[6 + k];
        double mn = a < b ? a : b, mx = a > b ? a : b;
        mn = mn < c ? mn : c;
        mx = mx > c ? mx : c;
        lo[k] = mn - pad;
        hi[k] = mx + pad;
    }
}

/* returns 0 on success; a grid over zero triangles (or non-finite vertices: then every query falls back to brute force) */
static inline int ah_grid_build(AhGrid* g, const float* rec, long long T) {
    memset(g, 0, sizeof(*g));
    g->rec = rec;
    g->T = T;
    if (T <= 0) return 0;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (long long t = 0; t < T; ++t) {
        double a[3], b[3];
        ah_tri_box(rec + 9 * t, 0.0, a, b);
        for (int k = 0; k < 3; ++k) {
            if (!(a[k] == a[k]) || !(b[k] == b[k]) || fabs(a[k]) > 1e30 || fabs(b[k]) > 1e30) return 0;   /* not built: brute force */
            if (a[k] < lo[k]) lo[k] = a[k];
            if (b[k] > hi[k]) hi[k] = b[k];
        }
    }
    double ext = 0;
    for (int k = 0; k < 3; ++k) ext = (hi[k] - lo[k]) > ext ? (hi[k] - lo[k]) : ext;
    if (!(ext > 0)) return 0;
    g->pad = AH_PAD * ext;
    int n = (int)floor(2.0 * cbrt((double)T) + 0.5);
    n = n < 4 ? 4 : (n > 160 ? 160 : n);
    double cell = ext / n;
    for (int k = 0; k < 3; ++k) {
        g->lo[k] = lo[k] - 2 * g->pad;
        double e = (hi[k] - lo[k]) + 4 * g->pad;
        int gk = (int)ceil(e / cell);
        g->G[k] = gk < 1 ? 1 : gk;
        g->h[k] = e / g->G[k];
    }
    const long long nc = (long long)g->G[0] * g->G[1] * g->G[2];
    g->start = (int64_t*)calloc((size_t)nc + 1, sizeof(int64_t));
    if (!g->start) return -1;
    for (int pass = 0; pass < 2; ++pass) {
        for (long long t = 0; t < T; ++t) {
            double a[3], b[3];
            int c0[3], c1[3];
            ah_tri_box(rec + 9 * t, g->pad, a, b);
            for (int k = 0; k < 3; ++k) ah_cells(g, k, a[k], b[k], &c0[k], &c1[k]);
            for (int z = c0[2]; z <= c1[2]; ++z)
                for (int y = c0[1]; y <= c1[1]; ++y)
                    for (int x = c0[0]; x <= c1[0]; ++x) {
                        const long long c = ((long long)z * g->G[1] + y) * g->G[0] + x;
                        if (pass == 0) ++g->start[c + 1];
                        else g->items[g->start[c]++] = (int32_t)t;
                    }
        }
        if (pass == 0) {
            for (long long c = 0; c < nc; ++c) g->start[c + 1] += g->start[c];
            g->items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(g->start[nc] > 0 ? g->start[nc] : 1));
            if (!g->items) return -1;
        } else {
            for (long long c = nc; c > 0; --c) g->start[c] = g->start[c - 1];   /* the fill advanced every start to its end */
            g->start[0] = 0;
        }
    }
    g->built = 1;
    return 0;
}

/* n_tests (optional) counts predicate evaluations */
static inline int ah_grid_query(const AhGrid* g, float ox, float oy, float oz, float dx, float dy, float dz, long long* n_tests) {
    if (!ah_dir_valid(dx, dy, dz)) return 0;
    if (g->T <= 0) return 0;
    if (!g->built || !(ox == ox && oy == oy && oz == oz) || fabsf(ox) > 1e30f || fabsf(oy) > 1e30f || fabsf(oz) > 1e30f || isinf(dx) || isinf(dy) ||
        isinf(dz)) {
        if (n_tests) *n_tests += g->T;
        return ah_brute(g->rec, g->T, ox, oy, oz, dx, dy, dz);
    }
    const double o[3] = {ox, oy, oz}, d[3] = {dx, dy, dz};
    /* the ray against the padded grid box, t in [0, inf) */
    double t0 = 0.0, t1 = 1e300;
    for (int k = 0; k < 3; ++k) {
        const double bl = g->lo[k] - g->pad, bh = g->lo[k] + g->h[k] * g->G[k] + g->pad;
        if (d[k] == 0.0) {
            if (o[k] < bl || o[k] > bh) return 0;
        } else {
            double ta = (bl - o[k]) / d[k], tb = (bh - o[k]) / d[k];
            if (ta > tb) { double s = ta; ta = tb; tb = s; }
            if (ta > t0) t0 = ta;
            if (tb < t1) t1 = tb;
        }
    }
    if (t0 > t1) return 0;
    int a = 0;
    if (fabs(d[1]) > fabs(d[a])) a = 1;
    if (fabs(d[2]) > fabs(d[a])) a = 2;
    const int b = (a + 1) % 3, c = (a + 2) % 3;
    const double pa0 = o[a] + t0 * d[a], pa1 = o[a] + t1 * d[a];
    int s0, s1;
    ah_cells(g, a, (pa0 < pa1 ? pa0 : pa1) - g->pad, (pa0 > pa1 ? pa0 : pa1) + g->pad, &s0, &s1);
    const int step = d[a] >= 0 ? 1 : -1;
    for (int s = step > 0 ? s0 : s1; s >= s0 && s <= s1; s += step) {
        /* the slab [lo + s h - pad, lo + (s+1) h + pad] of the major axis -> parameter range, clipped to [t0, t1] */
        const double sl = g->lo[a] + g->h[a] * s - g->pad, sh = g->lo[a] + g->h[a] * (s + 1) + g->pad;
        double ta = (sl - o[a]) / d[a], tb = (sh - o[a]) / d[a];
        if (ta > tb) { double q = ta; ta = tb; tb = q; }
        if (ta < t0) ta = t0;
        if (tb > t1) tb = t1;
        if (ta > tb) continue;
        int r0[3], r1[3];
        r0[a] = r1[a] = s;
        for (int j = 0; j < 2; ++j) {
            const int k = j == 0 ? b : c;
            const double p = o[k] + ta * d[k], q = o[k] + tb * d[k];
            ah_cells(g, k, (p < q ? p : q) - g->pad, (p > q ? p : q) + g->pad, &r0[k], &r1[k]);
        }
        for (int z = r0[2]; z <= r1[2]; ++z)
            for (int y = r0[1]; y <= r1[1]; ++y)
                for (int x = r0[0]; x <= r1[0]; ++x) {
                    const long long cell = ((long long)z * g->G[1] + y) * g->G[0] + x;
                    for (int64_t i = g->start[cell]; i < g->start[cell + 1]; ++i) {
                        if (n_tests) ++*n_tests;
                        if (ah_tri_hit(g->rec + 9 * (long long)g->items[i], ox, oy, oz, dx, dy, dz)) return 1;
                    }
                }
    }
    return 0;
}
#endif
