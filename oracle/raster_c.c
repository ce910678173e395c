/* CPU oracle of the rasteriser's INTEGER decisions at full frame / mesh size -- TEST INFRASTRUCTURE, checker only.
 *
 * PARITY UNPINNED (same status as oracle/raster_oracle.py): the reference delegates rasterisation to nvdiffrast (third party,
 * not in /root/reference; call sites render/render.py:26,358,377-379); no reference test or golden vector touches it.
 *
 * This file is a statement-for-statement C restatement of oracle/raster_oracle.py::rasterize_ids (the python loop over
 * triangles that is the specification of gshell_amd/csrc/raster.hip) so that the same specification can be evaluated on a
 * 10^5..10^6-triangle mesh at 512 x 512 in well under a second.  tests/test_raster_oracle.py pins it to the python loop bit for
 * bit on every small scene, including near-clipped triangles and depth ties.  Compiled with -ffp-contract=off and without
 * -ffast-math: every float32 operation below is the one numpy performs in raster_oracle.py, in the same order.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from this file.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define SUBPIX 256

static inline void zmin(uint64_t *cell, uint64_t key)
{
    uint64_t old = __atomic_load_n(cell, __ATOMIC_RELAXED);
    while (key < old && !__atomic_compare_exchange_n(cell, &old, key, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) { }
}

/* raster_oracle._depth_key */
static inline uint64_t depth_key(float zw)
{
    uint32_t u;
    memcpy(&u, &zw, 4);
    return (u & 0x80000000u) ? (uint64_t)(~u) : (uint64_t)(u | 0x80000000u);
}

/* raster_oracle._pix_ndc */
static inline float pix_ndc(int p, int n) { return ((float)p + 0.5f) * (2.0f / (float)n) - 1.0f; }

/* raster_oracle._project_fix for one vertex */
static inline int project_fix(const float *p, int H, int W, int64_t *fx, int64_t *fy)
{
    float x = p[0], y = p[1], w = p[3];
    float xn = x / w, yn = y / w;
    float sx = (xn * 0.5f + 0.5f) * (float)W;
    float sy = (yn * 0.5f + 0.5f) * (float)H;
    float gx = floorf(sx * (float)SUBPIX + 0.5f);
    float gy = floorf(sy * (float)SUBPIX + 0.5f);
    int ok = (w > 1e-6f) && (fabsf(gx) < 16777216.0f) && (fabsf(gy) < 16777216.0f);   /* NaN compares false, as in numpy */
    *fx = ok ? (int64_t)gx : 0;
    *fy = ok ? (int64_t)gy : 0;
    return ok;
}

/* raster_oracle._near_clipped: every pixel of the frame is tested */
static void near_clipped(uint64_t *zb, const float *p0, const float *p1, const float *p2, int64_t t, int H, int W)
{
    if (!((p0[2] + p0[3] >= 0) || (p1[2] + p1[3] >= 0) || (p2[2] + p2[3] >= 0))) return;
    for (int py = 0; py < H; ++py) {
        float fy = pix_ndc(py, H);
        for (int px = 0; px < W; ++px) {
            float fx = pix_ndc(px, W);
            float p0x = p0[0] - fx * p0[3], p0y = p0[1] - fy * p0[3];
            float p1x = p1[0] - fx * p1[3], p1y = p1[1] - fy * p1[3];
            float p2x = p2[0] - fx * p2[3], p2y = p2[1] - fy * p2[3];
            float a0 = p1x * p2y - p1y * p2x;
            float a1 = p2x * p0y - p2y * p0x;
            float a2 = p0x * p1y - p0y * p1x;
            float s = a0 + a1 + a2;
            float iw = 1.0f / s;
            float b0 = a0 * iw, b1 = a1 * iw;
            float b2 = 1.0f - b0 - b1;
            float zw = (p0[2] * a0 + p1[2] * a1 + p2[2] * a2) / (p0[3] * a0 + p1[3] * a1 + p2[3] * a2);
            float w = p0[3] * b0 + p1[3] * b1 + p2[3] * b2;
            int good = (s != 0) && (b0 >= 0) && (b1 >= 0) && (b2 >= 0) && (w > 0) && (zw >= -1.0f) && (zw <= 1.0f);
            if (good) zmin(&zb[(size_t)py * W + px], (depth_key(zw) << 32) | (uint64_t)t);
        }
    }
}

static inline int64_t min3(int64_t a, int64_t b, int64_t c) { int64_t m = a < b ? a : b; return m < c ? m : c; }
static inline int64_t max3(int64_t a, int64_t b, int64_t c) { int64_t m = a > b ? a : b; return m > c ? m : c; }

/* pos [B][V][4] float32 clip space, tri [T][3] int32 -> ids [B][H][W] int64 (-1 = empty).  zbuf: caller scratch [B][H][W] uint64.
 * fixpt: caller scratch [V][3] int64 (fx, fy, ok), rebuilt per view. */
int ro_rasterize_ids(const float *pos, int B, int V, const int32_t *tri, int64_t T, int H, int W, int64_t *ids, uint64_t *zbuf, int64_t *fixpt)
{
    const uint64_t EMPTY = ~(uint64_t)0;
    for (int b = 0; b < B; ++b) {
        const float *P = pos + (size_t)b * V * 4;
        uint64_t *zb = zbuf + (size_t)b * H * W;
        for (size_t i = 0; i < (size_t)H * W; ++i) zb[i] = EMPTY;
#pragma omp parallel for schedule(static)
        for (int v = 0; v < V; ++v)
            fixpt[3 * (size_t)v + 2] = project_fix(P + 4 * (size_t)v, H, W, &fixpt[3 * (size_t)v], &fixpt[3 * (size_t)v + 1]);
#pragma omp parallel for schedule(dynamic, 256)
        for (int64_t t = 0; t < T; ++t) {
            const int32_t *i = tri + 3 * t;
            const float *p0 = P + 4 * (size_t)i[0], *p1 = P + 4 * (size_t)i[1], *p2 = P + 4 * (size_t)i[2];
            int f0 = p0[3] > 1e-6f, f1 = p1[3] > 1e-6f, f2 = p2[3] > 1e-6f;
            if ((f0 || f1 || f2) && !(f0 && f1 && f2)) {
                near_clipped(zb, p0, p1, p2, t, H, W);
                continue;
            }
            if (!(fixpt[3 * (size_t)i[0] + 2] && fixpt[3 * (size_t)i[1] + 2] && fixpt[3 * (size_t)i[2] + 2])) continue;
            int64_t x[3] = {fixpt[3 * (size_t)i[0]], fixpt[3 * (size_t)i[1]], fixpt[3 * (size_t)i[2]]};
            int64_t y[3] = {fixpt[3 * (size_t)i[0] + 1], fixpt[3 * (size_t)i[1] + 1], fixpt[3 * (size_t)i[2] + 1]};
            int64_t area2 = (x[1] - x[0]) * (y[2] - y[0]) - (y[1] - y[0]) * (x[2] - x[0]);
            if (area2 == 0) continue;
            int64_t sgn = area2 > 0 ? 1 : -1;
            int64_t x0 = (min3(x[0], x[1], x[2]) - 128 + 255) >> 8, x1 = (max3(x[0], x[1], x[2]) - 128) >> 8;
            int64_t y0 = (min3(y[0], y[1], y[2]) - 128 + 255) >> 8, y1 = (max3(y[0], y[1], y[2]) - 128) >> 8;
            if (x0 < 0) x0 = 0;
            if (y0 < 0) y0 = 0;
            if (x1 > W - 1) x1 = W - 1;
            if (y1 > H - 1) y1 = H - 1;
            if (x0 > x1 || y0 > y1) continue;
            int64_t ea[3], eb[3], ec[3];
            int own[3];
            for (int e = 0; e < 3; ++e) {
                int a = (e + 1) % 3, c = (e + 2) % 3;
                int64_t dx = (x[c] - x[a]) * sgn, dy = (y[c] - y[a]) * sgn;
                ea[e] = -dy; eb[e] = dx; ec[e] = dy * x[a] - dx * y[a];
                own[e] = (dy > 0) || (dy == 0 && dx > 0);
            }
            for (int64_t py = y0; py <= y1; ++py) {
                int64_t cy = py * SUBPIX + 128;
                float fy = pix_ndc((int)py, H);
                for (int64_t px = x0; px <= x1; ++px) {
                    int64_t cx = px * SUBPIX + 128;
                    int inside = 1;
                    for (int e = 0; e < 3; ++e) {
                        int64_t v = ea[e] * cx + eb[e] * cy + ec[e];
                        inside &= (v > 0) || (v == 0 && own[e]);
                    }
                    if (!inside) continue;
                    /* raster_oracle._bary (only z/w is consumed here) */
                    float fx = pix_ndc((int)px, W);
                    float p0x = p0[0] - fx * p0[3], p0y = p0[1] - fy * p0[3];
                    float p1x = p1[0] - fx * p1[3], p1y = p1[1] - fy * p1[3];
                    float p2x = p2[0] - fx * p2[3], p2y = p2[1] - fy * p2[3];
                    float a0 = p1x * p2y - p1y * p2x;
                    float a1 = p2x * p0y - p2y * p0x;
                    float a2 = p0x * p1y - p0y * p1x;
                    float z = p0[2] * a0 + p1[2] * a1 + p2[2] * a2;
                    float w = p0[3] * a0 + p1[3] * a1 + p2[3] * a2;
                    float zw = z / w;
                    if ((zw >= -1.0f) && (zw <= 1.0f)) zmin(&zb[(size_t)py * W + px], (depth_key(zw) << 32) | (uint64_t)t);
                }
            }
        }
        int64_t *out = ids + (size_t)b * H * W;
        for (size_t i = 0; i < (size_t)H * W; ++i) out[i] = zb[i] == EMPTY ? -1 : (int64_t)(zb[i] & 0xFFFFFFFFu);
    }
    return 0;
}
