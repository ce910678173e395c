/* Batch front end of oracle/anyhit_grid.h -- TEST INFRASTRUCTURE, checker only (see that header for what is answered and why
 * the grid is only a candidate filter for ONE float32 predicate).  Built by oracle/Makefile into oracle/_c/anyhit_c.so with
 * -ffp-contract=off and without -ffast-math.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 *   ah_any_hit(org, dir, n, verts, tris, T, mode, out, stats)
 *       org, dir [n,3] float32; verts [V,3] float32; tris [T,3] int32; out [n] uint8 (1 = occluded)
 *       mode 0 = brute force (the definition), 1 = grid-filtered; stats[0] += predicate evaluations (may be NULL)
 *   The triangle records are (v0, v1 - v0, v2 - v0) in float32 -- what shade_oracle.any_hit_bruteforce and the product's BVH build form.
 */
#include "anyhit_grid.h"

#include <omp.h>

int ah_any_hit(const float* org, const float* dir, long long n, const float* verts, const int* tris, long long T, int mode,
               unsigned char* out, long long* stats) {
    float* rec = (float*)malloc(sizeof(float) * 9 * (size_t)(T > 0 ? T : 1));
    if (!rec) return -1;
    for (long long t = 0; t < T; ++t) {
        const float *a = verts + 3 * (long long)tris[3 * t], *b = verts + 3 * (long long)tris[3 * t + 1], *c = verts + 3 * (long long)tris[3 * t + 2];
        for (int k = 0; k < 3; ++k) {
            rec[9 * t + k] = a[k];
            rec[9 * t + 3 + k] = b[k] - a[k];
            rec[9 * t + 6 + k] = c[k] - a[k];
        }
    }
    AhGrid g;
    memset(&g, 0, sizeof(g));
    g.rec = rec;
    g.T = T;
    if (mode == 1 && ah_grid_build(&g, rec, T) != 0) {
        free(rec);
        return -2;
    }
    long long tests = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : tests)
    for (long long i = 0; i < n; ++i) {
        const float *o = org + 3 * i, *d = dir + 3 * i;
        long long nt = 0;
        int hit;
        if (mode == 1) hit = ah_grid_query(&g, o[0], o[1], o[2], d[0], d[1], d[2], &nt);
        else {
            hit = ah_brute(rec, T, o[0], o[1], o[2], d[0], d[1], d[2]);
            nt = T;
        }
        out[i] = (unsigned char)hit;
        tests += nt;
    }
    if (stats) stats[0] += tests;
    if (mode == 1) ah_grid_free(&g);
    free(rec);
    return 0;
}

void ah_set_threads(int n) { omp_set_num_threads(n > 0 ? n : omp_get_num_procs()); }
