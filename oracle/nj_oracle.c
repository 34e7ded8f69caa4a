/*
 * nj_oracle.c -- CPU restatement of the reference's two non-production lattice solvers:
 *   st_cy.solve_s_t_path_no_jerk_fast      st_cy.pyx:209-312   node (t, s)
 *   st_cy.solve_s_t_path_no_jerk_djikstra  st_cy.pyx:96-206    node (t, s, s_prev)
 * Both use the module's hard-coded constants (st_cy.pyx:21-31), seed the queue with the layer-1 nodes, order the heap by
 * (cost, entry_order) with entry_order counting DOWN (later pushes win ties), stop at the first pop in the last layer.
 *
 * TEST INFRASTRUCTURE ONLY (checker for tests/; never linked into or called from the product).
 * Parity status: PINNED by tests/golden/golden_nojerk.npz (outputs of the reference's own compiled st_cy functions,
 * tests/golden/make_golden_nojerk.py).  Compiled with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static const double NJ_MAX_SPEED = 40, NJ_A_POS = 4.5, NJ_A_NEG = -6.0, NJ_V_DES = 30, NJ_V_W = 0.5, NJ_A_W = 1.0, NJ_D_W = 1000.0,
                    NJ_MIN_ALLOWED = 7.5;                                   /* st_cy.pyx:21-31 */

static double nj_penalty(double d, double min_allowed) { return d < min_allowed ? 1000000.0 / (d > 1.0 ? d : 1.0) : 1 / d; }   /* :34-38 */
static double nj_cost(double s, double s1, double s2, double dt, double d)                                                     /* :41-44 */
{
    double v = (s - s1) / dt;
    double a = (s - 2 * s1 + s2) / (dt * dt);
    return NJ_V_W * ((v - NJ_V_DES) * (v - NJ_V_DES)) + NJ_A_W * (a * a) + NJ_D_W * nj_penalty(d, NJ_MIN_ALLOWED);
}
static void nj_range(double s, double prev_s, double dt, double start_s, double ds, int *lo, int *hi)                          /* :56-62, :78-93 */
{
    double v = (s - prev_s) / dt;
    double min_v = v + NJ_A_NEG * dt; if (!(min_v > 0)) min_v = 0;          /* max(x, 0) */
    double max_v = v + NJ_A_POS * dt; if (!(max_v < NJ_MAX_SPEED)) max_v = NJ_MAX_SPEED;
    double min_s = s + min_v * dt, max_s = s + max_v * dt;
    double x = (min_s - start_s) / ds;
    int mi = (int)x, ma = (int)((max_s - start_s) / ds);
    if (mi < x) mi += 1;
    *lo = mi; *hi = ma + 1;
}

typedef struct { double cost; long long order; int t, s, prev, second; } nj_item;
typedef struct { nj_item *a; size_t n, cap; } nj_heap;
static int nj_less(const nj_item *x, const nj_item *y) { return x->cost < y->cost || (x->cost == y->cost && x->order < y->order); }
static void nj_push(nj_heap *h, nj_item it)
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 4096; h->a = (nj_item *)realloc(h->a, h->cap * sizeof(nj_item)); }
    size_t i = h->n++;
    while (i > 0) { size_t p = (i - 1) >> 1; if (!nj_less(&it, &h->a[p])) break; h->a[i] = h->a[p]; i = p; }
    h->a[i] = it;
}
static nj_item nj_pop(nj_heap *h)
{
    nj_item top = h->a[0], last = h->a[--h->n];
    size_t i = 0, n = h->n;
    for (;;) { size_t c = 2 * i + 1; if (c >= n) break; if (c + 1 < n && nj_less(&h->a[c + 1], &h->a[c])) c++; if (!nj_less(&h->a[c], &last)) break; h->a[i] = h->a[c]; i = c; }
    if (n) h->a[i] = last;
    return top;
}

/* triple = 0: solve_s_t_path_no_jerk_fast; 1: solve_s_t_path_no_jerk_djikstra.  Returns 0, or -1 if the seeding loop would
 * index past the grid (the reference raises IndexError there). */
int orc_solve_no_jerk(int triple, const uint8_t *obstacles, const double *s_values, int S, const double *t_values, int H, double v0,
                      const double *distances, double *s_sequence, long long *pops_out)
{
    const double ds = s_values[1] - s_values[0], dt = t_values[1] - t_values[0], start_s = s_values[0];
    const double est_prev = start_s - v0 * dt;
    const size_t states = triple ? (size_t)H * S * S : (size_t)H * S;
    uint8_t *enc = (uint8_t *)calloc(states, 1);
    int32_t *prev = (int32_t *)calloc(states, sizeof(int32_t));
    nj_heap h = {0, 0, 0};
    long long order = 0, pops = 0;
    int lo, hi;
    nj_range(start_s, est_prev, dt, start_s, ds, &lo, &hi);
    if (hi > S || lo < 0) { free(enc); free(prev); return -1; }
    for (int n = lo; n < hi; n++) {
        if (obstacles[(size_t)1 * S + n]) continue;
        nj_item it = {nj_cost(s_values[n], start_s, est_prev, dt, distances[(size_t)1 * S + n]), order, 1, n, 0, 0};
        nj_push(&h, it); order -= 1;
    }
    int best_t = 0, best_s = 0, best_p = 0;
    while (h.n > 0) {
        nj_item it = nj_pop(&h); pops++;
        const size_t at = triple ? ((size_t)it.t * S + it.s) * S + it.prev : (size_t)it.t * S + it.s;
        if (enc[at]) continue;
        enc[at] = 1; prev[at] = triple ? it.second : it.prev;
        if (it.t == H - 1) { best_t = H - 1; best_s = it.s; best_p = it.prev; break; }
        else if (it.t > best_t) { best_t = it.t; best_s = it.s; best_p = it.prev; }
        const double sv = s_values[it.s], pv = s_values[it.prev];
        nj_range(sv, pv, dt, start_s, ds, &lo, &hi);
        const int nt = it.t + 1;
        for (int n = lo; n < hi; n++) {
            if (n >= S) break;
            const size_t nat = triple ? ((size_t)nt * S + n) * S + it.s : (size_t)nt * S + n;
            if (enc[nat]) continue;
            if (obstacles[(size_t)nt * S + n]) continue;
            nj_item ni = {it.cost + nj_cost(s_values[n], sv, pv, dt, distances[(size_t)nt * S + n]), order, nt, n, it.s, it.prev};
            nj_push(&h, ni); order -= 1;
        }
    }
    for (int t = 0; t < H; t++) s_sequence[t] = 0.0;
    if (triple) {                                                    /* st_cy.pyx:193-204 */
        int bs = best_s, bp = best_p;
        for (int t = best_t; t > 1; t--) {
            s_sequence[t] = s_values[bs];
            int second = prev[((size_t)t * S + bs) * S + bp];
            bs = bp; bp = second;
        }
        s_sequence[0] = s_values[bp];
        if (H > 1) s_sequence[1] = s_values[bs];
    } else {                                                         /* st_cy.pyx:303-310 */
        int bs = best_s;
        for (int t = best_t; t > 0; t--) { s_sequence[t] = s_values[bs]; bs = prev[(size_t)t * S + bs]; }
        s_sequence[0] = s_values[bs];
    }
    if (pops_out) *pops_out = pops;
    free(h.a); free(enc); free(prev);
    return 0;
}
