/*
 * st_oracle.c -- CPU restatement of the reference's ST ("MPC") hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path may include,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and only as the checker / reported baseline.
 *
 * Parity status: PINNED.  the .npz files under tests/golden/ hold outputs of the reference's own
 * prediction.py / st.py / st_cy.pyx (imported/compiled in the build container by
 * tests/golden/make_golden.py); tests/test_oracle_golden.py checks every function
 * below against them bit-for-bit.
 *
 * Every function cites the reference file:line it restates (paths are relative
 * to the reference checkout).  All arithmetic is fp64, compiled with
 * -ffp-contract=off -fno-builtin-pow so that each IEEE operation of the
 * reference is one IEEE operation here.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_KMAX 32

/* ---- parameters: Settings.* that reach the path (config.py:30-37,94-110,143,150) ---- */
typedef struct {
    /* grid, st.py:727-734 */
    double future_s;       /* Settings.FUTURE_S            */
    double ds;             /* Settings.S_DISCRETIZATION    */
    double dt;             /* Settings.T_DISCRETIZATION    */
    double future_t;       /* Settings.FUTURE_T            */
    double start_unc;      /* Settings.START_UNCERTAINTY   */
    double unc_per_s;      /* Settings.UNCERTAINTY_PER_SECOND */
    /* the 11 solver tunables in st_cy.solve_s_t_path_fast's order, st.py:740-746 */
    double d_w, v_w, a_w, j_w, v_des, v_max, a_min, a_max, j_min, j_max, min_allowed;
    /* constants used by grid build / predictor */
    double car_length;     /* Settings.CAR_LENGTH 5.0 (st.py:37,52-53) */
    double crash_min_s;    /* Settings.CRASH_MIN_S (st.py:46) */
    double max_pred_decel; /* Settings.MAX_PREDICTED_DECELERATION -4 (prediction.py:86) */
    double follow_gap;     /* literal 30 (prediction.py:85) */
    double react_thr;      /* HighwayState.ego_reaction_threshold 8 (prediction.py:11) */
    double crash_thr;      /* HighwayState.ego_crash_threshold 11 (prediction.py:12) */
    double comb_min_dist;  /* Settings.COMBINATION_MIN_DISTANCE 5.1 (st.py:800) */
} orc_params;

typedef struct {
    double ego_x, ego_y, ego_v, ego_a;
    int k;
    double xs[ORC_KMAX];
    double vs[ORC_KMAX];
} orc_state;

typedef struct {
    long long nodes;     /* settled (expanded) nodes  */
    long long edges;     /* relaxed edges (cost evaluations) */
    long long cells;     /* distinct target cells reached */
} orc_stats;

/* Scratch memory.  A batch worker thread owns one arena that it resets before every episode, so that the large per-solve
 * arrays (grids, back-pointers, label rows: ~4 MB at S = 7201, H = 40) are not mmap'ed and unmapped once per solve --
 * with many threads that serialises on the process's address-space lock and was what the cpu_baseline measured.
 * Outside a worker (single calls from the tests) the same macros fall through to malloc/free. */
typedef struct { char *base; size_t cap, off; int active; } orc_arena;
static __thread orc_arena orc_ws = {0, 0, 0, 0};
static void *orc_alloc(size_t n, int zero)
{
    if (orc_ws.active) {
        size_t at = (orc_ws.off + 63) & ~(size_t)63;
        if (at + n <= orc_ws.cap) { orc_ws.off = at + n; void *q = orc_ws.base + at; if (zero) memset(q, 0, n); return q; }
    }
    return zero ? calloc(n, 1) : malloc(n);
}
static void orc_release(void *q)
{
    if (orc_ws.active && (char *)q >= orc_ws.base && (char *)q < orc_ws.base + orc_ws.cap) return;   /* arena memory: reset wholesale */
    free(q);
}

/* libm pow through a volatile pointer: Python's float ** int and Cython's dt**3 are
 * runtime libm pow() calls (control.py:38, st_cy.pyx:49); gcc must not fold them. */
static double (*volatile orc_pow)(double, double) = pow;

/* ---- control.py:37-38 distance(); 366-370 constants; 373-380 get_ego_s ---- */
static const double MP_X = -50.9, MP_Y = 1.72;   /* merge_point  */
static const double MP2_X = 1.5, MP2_Y = -1.5;   /* merge_point2 */
static const double MP3_X = -51.0;               /* merge_point3[0] */

static double orc_distance(double x1, double y1, double x2, double y2)
{
    return sqrt(orc_pow(x1 - x2, 2.0) + orc_pow(y1 - y2, 2.0));
}

double orc_ego_s(double x, double y)
{
    const double common_s = MP2_X - MP3_X;        /* control.py:370 */
    if (x < MP_X) return -orc_distance(x, y, MP_X, MP_Y);
    else if (x < MP2_X) return orc_distance(x, y, MP_X, MP_Y);
    else return x - MP2_X + common_s;
}

/* control.py:388-389 get_obstacle_s_from_x */
static double orc_obstacle_s(double x) { return x - MP3_X; }

/* ---- prediction.py:46-105 HighwayState.predict_step_with_ego ---- */
int orc_predict_with_ego(const orc_params *p, const orc_state *in, double selected_speed,
                         double dt, double min_crash_distance, orc_state *out)
{
    double cx = in->ego_x, cy = in->ego_y, px, py;
    if (cx < MP2_X) {                                    /* :48 */
        double d0 = MP2_X - cx, d1 = MP2_Y - cy;        /* :50 */
        /* np.linalg.norm -> sqrt(x.dot(x)); this image's OpenBLAS ddot evaluates the
         * 2-vector as fma(d1,d1,d0*d0) (verified 20000/20000 in the build container). */
        double nrm = sqrt(fma(d1, d1, d0 * d0));         /* :51 */
        d0 /= nrm; d1 /= nrm;
        double step = selected_speed * dt;               /* :52 */
        d0 *= step; d1 *= step;
        px = cx + d0; py = cy + d1;                      /* :53-54 */
        if (py < -1.6) py = -1.6;                        /* :55-56 */
    } else {
        py = cy; px = cx + selected_speed * dt;          /* :58-59 */
    }
    double next_acc = (selected_speed - in->ego_v) / dt; /* :61 */
    double es = orc_ego_s(px, py);
    int ego_can_crash = es > p->crash_thr;               /* :64 */
    int ego_has_merged = es > p->react_thr;              /* :66 */

    double last_x = INFINITY, last_speed = 0.0;          /* :72-73 */
    int ego_encountered = 0;
    int k = in->k;
    double nx[ORC_KMAX], nv[ORC_KMAX];
    for (int i = 0; i < k; i++) {                        /* :75 */
        double other_speed = in->vs[i], other_x = in->xs[i];
        if (other_x < px && !ego_encountered) {          /* :78 */
            ego_encountered = 1;
            if (ego_has_merged) { last_x = px; last_speed = selected_speed; }
        }
        double speed_diff = last_speed - other_speed;    /* :83 */
        double x_diff = last_x - other_x;
        double new_speed;
        if (speed_diff < 0 && x_diff < p->follow_gap) {  /* :85 */
            double acc = (p->max_pred_decel > speed_diff) ? p->max_pred_decel : speed_diff; /* max(speed_diff, MPD) :86 */
            new_speed = other_speed + acc * dt;          /* :87 */
        } else {
            new_speed = other_speed;                     /* :90 */
        }
        double nxt = other_x + new_speed * dt;           /* :91 */
        last_x = nxt; last_speed = new_speed;            /* :93-94 */
        nx[i] = nxt; nv[i] = new_speed;
    }
    int crashed = 0;                                     /* :99-103 */
    double cdd = (min_crash_distance > p->car_length) ? min_crash_distance : p->car_length; /* max(CAR_LENGTH, mcd) */
    for (int i = 0; i < k; i++)
        if (fabs(nx[i] - px) < cdd && ego_can_crash) crashed = 1;

    out->ego_x = px; out->ego_y = py; out->ego_v = selected_speed; out->ego_a = next_acc;
    out->k = k;
    memcpy(out->xs, nx, sizeof(double) * (size_t)k);
    memcpy(out->vs, nv, sizeof(double) * (size_t)k);
    return crashed;
}

/* ---- prediction.py:22-44 HighwayState.predict_step_without_ego ---- */
int orc_predict_without_ego(const orc_params *p, const orc_state *in, double dt,
                            double min_crash_distance, orc_state *out)
{
    double ego_s = orc_ego_s(in->ego_x, in->ego_y);      /* :24 */
    double ego_x = in->ego_x;
    if (ego_s < p->react_thr || in->k == 0)              /* :26 */
        return orc_predict_with_ego(p, in, 0.0, dt, min_crash_distance, out);
    else if (in->xs[0] < ego_x) {                        /* :28 */
        orc_state m = *in;
        m.ego_x = -20.0; m.ego_y = -10.0; m.ego_v = 0.0; m.ego_a = 0.0;   /* :30 */
        return orc_predict_with_ego(p, &m, 0.0, dt, min_crash_distance, out);
    } else {
        double last_speed = 0.0, last_x = 0.0;           /* :33-34 */
        for (int i = 0; i < in->k; i++) {
            if (in->xs[i] < ego_x) {                     /* :36 */
                orc_state m = *in;
                m.ego_x = last_x - p->car_length - 5;    /* :39 */
                m.ego_v = last_speed; m.ego_a = 0.0;
                return orc_predict_with_ego(p, &m, last_speed, dt, min_crash_distance, out);
            } else {
                last_speed = in->vs[i]; last_x = in->xs[i];
            }
        }
        return orc_predict_with_ego(p, in, last_speed, dt, min_crash_distance, out); /* :44 */
    }
}

/* ---- np.arange sizes used at st.py:31-32 (numpy: len = ceil((stop-start)/step)) ---- */
int orc_num_s(const orc_params *p, double start_s)
{
    double stop = start_s + p->future_s + p->ds;
    return (int)ceil((stop - start_s) / p->ds);
}
int orc_num_t(const orc_params *p)
{
    double stop = p->future_t + p->dt;
    return (int)ceil((stop - 0.0) / p->dt);
}
/* numpy arange fill for float64: a[0]=start, a[1]=start+step, a[i]=start+i*(a[1]-a[0]) */
static void orc_arange(double start, double step, int n, double *out)
{
    if (n > 0) out[0] = start;
    if (n > 1) out[1] = start + step;
    if (n > 2) {
        double delta = out[1] - out[0];
        for (int i = 2; i < n; i++) out[i] = start + (double)i * delta;
    }
}

/* ---- st.py:25-70 find_s_t_obstacles_from_state ----
 * obstacles[H*S] u8, distances[H*S] f64, s_values[S], t_values[H];
 * obs_tab (optional) [H][ORC_KMAX] predicted other_xs per layer, k per layer is state k. */
int orc_build_grid(const orc_params *p, const orc_state *state, double start_s,
                   int S, int H, uint8_t *obstacles, double *distances,
                   double *s_values, double *t_values, double *obs_tab)
{
    orc_arange(start_s, p->ds, S, s_values);                      /* :31 */
    orc_arange(0.0, p->dt, H, t_values);                          /* :32 */
    memset(obstacles, 0, (size_t)H * S);
    for (size_t i = 0; i < (size_t)H * S; i++) distances[i] = 0.0 + 1e10;   /* :34-35 */
    int discrete_length = (int)(p->car_length / p->ds);           /* :37 */
    orc_state cur = *state, nxt;
    for (int t = 0; t < H; t++) {
        double unc = p->start_unc + p->unc_per_s * t_values[t];   /* :40 */
        int dunc = (int)(unc / p->ds);                            /* :41 */
        if (t != 0) { orc_predict_without_ego(p, &cur, p->dt, 5.0, &nxt); cur = nxt; } /* :42-43 */
        if (obs_tab) for (int i = 0; i < ORC_KMAX; i++) obs_tab[(size_t)t * ORC_KMAX + i] = (i < cur.k) ? cur.xs[i] : NAN;
        uint8_t *ob = obstacles + (size_t)t * S;
        double *di = distances + (size_t)t * S;
        for (int c = 0; c < cur.k; c++) {                         /* :44 */
            double o = orc_obstacle_s(cur.xs[c]);                 /* :45 */
            if (o < p->crash_min_s - p->min_allowed) break;       /* :46-47 */
            else if (o > s_values[S - 1] + p->car_length) continue; /* :48-49 */
            double front = o - p->car_length - unc;               /* :52 */
            double back = o + p->car_length + unc;                /* :53 */
            for (int i = 0; i < S; i++) {
                double f = fabs(s_values[i] - front);
                double b = fabs(s_values[i] - back);
                double d = di[i];
                d = (f < d) ? f : d;                              /* :56 */
                d = (b < d) ? b : d;                              /* :57 */
                di[i] = d;
            }
            int i0 = (int)((o - start_s) / p->ds);                /* :60 -> :20-22, trunc toward 0 */
            int imin = i0 - discrete_length - dunc; if (imin < 0) imin = 0;   /* :61 */
            int imax = i0 + discrete_length + dunc; if (imax > S) imax = S;   /* :62 */
            if (imin < S && imax > 0)                             /* :63 */
                for (int i = imin; i < imax; i++) { ob[i] = 1; di[i] = 0.0; }  /* :64-65 */
        }
    }
    return 0;
}

/* ---- st_cy.pyx:34-38 distance_penalty ---- */
static inline double orc_distance_penalty(double d, double min_allowed)
{
    if (d < min_allowed) return 1000000.0 / ((1.0 > d) ? 1.0 : d);   /* max(d, 1.0) */
    else return 1 / d;
}
/* ---- st_cy.pyx:46-50 cost_with_jerk ---- */
static inline double orc_cost_with_jerk(double s, double s1, double s2, double s3, double dt,
                                        double dt3, double d, double min_allowed, double v_w,
                                        double v_des, double a_w, double j_w, double d_w)
{
    double v = (s - s1) / dt;
    double a = (s - 2 * s1 + s2) / (dt * dt);
    double j = (s - 3 * s1 + 3 * s2 - s3) / dt3;
    double dv = v - v_des;
    return v_w * (dv * dv) + a_w * (a * a) + j_w * (j * j) + d_w * orc_distance_penalty(d, min_allowed);
}
/* ---- st_cy.pyx:65-75 get_feasible_next_s_range_with_jerk_limits ---- */
static inline void orc_next_s_range(double s, double s1, double s2, double dt, double j_min,
                                    double j_max, double a_min, double a_max, double v_max,
                                    double *min_s, double *max_s)
{
    double prev_v = (s1 - s2) / dt;
    double v = (s - s1) / dt;
    double a = (v - prev_v) / dt;
    double lo_a = a + j_min * dt; if (a_min > lo_a) lo_a = a_min;        /* max(x, a_min) */
    double hi_a = a + j_max * dt; if (a_max < hi_a) hi_a = a_max;        /* min(x, a_max) */
    double lo_v = v + lo_a * dt; if (0 > lo_v) lo_v = 0;                 /* max(x, 0) */
    double hi_v = v + hi_a * dt; if (v_max < hi_v) hi_v = v_max;         /* min(x, v_max) */
    *min_s = s + lo_v * dt;
    *max_s = s + hi_v * dt;
}
/* ---- st_cy.pyx:78-93 get_all_range_indices -> half-open [lo, hi) ---- */
static inline void orc_range_indices(double start_s, double delta_s, double rmin, double rmax,
                                     int *lo, int *hi)
{
    double x = (rmin - start_s) / delta_s;
    int mi = (int)x;
    int ma = (int)((rmax - start_s) / delta_s);
    if (mi < x) mi += 1;
    *lo = mi; *hi = ma + 1;
}

/* ---- heap of the reference's 8-tuples; Python tuple order = lexicographic ---- */
typedef struct {
    double cost; int t; int s_idx; double s_val; int p_idx; double p_val; int q_idx; double q_val;
} orc_item;

static inline int orc_item_less(const orc_item *a, const orc_item *b)
{
    if (a->cost != b->cost) return a->cost < b->cost;
    if (a->t != b->t) return a->t < b->t;
    if (a->s_idx != b->s_idx) return a->s_idx < b->s_idx;
    if (a->s_val != b->s_val) return a->s_val < b->s_val;
    if (a->p_idx != b->p_idx) return a->p_idx < b->p_idx;
    if (a->p_val != b->p_val) return a->p_val < b->p_val;
    if (a->q_idx != b->q_idx) return a->q_idx < b->q_idx;
    return a->q_val < b->q_val;
}
typedef struct { orc_item *a; size_t n, cap; } orc_heap;
static void heap_push(orc_heap *h, orc_item it)
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 4096; h->a = (orc_item *)realloc(h->a, h->cap * sizeof(orc_item)); }
    size_t i = h->n++;
    while (i > 0) {
        size_t par = (i - 1) >> 1;
        if (!orc_item_less(&it, &h->a[par])) break;
        h->a[i] = h->a[par]; i = par;
    }
    h->a[i] = it;
}
/* the heap's item array is kept per thread between solves of a batch worker (it grows to ~10 MB on the wide lattice) */
static __thread orc_heap orc_heap_keep = {0, 0, 0};
static orc_heap orc_heap_get(void)
{
    orc_heap h = {0, 0, 0};
    if (orc_ws.active) { h = orc_heap_keep; h.n = 0; orc_heap_keep.a = NULL; orc_heap_keep.cap = 0; }
    return h;
}
static void orc_heap_done(orc_heap *h)
{
    if (orc_ws.active) { orc_heap_keep = *h; orc_heap_keep.n = 0; }
    else free(h->a);
    h->a = NULL;
}
static orc_item heap_pop(orc_heap *h)
{
    orc_item top = h->a[0];
    orc_item last = h->a[--h->n];
    size_t i = 0, n = h->n;
    for (;;) {
        size_t c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && orc_item_less(&h->a[c + 1], &h->a[c])) c++;
        if (!orc_item_less(&h->a[c], &last)) break;
        h->a[i] = h->a[c]; i = c;
    }
    if (n) h->a[i] = last;
    return top;
}

/* ---- st_cy.pyx:315-399 solve_s_t_path_fast: literal heap Dijkstra ----
 * Outputs: s_sequence[H] (0.0 past best_t), path_idx[H] (-1 past best_t), *best_t_out,
 * *cost_out = the popped total_cost of the terminal node (dropped by the reference, :393-399). */
int orc_solve_heap(const uint8_t *obstacles, const double *s_values, int S, const double *t_values,
                   int H, double v0, double a0, const double *distances, double d_w, double v_w,
                   double a_w, double j_w, double v_des, double v_max, double a_min, double a_max,
                   double j_min, double j_max, double min_allowed, double *s_sequence,
                   int *path_idx, int *best_t_out, double *cost_out, orc_stats *stats)
{
    double delta_s = s_values[1] - s_values[0];          /* :318 */
    double delta_t = t_values[1] - t_values[0];          /* :319 */
    double start_s = s_values[0];                        /* :320 */
    double dt3 = orc_pow(delta_t, 3.0);                  /* delta_t**3 :49 */
    uint8_t *encountered = (uint8_t *)orc_alloc((size_t)H * S, 1);       /* :323 */
    int32_t *previous = (int32_t *)orc_alloc((size_t)H * S * sizeof(int32_t), 1); /* :324 */
    double est_prev = start_s - v0 * delta_t;            /* :329 */
    double est_second = est_prev - delta_t * (v0 - a0 * delta_t); /* :330 */
    orc_heap h = orc_heap_get();
    orc_item first = {0.0, 0, 0, start_s, 0, est_prev, 0, est_second};   /* :342 */
    heap_push(&h, first);
    int best_last_s = 0, best_t = 0;                     /* :352-353 */
    double best_cost = 0.0;
    long long nodes = 0, edges = 0;
    while (h.n > 0) {                                    /* :355 */
        orc_item it = heap_pop(&h);
        size_t at = (size_t)it.t * S + it.s_idx;
        if (encountered[at]) continue;                   /* :359 */
        encountered[at] = 1; previous[at] = it.p_idx;    /* :362-363 */
        nodes++;
        if (it.t > best_t) { best_t = it.t; best_last_s = it.s_idx; best_cost = it.cost; } /* :365-367 */
        if (it.t == H - 1) break;                        /* :368 */
        double mn, mx; int lo, hi;
        orc_next_s_range(it.s_val, it.p_val, it.q_val, delta_t, j_min, j_max, a_min, a_max, v_max, &mn, &mx); /* :372 */
        orc_range_indices(start_s, delta_s, mn, mx, &lo, &hi);           /* :373 */
        int nt = it.t + 1;
        for (int n = lo; n < hi; n++) {                  /* :376 */
            if (n >= S) break;                           /* :379 */
            size_t nat = (size_t)nt * S + n;
            if (!encountered[nat]) {                     /* :382 */
                if (obstacles[nat]) continue;            /* :383 */
                double nv = s_values[n];
                double c = orc_cost_with_jerk(nv, it.s_val, it.p_val, it.q_val, delta_t, dt3,
                                              distances[nat], min_allowed, v_w, v_des, a_w, j_w, d_w); /* :387 */
                orc_item ni = {it.cost + c, nt, n, nv, it.s_idx, it.s_val, it.p_idx, it.p_val};       /* :388 */
                heap_push(&h, ni);
                edges++;
            }
        }
    }
    int bs = best_last_s;                                /* :391 */
    for (int t = 0; t < H; t++) { s_sequence[t] = 0.0; path_idx[t] = -1; }
    for (int t = best_t; t > 0; t--) {                   /* :394 */
        s_sequence[t] = s_values[bs]; path_idx[t] = bs;
        bs = previous[(size_t)t * S + bs];
    }
    s_sequence[0] = s_values[bs]; path_idx[0] = bs;      /* :398 */
    *best_t_out = best_t; *cost_out = best_cost;
    if (stats) { stats->nodes = nodes; stats->edges = edges; stats->cells = 0; }
    orc_heap_done(&h); orc_release(encountered); orc_release(previous);
    return 0;
}

/* ---- layer-synchronous forward DP, equivalent to the heap search above ----
 * C[t+1,n] = min over settled (t,s), n in window(t,s), !obstacles[t+1,n] of C[t,s]+edge,
 * ties -> smaller predecessor index (heap tuple order (cost,t,n,s_n,s,...), st_cy.pyx:388);
 * terminal = argmin (C, s) at the deepest non-empty layer (st_cy.pyx:365-369).
 * Expands every reachable node (a superset of what the heap search settles). */
int orc_solve_layered(const uint8_t *obstacles, const double *s_values, int S, const double *t_values,
                      int H, double v0, double a0, const double *distances, double d_w, double v_w,
                      double a_w, double j_w, double v_des, double v_max, double a_min, double a_max,
                      double j_min, double j_max, double min_allowed, double *s_sequence,
                      int *path_idx, int *best_t_out, double *cost_out, orc_stats *stats)
{
    double delta_s = s_values[1] - s_values[0];
    double delta_t = t_values[1] - t_values[0];
    double start_s = s_values[0];
    double dt3 = orc_pow(delta_t, 3.0);
    double est_prev = start_s - v0 * delta_t;
    double est_second = est_prev - delta_t * (v0 - a0 * delta_t);
    int32_t *previous = (int32_t *)orc_alloc((size_t)H * S * sizeof(int32_t), 0);
    double *cc = (double *)orc_alloc(sizeof(double) * S * 2, 0);
    double *p1 = (double *)orc_alloc(sizeof(double) * S * 2, 0);   /* history value s_{t-1} per node */
    double *p2 = (double *)orc_alloc(sizeof(double) * S * 2, 0);   /* history value s_{t-2} per node */
    double *cur_c = cc, *nxt_c = cc + S, *cur_p1 = p1, *nxt_p1 = p1 + S, *cur_p2 = p2, *nxt_p2 = p2 + S;
    for (int i = 0; i < S; i++) cur_c[i] = INFINITY;
    cur_c[0] = 0.0; cur_p1[0] = est_prev; cur_p2[0] = est_second;
    int lo_w = 0, hi_w = 1;            /* window of possibly-reached cells in the current layer */
    int best_t = 0, best_s = 0; double best_cost = 0.0;
    long long nodes = 0, edges = 0, cells = 0;
    for (int t = 0; t < H - 1; t++) {
        int nlo = S, nhi = 0;
        for (int i = 0; i < S; i++) nxt_c[i] = INFINITY;
        int32_t *prev_n = previous + (size_t)(t + 1) * S;
        for (int s = lo_w; s < hi_w; s++) {
            double C = cur_c[s];
            if (!(C < INFINITY)) continue;
            nodes++;
            double sv = s_values[s], mn, mx; int lo, hi;
            orc_next_s_range(sv, cur_p1[s], cur_p2[s], delta_t, j_min, j_max, a_min, a_max, v_max, &mn, &mx);
            orc_range_indices(start_s, delta_s, mn, mx, &lo, &hi);
            for (int n = lo; n < hi; n++) {
                if (n >= S) break;
                size_t nat = (size_t)(t + 1) * S + n;
                if (obstacles[nat]) continue;
                double c = C + orc_cost_with_jerk(s_values[n], sv, cur_p1[s], cur_p2[s], delta_t, dt3,
                                                  distances[nat], min_allowed, v_w, v_des, a_w, j_w, d_w);
                edges++;
                if (c < nxt_c[n]) {          /* sources visited in increasing s: strict < keeps the smaller index on ties */
                    if (!(nxt_c[n] < INFINITY)) cells++;
                    nxt_c[n] = c; prev_n[n] = s; nxt_p1[n] = sv; nxt_p2[n] = cur_p1[s];
                    if (n < nlo) nlo = n;
                    if (n + 1 > nhi) nhi = n + 1;
                }
            }
        }
        if (nlo >= nhi) break;                /* nothing reached: failure, deepest layer is t */
        double bc = INFINITY; int bs = -1;
        for (int n = nlo; n < nhi; n++) if (nxt_c[n] < bc) { bc = nxt_c[n]; bs = n; }
        best_t = t + 1; best_s = bs; best_cost = bc;
        double *tmp;
        tmp = cur_c; cur_c = nxt_c; nxt_c = tmp;
        tmp = cur_p1; cur_p1 = nxt_p1; nxt_p1 = tmp;
        tmp = cur_p2; cur_p2 = nxt_p2; nxt_p2 = tmp;
        lo_w = nlo; hi_w = nhi;
    }
    int bs = best_s;
    for (int t = 0; t < H; t++) { s_sequence[t] = 0.0; path_idx[t] = -1; }
    for (int t = best_t; t > 0; t--) {
        s_sequence[t] = s_values[bs]; path_idx[t] = bs;
        bs = previous[(size_t)t * S + bs];
    }
    s_sequence[0] = s_values[bs]; path_idx[0] = bs;
    *best_t_out = best_t; *cost_out = best_cost;
    if (stats) { stats->nodes = nodes; stats->edges = edges; stats->cells = cells; }
    orc_release(previous); orc_release(cc); orc_release(p1); orc_release(p2);
    return 0;
}

/* Replay of the accumulated cost along a returned path, in the accumulation order of
 * st_cy.pyx:388 (total = total + edge), history per st_cy.pyx:329-330,342,388. */
double orc_replay_cost(const int *path_idx, int best_t, const double *s_values, int S,
                       const double *t_values, double v0, double a0, const double *distances,
                       const orc_params *p)
{
    double delta_t = t_values[1] - t_values[0];
    double start_s = s_values[0];
    double dt3 = orc_pow(delta_t, 3.0);
    double s1 = start_s - v0 * delta_t;
    double s2 = s1 - delta_t * (v0 - a0 * delta_t);
    double s0 = start_s, total = 0.0;
    for (int t = 1; t <= best_t; t++) {
        int n = path_idx[t];
        double sv = s_values[n];
        double c = orc_cost_with_jerk(sv, s0, s1, s2, delta_t, dt3, distances[(size_t)t * S + n],
                                      p->min_allowed, p->v_w, p->v_des, p->a_w, p->j_w, p->d_w);
        total = total + c;
        s2 = s1; s1 = s0; s0 = sv;
    }
    return total;
}

/* ---- st.py:726-754 get_appropriate_base_st_path_and_obstacles + st.py:790-802 ----
 * solver: 0 = heap (literal), 1 = layered.  path_dist[t] = distances[t, int((s_t - s0)/delta_s)]
 * for t <= best_t (st.py:797-799), NaN past best_t.  crash_guaranteed per st.py:790-802. */
int orc_solve_state(const orc_params *p, const orc_state *st, double start_s, int solver,
                    int *path_idx, double *s_sequence, int *best_t, double *cost,
                    double *path_dist, int *crash_guaranteed, orc_stats *stats)
{
    int S = orc_num_s(p, start_s), H = orc_num_t(p);
    uint8_t *ob = (uint8_t *)orc_alloc((size_t)H * S, 0);
    double *di = (double *)orc_alloc(sizeof(double) * (size_t)H * S, 0);
    double *sv = (double *)orc_alloc(sizeof(double) * S, 0);
    double *tv = (double *)orc_alloc(sizeof(double) * H, 0);
    orc_build_grid(p, st, start_s, S, H, ob, di, sv, tv, NULL);
    double c;
    if (solver == 0)
        orc_solve_heap(ob, sv, S, tv, H, st->ego_v, st->ego_a, di, p->d_w, p->v_w, p->a_w, p->j_w, p->v_des,
                       p->v_max, p->a_min, p->a_max, p->j_min, p->j_max, p->min_allowed, s_sequence, path_idx, best_t, &c, stats);
    else
        orc_solve_layered(ob, sv, S, tv, H, st->ego_v, st->ego_a, di, p->d_w, p->v_w, p->a_w, p->j_w, p->v_des,
                          p->v_max, p->a_min, p->a_max, p->j_min, p->j_max, p->min_allowed, s_sequence, path_idx, best_t, &c, stats);
    if (cost) *cost = c;
    int crash = (*best_t != H - 1);                      /* st.py:792-796 (trailing zeros) */
    double delta_s = sv[1] - sv[0];
    for (int t = 0; t < H; t++) {
        if (t <= *best_t) {
            int qi = (int)((s_sequence[t] - sv[0]) / delta_s);       /* st.py:798 -> :20-22 */
            double d = di[(size_t)t * S + qi];
            if (path_dist) path_dist[t] = d;
            if (d < p->comb_min_dist - p->car_length) crash = 1;     /* st.py:800 */
        } else if (path_dist) path_dist[t] = NAN;
    }
    if (crash_guaranteed) *crash_guaranteed = crash;
    orc_release(ob); orc_release(di); orc_release(sv); orc_release(tv);
    return 0;
}

/* ---- batched driver over independent episodes (threads), used for tests and for the
 * cpu_baseline leg of bench.py.  ego[N][5] = x, y, v, a, start_s. ---- */
typedef struct {
    const orc_params *p; int N, Kmax, H, solver; const double *ego; const int32_t *k_count;
    const double *ox, *ov; int32_t *path_idx; int32_t *best_t; double *cost; double *path_dist;
    int32_t *crash; long long *counters; int tid, nthreads; volatile int *next;
} orc_job;

static void *orc_worker(void *arg)
{
    orc_job *j = (orc_job *)arg;
    int H = j->H;
    double *sseq = (double *)malloc(sizeof(double) * H);
    int *pidx = (int *)malloc(sizeof(int) * H);
    double *pd = (double *)malloc(sizeof(double) * H);
    long long nodes = 0, edges = 0, cells = 0;
    {   /* per-thread arena for everything one solve allocates (see orc_alloc) */
        int Smax = orc_num_s(j->p, 0.0) + 8;
        size_t need = (size_t)H * Smax * (1 + 8 + 4 + 1) + (size_t)Smax * 8 * 8 + (1u << 16);
        orc_ws.base = (char *)malloc(need); orc_ws.cap = orc_ws.base ? need : 0; orc_ws.off = 0; orc_ws.active = orc_ws.base != NULL;
    }
    for (;;) {
        int e = __sync_fetch_and_add(j->next, 1);
        if (e >= j->N) break;
        orc_ws.off = 0;
        orc_state st;
        st.ego_x = j->ego[e * 5 + 0]; st.ego_y = j->ego[e * 5 + 1];
        st.ego_v = j->ego[e * 5 + 2]; st.ego_a = j->ego[e * 5 + 3];
        st.k = j->k_count[e];
        for (int i = 0; i < st.k; i++) { st.xs[i] = j->ox[(size_t)e * j->Kmax + i]; st.vs[i] = j->ov[(size_t)e * j->Kmax + i]; }
        int bt, cr; double c; orc_stats s;
        orc_solve_state(j->p, &st, j->ego[e * 5 + 4], j->solver, pidx, sseq, &bt, &c, pd, &cr, &s);
        for (int t = 0; t < H; t++) j->path_idx[(size_t)e * H + t] = pidx[t];
        j->best_t[e] = bt; j->cost[e] = c;
        if (j->path_dist) for (int t = 0; t < H; t++) j->path_dist[(size_t)e * H + t] = pd[t];
        if (j->crash) j->crash[e] = cr;
        nodes += s.nodes; edges += s.edges; cells += s.cells;
    }
    __sync_fetch_and_add(&j->counters[0], nodes);
    __sync_fetch_and_add(&j->counters[1], edges);
    __sync_fetch_and_add(&j->counters[2], cells);
    free(sseq); free(pidx); free(pd);
    free(orc_heap_keep.a); orc_heap_keep.a = NULL; orc_heap_keep.cap = 0;
    free(orc_ws.base); orc_ws.base = NULL; orc_ws.cap = 0; orc_ws.active = 0;
    return NULL;
}

int orc_solve_batch(const orc_params *p, int N, int Kmax, const double *ego, const int32_t *k_count,
                    const double *other_x, const double *other_v, int solver, int nthreads,
                    int32_t *path_idx, int32_t *best_t, double *cost, double *path_dist,
                    int32_t *crash, long long *counters /* [3] nodes, edges, cells */)
{
    if (Kmax > ORC_KMAX) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    int H = orc_num_t(p);
    volatile int next = 0;
    long long local_counters[3] = {0, 0, 0};
    pthread_t th[256];
    orc_job jobs[256];
    for (int i = 0; i < nthreads; i++) {
        orc_job j = {p, N, Kmax, H, solver, ego, k_count, other_x, other_v, path_idx, best_t, cost, path_dist,
                     crash, local_counters, i, nthreads, &next};
        jobs[i] = j;
    }
    if (nthreads == 1) orc_worker(&jobs[0]);
    else {
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, orc_worker, &jobs[i]);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    }
    if (counters) { counters[0] = local_counters[0]; counters[1] = local_counters[1]; counters[2] = local_counters[2]; }
    return 0;
}

/* ---- st.py:274-288 get_path_mean_abs_jerk ---- */
double orc_path_mean_abs_jerk(const double *s_sequence, int n, double v0, double a0, double dt)
{
    double prev_a = a0, prev_v = v0, acc = 0.0;          /* path_cost = 0 (int) + abs(j) */
    for (int i = 1; i < n; i++) {
        double v = (s_sequence[i] - s_sequence[i - 1]) / dt;
        double a = (v - prev_v) / dt;
        double j = (a - prev_a) / dt;
        prev_v = v; prev_a = a;
        acc += fabs(j);
    }
    return acc / (double)(n - 1);
}

double orc_pow3(double dt) { return orc_pow(dt, 3.0); }
int orc_kmax(void) { return ORC_KMAX; }
