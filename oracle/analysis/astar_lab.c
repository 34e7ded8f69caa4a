/* astar_lab.c -- offline laboratory (TEST/ANALYSIS INFRASTRUCTURE ONLY, nothing in the product includes or links it):
 * how many nodes does the bounded exact pass expand when, besides "cost so far <= U", it also requires
 * "cost so far + a LOWER BOUND of the cost still to come <= U"?  The lower bound is the optimal cost-to-go of the
 * obstacle-free problem on the same lattice from the node's own state (cells covered in the last step, and in the step before),
 * by backward dynamic programming over ranges that CONTAIN the reference's (st_cy.pyx:65-93) -- the gap penalty only adds cost,
 * so every completion of a node costs at least that much, and a node whose sum exceeds U cannot lie on a path of cost <= U:
 * dropping it leaves every node of such a path exactly as the reference settles it (same argument as for the plain bound).
 * Driven by oracle/analysis/astar_lab.py. */
#include "../st_oracle.c"

/* F[r][i1][dd + D], r = steps still to go (0..R); i1 = cells of the last step, dd = i1 - cells of the step before.
 * margin: cells by which the feasible range of the next step is widened on each side (>= 1e-6: rounding of the range arithmetic). */
int astar_table(double ds, double dt, int H, double v_w, double a_w, double j_w, double v_des, double v_max, double a_min, double a_max,
                double j_min, double j_max, double margin, int imax, int D, double *F)
{
    const double u = ds / dt;
    const int nd = 2 * D + 1, nS = (imax + 1) * nd, R = H - 1;
    for (int i = 0; i < nS; i++) F[i] = 0.0;
    for (int r = 1; r <= R; ++r) {
        const double *Fp = F + (size_t)(r - 1) * nS;
        double *Fc = F + (size_t)r * nS;
        for (int i1 = 0; i1 <= imax; ++i1) for (int dd = -D; dd <= D; ++dd) {
            const int st = i1 * nd + dd + D, i2 = i1 - dd;
            Fc[st] = 0.0;                                   /* states the search cannot be in: no information */
            if (i2 < 0 || i2 > imax) continue;
            const double v = i1 * u, pv = i2 * u, a = (v - pv) / dt;
            const double lo_a = fmax(a + j_min * dt, a_min), hi_a = fmin(a + j_max * dt, a_max);
            const double lo_v = fmax(v + lo_a * dt, 0.0), hi_v = fmin(v + hi_a * dt, v_max);
            int lo = (int)ceil(lo_v * dt / ds - margin), hi = (int)floor(hi_v * dt / ds + margin);
            lo = lo < 0 ? 0 : lo; hi = hi > imax ? imax : hi;
            double best = INFINITY;
            for (int i0 = lo; i0 <= hi; ++i0) {
                const int d0 = i0 - i1;
                const double vv = i0 * u - v_des, aa = (i0 - i1) * u / dt, jj = (i0 - 2 * i1 + i2) * u / (dt * dt);
                const double nxt = (d0 < -D || d0 > D) ? 0.0 : Fp[(size_t)i0 * nd + d0 + D];
                const double tot = v_w * vv * vv + a_w * aa * aa + j_w * jj * jj + nxt;
                if (tot < best) best = tot;
            }
            Fc[st] = best < INFINITY ? best : 1e300;        /* no feasible next step at all: any completion is impossible */
        }
    }
    return 0;
}

typedef struct { long long nodes, edges, edges_filt, cut_h, reached; int best_t; double cost; int pruned; long long span_sum, layers;
                 int first_cut_layer; long long nodes_before_cut; } astar_out;
/* retry study (oracle/analysis/retry_depth.py): a pass under the bound U and one under U2 > U are the same up to the first layer that holds a node with
 * U < cost <= U2 -- the failed pass dropped it, the retry expands it.  astar_U2 > 0: record that layer and the nodes expanded before it. */
double astar_U2 = 0.0;

/* Bounded layered pass (orc_solve_layered with "expand only nodes <= U"), optionally with the cost-to-go test.
 * use_h: 0 plain bound, 1 + cost-to-go at expansion.  deflate: factor (< 1) on F against the rounding of the lattice. */
int astar_pass(const uint8_t *obstacles, const double *s_values, int S, const double *t_values, int H, double v0, double a0,
               const double *distances, double d_w, double v_w, double a_w, double j_w, double v_des, double v_max, double a_min,
               double a_max, double j_min, double j_max, double min_allowed, double U, int use_h, const double *F, int imax, int D,
               double deflate, int *path_idx, astar_out *out)
{
    double delta_s = s_values[1] - s_values[0], delta_t = t_values[1] - t_values[0], start_s = s_values[0];
    double dt3 = orc_pow(delta_t, 3.0);
    double est_prev = start_s - v0 * delta_t, est_second = est_prev - delta_t * (v0 - a0 * delta_t);
    const int nd = 2 * D + 1, nS = (imax + 1) * nd;
    int32_t *previous = (int32_t *)malloc((size_t)H * S * sizeof(int32_t));
    double *cc = (double *)malloc(sizeof(double) * S * 2);
    int *hp = (int *)malloc(sizeof(int) * S * 4);
    double *cur_c = cc, *nxt_c = cc + S;
    int *cur_p1 = hp, *nxt_p1 = hp + S, *cur_p2 = hp + 2 * S, *nxt_p2 = hp + 3 * S;      /* history as indices; -1 / -2 = the virtual points */
    for (int i = 0; i < S; i++) cur_c[i] = INFINITY;
    cur_c[0] = 0.0; cur_p1[0] = -1; cur_p2[0] = -2;
    int lo_w = 0, hi_w = 1, best_t = 0, best_s = 0; double best_cost = 0.0;
    memset(out, 0, sizeof *out);
    out->first_cut_layer = -1;
    const double Kq = v_w / (delta_t * delta_t) + a_w / (delta_t * delta_t * delta_t * delta_t) + j_w / (dt3 * dt3);
    for (int t = 0; t < H - 1; t++) {
        int nlo = S, nhi = 0;
        for (int i = 0; i < S; i++) nxt_c[i] = INFINITY;
        int32_t *prev_n = previous + (size_t)(t + 1) * S;
        const int r = H - 1 - t;
        out->span_sum += hi_w - lo_w; out->layers++;
        for (int s = lo_w; s < hi_w; s++) {
            double C = cur_c[s];
            if (!(C < INFINITY)) continue;
            out->reached++;
            if (C > U) {
                out->pruned = 1;
                if (astar_U2 > 0.0 && C <= astar_U2 && out->first_cut_layer < 0) { out->first_cut_layer = t; out->nodes_before_cut = out->nodes; }
                continue;
            }
            const double sv = s_values[s];
            const double h1 = cur_p1[s] == -1 ? est_prev : s_values[cur_p1[s]];
            const double h2 = cur_p2[s] == -2 ? est_second : (cur_p2[s] == -1 ? est_prev : s_values[cur_p2[s]]);
            if (use_h && cur_p1[s] >= 0 && cur_p2[s] >= 0) {
                const int i1 = s - cur_p1[s], i2 = cur_p1[s] - cur_p2[s], dd = i1 - i2;
                if (i1 >= 0 && i1 <= imax && dd >= -D && dd <= D) {
                    const double h = F[(size_t)r * nS + (size_t)i1 * nd + dd + D] * deflate;
                    if (C + h > U) { out->pruned = 1; out->cut_h++; continue; }
                }
            }
            out->nodes++;
            double mn, mx; int lo, hi;
            orc_next_s_range(sv, h1, h2, delta_t, j_min, j_max, a_min, a_max, v_max, &mn, &mx);
            orc_range_indices(start_s, delta_s, mn, mx, &lo, &hi);
            for (int n = lo; n < hi; n++) {
                if (n >= S) break;
                size_t nat = (size_t)(t + 1) * S + n;
                if (obstacles[nat]) continue;
                double c = C + orc_cost_with_jerk(s_values[n], sv, h1, h2, delta_t, dt3, distances[nat], min_allowed, v_w, v_des, a_w, j_w, d_w);
                out->edges++;
                {   /* what the kernel's candidate filter would evaluate: quadratic part alone within the bound */
                    const double vv = (s_values[n] - sv) / delta_t - v_des, aa = (s_values[n] - 2 * sv + h1) / (delta_t * delta_t), jj = (s_values[n] - 3 * sv + 3 * h1 - h2) / dt3;
                    const double q = v_w * vv * vv + a_w * aa * aa + j_w * jj * jj;
                    double hh = 0.0;
                    if (use_h && cur_p1[s] >= 0 && r >= 2) {
                        const int i0 = n - s, i1 = s - cur_p1[s], d0 = i0 - i1;
                        if (i0 >= 0 && i0 <= imax && d0 >= -D && d0 <= D) hh = F[(size_t)(r - 1) * nS + (size_t)i0 * nd + d0 + D] * deflate;
                    }
                    if (C + q + (use_h == 2 ? hh : 0.0) <= U * (1 + 1e-9)) out->edges_filt++;
                    (void)Kq;
                }
                if (c < nxt_c[n]) {
                    nxt_c[n] = c; prev_n[n] = s; nxt_p1[n] = s; nxt_p2[n] = cur_p1[s];
                    if (n < nlo) nlo = n;
                    if (n + 1 > nhi) nhi = n + 1;
                }
            }
        }
        if (nlo >= nhi) break;
        double bc = INFINITY; int bs = -1;
        for (int n = nlo; n < nhi; n++) if (nxt_c[n] < bc) { bc = nxt_c[n]; bs = n; }
        best_t = t + 1; best_s = bs; best_cost = bc;
        double *tmp = cur_c; cur_c = nxt_c; nxt_c = tmp;
        int *ti = cur_p1; cur_p1 = nxt_p1; nxt_p1 = ti;
        ti = cur_p2; cur_p2 = nxt_p2; nxt_p2 = ti;
        lo_w = nlo; hi_w = nhi;
    }
    int bs = best_s;
    for (int t = 0; t < H; t++) path_idx[t] = -1;
    for (int t = best_t; t > 0; t--) { path_idx[t] = bs; bs = previous[(size_t)t * S + bs]; }
    path_idx[0] = bs;
    out->best_t = best_t; out->cost = best_cost;
    free(previous); free(cc); free(hp);
    return 0;
}
