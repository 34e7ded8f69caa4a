/* lab.c -- offline node-count laboratory for bounding strategies (TEST/ANALYSIS INFRASTRUCTURE ONLY).
 * Includes the oracle and adds a generic layered pass whose expansion set is configurable.  Nothing in the product
 * includes or links this file; it is driven by oracle/analysis/lab.py on the build host to choose pre-pass designs
 * before they are ported to the HIP kernel. */
#include "../st_oracle.c"

typedef struct { double key; int s; } lab_ent;
static int lab_cmp(const void *a, const void *b) { double x = ((const lab_ent *)a)->key, y = ((const lab_ent *)b)->key; return x < y ? -1 : (x > y ? 1 : 0); }

typedef struct {
    double U;          /* expand only nodes with g <= U */
    int beamK;         /* > 0: at most K nodes per layer (smallest key) */
    double band;       /* > 0: only nodes with key <= min key + band */
    int hmode;         /* 0: key = g;  1: key = g + kh[t] * (v_des - v)^2 */
    double hscale;     /* multiplier of the heuristic */
    int hardsoft;      /* 1: cells closer than min_allowed are blocked */
    int stride;        /* > 1: only target cells with n % stride == 0 (last layer: all) */
    int cap;           /* > 0 with band > 0: the band adapts to keep ~cap nodes per layer (kernel's rule) */
    int twin;          /* > 0: target window cells of a single-wave beam pass: lowest sources whose targets do not fit are dropped */
    int filt;          /* 1: count only candidate edges whose quadratic part can stay within U (exact-pass filter emulation) */
    int divchunk;      /* > 0 with band > 0: besides the band, the LOWEST reached cell of every divchunk-cell block whose key is within divmult x band */
    double divmult;
    int sections;      /* hmode 7: number of position sections with their own band */
    int switch_t;      /* > 0 with band > 0: layers before switch_t expand everything <= U (the exact pass), the band applies from switch_t on */
    int gpu_round;     /* > 0: expand in the kernel's order (descending rounds of this many sources, candidate slots in lock-step) and count the offers that
                          could still matter when they are made (flat3: cost <= the cell's value at the start of the round, flat10: at the start of the slot,
                          flat30: 64-lane groups x slots with at least one such offer; tspan_over: all such groups) */
} lab_cfg;

typedef struct { long long nodes, edges, maxspan, maxlayer, rounds64, flat3, flat10, flat30, tspan, tspan_over; int best_t; double cost; int complete; long long per_layer[64]; double lay_kmin[64], lay_band[64], watch_c[64]; int watch_sel[64]; long long per_layer_span[64]; long long per_layer_off[64]; } lab_out;
static __thread const int *lab_watch = 0;   /* optional: cells of a path to watch (set through lab_set_watch) */
void lab_set_watch(const int *w) { lab_watch = w; }

int lab_pass(const lab_cfg *cfg, const uint8_t *obstacles, const double *s_values, int S, const double *t_values, int H,
             double v0, double a0, const double *distances, double d_w, double v_w, double a_w, double j_w, double v_des,
             double v_max, double a_min, double a_max, double j_min, double j_max, double min_allowed, int *path_idx, lab_out *out)
{
    double delta_s = s_values[1] - s_values[0], delta_t = t_values[1] - t_values[0], start_s = s_values[0];
    double dt3 = orc_pow(delta_t, 3.0);
    double est_prev = start_s - v0 * delta_t, est_second = est_prev - delta_t * (v0 - a0 * delta_t);
    int32_t *previous = (int32_t *)malloc((size_t)H * S * sizeof(int32_t));
    double *cc = (double *)malloc(sizeof(double) * S * 6);
    double *cur_c = cc, *nxt_c = cc + S, *cur_p1 = cc + 2 * S, *nxt_p1 = cc + 3 * S, *cur_p2 = cc + 4 * S, *nxt_p2 = cc + 5 * S;
    lab_ent *ents = (lab_ent *)malloc(sizeof(lab_ent) * S);
    for (int i = 0; i < S; i++) cur_c[i] = INFINITY;
    cur_c[0] = 0.0; cur_p1[0] = est_prev; cur_p2[0] = est_second;
    int lo_w = 0, hi_w = 1, best_t = 0, best_s = 0; double best_cost = 0.0;
    long long nodes = 0, edges = 0, maxspan = 0, maxlayer = 0;
    memset(out, 0, sizeof *out);
    const double U = cfg->U;
    double bandt = cfg->band; long long rounds64 = 0;
    const double lam = sqrt(v_w / (a_w > 0 ? a_w : 1.0)), Jc = sqrt(v_w * a_w) / delta_t;
    for (int t = 0; t < H - 1; t++) {
        int nlo = S, nhi = 0;
        /* select the expansion set */
        int cnt = 0; double kmin = INFINITY;
        const double Trem = (double)(H - 1 - t) * delta_t;
        const double kh = cfg->hmode ? cfg->hscale * Jc * tanh(lam * Trem) : 0.0;
        for (int s = lo_w; s < hi_w; s++) {
            double C = cur_c[s];
            if (!(C < INFINITY) || C > U) continue;
            double key = C;
            if (cfg->hmode && cfg->hmode < 7) { double v = (s_values[s] - cur_p1[s]) / delta_t; double D = v_des - v; key = C + kh * D * D; }
            ents[cnt].key = key; ents[cnt].s = s; cnt++;
            if (key < kmin) kmin = key;
        }
        if (cfg->gpu_round <= 0) { for (int i = 0; i < cnt; i++) { if (ents[i].key <= kmin * 1.003) out->flat3++; if (ents[i].key <= kmin * 1.01) out->flat10++; if (ents[i].key <= kmin * 1.03) out->flat30++; } }
        if (cfg->hmode == 9 && lab_watch) {
            /* tube around a guide path (lab_set_watch): only the cells within `twin` cells of the guide's cell of this layer; the band applies on top */
            int m = 0; const int c0 = lab_watch[t];
            for (int i = 0; i < cnt; i++) { int dd = ents[i].s - c0; if (dd < 0) dd = -dd; if (dd <= cfg->twin && (cfg->band <= 0 || ents[i].key <= kmin + cfg->band)) ents[m++] = ents[i]; }
            cnt = m;
        } else
        if (cfg->band > 0 && cfg->hmode == 3) {
            /* mixed band selection as the kernel can do it in one scan: f within its band OR g within its band; each band is
             * steered towards cap/2 nodes per layer */
            static __thread double bandf, bandg; if (t == 0) { bandf = cfg->band; bandg = cfg->band; }
            double gmin = INFINITY; for (int i = 0; i < cnt; i++) { double g_ = cur_c[ents[i].s]; if (g_ < gmin) gmin = g_; }
            int m = 0, nf = 0, ng = 0;
            for (int i = 0; i < cnt; i++) { int byf = ents[i].key <= kmin + bandf, byg = cur_c[ents[i].s] <= gmin + bandg; nf += byf; ng += byg; if (byf || byg) ents[m++] = ents[i]; }
            cnt = m;
            if (cfg->cap > 0) {
                double half = 0.5 * cfg->cap, f;
                if (nf > 0) { f = half / nf; bandf *= (f < 1.0 ? f : sqrt(f)); bandf = bandf > cfg->band ? cfg->band : (bandf < 0.02 * cfg->band ? 0.02 * cfg->band : bandf); }
                if (ng > 0) { f = half / ng; bandg *= (f < 1.0 ? f : sqrt(f)); bandg = bandg > cfg->band ? cfg->band : (bandg < 0.02 * cfg->band ? 0.02 * cfg->band : bandg); }
            }
        } else
        if (cfg->band > 0 && cfg->switch_t > 0 && t < cfg->switch_t) { /* exact part: everything <= U */ } else
        if (cfg->band > 0 && cfg->hmode == 8) {
            /* fixed band, then every k-th selected node (position order) so that at most cap remain */
            int m = 0; for (int i = 0; i < cnt; i++) if (ents[i].key <= kmin + cfg->band) ents[m++] = ents[i]; cnt = m;
            if (cfg->cap > 0 && cnt > cfg->cap) { int k = (cnt + cfg->cap - 1) / cfg->cap; m = 0; for (int i = 0; i < cnt; i += k) ents[m++] = ents[i]; cnt = m; }
        } else
        if (cfg->band > 0 && cfg->hmode == 7) {
            /* per-section steering: the live window is cut into `twin` equal position ranges (the kernel: one per wave); each keeps its own band
             * (relative to the layer's global minimum) and steers it towards cap / twin nodes */
            static __thread double bq[16]; const int Q = cfg->sections > 0 && cfg->sections <= 16 ? cfg->sections : 4;
            if (t == 0) for (int q = 0; q < Q; q++) bq[q] = cfg->band;
            int cq[16] = {0}; int m = 0; const int span_ = hi_w - lo_w > 0 ? hi_w - lo_w : 1;
            /* sections follow the kernel's chunking: 64-cell chunks from the top, ceil(nch / Q) chunks per section */
            const int top0 = (hi_w + 63) & ~63, nch = (top0 - (lo_w & ~63)) >> 6, cpw = (nch + Q - 1) / Q; (void)span_;
            for (int i = 0; i < cnt; i++) {
                int ch = (top0 - 1 - ents[i].s) >> 6, q = ch / (cpw > 0 ? cpw : 1); if (q >= Q) q = Q - 1;
                if (cur_c[ents[i].s] <= kmin + bq[q]) { cq[q]++; ents[m++] = ents[i]; }
            }
            cnt = m;
            if (cfg->cap > 0) for (int q = 0; q < Q; q++) if (cq[q] > 0) { double f = (double)cfg->cap / (double)Q / (double)cq[q]; bq[q] *= (f < 1.0 ? f : sqrt(f)); bq[q] = bq[q] > cfg->band ? cfg->band : (bq[q] < 0.05 * cfg->band ? 0.05 * cfg->band : bq[q]); }
        } else
        if (cfg->band > 0 && cfg->divchunk < 0) {
            /* wide-and-coarse: every cell within the steered (tight) band, plus every stride-th cell within the FULL band; the tight band
             * is steered towards cap/2 nodes, the stride (power of two, at most -divchunk) towards cap nodes in all */
            static __thread int stride_; if (t == 0) stride_ = 1;
            int m = 0, tight = 0;
            for (int i = 0; i < cnt; i++) {
                int in_t = ents[i].key <= kmin + bandt, in_w = ents[i].key <= kmin + cfg->band && (ents[i].s % stride_) == 0;
                tight += in_t;
                if (in_t || in_w) ents[m++] = ents[i];
            }
            cnt = m;
            if (cfg->cap > 0 && tight > 0) { double f = 0.5 * (double)cfg->cap / (double)tight; bandt = bandt * (f < 1.0 ? f : sqrt(f)); bandt = bandt > cfg->band ? cfg->band : (bandt < 0.02 * cfg->band ? 0.02 * cfg->band : bandt); }
            if (cfg->cap > 0) { if (cnt > cfg->cap && stride_ < -cfg->divchunk) stride_ *= 2; else if (cnt < cfg->cap / 2 && stride_ > 1) stride_ /= 2; }
        } else
        if (cfg->band > 0 && cfg->divchunk > 0) {
            int m = 0, inband = 0, lastblk = -1;
            for (int i = 0; i < cnt; i++) {          /* ents ascending in s */
                int blk = ents[i].s / cfg->divchunk;
                int take = ents[i].key <= kmin + bandt;
                if (take) inband++;
                if (!take && blk != lastblk && ents[i].key <= kmin + cfg->divmult * cfg->band) take = 1;     /* first (lowest) reached cell of its block */
                if (blk != lastblk && (take || ents[i].key <= kmin + cfg->divmult * cfg->band)) lastblk = blk;
                if (take) ents[m++] = ents[i];
            }
            cnt = m;
            if (cfg->cap > 0 && inband > 0) { double f = (double)cfg->cap / (double)inband; bandt = bandt * (f < 1.0 ? f : sqrt(f)); bandt = bandt > cfg->band ? cfg->band : (bandt < 0.05 * cfg->band ? 0.05 * cfg->band : bandt); }
        } else
        if (cfg->band > 0) { int m = 0; for (int i = 0; i < cnt; i++) if (ents[i].key <= kmin + bandt) ents[m++] = ents[i]; cnt = m;
            if (cfg->cap > 0 && cnt > 0) { double f = (double)cfg->cap / (double)cnt; const double up = (cfg->divchunk == 0 && cfg->divmult > 1.0) ? cfg->divmult * cfg->band : cfg->band;   /* divmult doubles as the band's upper limit factor */
                bandt = bandt * (f < 1.0 ? f : sqrt(f)); bandt = bandt > up ? up : (bandt < 0.05 * cfg->band ? 0.05 * cfg->band : bandt); } }
        if (cfg->beamK > 0 && cnt > cfg->beamK && cfg->hmode == 3) {
            /* mixed beam: K/2 smallest by f = g + h, then K/2 smallest by g among the rest */
            qsort(ents, cnt, sizeof(lab_ent), lab_cmp);
            int kf = cfg->beamK / 2, m = kf;
            for (int i = kf; i < cnt; i++) ents[i].key = cur_c[ents[i].s];
            qsort(ents + kf, cnt - kf, sizeof(lab_ent), lab_cmp);
            cnt = cfg->beamK; (void)m;
        } else
        if (cfg->beamK > 0 && cnt > cfg->beamK) { qsort(ents, cnt, sizeof(lab_ent), lab_cmp); cnt = cfg->beamK; }
        /* restore ascending-s order so ties resolve like the reference */
        if (cfg->beamK > 0) { for (int i = 1; i < cnt; i++) { lab_ent e = ents[i]; int j = i - 1; while (j >= 0 && ents[j].s > e.s) { ents[j + 1] = ents[j]; j--; } ents[j + 1] = e; } }
        out->lay_kmin[t] = kmin; out->lay_band[t] = bandt;
        if (lab_watch) { int w = lab_watch[t]; out->watch_c[t] = (w >= 0 && w < S) ? cur_c[w] : -1.0; out->watch_sel[t] = 0; for (int i = 0; i < cnt; i++) if (ents[i].s == w) out->watch_sel[t] = 1; }
        if (cnt == 0) break;
        if (cnt > maxlayer) maxlayer = cnt;
        rounds64 += (cnt + 63) / 64;
        out->per_layer[t] = cnt;
        out->per_layer_span[t] = ents[cnt - 1].s - ents[0].s + 1;   /* extent of the expanded cells (ents ascending in s unless a beam re-sorted them) */
        for (int i = lo_w > 0 ? lo_w : 0; i < S; i++) nxt_c[i] = INFINITY;
        int32_t *prev_n = previous + (size_t)(t + 1) * S;
        const int last = (t + 1 == H - 1);
        int q0 = 0;
        {   /* span of this layer's targets; a single-wave pass with a target window drops the lowest sources that do not fit */
            int tlo = S, thi = 0;
            for (int q = cnt - 1; q >= 0; q--) {
                int s = ents[q].s; double mn, mx; int lo, hi;
                orc_next_s_range(s_values[s], cur_p1[s], cur_p2[s], delta_t, j_min, j_max, a_min, a_max, v_max, &mn, &mx);
                orc_range_indices(start_s, delta_s, mn, mx, &lo, &hi);
                if (hi > S) hi = S;
                if (hi <= lo) continue;
                int nlo2 = lo < tlo ? lo : tlo, nhi2 = hi > thi ? hi : thi;
                if (cfg->twin > 0 && cfg->hmode != 9 && nhi2 - nlo2 > cfg->twin) { q0 = q + 1; out->tspan_over++; break; }
                tlo = nlo2; thi = nhi2;
            }
            if (thi - tlo > out->tspan) out->tspan = thi - tlo;
        }
        int off_lo = S, off_hi = 0;   /* extent of the candidates offered to layer t+1 (after the quadratic filter, blocked cells included) */
        if (cfg->gpu_round > 0) {
            const int R = cfg->gpu_round;
            double *snap = (double *)malloc(sizeof(double) * S);
            for (int r1 = cnt; r1 > q0; r1 -= R) {                 /* sources ents[r0..r1), highest first */
                const int r0 = r1 - R > q0 ? r1 - R : q0;
                memcpy(snap, nxt_c, sizeof(double) * S);
                int maxfan = 0;
                int *lo_ = (int *)malloc(sizeof(int) * (r1 - r0)), *hi_ = (int *)malloc(sizeof(int) * (r1 - r0));
                for (int q = r0; q < r1; q++) {
                    int s = ents[q].s; double mn, mx; int lo, hi;
                    orc_next_s_range(s_values[s], cur_p1[s], cur_p2[s], delta_t, j_min, j_max, a_min, a_max, v_max, &mn, &mx);
                    orc_range_indices(start_s, delta_s, mn, mx, &lo, &hi);
                    if (hi > S) hi = S;
                    if (cfg->filt && U < 1e300) {      /* the kernel narrows the range to the candidates whose quadratic part stays within U */
                        double C = cur_c[s], sv = s_values[s];
                        while (lo < hi) { double sn = s_values[lo]; double v = (sn - sv) / delta_t, a = (sn - 2 * sv + cur_p1[s]) / (delta_t * delta_t), j = (sn - 3 * sv + 3 * cur_p1[s] - cur_p2[s]) / dt3;
                            if (C + v_w * (v - v_des) * (v - v_des) + a_w * a * a + j_w * j * j > U) lo++; else break; }
                        while (hi > lo) { double sn = s_values[hi - 1]; double v = (sn - sv) / delta_t, a = (sn - 2 * sv + cur_p1[s]) / (delta_t * delta_t), j = (sn - 3 * sv + 3 * cur_p1[s] - cur_p2[s]) / dt3;
                            if (C + v_w * (v - v_des) * (v - v_des) + a_w * a * a + j_w * j * j > U) hi--; else break; }
                    }
                    lo_[q - r0] = lo; hi_[q - r0] = hi; if (hi - lo > maxfan) maxfan = hi - lo;
                    nodes++;
                }
                for (int k = 0; k < maxfan; k++) {
                    double *snap2 = (double *)malloc(sizeof(double) * S); memcpy(snap2, nxt_c, sizeof(double) * S);
                    for (int g1 = r1; g1 > r0; g1 -= 64) {        /* one wave's 64 sources */
                        int g0 = g1 - 64 > r0 ? g1 - 64 : r0, any = 0, anyoff = 0;
                        for (int q = g1 - 1; q >= g0; q--) {
                            int s = ents[q].s, n = lo_[q - r0] + k;
                            if (n >= hi_[q - r0]) continue;
                            anyoff = 1;
                            size_t nat = (size_t)(t + 1) * S + n;
                            if (obstacles[nat]) continue;
                            double C = cur_c[s], sv = s_values[s];
                            double c = C + orc_cost_with_jerk(s_values[n], sv, cur_p1[s], cur_p2[s], delta_t, dt3, distances[nat], min_allowed, v_w, v_des, a_w, j_w, d_w);
                            edges++;
                            if (c <= snap[n]) out->flat3++;
                            if (c <= snap2[n]) { out->flat10++; any = 1; }
                            if (c < nxt_c[n] || (c == nxt_c[n] && s < prev_n[n])) {
                                nxt_c[n] = c; prev_n[n] = s; nxt_p1[n] = sv; nxt_p2[n] = cur_p1[s];
                                if (n < nlo) nlo = n;
                                if (n + 1 > nhi) nhi = n + 1;
                            }
                        }
                        out->flat30 += any; out->tspan_over += anyoff;
                    }
                    free(snap2);
                }
                free(lo_); free(hi_);
            }
            free(snap);
        } else
        for (int q = q0; q < cnt; q++) {
            int s = ents[q].s; double C = cur_c[s];
            nodes++; if (cfg->switch_t > 0 && t >= cfg->switch_t) out->tspan_over++;   /* (switch mode: nodes of the completion part) */
            double sv = s_values[s], mn, mx; int lo, hi;
            orc_next_s_range(sv, cur_p1[s], cur_p2[s], delta_t, j_min, j_max, a_min, a_max, v_max, &mn, &mx);
            orc_range_indices(start_s, delta_s, mn, mx, &lo, &hi);
            for (int n = lo; n < hi; n++) {
                if (n >= S) break;
                size_t nat = (size_t)(t + 1) * S + n;
                if (obstacles[nat]) continue;
                if (cfg->stride > 1 && !last && (n % cfg->stride)) continue;
                if (cfg->hardsoft && distances[nat] < min_allowed) continue;
                if (cfg->filt && U < 1e300) {
                    double sn = s_values[n];
                    double v = (sn - sv) / delta_t, a = (sn - 2 * sv + cur_p1[s]) / (delta_t * delta_t), j = (sn - 3 * sv + 3 * cur_p1[s] - cur_p2[s]) / dt3;
                    double q_ = v_w * (v - v_des) * (v - v_des) + a_w * a * a + j_w * j * j;
                    if (C + q_ > U) continue;
                }
                if (n < off_lo) off_lo = n;
                if (n + 1 > off_hi) off_hi = n + 1;
                double c = C + orc_cost_with_jerk(s_values[n], sv, cur_p1[s], cur_p2[s], delta_t, dt3, distances[nat], min_allowed, v_w, v_des, a_w, j_w, d_w);
                edges++;
                if (c < nxt_c[n]) {
                    nxt_c[n] = c; prev_n[n] = s; nxt_p1[n] = sv; nxt_p2[n] = cur_p1[s];
                    if (n < nlo) nlo = n;
                    if (n + 1 > nhi) nhi = n + 1;
                }
            }
        }
        if (cfg->gpu_round <= 0 && t + 1 < 64) out->per_layer_off[t + 1] = off_hi > off_lo ? off_hi - off_lo : 0;
        if (nlo >= nhi) break;
        { int span = nhi - ents[0].s; if (span > maxspan) maxspan = span; }
        double bc = INFINITY; int bs = -1;
        for (int n = nlo; n < nhi; n++) if (nxt_c[n] < bc) { bc = nxt_c[n]; bs = n; }
        best_t = t + 1; best_s = bs; best_cost = bc;
        double *tmp;
        tmp = cur_c; cur_c = nxt_c; nxt_c = tmp; tmp = cur_p1; cur_p1 = nxt_p1; nxt_p1 = tmp; tmp = cur_p2; cur_p2 = nxt_p2; nxt_p2 = tmp;
        lo_w = nlo; hi_w = nhi;
    }
    if (path_idx) {
        int bs = best_s;
        for (int t = 0; t < H; t++) path_idx[t] = -1;
        for (int t = best_t; t > 0; t--) { path_idx[t] = bs; bs = previous[(size_t)t * S + bs]; }
        path_idx[0] = bs;
    }
    out->nodes = nodes; out->edges = edges; out->maxspan = maxspan; out->maxlayer = maxlayer; out->rounds64 = rounds64; out->best_t = best_t; out->cost = best_cost;
    out->complete = (best_t == H - 1) && !(best_cost > U);   /* the kernel's exact pass only accepts terminals within the bound */
    free(previous); free(cc); free(ents);
    return 0;
}
