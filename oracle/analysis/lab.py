"""Offline laboratory for bounding strategies (analysis infrastructure, not product code).
python oracle/analysis/lab.py <n_states> -- prints pre-pass / exact-pass node counts for a set of pre-pass designs."""
import ctypes as C, os, subprocess, sys
import numpy as np
from multiprocessing import Pool
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, synth
from oracle import st_oracle as orc

LIB = os.path.join(HERE, "liblab.so")
def build():
    src = os.path.join(HERE, "lab.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(REPO, "oracle", "st_oracle.c"))):
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fno-builtin-pow", "-fPIC", "-shared", src, "-o", LIB, "-lm", "-lpthread"], check=True)

class Cfg(C.Structure):
    _fields_ = [("U", C.c_double), ("beamK", C.c_int), ("band", C.c_double), ("hmode", C.c_int), ("hscale", C.c_double),
                ("hardsoft", C.c_int), ("stride", C.c_int), ("cap", C.c_int), ("twin", C.c_int), ("filt", C.c_int), ("divchunk", C.c_int), ("divmult", C.c_double), ("sections", C.c_int), ("switch_t", C.c_int), ("gpu_round", C.c_int)]
class Out(C.Structure):
    _fields_ = [("nodes", C.c_longlong), ("edges", C.c_longlong), ("maxspan", C.c_longlong), ("maxlayer", C.c_longlong), ("rounds64", C.c_longlong), ("flat3", C.c_longlong), ("flat10", C.c_longlong), ("flat30", C.c_longlong), ("tspan", C.c_longlong), ("tspan_over", C.c_longlong),
                ("best_t", C.c_int), ("cost", C.c_double), ("complete", C.c_int), ("per_layer", C.c_longlong * 64),
                ("lay_kmin", C.c_double * 64), ("lay_band", C.c_double * 64), ("watch_c", C.c_double * 64), ("watch_sel", C.c_int * 64), ("per_layer_span", C.c_longlong * 64), ("per_layer_off", C.c_longlong * 64)]

def setup(wl="h40a21"):
    pkg.apply_overrides(pkg.REFERENCE_DEFAULT)
    if wl == "h40a21":
        pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
    return _capi.Params.from_settings(pkg.Settings)

INF = 1e300
_g = {}
def _init(wl):
    p = setup(wl)
    _g["p"] = p
    _g["op"] = orc.OrcParams.from_dict(p.as_dict())
    _g["tun"] = orc.tunables_from_params(_g["op"])
    _g["L"] = C.CDLL(LIB)
    _g["states"] = synth.generate_states(4096, k=6, kmax=8, seed=1000)

def run_pass(grid, v0, a0, **kw):
    ob, sv, tv, di = grid
    cfg = Cfg(U=kw.get("U", INF), beamK=kw.get("K", 0), band=kw.get("band", 0.0), hmode=kw.get("hmode", 0), hscale=kw.get("hscale", 1.0),
              hardsoft=kw.get("hs", 0), stride=kw.get("stride", 1), cap=kw.get("cap", 0), twin=kw.get("twin", 0), filt=kw.get("filt", 0), divchunk=kw.get("divchunk", 0), divmult=kw.get("divmult", 1.0), sections=kw.get("sections", 0), switch_t=kw.get("switch_t", 0), gpu_round=kw.get("gpu_round", 0))
    out = Out()
    _g["L"].lab_pass(C.byref(cfg), ob.view(np.uint8).ctypes.data_as(C.POINTER(C.c_uint8)), orc._dp(sv), sv.size, orc._dp(tv), tv.size,
                     C.c_double(v0), C.c_double(a0), orc._dp(di), *[C.c_double(x) for x in _g["tun"]], None, C.byref(out))
    return out

def grid_of(i):
    ego, k, ox, ov = _g["states"]
    st = orc.make_state(ego[i, 0], ego[i, 1], ego[i, 2], ego[i, 3], ox[i, :k[i]], ov[i, :k[i]])
    return orc.build_grid(_g["op"], st, ego[i, 4]), ego[i, 2], ego[i, 3]

def exact_with_retries(grid, v0, a0, U, H):
    xn = xe = 0; u = U; att = 0; span = 0
    while True:
        o = run_pass(grid, v0, a0, U=u, filt=1)
        xn += o.nodes; xe += o.edges; span = max(span, o.maxspan)
        if o.complete or u >= INF: break
        u = INF if att >= 3 else u * (1.02 if att == 0 else (1.08 if att == 1 else 1.3)); att += 1
    return xn, xe, att, span, o.cost

VARIANTS = {}
def variant(name):
    def deco(f): VARIANTS[name] = f; return f
    return deco

def work(args):
    i, names = args
    grid, v0, a0 = grid_of(i)
    H = grid[2].size
    full = run_pass(grid, v0, a0)
    res = {}
    for nm in names:
        pn, pe, pr, U = VARIANTS[nm](grid, v0, a0, H)
        if not full.complete: U = INF
        xn, xe, att, span, cost = exact_with_retries(grid, v0, a0, U, H)
        assert cost == full.cost, (nm, i)
        res[nm] = (pn, pe, xn, xe, U, att, span, pr)
    return i, full.complete, full.cost, full.nodes, res

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    names = sys.argv[2].split(",") if len(sys.argv) > 2 else list(VARIANTS)
    build()
    with Pool(8, initializer=_init, initargs=("h40a21",)) as pool:
        res = pool.map(work, [(i, names) for i in range(n)], chunksize=4)
    succ = np.array([bool(r[1]) for r in res]); cost = np.array([r[2] for r in res])
    print("states %d, complete %d, full-DP nodes %.0f" % (n, succ.sum(), np.mean([r[3] for r in res])))
    print("%-22s %7s %7s %7s %7s %7s %6s %6s %6s  %7s %7s %7s %7s  %s" % ("variant", "pre_n", "pre_r64", "ex_n", "ex_e", "tot_n", "nobnd", "retry", "U<C*", "q05", "q50", "q95", "q99", "span>2048"))
    for nm in names:
        a = np.array([r[4][nm] for r in res], dtype=float)
        ratio = a[succ, 4] / cost[succ]; fin = ratio < 1e200
        qs = np.quantile(ratio[fin], [.05, .5, .95, .99]) if fin.any() else (0, 0, 0, 0)
        print("%-22s %7.0f %7.1f %7.0f %7.0f %7.0f %6d %6d %6d  %7.4f %7.4f %7.4f %7.3f  %d" % (nm, a[:, 0].mean(), a[:, 7].mean(), a[:, 2].mean(), a[:, 3].mean(), (a[:, 0] + a[:, 2]).mean(),
              (~fin).sum(), a[:, 5].sum(), (ratio[fin] < 1.0).sum(), qs[0], qs[1], qs[2], qs[3], (a[:, 6] > 2048).sum()))
    if os.environ.get("LAB_DUMP"):
        np.save(os.environ["LAB_DUMP"], {"names": names, "res": res}, allow_pickle=True)

# ---- pre-pass designs -------------------------------------------------------------------------
# spec: (attempts, margin, combine) -- attempts = list of run_pass keyword dicts; combine "first": stop at the first complete
# attempt; "min": run all, take the cheapest complete one.
BAND = 225.0
SPECS = {
    "perfect": ([dict()], 1.0, "first"),
    "none": ([], 1.0, "first"),
    # the kernel of round 1: band 8x nominal steered to 300 nodes/layer, penalty zone hard; then 4x that band, zone allowed
    "r1": ([dict(band=8 * BAND, cap=300, hs=1), dict(band=32 * BAND, cap=300, hs=0)], 1.0, "first"),
}
def _add(name, attempts, margin=1.0, combine="first"): SPECS[name] = (attempts, margin, combine)
for cap in (48, 64, 96, 128, 192):
    for mg in (1.0, 1.003, 1.006, 1.01):
        _add("fcap%d_m%.3f" % (cap, mg), [dict(band=8 * BAND, cap=cap, hs=1, hmode=1), dict(band=32 * BAND, cap=cap, hs=0, hmode=1)], mg)
        _add("gcap%d_m%.3f" % (cap, mg), [dict(band=8 * BAND, cap=cap, hs=1), dict(band=32 * BAND, cap=cap, hs=0)], mg)
        _add("fg%d_m%.3f" % (cap, mg), [dict(band=8 * BAND, cap=cap, hs=1, hmode=1), dict(band=8 * BAND, cap=cap, hs=1)], mg, "min")
        _add("fK%d_m%.3f" % (cap, mg), [dict(K=cap, hs=1, hmode=1), dict(K=cap, hs=0, hmode=1)], mg)
        _add("fgK%d_m%.3f" % (cap, mg), [dict(K=cap, hs=1, hmode=1), dict(K=cap, hs=1)], mg, "min")

R1A = [dict(band=8 * BAND, cap=300, hs=1), dict(band=32 * BAND, cap=300, hs=0)]
for K in (32, 48, 64, 96, 128):
    for mg in (1.003, 1.006):
        _add("cK%d_m%.3f" % (K, mg), [dict(K=K, hs=1, hmode=1)] + R1A, mg)
        _add("ccap%d_m%.3f" % (K, mg), [dict(band=8 * BAND, cap=K, hs=1, hmode=1)] + R1A, mg)
        _add("cK%dh0.5_m%.3f" % (K, mg), [dict(K=K, hs=1, hmode=1, hscale=0.5)] + R1A, mg)
        _add("cK%dh1.5_m%.3f" % (K, mg), [dict(K=K, hs=1, hmode=1, hscale=1.5)] + R1A, mg)
        _add("cK%dx2_m%.3f" % (K, mg), [dict(K=K, hs=1, hmode=1), dict(K=2 * K, hs=1, hmode=1, hscale=0.5)] + R1A, mg)

for K in (48, 64):
    for tw in (384, 448, 512, 640):
        _add("cK%dw%d_m1.003" % (K, tw), [dict(K=K, hs=1, hmode=1, twin=tw)] + R1A, 1.003)

for K1, K2 in ((64, 128), (128, 256), (256, 256), (128, 128), (192, 256), (64, 256)):
    for mg in (1.002, 1.003, 1.005):
        _add("s%d_%d_m%.3f" % (K1, K2, mg), [dict(K=K1, hs=0, hmode=1), dict(K=K2, hs=1, hmode=1)], mg, "soft")
_add("s128_256_r1_m1.003", [dict(K=128, hs=0, hmode=1), dict(K=256, hs=1, hmode=1), dict(band=8 * BAND, cap=300, hs=1)], 1.003, "soft")

def run_spec(spec, grid, v0, a0, H):
    attempts, margin, combine = spec
    pn = pe = pr = 0; U = INF
    for kw in attempts:
        o = run_pass(grid, v0, a0, **kw)
        pn += o.nodes; pe += o.edges; pr += o.rounds64
        if o.complete:
            U = min(U, o.cost)
            if combine == "first": break
            if combine == "soft" and U < 1e5: break      # a path free of min-distance penalties: accept
    return pn, pe, pr, (U * margin if U < INF else U)
for _nm, _sp in SPECS.items():
    VARIANTS[_nm] = (lambda sp: lambda grid, v0, a0, H: run_spec(sp, grid, v0, a0, H))(_sp)

if __name__ == "__main__":
    main()
