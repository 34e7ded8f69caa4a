"""Throughput of back-to-back batches when S batches are in flight (one context + stream + output buffers each)."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
import rl_mpc_lanemerging_amd as pkg
from rl_mpc_lanemerging_amd import _capi, synth
pkg.apply_overrides(pkg.REFERENCE_DEFAULT); pkg.apply_overrides(pkg.SYNTHETIC_H40A21)
params = _capi.Params.from_settings(pkg.Settings)
H = _capi.num_t(params); n = 4096; Kmax = 8
dev = torch.device("cuda", 0)
ego, kc, ox, ov = synth.generate_states(n, k=6, kmax=Kmax, seed=1000)
d_in = [torch.as_tensor(a, device=dev) for a in (ego, kc, ox, ov)]
for S in (1, 2, 3):
    ctxs = [_capi.Context(0) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [(torch.empty((n, H), dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float64, device=dev),
             torch.empty((n, H), dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev)) for _ in range(S)]
    def step(i):
        j = i % S
        o = outs[j]
        ctxs[j].solve_batch_device(params, n, Kmax, d_in[0].data_ptr(), d_in[1].data_ptr(), d_in[2].data_ptr(), d_in[3].data_ptr(),
                                   o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), o[4].data_ptr(), streams[j].cuda_stream)
    for i in range(2 * S): step(i)
    torch.cuda.synchronize()
    K = 12
    t0 = time.perf_counter()
    for i in range(K): step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][2], o[2]) for o in outs)
    print("in flight %d: %.3f ms/step, %.0f solves/s, outputs identical across buffers: %s" % (S, dt / K * 1e3, n * K / dt, ok))
