// Micro-benchmark: cycles per instruction of a lone wavefront on a SIMD for the instruction patterns of k_predict's recurrence.
// build: hipcc --offload-arch=gfx950 -O2 scripts/lab/exp/issue_latency.hip -o variants/issue_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, unsigned long long *cyc, unsigned long long *wall, double a0, double b0, int lanes) {
    if ((int)threadIdx.x >= lanes) return;
    double a = a0 + threadIdx.x, b = b0, c = a0 * 0.5, d = b0 * 0.25, e = 1.5;
    int m = 0;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < N / 16; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) a = __builtin_fma(a, b, c);                                     // dependent fp64 fma chain
            if (MODE == 1) { a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); }     // two independent chains (2 instrs per step)
            if (MODE == 2) { a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); e = __builtin_fma(e, b, c); c = __builtin_fma(c, b, a0); }   // four chains
            if (MODE == 3) a = a + b;                                                      // dependent add chain
            if (MODE == 4) { a = (a < c) ? a + b : a - b; }                                // cmp -> select chain (compiler's choice of form)
            if (MODE == 5) { a = __builtin_fmin(__builtin_fmax(a + b, c), d); }            // add, max, min
            if (MODE == 6) { asm volatile("v_add_f64 %0, %0, %1\n\tv_cmp_lt_f64 vcc, %0, %2\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %3, 0, 1, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(m) :: "vcc"); }
            if (MODE == 7) { asm volatile("v_mov_b32 %0, %0" : "+v"(m)); }                 // dependent 32-bit movs
            if (MODE == 8) { asm volatile("s_nop 0"); }                                    // scalar no-ops
            if (MODE == 9) { asm volatile("v_add_f64 %0, %0, %1\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" : "+v"(a) : "v"(b)); }   // add + 3 scalar no-ops
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a + d + e + c + m;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; wall[blockIdx.x] = w1 - w0; }
}
template <int MODE> void run(const char *name, int instr_per_step, int blocks, int lanes) {
    double *out; unsigned long long *cyc, *wall;
    hipMalloc(&out, blocks * 64 * 8); hipMalloc(&cyc, blocks * 8); hipMalloc(&wall, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, wall, 1.0000001, 0.9999999, lanes);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, out, cyc, wall, 1.0000001, 0.9999999, lanes);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c, w; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
    printf("%-44s blocks %5d lanes %2d: %6.2f counter-ticks/instr, %6.2f ns/instr (wall clock 100 MHz), kernel %.1f us -> %.2f ns/instr\n", name, blocks, lanes,
           (double)c / (N * instr_per_step), (double)w * 10.0 / (N * instr_per_step), ms * 1e3, ms * 1e6 / (N * instr_per_step));
    hipFree(out); hipFree(cyc); hipFree(wall);
}
int main() {
    for (int blocks : {1, 1024, 4096}) {
        run<0>("dependent v_fma_f64", 1, blocks, 64);
        run<1>("2 independent fma chains", 2, blocks, 64);
        run<2>("4 independent fma chains", 4, blocks, 64);
        run<3>("dependent v_add_f64", 1, blocks, 64);
        run<4>("cmp + select (fp64)", 1, blocks, 64);
        run<5>("add, max, min (fp64)", 3, blocks, 64);
        run<6>("add, cmp, s_and, cndmask", 4, blocks, 64);
        run<7>("dependent v_mov_b32", 1, blocks, 64);
        run<8>("s_nop", 1, blocks, 64);
        run<9>("v_add_f64 + 3 s_nop", 4, blocks, 64);
    }
    run<0>("dependent v_fma_f64, 4 lanes", 1, 1024, 4);
    run<2>("4 independent fma chains, 4 lanes", 4, 1024, 4);
    return 0;
}
