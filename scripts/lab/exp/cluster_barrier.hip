// Micro-experiment: what does a barrier across G workgroups on G different compute units cost on gfx950?  (Would a heavy second-window search be
// worth spreading over several units, with two such barriers per lattice layer?)  Each cluster of G workgroups of 512 threads runs K barriers:
// arrive = one agent-scope atomic add (release) by thread 0 after a workgroup barrier, wait = spin on an acquire load until G * round arrivals.
// Reported: ns per barrier for G = 1, 2, 4, 8, with the clusters' workgroups adjacent in the grid (same XCD every 8th: blockIdx -> XCD is round robin,
// so adjacent workgroups sit on DIFFERENT XCDs) and with a stride of 8 (same XCD).
// build: hipcc --offload-arch=gfx950 -O3 scripts/lab/exp/cluster_barrier.hip -o /tmp/cluster_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(512) k_bar(unsigned *ctr, int G, int stride, int K, unsigned long long *ns, double *sink) {
    // cluster c = workgroups { base + j * stride }: with stride 1 the members are adjacent, with stride 8 they share an XCD
    const int b = blockIdx.x;
    const int cl = stride == 1 ? b / G : (b / (G * stride)) * stride + (b % stride);
    extern __shared__ unsigned char sm[];                        // (forces one workgroup per unit when 150 KB are requested)
    double x = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 1; r <= K; ++r) {
        x = x * 1.0000001 + 1.0;                                 // (a little work between the barriers)
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr + cl * 32, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(ctr + cl * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * r)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { ns[b] = (t1 - t0) * 10ull; sm[0] = 1; }          // wall_clock64: 100 MHz
    if (x == -1.0) sink[0] = x;
}
int main() {
    unsigned *ctr; unsigned long long *ns; double *sink;
    const int grid = 64, K = 2000;
    CHK(hipMalloc(&ctr, 4096 * 4)); CHK(hipMalloc(&ns, grid * 8)); CHK(hipMalloc(&sink, 8));
    CHK(hipFuncSetAttribute((const void *)k_bar, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    for (int stride : {1, 8}) for (int G : {1, 2, 4, 8}) {
        if (stride == 8 && G == 1) continue;
        CHK(hipMemset(ctr, 0, 4096 * 4));
        hipLaunchKernelGGL(k_bar, dim3(grid), dim3(512), 150 * 1024, 0, ctr, G, stride, K, ns, sink);
        CHK(hipDeviceSynchronize());
        unsigned long long h[64]; CHK(hipMemcpy(h, ns, sizeof h, hipMemcpyDeviceToHost));
        double s = 0; for (int i = 0; i < grid; ++i) s += (double)h[i];
        printf("G = %d workgroups per cluster, members %s: %.0f ns per barrier (mean over %d workgroups, %d barriers each)\n", G,
               stride == 1 ? "adjacent (different XCDs)" : "8 apart (same XCD)      ", s / grid / K, grid, K);
    }
    return 0;
}
