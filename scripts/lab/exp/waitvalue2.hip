// Micro-experiment 2: do workgroups of a second kernel (38 KB LDS, 4 waves) take over slots one by one as the first kernel's workgroups (26 KB LDS, 4 waves,
// 5 per CU) exit at different times?  With and without hipStreamWaitValue32 in front of the second kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void __launch_bounds__(256, 5) kA(unsigned *sig, unsigned long long *t) {
    extern __shared__ unsigned char sm[];
    if (threadIdx.x == 0) {
        sm[0] = 1;
        t[blockIdx.x * 2] = wall_clock64();
        __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64(), dur = 50000ull + (unsigned long long)(blockIdx.x % 64) * 2000ull;     // 0.5 .. 1.76 ms
        while (wall_clock64() - t0 < dur) __builtin_amdgcn_s_sleep(32);
        t[blockIdx.x * 2 + 1] = wall_clock64();
    }
}
__global__ void __launch_bounds__(256, 4) kB(unsigned long long *t) {
    extern __shared__ unsigned char sm[];
    if (threadIdx.x == 0) { sm[0] = 1; t[blockIdx.x] = wall_clock64(); const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < 20000ull) __builtin_amdgcn_s_sleep(32); }
}
int main() {
    unsigned *sig = nullptr; CHK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    const int gridA = 1280, gridB = 1024;
    unsigned long long *tA, *tB; CHK(hipMalloc(&tA, gridA * 16)); CHK(hipMalloc(&tB, gridB * 8));
    int lo = 0, hi = 0; CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 2; ++rep) {
        CHK(hipMemset(sig, 0, 8)); CHK(hipMemset(tB, 0, gridB * 8)); CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(kA, dim3(gridA), dim3(256), 26 * 1024, s1, sig, tA);
        if (mode == 0) CHK(hipStreamWaitValue32(s2, sig, gridA, hipStreamWaitValueGte, 0xffffffffu));
        if (mode == 2) CHK(hipStreamWaitValue32(s2, sig, 1, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(kB, dim3(gridB), dim3(256), 38 * 1024, s2, tB);
        CHK(hipDeviceSynchronize());
        std::vector<unsigned long long> a(gridA * 2), b(gridB);
        CHK(hipMemcpy(a.data(), tA, gridA * 16, hipMemcpyDeviceToHost)); CHK(hipMemcpy(b.data(), tB, gridB * 8, hipMemcpyDeviceToHost));
        unsigned long long a0 = ~0ull; for (int i = 0; i < gridA; ++i) a0 = std::min(a0, a[2 * i]);
        std::vector<double> as, ae, bs;
        for (int i = 0; i < gridA; ++i) { as.push_back((a[2 * i] - a0) / 100.0); ae.push_back((a[2 * i + 1] - a0) / 100.0); }
        for (int i = 0; i < gridB; ++i) bs.push_back(((double)b[i] - (double)a0) / 100.0);
        std::sort(as.begin(), as.end()); std::sort(ae.begin(), ae.end()); std::sort(bs.begin(), bs.end());
        printf("mode %d (%s): A starts ..%.0f us (q50 %.0f), A ends %.0f..%.0f (q50 %.0f); B starts %.0f (q10 %.0f q50 %.0f q90 %.0f) .. %.0f us\n", mode,
               mode == 0 ? "wait for all A resident" : (mode == 1 ? "no wait" : "wait for 1"), as.back(), as[gridA / 2], ae.front(), ae.back(), ae[gridA / 2], bs.front(), bs[gridB / 10], bs[gridB / 2], bs[gridB * 9 / 10], bs.back());
    }
    return 0;
}
