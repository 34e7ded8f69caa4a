// Micro-experiment: can a second stream's kernel be held back until all workgroups of a first kernel are resident (hipStreamWaitValue32 on a
// counter the first kernel increments), and does it then start while the first is still running?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void kA(unsigned *sig, unsigned long long *t, unsigned long long spin_ticks) {
    if (threadIdx.x == 0) {
        t[blockIdx.x * 2] = wall_clock64();
        __hip_atomic_fetch_add(sig, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(32);
        t[blockIdx.x * 2 + 1] = wall_clock64();
    }
}
__global__ void kB(unsigned long long *t) { if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64(); }
int main() {
    int can = 0; CHK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0)); printf("CanUseStreamWaitValue %d\n", can);
    unsigned *sig = nullptr; CHK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    CHK(hipMemset(sig, 0, 8));
    const int gridA = 1024, gridB = 1024;
    unsigned long long *tA, *tB; CHK(hipMalloc(&tA, gridA * 16)); CHK(hipMalloc(&tB, gridB * 8));
    hipStream_t s1, s2; CHK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipMemset(sig, 0, 8)); CHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(kA, dim3(gridA), dim3(256), 38 * 1024, s1, sig, tA, 200000ull);        // 2 ms, 4 workgroups per CU by LDS
        CHK(hipStreamWaitValue32(s2, sig, gridA, hipStreamWaitValueGte, 0xffffffffu));
        hipLaunchKernelGGL(kB, dim3(gridB), dim3(256), 38 * 1024, s2, tB);
        CHK(hipDeviceSynchronize());
        std::vector<unsigned long long> a(gridA * 2), b(gridB);
        CHK(hipMemcpy(a.data(), tA, gridA * 16, hipMemcpyDeviceToHost)); CHK(hipMemcpy(b.data(), tB, gridB * 8, hipMemcpyDeviceToHost));
        unsigned long long a0 = ~0ull, a1 = 0, ae0 = ~0ull, ae1 = 0, b0 = ~0ull, b1 = 0;
        for (int i = 0; i < gridA; ++i) { a0 = std::min(a0, a[2 * i]); a1 = std::max(a1, a[2 * i]); ae0 = std::min(ae0, a[2 * i + 1]); ae1 = std::max(ae1, a[2 * i + 1]); }
        for (int i = 0; i < gridB; ++i) { b0 = std::min(b0, b[i]); b1 = std::max(b1, b[i]); }
        printf("rep %d: A starts %.1f..%.1f us, A ends %.1f..%.1f us, B starts %.1f..%.1f us (relative to first A start)\n", rep, 0.0, (a1 - a0) / 100.0, (ae0 - a0) / 100.0, (ae1 - a0) / 100.0,
               ((double)b0 - (double)a0) / 100.0, ((double)b1 - (double)a0) / 100.0);
    }
    return 0;
}
