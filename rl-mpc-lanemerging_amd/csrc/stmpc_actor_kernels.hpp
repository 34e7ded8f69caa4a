// stmpc_actor_kernels.hpp -- the combined controller's policy network on the device, one launch per evaluation:
// the reference's DDPGAgent.get_control (ddpg.py:83-87) for N states = state vector (dqn.py:389-446, k_policy_features' code) ->
// Linear(n_in, h1) -> ReLU -> Linear(h1, h2) -> ReLU -> Linear(h2, 1) -> tanh * scale + mean (the `all` library's fc_deterministic_policy,
// ddpg.py:29-41; 21 -> 400 -> 300 -> 1 for the shipped actors), in float32 like the reference's torch modules.
//
// gfx950 mapping: one workgroup of eight wavefronts per AT_TM = 16 states (4096 states = one workgroup per compute unit).  Both hidden layers
// run on the matrix cores with v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact fmaf chains, at the f32 vector rate): the 16-column
// tiles of a layer are dealt to the waves, two waves per SIMD cover the instruction's 40-cycle dependent latency.  Activations never leave LDS
// (X 16x32, H1 16 x h1, H2 16 x h2: 48 KB for the shipped shape); the weights are PRE-PACKED on the host in the order the lanes consume
// them -- [column tile][16-wide k block][lane][4]: lane (j, kk) holds W[n0 + j][k0 + 4 kk .. + 3], so a wave's B-operand load is one
// coalesced 1 KB read that feeds four MFMAs (the k permutation inside a block is the same for A and B), five such loads in flight per
// wave -- and stream from L2 (486 KB for the 400 x 300 layer).  The last layer is a 300-term dot product per state: 32 lanes per state, a
// five-step xor butterfly.  Summation orders are fixed, so results do not depend on the launch; they differ from torch's GEMM by float32
// rounding of another order (tests: 5e-5 absolute on a jerk in [-5, 5]).
#pragma once
#include "stmpc_cc_kernels.hpp"

namespace stmpc {

constexpr int AT_TM = 16;           // states per workgroup (one 16-row tile)
constexpr int AT_KIN = 32;          // padded input width (n_in <= 32)
constexpr int AT_THREADS = 512;

struct ActorDev {                   // device pointers + shape of one packed actor
    const float *p0, *b0;           // [h1p / 16][AT_KIN / 16][64][4], [h1p]
    const float *p1, *b1;           // [h2p / 16][h1p / 16][64][4], [h2p]
    const float *w2;                // [h2p]
    float b2, scale, mean;
    int n_in, h1p, h2p;             // h1p, h2p: hidden widths padded to multiples of 16 (pad weights and biases are zero)
};

typedef float at_f4 __attribute__((ext_vector_type(4)));

// one hidden layer: out[16][np] = relu(in[16][kp] * W^T + b), in / out in LDS (row strides in_ld / out_ld floats), W packed as above
__device__ __forceinline__ void actor_layer(const float *in, int in_ld, int kp, const float *packed, const float *bias, float *out, int out_ld, int np) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int kblocks = kp >> 4;
    const float *ap = in + (size_t)j * in_ld + 4 * kk;
    for (int nt = wave; nt < (np >> 4); nt += nwaves) {
        at_f4 acc = {0.f, 0.f, 0.f, 0.f};
        const at_f4 *bp = (const at_f4 *)packed + ((size_t)nt * kblocks) * 64 + lane;
        int kb = 0;
        for (; kb + 5 <= kblocks; kb += 5) {                        // five coalesced 1 KB weight reads in flight, then twenty MFMAs
            at_f4 b[5], a[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) b[q] = bp[(size_t)(kb + q) * 64];
#pragma unroll
            for (int q = 0; q < 5; ++q) a[q] = *(const at_f4 *)(ap + (kb + q) * 16);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b[q].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b[q].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b[q].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b[q].w, acc, 0, 0, 0);
            }
        }
        for (; kb < kblocks; ++kb) {
            const at_f4 b = bp[(size_t)kb * 64];
            const at_f4 a = *(const at_f4 *)(ap + kb * 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
        }
        // C / D: column = lane & 15, row = (lane >> 4) * 4 + register
        const int n = nt * 16 + j;
        const float bv = bias[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = acc[r] + bv;
            out[(size_t)(kk * 4 + r) * out_ld + n] = v > 0.f ? v : 0.f;
        }
    }
}

// feat_out (may be null): the input vectors as the network saw them, [N][feat_stride]; jerk_out [N] fp64.
__global__ void __launch_bounds__(AT_THREADS) k_actor_eval(FeatCfg f, ActorDev A, int N, int Kmax, const double *__restrict__ ego4, const int *__restrict__ k_count,
                                                           const double *__restrict__ ox, const double *__restrict__ ov, const double *__restrict__ oa,
                                                           const int *__restrict__ live, int *evals, float *feat_out, int feat_stride, double *jerk_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char at_smem[];
    const int h1_ld = A.h1p + 4, h2_ld = A.h2p + 4;                 // (+4 floats: rows start 16 B apart modulo the banks)
    float *X = (float *)at_smem;                                   // [AT_TM][AT_KIN]
    float *H1 = X + AT_TM * AT_KIN;                                // [AT_TM][h1_ld]
    float *H2 = H1 + (size_t)AT_TM * h1_ld;                        // [AT_TM][h2_ld]
    const int tid = threadIdx.x, e0 = blockIdx.x * AT_TM;
    for (int x = tid; x < AT_TM * AT_KIN; x += blockDim.x) X[x] = 0.f;
    __syncthreads();
    if (tid < AT_TM && e0 + tid < N) {
        const int e = e0 + tid;
        dev_policy_features(f, e, Kmax, ego4, k_count, ox, ov, oa, live, evals, X + tid * AT_KIN);
        if (feat_out) for (int q = 0; q < A.n_in; ++q) feat_out[(size_t)e * feat_stride + q] = X[tid * AT_KIN + q];
    }
    __syncthreads();
    actor_layer(X, AT_KIN, AT_KIN, A.p0, A.b0, H1, h1_ld, A.h1p);
    __syncthreads();
    actor_layer(H1, h1_ld, A.h1p, A.p1, A.b1, H2, h2_ld, A.h2p);
    __syncthreads();
    // output layer: 32 lanes per state (AT_THREADS / AT_TM)
    const int row = tid >> 5, part = tid & 31;
    float s = 0.f;
    for (int n = part; n < A.h2p; n += 32) s = __builtin_fmaf(H2[(size_t)row * h2_ld + n], A.w2[n], s);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16);
    if (part == 0 && e0 + row < N) jerk_out[e0 + row] = (double)(tanhf(s + A.b2) * A.scale + A.mean);
}

}  // namespace stmpc
