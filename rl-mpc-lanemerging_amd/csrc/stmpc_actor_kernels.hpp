// stmpc_actor_kernels.hpp -- the combined controller's policy network on the device, one launch per evaluation:
// the reference's DDPGAgent.get_control (ddpg.py:83-87) for N states = state vector (dqn.py:389-446, k_policy_features' code) ->
// Linear(n_in, h1) -> ReLU -> Linear(h1, h2) -> ReLU -> Linear(h2, 1) -> tanh * scale + mean (the `all` library's fc_deterministic_policy,
// ddpg.py:29-41; 21 -> 400 -> 300 -> 1 for the shipped actors), in float32 like the reference's torch modules.
//
// gfx950 mapping: one workgroup of four wavefronts per AT_TM = 32 states.  Both hidden layers run on the matrix cores with
// v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: exact fmaf chains, at the f32 vector rate) -- 2 row tiles x (h / 16) column tiles, the
// column tiles dealt to the waves, two independent accumulators per wave (the 40-cycle dependent latency of the 32-cycle instruction).
// Activations never leave LDS (X 32x32, H1 32 x h1, H2 32 x h2: 94 KB for the shipped shape); the weights are PRE-PACKED on the host in the
// order the lanes consume them -- tile [column tile][16-wide k block][lane][4]: lane (j, kk) holds W[n0 + j][k0 + 4 kk .. + 3], so a wave's
// B-operand load is one coalesced 1 KB read and feeds four MFMAs (the k permutation inside a block is the same for A and B) -- and stream
// from L2 (486 KB for the 400 x 300 layer, read by all 128 workgroups of a 4096-state batch).  The last layer is a 300-term dot product per
// state: eight lanes per state, a three-step xor butterfly.  Summation orders are fixed, so results do not depend on the launch; they differ from
// torch's GEMM by float32 rounding of another order (tests: 5e-5 absolute on a jerk in [-5, 5]).
#pragma once
#include "stmpc_cc_kernels.hpp"

namespace stmpc {

constexpr int AT_TM = 32;           // states per workgroup (two 16-row tiles)
constexpr int AT_KIN = 32;          // padded input width (n_in <= 32)

struct ActorDev {                   // device pointers + shape of one packed actor
    const float *p0, *b0;           // [h1p / 16][AT_KIN / 16][64][4], [h1p]
    const float *p1, *b1;           // [h2p / 16][h1p / 16][64][4], [h2p]
    const float *w2;                // [h2p]
    float b2, scale, mean;
    int n_in, h1p, h2p;             // h1p, h2p: hidden widths padded to multiples of 16 (pad weights and biases are zero)
};

typedef float at_f4 __attribute__((ext_vector_type(4)));

// one hidden layer: out[AT_TM][np] = relu(in[AT_TM][kp] * W^T + b), in / out in LDS (row strides in_ld / out_ld floats), W packed as above
__device__ __forceinline__ void actor_layer(const float *in, int in_ld, int kp, const float *packed, const float *bias, float *out, int out_ld, int np) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int kblocks = kp >> 4;
    for (int nt = wave; nt < (np >> 4); nt += nwaves) {
        at_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const at_f4 *bp = (const at_f4 *)packed + ((size_t)nt * kblocks) * 64 + lane;
        const float *a0p = in + (size_t)j * in_ld + 4 * kk, *a1p = in + (size_t)(16 + j) * in_ld + 4 * kk;
        for (int kb = 0; kb < kblocks; ++kb) {
            const at_f4 b = bp[(size_t)kb * 64];
            const at_f4 a0 = *(const at_f4 *)(a0p + kb * 16), a1 = *(const at_f4 *)(a1p + kb * 16);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b.w, acc1, 0, 0, 0);
        }
        // C / D: column = lane & 15, row = (lane >> 4) * 4 + register
        const int n = nt * 16 + j;
        const float bv = bias[n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kk * 4 + r;
            const float v0 = acc0[r] + bv, v1 = acc1[r] + bv;
            out[(size_t)row * out_ld + n] = v0 > 0.f ? v0 : 0.f;
            out[(size_t)(16 + row) * out_ld + n] = v1 > 0.f ? v1 : 0.f;
        }
    }
}

// feat_out (may be null): the input vectors as the network saw them, [N][feat_stride]; jerk_out [N] fp64.
__global__ void __launch_bounds__(256) k_actor_eval(FeatCfg f, ActorDev A, int N, int Kmax, const double *__restrict__ ego4, const int *__restrict__ k_count,
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
    // output layer: eight lanes per state
    const int row = tid >> 3, part = tid & 7;
    float s = 0.f;
    for (int n = part; n < A.h2p; n += 8) s = __builtin_fmaf(H2[(size_t)row * h2_ld + n], A.w2[n], s);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (part == 0 && e0 + row < N) jerk_out[e0 + row] = (double)(tanhf(s + A.b2) * A.scale + A.mean);
}

}  // namespace stmpc
