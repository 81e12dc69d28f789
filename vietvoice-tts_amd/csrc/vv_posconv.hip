// K3: convolutional position embedding -- grouped Conv1d (k = 31, 64 channels per group) + Mish,
// token-major activations [rows][D].  Per group this is an implicit GEMM with K = 31 * 64:
//   out[t][co] = bias[co] + sum_kw sum_ci  W[g][kw][co][ci] * in[t + kw - pad][g*64 + ci]
//
// bf16: one workgroup = 256 tokens x 64 output channels of one (sequence, group).  The
// (256 + 30)-token input window is staged ONCE in LDS (128-byte rows, swz128 image) with zero fill
// outside [0, len); every tap re-reads it at a row offset, so HBM sees each activation once.
// The tap's 8 KiB weight tile is staged by LDS-DMA into a double buffer one tap ahead.  MFMA
// v_mfma_f32_16x16x32_bf16 with the transposed product (A = weights, B = tokens) so a lane owns 4
// consecutive channels of a token.
//
// fp32: exact-f32 VALU kernel for the numerics configuration (not the throughput path).
#include "vv_common.h"
#include "vv_kernels.h"

namespace {

constexpr int PC_TOK = 256;

// bf16: one workgroup = 256 tokens x 64 output channels of one (sequence, group), 8 waves x 32 tokens.
// LDS: the (256 + KW - 1)-token input window (128-byte rows, swz128, zero fill outside [0, len)) staged once, plus a
// double-buffered 8 KiB weight tile per tap filled by LDS-DMA one tap ahead -- the tap's weights are read from L2
// once per workgroup instead of once per wave (the first version moved ~9 TB/s of L2->CU traffic that way).
template <typename To>
__global__ __launch_bounds__(512, 2) void posconv_bf16_kernel(const bf16* __restrict__ in, int ldi,
                                                              const bf16* __restrict__ W /*[G][KW][64co][64ci]*/,
                                                              const float* __restrict__ bias, To* __restrict__ out, int ldo,
                                                              const bf16* __restrict__ resid, int ldr, int seq_n,
                                                              const int* __restrict__ seq_len, int B, int KW,
                                                              const int* __restrict__ row_start) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][64 x 128 B] weight taps | (PC_TOK + KW - 1) rows x 128 B
    char* wbuf = smem;
    char* xwin = smem + 2 * 8192;
    const int g = blockIdx.y, seq = blockIdx.z;
    const int t0 = blockIdx.x * PC_TOK;
    const int pad = KW / 2;
    const int len = seq_len ? min(seq_len[seq % B], seq_n) : seq_n;
    // packed rows: the sequence owns rows [row_start[seq], +len) only; tokens beyond len are neither read nor written
    const size_t row0 = row_start ? (size_t)row_start[seq] : (size_t)seq * seq_n;
    const int t_lim = row_start ? len : seq_n;
    if (t0 >= t_lim) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rows = PC_TOK + KW - 1;

    // weight tap kw -> LDS: 8 pieces of 8 rows x 128 B, one per wave; swizzle on the source address
    const int wrow = wave * 8 + (lane >> 3);
    const bf16* wsrc = W + (size_t)g * KW * 4096 + (size_t)wrow * 64 + (((lane & 7) ^ ((wrow >> 1) & 7)) * 8);
    auto stage_w = [&](int kw, int buf) { glds16(wsrc + (size_t)kw * 4096, wbuf + buf * 8192 + wave * 1024); };
    stage_w(0, 0);

    // ---- stage the window (register path: needs zero fill at the sequence edges)
    for (int i = threadIdx.x; i < rows * 8; i += 512) {
        const int r = i >> 3, c = i & 7;
        const int t = t0 + r - pad;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (t >= 0 && t < len) v = *(const uint4*)(in + (row0 + t) * ldi + g * 64 + c * 8);
        *(uint4*)(xwin + swz128(r, c)) = v;
    }

    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, cq = lane >> 4;

    for (int kw = 0; kw < KW; ++kw) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // tap kw landed (and the window, first time); everyone left tap kw-1
        if (kw + 1 < KW) stage_w(kw + 1, (kw + 1) & 1);
        const char* wt = wbuf + (kw & 1) * 8192;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[4], xf[2];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) wf[ni] = *(const bf16x8*)(wt + swz128(ni * 16 + r16, ks * 4 + cq));
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) xf[mi] = *(const bf16x8*)(xwin + swz128(wave * 32 + mi * 16 + r16 + kw, ks * 4 + cq));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
    }

    // D[co_local = cq*4 + j][token_local = r16]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int t = t0 + wave * 32 + mi * 16 + r16;
        if (t >= t_lim) continue;
        const size_t row = row0 + t;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int co = g * 64 + ni * 16 + cq * 4;
            const float4 b = *(const float4*)(bias + co);
            float v0 = act_apply(acc[ni][mi][0] + b.x, VV_ACT_MISH), v1 = act_apply(acc[ni][mi][1] + b.y, VV_ACT_MISH);
            float v2 = act_apply(acc[ni][mi][2] + b.z, VV_ACT_MISH), v3 = act_apply(acc[ni][mi][3] + b.w, VV_ACT_MISH);
            if (resid) {
                const float4 r = load4<bf16>(resid + row * ldr + co);
                v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
            }
            store4<To>(out + row * ldo + co, v0, v1, v2, v3);
        }
    }
}

// fp32 VALU version: block = 64 tokens x 64 channels of a group; thread = 1 channel x 16 tokens.
// Weights in [G][KW][ci][co] order so a wave reads 64 consecutive channels per (kw, ci).
constexpr int PF_TOK = 64;
__global__ __launch_bounds__(256) void posconv_f32_kernel(const float* __restrict__ in, int ldi,
                                                          const float* __restrict__ W /*[G][KW][64ci][64co]*/,
                                                          const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                          const float* __restrict__ resid, int ldr, int seq_n,
                                                          const int* __restrict__ seq_len, int B, int KW,
                                                          const int* __restrict__ row_start) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // (PF_TOK + KW - 1) rows x 64 floats
    float* xs = (float*)smem;
    const int g = blockIdx.y, seq = blockIdx.z;
    const int t0 = blockIdx.x * PF_TOK;
    const int pad = KW / 2;
    const int len = seq_len ? min(seq_len[seq % B], seq_n) : seq_n;
    const size_t row0 = row_start ? (size_t)row_start[seq] : (size_t)seq * seq_n;       // packed rows: see the bf16 kernel
    const int t_lim = row_start ? len : seq_n;
    if (t0 >= t_lim) return;
    const int rows = PF_TOK + KW - 1;
    for (int i = threadIdx.x; i < rows * 16; i += 256) {
        const int r = i >> 4, c = i & 15;
        const int t = t0 + r - pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < len) v = *(const float4*)(in + (row0 + t) * ldi + g * 64 + c * 4);
        *(float4*)(xs + r * 64 + c * 4) = v;
    }
    __syncthreads();
    const int co = threadIdx.x & 63, tg = threadIdx.x >> 6;   // tokens tg*16 .. tg*16+15
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* wp = W + (size_t)g * KW * 4096 + co;
    for (int kw = 0; kw < KW; ++kw) {
        for (int ci = 0; ci < 64; ci += 4) {
            const float w0 = wp[(kw * 64 + ci + 0) * 64], w1 = wp[(kw * 64 + ci + 1) * 64];
            const float w2 = wp[(kw * 64 + ci + 2) * 64], w3 = wp[(kw * 64 + ci + 3) * 64];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float4 x = *(const float4*)(xs + (tg * 16 + i + kw) * 64 + ci);   // wave-uniform address: broadcast
                acc[i] = fmaf(w0, x.x, acc[i]);
                acc[i] = fmaf(w1, x.y, acc[i]);
                acc[i] = fmaf(w2, x.z, acc[i]);
                acc[i] = fmaf(w3, x.w, acc[i]);
            }
        }
    }
    const float b = bias[g * 64 + co];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int t = t0 + tg * 16 + i;
        if (t >= t_lim) break;
        const size_t row = row0 + t;
        float v = act_apply(acc[i] + b, VV_ACT_MISH);
        if (resid) v += resid[row * ldr + g * 64 + co];
        out[row * ldo + g * 64 + co] = v;
    }
}

}  // namespace

int vvk_posconv(const vv_posconv_args* a, hipStream_t st, const char** err) {
    if (a->groups <= 0 || a->n_seq <= 0 || a->seq_n <= 0) { *err = "posconv: empty shape"; return -22; }
    if (a->KW < 1 || a->KW > 63 || !(a->KW & 1)) { *err = "posconv: odd kernel width expected"; return -22; }
    if (a->ld_in < a->groups * 64 || a->ld_out < a->groups * 64) { *err = "posconv: 64 channels per group expected"; return -22; }
    if (a->row_start && !a->seq_len) { *err = "posconv: packed rows need seq_len"; return -22; }
    if (a->dtype == VV_BF16) {
        if ((a->ld_in * 2) % 16 || (uintptr_t)a->in % 16 || (uintptr_t)a->W % 16) { *err = "posconv: alignment"; return -22; }
        dim3 grid((a->seq_n + PC_TOK - 1) / PC_TOK, a->groups, a->n_seq);
        const size_t lds = (size_t)(PC_TOK + a->KW - 1) * 128 + 2 * 8192;
        if (a->out_dtype == VV_BF16)
            posconv_bf16_kernel<bf16><<<grid, 512, lds, st>>>((const bf16*)a->in, a->ld_in, (const bf16*)a->W, a->bias, (bf16*)a->out,
                                                              a->ld_out, (const bf16*)a->resid, a->ld_resid, a->seq_n, a->seq_len, a->B, a->KW, a->row_start);
        else
            posconv_bf16_kernel<float><<<grid, 512, lds, st>>>((const bf16*)a->in, a->ld_in, (const bf16*)a->W, a->bias, (float*)a->out,
                                                               a->ld_out, (const bf16*)a->resid, a->ld_resid, a->seq_n, a->seq_len, a->B, a->KW, a->row_start);
    } else {
        if (a->out_dtype != VV_F32) { *err = "posconv: f32 path writes f32"; return -22; }
        if ((a->ld_in * 4) % 16 || (uintptr_t)a->in % 16) { *err = "posconv: alignment"; return -22; }
        dim3 grid((a->seq_n + PF_TOK - 1) / PF_TOK, a->groups, a->n_seq);
        const size_t lds = (size_t)(PF_TOK + a->KW - 1) * 256;
        posconv_f32_kernel<<<grid, 256, lds, st>>>((const float*)a->in, a->ld_in, (const float*)a->W, a->bias, (float*)a->out, a->ld_out,
                                                   (const float*)a->resid, a->ld_resid, a->seq_n, a->seq_len, a->B, a->KW, a->row_start);
    }
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}
