// K6: MFMA GEMM  C[M,N] = A[M,K] * W[N,K]^T  (+ fused epilogues) for gfx950.
//
// Both operands are K-contiguous (activations row-major, weights in torch Linear [out,in] order),
// so both stream into LDS as 128-byte rows with global_load_lds_dwordx4 (LDS-DMA, no VGPR
// staging) into a double-buffered, XOR-swizzled image (swz128).  The product is computed
// TRANSPOSED -- W rows are the MFMA A operand, activation rows the B operand -- so each lane
// ends up with 4 consecutive output features of one token: bias/activation/RoPE-pair/gated
// residual epilogues are lane-local and stores are 8/16-byte vectors.
//
//   bf16:  v_mfma_f32_16x16x32_bf16, BK = 64       fp32:  v_mfma_f32_32x32x2_f32, BK = 32
//   BIG   tile 256(m) x 256(n), 8 waves (2 x 4), wave = 128 m x 64 n, 128 accumulator registers,
//         128 KiB LDS: one K-step is 64 MFMAs per wave (2 waves per SIMD), long enough to cover the
//         LDS-DMA latency of the next tile with a plain double buffer.  Used when M >= 4096.
//   SMALL tile 128 x 128, 4 waves (2 x 2), wave = 64 x 64: latency-sized problems (B = 1, time grid).
//
// Workgroup -> tile map is XCD-aware: the 8 XCDs (blockIdx % 8) each own whole activation panels
// and walk all n-tiles of a panel back-to-back, so a panel is fetched from HBM once into that
// XCD's L2 while the (small) weight matrix stays hot in every L2 / MALL.
#include "vv_common.h"
#include "vv_kernels.h"

namespace {

template <typename T> struct GemmTraits;
template <> struct GemmTraits<bf16> { static constexpr int BK = 64; };
template <> struct GemmTraits<float> { static constexpr int BK = 32; };

enum { MODE_STORE = 0, MODE_QKV_ROPE = 1, MODE_GATE_RES = 2 };

struct EpiArgs {
    const float* bias;
    const float* gate;
    const float* cos_q;
    const float* sin_q;
    const float* cos_k;
    const float* sin_k;
    int act;
    int n_store;
    int seq_n;
    int rope_dim;
};

// fast epilogue activations: v_exp_f32 / v_rcp_f32 forms (about 1 ulp each), no libm calls
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float epi_act(float x, int act) {
    if (act == VV_ACT_GELU_TANH) {      // 0.5 x (1 + tanh(u)) == x * sigmoid(2u)
        const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
        return x * fast_sigmoid(2.0f * u);
    }
    if (act == VV_ACT_SILU) return x * fast_sigmoid(x);
    if (act == VV_ACT_GELU_ERF) return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
    return x;
}

template <int MODE, typename To>
__device__ __forceinline__ void epi_store(const EpiArgs& e, To* C, int ldc, int m, int pos, int n0, float v0, float v1,
                                          float v2, float v3) {
    if constexpr (MODE == MODE_STORE) {
        if (n0 < e.n_store) store4<To>(C + (size_t)m * ldc + n0, v0, v1, v2, v3);
    } else if constexpr (MODE == MODE_QKV_ROPE) {
        if (n0 < 2 * e.rope_dim) {
            const bool is_k = n0 >= e.rope_dim;
            const int d = n0 & 63;
            const float4 c = *(const float4*)((is_k ? e.cos_k : e.cos_q) + (size_t)pos * 64 + d);
            const float4 s = *(const float4*)((is_k ? e.sin_k : e.sin_q) + (size_t)pos * 64 + d);
            const float o0 = v0 * c.x - v1 * s.x, o1 = v1 * c.y + v0 * s.y;
            const float o2 = v2 * c.z - v3 * s.z, o3 = v3 * c.w + v2 * s.w;
            v0 = o0; v1 = o1; v2 = o2; v3 = o3;
        }
        store4<To>(C + (size_t)m * ldc + n0, v0, v1, v2, v3);
    } else {   // MODE_GATE_RES: C is the fp32 residual stream, updated in place
        float* x = (float*)C + (size_t)m * ldc + n0;
        float4 r = *(const float4*)x;
        if (e.gate) {
            const float4 g = *(const float4*)(e.gate + n0);
            r.x += g.x * v0; r.y += g.y * v1; r.z += g.z * v2; r.w += g.w * v3;
        } else {
            r.x += v0; r.y += v1; r.z += v2; r.w += v3;
        }
        *(float4*)x = r;
    }
}

template <typename T, int MODE, typename To, bool BIG>
__global__ __launch_bounds__(BIG ? 512 : 256, 2) void gemm_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W,
                                                                  int ldw, To* __restrict__ C, int ldc, int M, int N, int K,
                                                                  EpiArgs e, int m_tiles, int n_tiles) {
    constexpr int BK = GemmTraits<T>::BK;
    constexpr int BT = BIG ? 256 : 128;            // tile edge (rows of A and rows of W)
    constexpr int TILE_BYTES = BT * 128;
    constexpr int STAGE_BYTES = 2 * TILE_BYTES;
    constexpr int MW = BIG ? 128 : 64;             // tokens per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A tile | W tile)

    const int id = blockIdx.x;
    const int xcd = id & 7, L = id >> 3;
    const int mt = (L / n_tiles) * 8 + xcd;
    const int nt = L % n_tiles;
    if (mt >= m_tiles) return;
    const int bm = mt * BT, bn = nt * BT;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = BIG ? (wave >> 2) : (wave >> 1);
    const int wc = BIG ? (wave & 3) : (wave & 1);

    // ---- staging: a tile is BT/8 LDS-DMA pieces (8 rows x 128 B); every wave issues 4 pieces of each tile
    const char* a_src[4];
    const char* w_src[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = wave * 4 + u;
        const int row = q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int gm = min(bm + row, M - 1);          // clamp: rows >= M are computed but never stored
        a_src[u] = (const char*)A + (size_t)gm * lda * sizeof(T) + c * 16;
        w_src[u] = (const char*)W + (size_t)(bn + row) * ldw * sizeof(T) + c * 16;
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE_BYTES;
        const size_t koff = (size_t)kt * BK * sizeof(T);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = wave * 4 + u;
            glds16(a_src[u] + koff, base + q * 1024);
            glds16(w_src[u] + koff, base + TILE_BYTES + q * 1024);
        }
    };

    const int nk = K / BK;
    stage(0, 0);

    if constexpr (sizeof(T) == 2) {
        // ------------------------------------------------ bf16: 4 x (MW/16) tiles of 16x16x32
        constexpr int MI = MW / 16;
        f32x4 acc[4][MI];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int r16 = lane & 15, cq = lane >> 4;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
            const char* sa = smem + (kt & 1) * STAGE_BYTES;
            const char* sw = sa + TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 wf[4], af[MI];
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[i] = *(const bf16x8*)(sw + swz128(wc * 64 + i * 16 + r16, ks * 4 + cq));
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(sa + swz128(wr * MW + i * 16 + r16, ks * 4 + cq));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
            }
        }
        // ---- epilogue.  D[n_local = cq*4 + j][m_local = r16]; bias / activation hoisted out of the store loop
        if (e.bias) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const float4 b = *(const float4*)(e.bias + bn + wc * 64 + ni * 16 + cq * 4);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) { acc[ni][mi][0] += b.x; acc[ni][mi][1] += b.y; acc[ni][mi][2] += b.z; acc[ni][mi][3] += b.w; }
            }
        }
        if (MODE == MODE_STORE && e.act != VV_ACT_NONE) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[ni][mi][j] = epi_act(acc[ni][mi][j], e.act);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = bm + wr * MW + mi * 16 + r16;
            if (m >= M) continue;
            const int pos = (MODE == MODE_QKV_ROPE) ? (m % e.seq_n) : 0;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
                epi_store<MODE, To>(e, C, ldc, m, pos, bn + wc * 64 + ni * 16 + cq * 4, acc[ni][mi][0], acc[ni][mi][1],
                                    acc[ni][mi][2], acc[ni][mi][3]);
        }
    } else {
        // ------------------------------------------------ fp32: 2 x (MW/32) tiles of 32x32x2 (exact f32 MFMA)
        constexpr int MI = MW / 32;
        f32x16 acc[2][MI];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int r32 = lane & 31, h = lane >> 5;
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
            const char* sa = smem + (kt & 1) * STAGE_BYTES;
            const char* sw = sa + TILE_BYTES;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                // lane half h takes chunk 2kk+h: the k pairing {4(2kk)+j, 4(2kk+1)+j} is the same for both operands
                f32x4 wf[2], af[MI];
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[i] = *(const f32x4*)(sw + swz128(wc * 64 + i * 32 + r32, kk * 2 + h));
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *(const f32x4*)(sa + swz128(wr * MW + i * 32 + r32, kk * 2 + h));
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[ni][j], af[mi][j], acc[ni][mi], 0, 0, 0);
            }
        }
        // D[n_local = (reg&3) + 8(reg>>2) + 4h][m_local = r32]
        if (e.bias) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = *(const float4*)(e.bias + bn + wc * 64 + ni * 32 + g * 8 + h * 4);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        acc[ni][mi][g * 4 + 0] += b.x; acc[ni][mi][g * 4 + 1] += b.y;
                        acc[ni][mi][g * 4 + 2] += b.z; acc[ni][mi][g * 4 + 3] += b.w;
                    }
                }
        }
        if (MODE == MODE_STORE && e.act != VV_ACT_NONE) {
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ni][mi][r] = epi_act(acc[ni][mi][r], e.act);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = bm + wr * MW + mi * 32 + r32;
            if (m >= M) continue;
            const int pos = (MODE == MODE_QKV_ROPE) ? (m % e.seq_n) : 0;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    epi_store<MODE, To>(e, C, ldc, m, pos, bn + wc * 64 + ni * 32 + g * 8 + h * 4, acc[ni][mi][g * 4 + 0],
                                        acc[ni][mi][g * 4 + 1], acc[ni][mi][g * 4 + 2], acc[ni][mi][g * 4 + 3]);
        }
    }
}

template <typename T, int MODE, typename To, bool BIG>
hipError_t launch_t(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const EpiArgs& e,
                    hipStream_t st) {
    constexpr int BT = BIG ? 256 : 128;
    constexpr int LDS = 2 * 2 * BT * 128;
    static bool attr_set = false;
    auto kern = gemm_kernel<T, MODE, To, BIG>;
    if (!attr_set) {
        hipError_t he = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (he != hipSuccess) return he;
        attr_set = true;
    }
    const int m_tiles = (M + BT - 1) / BT, n_tiles = N / BT;
    const int grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
    kern<<<grid, BIG ? 512 : 256, LDS, st>>>((const T*)A, lda, (const T*)W, ldw, (To*)C, ldc, M, N, K, e, m_tiles, n_tiles);
    return hipGetLastError();
}

template <typename T, int MODE, typename To>
hipError_t launch(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const EpiArgs& e,
                  hipStream_t st, int force_tile) {
    const bool big = force_tile == 256 || (force_tile == 0 && M >= 4096 && N % 256 == 0);
    if (big) return launch_t<T, MODE, To, true>(A, lda, W, ldw, C, ldc, M, N, K, e, st);
    return launch_t<T, MODE, To, false>(A, lda, W, ldw, C, ldc, M, N, K, e, st);
}

}  // namespace

// Host launcher.  Returns 0 or a negative errno-style code with a message in err.
int vvk_gemm(const vvk_gemm_args* g, hipStream_t st, const char** err) {
    const int esz = g->dtype == VV_BF16 ? 2 : 4;
    const int BK = g->dtype == VV_BF16 ? 64 : 32;
    if (g->M <= 0 || g->N <= 0 || g->K <= 0) { *err = "gemm: empty shape"; return -22; }
    if (g->N % 128 != 0) { *err = "gemm: N must be a multiple of 128 (pad the weight rows)"; return -22; }
    if (g->K % BK != 0) { *err = "gemm: K must be a multiple of BK (64 bf16 / 32 f32)"; return -22; }
    if (((size_t)g->lda * esz) % 16 || ((size_t)g->ldw * esz) % 16 || ((uintptr_t)g->A % 16) || ((uintptr_t)g->W % 16) ||
        ((uintptr_t)g->C % 16) || (g->ldc % 4)) { *err = "gemm: operands must be 16-byte aligned"; return -22; }
    if (g->lda < g->K || g->ldw < g->K) { *err = "gemm: leading dimension smaller than K"; return -22; }
    if (g->tile != 0 && g->tile != 128 && g->tile != 256) { *err = "gemm: tile must be 0 (auto), 128 or 256"; return -22; }
    if (g->tile == 256 && g->N % 256) { *err = "gemm: the 256 tile needs N % 256 == 0"; return -22; }
    EpiArgs e;
    e.bias = g->bias; e.gate = g->gate; e.cos_q = g->cos_q; e.sin_q = g->sin_q; e.cos_k = g->cos_k; e.sin_k = g->sin_k;
    e.act = g->act; e.n_store = g->n_store > 0 ? g->n_store : g->N; e.seq_n = g->seq_n > 0 ? g->seq_n : 1;
    e.rope_dim = g->rope_dim;
    if (g->mode == MODE_QKV_ROPE && (!g->cos_q || !g->sin_q || !g->cos_k || !g->sin_k || g->rope_dim % 64)) {
        *err = "gemm: rope epilogue needs the four tables and rope_dim % 64 == 0"; return -22;
    }
    if (g->mode == MODE_GATE_RES && g->out_dtype != VV_F32) { *err = "gemm: residual stream is fp32"; return -22; }
    const bool bf = g->dtype == VV_BF16, obf = g->out_dtype == VV_BF16;
    hipError_t he = hipSuccess;
#define GO(T, MODE, To) he = launch<T, MODE, To>(g->A, g->lda, g->W, g->ldw, g->C, g->ldc, g->M, g->N, g->K, e, st, g->tile)
    if (g->mode == MODE_STORE) {
        if (bf && obf) GO(bf16, MODE_STORE, bf16);
        else if (bf) GO(bf16, MODE_STORE, float);
        else if (!obf) GO(float, MODE_STORE, float);
        else { *err = "gemm: f32 operands with bf16 output not built"; return -22; }
    } else if (g->mode == MODE_QKV_ROPE) {
        if (bf && obf) GO(bf16, MODE_QKV_ROPE, bf16);
        else if (!bf && !obf) GO(float, MODE_QKV_ROPE, float);
        else { *err = "gemm: qkv epilogue writes the operand dtype"; return -22; }
    } else if (g->mode == MODE_GATE_RES) {
        if (bf) GO(bf16, MODE_GATE_RES, float); else GO(float, MODE_GATE_RES, float);
    } else { *err = "gemm: unknown epilogue mode"; return -22; }
#undef GO
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}
