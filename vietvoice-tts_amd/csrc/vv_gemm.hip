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
// Round-2 structure of the persistent kernel, all measured in profiles/r02/gemm_notes.md (the rejected variants live in the
// notes and in git history, not here): G1 = counted output stores + async bias / gate prefetch; E2 = both wave groups run their
// epilogue in the same barrier interval; A1 = scalar tile origin for the activation LDS-DMA.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace {

template <typename T> struct GemmTraits;
template <> struct GemmTraits<bf16> { static constexpr int BK = 64; };
template <> struct GemmTraits<float> { static constexpr int BK = 32; };

enum { MODE_STORE = 0, MODE_QKV_ROPE = 1, MODE_GATE_RES = 2, MODE_GATE_STORE = 3 };

struct EpiArgs {
    const float* bias;
    const float* gate;
    const float* cos_q;
    const float* sin_q;
    const float* cos_k;
    const float* sin_k;
    const float* cs_q;       // optional compact tables [pos][64] = (cos, sin) per pair, pair-duplicated tables only
    const float* cs_k;
    int act;
    int n_store;
    int seq_n;
    const int* pos_tab;      // optional: rope position of row m (packed ragged rows); default m % seq_n
    int ks;                  // split-K tail of the persistent gate-store kernel (tail_plan): K parts of the tail tiles, 1 = no tail
    int tail_panel0;         // ks > 1: first 256-row panel of the tail (a multiple of 8)
    char* c_part;            // ks > 1: fp32 [ks][M - 256 * tail_panel0][ldc] gated products of the K parts of the tail rows (C's tail rows stay unwritten)
    unsigned seq_rcp;        // ceil(2^32 / seq_n): row -> position without a table when every sequence has seq_n rows (m * seq_n < 2^32)
    int cs_by_row;           // cs_q / cs_k are [M][64] tables already gathered per ROW (vv_rope_rows): no position lookup in the epilogue
    int rope_dim;
    int rope_lo;             // first roped column: 0, or rope_dim when the q columns are roped by the attention kernel (rope_skip_q)
    float rope_k1, rope_k0;  // computed rope (rope_k1 > 0): revolutions per position of pair i = exp2(-(i * rope_k1 + rope_k0)), k1 = log2(theta) / 32, k0 = log2(2 pi)
#ifdef VV_GEMM_EXP           // diagnostic build only (profiles/r04/gemm_notes.md)
    int n_group;             // persistent kernel: walk the n-tiles in groups of n_group (each XCD finishes all its panels for one group of
                             // weight tiles before the next group: the group's weights stay in that XCD's L2); 0 = all n-tiles of a panel together
    int a_nt;                // persistent kernel: activation LDS-DMA loads carry the non-temporal hint (streamed once per n-group)
#endif
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

// ONE arithmetic for both bf16 kernels (round 4).  A row's result must not depend on which kernel its launch took (M < 4096 takes
// gemm_kernel, larger launches the persistent gemm_pp_kernel): the reference synthesises every unit as an independent B = 1 call
// (core/tts_engine.py:47,121), so an item inside a batch of 32 has to equal the same item alone BIT FOR BIT.  Both kernels contract
// K in the same order with the same MFMA instruction and start their accumulators at the bias; what is left are the epilogue
// formulas, written here once with explicit fma / mul steps so that hipcc's contraction cannot differ between the two contexts.
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
// tanh-GELU / SiLU of two elements: x * sigmoid(w), sigmoid(w) = 1 / (1 + 2^(-w log2 e)), the -log2(e) folded into k1 / k3
__device__ __forceinline__ void act_consts(int act, float& k1, float& k3) {
    k1 = -1.4426950408889634f * (act == VV_ACT_GELU_TANH ? 2.0f * 0.7978845608028654f : 1.0f);
    k3 = -1.4426950408889634f * (act == VV_ACT_GELU_TANH ? 2.0f * 0.7978845608028654f * 0.044715f : 0.0f);
}
__device__ __forceinline__ f32x2_t act_pair(f32x2_t xv, float k1, float k3) {
    const f32x2_t q = xv * xv;
    const f32x2_t t = xv * __builtin_elementwise_fma(q, (f32x2_t){k3, k3}, (f32x2_t){k1, k1});
    const f32x2_t d = (f32x2_t){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)} + 1.0f;
    return xv * (f32x2_t){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
}
// one interleaved rope pair: (a, b) -> (a c - b s, b c + a s)
__device__ __forceinline__ void rope_pair(float& a, float& b, float c, float sn) {
    const float na = __builtin_fmaf(a, c, -(b * sn));
    const float nb = __builtin_fmaf(b, c, a * sn);
    a = na; b = nb;
}
// The same with cos / sin COMPUTED from the position (bf16 model, standard tables: vv_gemm_args.rope_theta): pair i of a head turns by
// pos * theta^(-i/32) rad.  In revolutions: pos * exp2(-(i k1 + k0)), fract, v_cos / v_sin (their argument is in revolutions) -- six
// vector instructions instead of a table load whose in-order wait stands behind the previous pass's stores (the rope epilogue was
// 3-6 x a plain one: profiles/r04/gemm_notes.md).  fp32 angle: ~3e-4 rad off at position 4096 for pair 0, geometrically less for
// the others -- a tenth of the bf16 rounding the roped value gets anyway.  One formula for both kernels.
__device__ __forceinline__ void rope_pair_computed(float& a, float& b, int pos, int pair, float k1, float k0) {
    const float rev = (float)pos * __builtin_amdgcn_exp2f(-__builtin_fmaf((float)pair, k1, k0));
    const float fr = __builtin_amdgcn_fractf(rev);
    rope_pair(a, b, __builtin_amdgcn_cosf(fr), __builtin_amdgcn_sinf(fr));
}

template <int MODE, typename To>
__device__ __forceinline__ void epi_store(const EpiArgs& e, To* C, int ldc, int m, int pos, int n0, float v0, float v1,
                                          float v2, float v3) {
    if constexpr (MODE == MODE_STORE) {
        if (n0 < e.n_store) store4<To>(C + (size_t)m * ldc + n0, v0, v1, v2, v3);
    } else if constexpr (MODE == MODE_GATE_STORE) {      // C = gate * (A W^T + bias): the residual add is fused into the next LayerNorm
        const float4 g = *(const float4*)(e.gate + n0);
        store4<To>(C + (size_t)m * ldc + n0, g.x * v0, g.y * v1, g.z * v2, g.w * v3);
    } else if constexpr (MODE == MODE_QKV_ROPE) {
        if (n0 >= e.rope_lo && n0 < 2 * e.rope_dim) {
            const bool is_k = n0 >= e.rope_dim;
            const int d = n0 & 63;
            if (e.rope_k1 > 0.f) {
                rope_pair_computed(v0, v1, pos, d >> 1, e.rope_k1, e.rope_k0);
                rope_pair_computed(v2, v3, pos, (d >> 1) + 1, e.rope_k1, e.rope_k0);
            } else {
                const float4 c = *(const float4*)((is_k ? e.cos_k : e.cos_q) + (size_t)pos * 64 + d);       // pair-duplicated tables: c.x == c.y
                const float4 s = *(const float4*)((is_k ? e.sin_k : e.sin_q) + (size_t)pos * 64 + d);
                rope_pair(v0, v1, c.x, s.x);
                rope_pair(v2, v3, c.z, s.z);
            }
        }
        store4<To>(C + (size_t)m * ldc + n0, v0, v1, v2, v3);
    } else {   // MODE_GATE_RES: C is the fp32 residual stream, updated in place
        float* x = (float*)C + (size_t)m * ldc + n0;
        float4 r = *(const float4*)x;
        if (e.gate) {
            const float4 g = *(const float4*)(e.gate + n0);
            r.x = __builtin_fmaf(g.x, v0, r.x); r.y = __builtin_fmaf(g.y, v1, r.y); r.z = __builtin_fmaf(g.z, v2, r.z); r.w = __builtin_fmaf(g.w, v3, r.w);
        } else {
            r.x += v0; r.y += v1; r.z += v2; r.w += v3;
        }
        *(float4*)x = r;
    }
}

// LDS-DMA as an asm statement (as in vv_attention.hip): outside hipcc's waitcnt bookkeeping, so a COUNTED s_waitcnt vmcnt(N) can leave a
// younger stage in flight (the builtin form makes hipcc put vmcnt(0) in front of the next LDS read).  Used by the three-stage ring of CFG 4.
typedef __attribute__((ext_vector_type(4))) int gemm_i32x4_t;
__device__ __forceinline__ gemm_i32x4_t gemm_rsrc4(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    gemm_i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void gemm_glds16_asm(gemm_i32x4_t rs, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

template <typename T, int MODE, typename To, int CFG>
__global__ __launch_bounds__(CFG == 2 ? 1024 : (CFG == 1 ? 512 : 256), CFG == 2 ? 4 : ((CFG == 3 || CFG == 4) ? 3 : 2)) void gemm_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W,
                                                                  int ldw, To* __restrict__ C, int ldc, int M, int N, int K,
                                                                  EpiArgs e, int m_tiles, int n_tiles) {
    constexpr int BK = GemmTraits<T>::BK;
    // CFG 0: 128 x 128 tile, 4 waves (64 tokens x 64 features each); 1 / 2: 256 x 256 with 8 / 16 waves; 3 (bf16 only): 64 tokens x 128
    // features, 4 waves of 32 x 64 -- twice the workgroups for launches whose 128-tiles do not fill the chip (single utterances);
    // 4 (bf16 only, round 5): 64 tokens x 64 features, 4 waves of 32 x 32, THREE-stage LDS ring with a counted wait -- for the N = 1024
    // GEMMs of a single utterance's CFG branch, whose 64 x 128 tiles are fewer than the CUs: a workgroup alone on its CU pays the full
    // L2 latency per K-tile in the two-stage loop (0.9 - 1.1 us per K-tile measured), the ring keeps two K-tiles in flight
    constexpr bool BIG = CFG == 1 || CFG == 2;
    constexpr bool RING3 = CFG == 4;                              // (the 64 x 128 tile on the same ring was measured too: FF2 alike, QKV / FF1 10 - 30 % slower)
    constexpr int BT = BIG ? 256 : (RING3 ? 64 : 128);            // rows of W (features) per tile
    constexpr int BTM = (CFG == 3 || RING3) ? 64 : BT;            // rows of A (tokens) per tile
    constexpr int A_BYTES = BTM * 128, W_BYTES = BT * 128;
    constexpr int STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int MW = CFG == 1 ? 128 : ((CFG == 3 || RING3) ? 32 : 64);        // tokens per wave
    constexpr int NW16 = RING3 ? 2 : 4;                           // 16-feature blocks per wave (features per wave = 32 or 64)
    constexpr int NWAVE = CFG == 2 ? 16 : (CFG == 1 ? 8 : 4);
    constexpr int PPA = BTM / 8 / NWAVE, PPW = BT / 8 / NWAVE;       // LDS-DMA pieces per wave per tile (A, W)
    static_assert((CFG != 3 && CFG != 4) || sizeof(T) == 2, "the 64-token tiles exist for bf16 only");
    // (a four-stage ring with counted waits, one workgroup per CU, was measured SLOWER for the single-utterance shapes than this
    // two-stage loop at two workgroups per CU: 97.9 against 94.1 ms of GEMM per utterance, profiles/r03/gemm_notes.md)
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x (A tile | W tile)

    const int id = blockIdx.x;
    const int xcd = id & 7, L = id >> 3;
    const int mt = (L / n_tiles) * 8 + xcd;
    const int nt = L % n_tiles;
    if (mt >= m_tiles) return;
    const int bm = mt * BTM, bn = nt * BT;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = BIG ? (wave >> 2) : (wave >> 1);
    const int wc = BIG ? (wave & 3) : (wave & 1);

    // ---- staging: a tile is BT/8 LDS-DMA pieces (8 rows x 128 B); every wave issues PPW pieces of each tile
    const char* a_src[PPA];
    const char* w_src[PPW];
#pragma unroll
    for (int u = 0; u < PPA; ++u) {
        const int row = (wave * PPA + u) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int gm = min(bm + row, M - 1);          // clamp: rows >= M are computed but never stored
        a_src[u] = (const char*)A + (size_t)gm * lda * sizeof(T) + c * 16;
    }
#pragma unroll
    for (int u = 0; u < PPW; ++u) {
        const int row = (wave * PPW + u) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        w_src[u] = (const char*)W + (size_t)(bn + row) * ldw * sizeof(T) + c * 16;
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE_BYTES;
        const size_t koff = (size_t)kt * BK * sizeof(T);
#pragma unroll
        for (int u = 0; u < PPA; ++u) glds16(a_src[u] + koff, base + (wave * PPA + u) * 1024);
#pragma unroll
        for (int u = 0; u < PPW; ++u) glds16(w_src[u] + koff, base + A_BYTES + (wave * PPW + u) * 1024);
    };

    const int nk = K / BK;
    constexpr int NST = RING3 ? 3 : 2;
    // ---- CFG 4: the same pieces through buffer_load ... lds as asm statements (rows past M read zeros: their offset is past num_records)
    const gemm_i32x4_t rs_a = gemm_rsrc4(A, (unsigned)min((size_t)M * lda * sizeof(T), (size_t)0x7fffffff));
    const gemm_i32x4_t rs_w = gemm_rsrc4(W, (unsigned)min((size_t)N * ldw * sizeof(T), (size_t)0x7fffffff));
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned voff_a[PPA], voff_w[PPW];
    if constexpr (RING3) {
#pragma unroll
        for (int u = 0; u < PPA; ++u) {
            const int row = (wave * PPA + u) * 8 + (lane >> 3);
            voff_a[u] = (unsigned)(bm + row) * (unsigned)lda * 2u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
        }
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
            const int row = (wave * PPW + u) * 8 + (lane >> 3);
            voff_w[u] = (unsigned)(bn + row) * (unsigned)ldw * 2u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) * 16);
        }
    }
    auto stage3 = [&](int kt, int buf) __attribute__((always_inline)) {
        const unsigned dst = lds0 + (unsigned)(buf * STAGE_BYTES), koff = (unsigned)kt * 128u;
#pragma unroll
        for (int u = 0; u < PPA; ++u) gemm_glds16_asm(rs_a, voff_a[u] + koff, dst + (unsigned)((wave * PPA + u) * 1024));
#pragma unroll
        for (int u = 0; u < PPW; ++u) gemm_glds16_asm(rs_w, voff_w[u] + koff, dst + (unsigned)(A_BYTES + (wave * PPW + u) * 1024));
    };
    if constexpr (RING3) { stage3(0, 0); if (nk > 1) stage3(1, 1); }
    else stage(0, 0);
    auto ring_step = [&](int kt) __attribute__((always_inline)) {       // top of K-tile kt: its pieces have landed, the other stage is free
        if constexpr (RING3) {
            // INVARIANT (asm LDS-DMA is outside hipcc's bookkeeping): a wave has issued stages <= kt + 1 here, PPA + PPW pieces each, and
            // nothing else counts in vmcnt inside the loop -- so "all but the youngest PPA + PPW" = stage kt has landed for this wave, and
            // behind the barrier for every wave.  Stage kt + 2 goes into the buffer tile kt - 1 was read from: every wave has passed this
            // barrier only after its last LDS read of tile kt - 1 was consumed by an MFMA.
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPA + PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");          // compiler fence: no LDS read of this K-tile may be scheduled above the barrier (the last two
                                                    // K-tiles issue no stage3 asm behind it to do that job)
            if (kt + 2 < nk) stage3(kt + 2, (kt + 2) % 3);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        }
    };

    if constexpr (sizeof(T) == 2) {
        // ------------------------------------------------ bf16: 4 x (MW/16) tiles of 16x16x32
        constexpr int MI = MW / 16;
        f32x4 acc[NW16][MI];
        constexpr int WF = NW16 * 16;      // features per wave
        const int r16 = lane & 15, cq = lane >> 4;
        // accumulators start at the bias (feature-only), as in the persistent kernel: the same fp32 sum, bit for bit
#pragma unroll
        for (int i = 0; i < NW16; ++i) {
            f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (e.bias) { const float4 t = *(const float4*)(e.bias + bn + wc * WF + i * 16 + cq * 4); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = b4;
        }
        for (int kt = 0; kt < nk; ++kt) {
            ring_step(kt);
            const char* sa = smem + (kt % NST) * STAGE_BYTES;
            const char* sw = sa + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 wf[NW16], af[MI];
#pragma unroll
                for (int i = 0; i < NW16; ++i) wf[i] = *(const bf16x8*)(sw + swz128(wc * WF + i * 16 + r16, ks * 4 + cq));
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(sa + swz128(wr * MW + i * 16 + r16, ks * 4 + cq));
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NW16; ++ni)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
            }
        }
        // ---- epilogue.  D[n_local = cq*4 + j][m_local = r16]; the bias is already in the accumulators
        if (MODE == MODE_STORE && e.act != VV_ACT_NONE) {
            if (e.act == VV_ACT_GELU_ERF) {              // the persistent kernel never takes erf-GELU: no second formula to agree with
#pragma unroll
                for (int ni = 0; ni < NW16; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[ni][mi][j] = epi_act(acc[ni][mi][j], e.act);
            } else {
                float k1, k3;
                act_consts(e.act, k1, k3);
#pragma unroll
                for (int ni = 0; ni < NW16; ++ni)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {
                            const f32x2_t o = act_pair((f32x2_t){acc[ni][mi][j], acc[ni][mi][j + 1]}, k1, k3);
                            acc[ni][mi][j] = o.x; acc[ni][mi][j + 1] = o.y;
                        }
            }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int m = bm + wr * MW + mi * 16 + r16;
            if (m >= M) continue;
            const int pos = (MODE == MODE_QKV_ROPE) ? (e.pos_tab ? e.pos_tab[m] : m % e.seq_n) : 0;
#pragma unroll
            for (int ni = 0; ni < NW16; ++ni)
                epi_store<MODE, To>(e, C, ldc, m, pos, bn + wc * WF + ni * 16 + cq * 4, acc[ni][mi][0], acc[ni][mi][1],
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
            ring_step(kt);
            const char* sa = smem + (kt % NST) * STAGE_BYTES;
            const char* sw = sa + A_BYTES;
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
            const int pos = (MODE == MODE_QKV_ROPE) ? (e.pos_tab ? e.pos_tab[m] : m % e.seq_n) : 0;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    epi_store<MODE, To>(e, C, ldc, m, pos, bn + wc * 64 + ni * 32 + g * 8 + h * 4, acc[ni][mi][g * 4 + 0],
                                        acc[ni][mi][g * 4 + 1], acc[ni][mi][g * 4 + 2], acc[ni][mi][g * 4 + 3]);
        }
    }
}


// =====================================================================================================
// bf16 ping-pong kernel (the throughput path): 256 x 256 tile, BK = 64, 8 waves = 2 groups x 4.
//
// Wave (g, wc) owns tokens g*128..+127 and features wc*64..+63; its K-tile work is four QUADRANT phases
// (m-half, n-half) = (0,0) (0,1) (1,1) (1,0), 16 MFMAs each.  Each phase is two barrier-delimited
// segments: L (LDS fragment reads for the phase + 2 LDS-DMA pieces of a future unit + counted vmcnt)
// and C (the MFMA cluster).  Group 1 executes one extra barrier up front, so it always runs one
// segment behind group 0: while one group's waves are in C, their SIMD partners are in L -- the
// matrix pipe is fed across every barrier.
//
// LDS (128 KiB) = 2 K-tile parities x 4 units of 128 rows x 128 B (swz128), cut by CONSUMPTION ORDER:
//   Wn0 (phase 0)  rows = for each wc: features wc*64 + 0..31      Am0 (phase 0)  rows = for each g: tokens g*128 + 0..63
//   Wn1 (phase 1)                    features wc*64 + 32..63      Am1 (phase 2)                  tokens g*128 + 64..127
// A unit is re-staged as soon as its last reader is a barrier behind, which is 5 phases (6 for Wn0)
// before its next use: in K-tile T phase 0/1/2/3 the block stages Wn1(T+1) / Am1(T+1) / Wn0(T+2) /
// Am0(T+2).  Every wave issues 2 pieces per phase, so "everything staged <= 4 phases ago has landed"
// is s_waitcnt vmcnt(8) -- never 0 in steady state -- placed at the end of L(q), one barrier (two for
// the other group) before the reads of phase q+1.
// =====================================================================================================
#define VV_WAITVM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#ifdef VV_GEMM_STAMP
// Diagnostic build ONLY (tools/gemm_stamp.py; never the shipped library): s_memtime stamps around the load segment, the barriers
// and the MFMA cluster of every phase, and around the epilogue, summed per wave in scalar registers; lane 0 of waves 0 and 4 stores
// the sums to the debug buffer handed in through vv_gemm_args.C_tail.  The stamps' own waits change the timing: read SHARES.
#define VV_STAMP(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VV_ST_DECL unsigned long long st_t0 = 0, st_t1 = 0, st_t2 = 0, st_t3 = 0, st_t4 = 0
#define VV_ST_ACC() do { st_sumL += st_t1 - st_t0; st_sumB += (st_t2 - st_t1) + (st_t4 - st_t3); st_sumC += st_t3 - st_t2; ++st_n; } while (0)
#else
#define VV_STAMP(v) do { } while (0)
#define VV_ST_DECL do { } while (0)
#define VV_ST_ACC() do { } while (0)
#endif
#define VV_PHASE_WAIT(relaxed) do { if constexpr (relaxed) VV_WAITVM(24); else VV_WAITVM(8); } while (0)
#define VV_CLUSTER(mh, nh, ws) do { cluster(mh, nh, ws); VV_STAMP(st_t3); bar(); } while (0)

template <int MODE, typename To>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                                                         To* __restrict__ C, int ldc, int M, int N, int K, EpiArgs e, int m_tiles,
                                                         int n_tiles) {
    constexpr int UNIT = 128 * 128;                       // bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [parity 2][unit 4][UNIT] | 8 x 4 KiB epilogue staging
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = wave >> 2, wc = wave & 3;
    const int r16 = lane & 15, cq = lane >> 4;

    // ---- persistent tile walk, XCD-aware: block b serves XCD label x = b & 7 (blocks b, b+8 share an L2); that
    // XCD owns activation panels x, x+8, ... and its blocks walk the (panel, n-tile) list panel-major, so
    // the panel's re-reads by the other n-tiles hit the same L2.  Placement changes speed only.
    const int x = blockIdx.x & 7, jb = blockIdx.x >> 3, bpx = gridDim.x >> 3;
    // Split-K tail (gate-store mode only, ks > 1): the tile count left a partial last round, so the last row panels
    // (tail_panel0 .., a multiple of 8: the panel -> XCD label rule is unchanged) are walked AFTER the whole-round part as
    // (tile, K part) entries; the ks parts of a tile are consecutive entries, part p contracts K-tiles [p, p + 1) * nk / ks.
    // In the other modes (not linear in the product) ks folds to the constant 1 and the tail walk compiles away.
    const int ks = MODE == MODE_GATE_STORE ? e.ks : 1;
    const int mt_main = ks > 1 ? e.tail_panel0 : m_tiles;
    const int cols = n_tiles * ks;
    const int n_main = (mt_main > x ? (mt_main - x + 7) / 8 : 0) * n_tiles;
    const int n_tail = ks > 1 ? (m_tiles - mt_main > x ? (m_tiles - mt_main - x + 7) / 8 : 0) * cols : 0;
    const int n_my_main = jb < n_main ? (n_main - jb + bpx - 1) / bpx : 0;
    const int n_my = n_my_main + (jb < n_tail ? (n_tail - jb + bpx - 1) / bpx : 0);
    if (n_my == 0) return;
    const int nk_full = K >> 6;                                // >= 2 per entry (host-checked)
    auto entry = [&](int i, int& bm_, int& bn_, int& part_, int& nk_) __attribute__((always_inline)) {
        if (ks == 1 || i < n_my_main) {
            const int E = jb + i * bpx;
#ifdef VV_GEMM_EXP
            if (e.n_group > 0) {                       // n-group-major: (group, panel, n-tile inside the group)
                const int P = n_main / n_tiles, per = e.n_group * P;
                const int gq = E / per, r = E - gq * per, p = r / e.n_group;
                bm_ = (p * 8 + x) * 256; bn_ = (gq * e.n_group + (r - p * e.n_group)) * 256;
            } else
#endif
            { bm_ = ((E / n_tiles) * 8 + x) * 256; bn_ = (E % n_tiles) * 256; }
            part_ = 0; nk_ = nk_full;
        } else {
            const int E = jb + (i - n_my_main) * bpx, c = E % cols;
            bm_ = (mt_main + (E / cols) * 8 + x) * 256; bn_ = (c / ks) * 256; part_ = c % ks; nk_ = nk_full / ks;
        }
    };

    // ---- LDS-DMA sources (buffer_load ... lds): per-lane 32-bit voffset + SGPR soffset (tile origin, K advance).
    // unit types 0 Wn0, 1 Am0, 2 Wn1, 3 Am1; this wave fills pieces 2*wave, 2*wave+1 of every unit.
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)min((size_t)M * lda * 2, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)min((size_t)N * ldw * 2, (size_t)0x7fffffff), 0x00020000);
    // G1: (1) every tile issues exactly 16 output stores per wave -- buffer stores whose out-of-range lanes carry an offset past
    // num_records (dropped by the hardware) instead of an exec-masked global store that the compiler may branch around -- so the
    // first K-tile of EVERY tile may leave the previous tile's stores in flight (vmcnt(24)); 16 dropped stores in the prologue
    // make that true for a block's first tile too.  (2) bias / gate vectors are prefetched one tile ahead, one coalesced dword
    // per lane (lane i = feature wc*64 + i), and handed to the lanes that need them with ds_bpermute: no VGPR-destination
    // vector load sits at the top of a tile or in the epilogue any more (its in-order vmcnt wait drained the 16 stores).
    constexpr bool G1 = sizeof(To) == 2 && MODE != MODE_GATE_RES;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, (int)min((size_t)M * ldc * sizeof(To), (size_t)0x7fffffff), 0x00020000);
    const unsigned st_lane = ((unsigned)(lane >> 3) * (unsigned)ldc + (unsigned)(lane & 7) * 8u) * 2u;     // store: row lane>>3, 16-byte chunk lane&7
    auto lane_get = [&](float v, int src_lane) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
    };
    // a dword load hipcc does not count: no compiler-inserted vmcnt for it (that wait would drain the stores and the LDS-DMA
    // ring); its completion is covered by the explicit counted wait at the end of the epilogue, which also names the
    // destination so that nothing reads or copies it earlier (cdna guide 5.7 item 1, form ii)
    auto load_async = [&](const float* p) __attribute__((always_inline)) {
        float v;
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
        return v;
    };
    unsigned voff_w[2][2];     // [n-half][piece]  relative to the tile's first weight row
    int loc_a[2];              // [piece]          token row of the piece inside the tile (m-half 0); + 64 for m-half 1
    unsigned cbyte[2];
    unsigned voff_a[2][2];     // [m-half][piece]  per-lane byte offset of the piece's token row inside the tile (A1, below)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int ur = (wave * 2 + u) * 8 + (lane >> 3);              // row inside the unit
        cbyte[u] = (unsigned)(((lane & 7) ^ ((ur >> 1) & 7)) * 16);
        loc_a[u] = (ur >> 6) * 128 + (ur & 63);
#pragma unroll
        for (int h = 0; h < 2; ++h) voff_w[h][u] = (unsigned)((ur >> 5) * 64 + h * 32 + (ur & 31)) * (unsigned)ldw * 2u + cbyte[u];
#pragma unroll
        for (int h = 0; h < 2; ++h) voff_a[h][u] = (unsigned)(loc_a[u] + h * 64) * (unsigned)lda * 2u + cbyte[u];
    }
    // stage unit `type` of K-tile Tk (parity par) of the tile whose origin is (bmS, bnS)
    auto stage = [&](auto type_c, int bmS, int bnS, int Tk, int par) __attribute__((always_inline)) {
        constexpr int type = decltype(type_c)::value;
        constexpr int h = type >> 1;
        char* slot = smem + (par * 4 + type) * UNIT + wave * 2048;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (type & 1) {
                // A1: one v_add per piece (tile origin is a scalar) instead of add + clamp + 64-bit multiply in every phase.  Rows past
                // M need no clamp: their offset is past the resource's num_records (the origin is part of the VECTOR offset, the
                // part the hardware range-checks), so the DMA writes zeros; those rows are computed but never stored.
                const unsigned v = voff_a[h][u] + (unsigned)bmS * (unsigned)lda * 2u;
#ifdef VV_GEMM_EXP
                if (e.a_nt) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(slot + u * 1024), 16, (int)v, Tk * 128, 0, 2);
                else
#endif
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(slot + u * 1024), 16, (int)v, Tk * 128, 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(slot + u * 1024), 16, (int)voff_w[h][u], bnS * ldw * 2 + Tk * 128, 0, 0);
            }
        }
    };
    using T0 = std::integral_constant<int, 0>; using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>; using T3 = std::integral_constant<int, 3>;

    // fragment addressing inside a unit: row = base16 + r16, chunk = ks*4 + cq  ->  swz128
    unsigned lane_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) lane_off[ks] = (unsigned)((r16 << 7) | (((ks * 4 + cq) ^ ((r16 >> 1) & 7)) << 4));
    const unsigned w_row0 = (unsigned)(wc * 32) << 7, a_row0 = (unsigned)(g * 64) << 7;

    f32x4 acc[2][2][4][2];     // [m-half][n-half][mi][ni]
    bf16x8 wfr[2][2][2];       // [n-half][ni][ks]
    bf16x8 afr[4][2];          // [mi][ks]   (one m-half at a time)

    auto read_w = [&](int nh, const char* unit) __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wfr[nh][ni][ks] = *(const bf16x8*)(unit + w_row0 + (ni << 11) + lane_off[ks]);
    };
    auto read_a = [&](const char* unit) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) afr[mi][ks] = *(const bf16x8*)(unit + a_row0 + (mi << 11) + lane_off[ks]);
    };
    auto cluster = [&](int mh, int nh, int ws) __attribute__((always_inline)) {       // quadrant (mh, nh); the n-half's W fragments are in register set ws
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mh][nh][mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[ws][ni][ks], afr[mi][ks], acc[mh][nh][mi][ni], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto bar = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
#ifdef VV_GEMM_STAMP
    unsigned long long st_sumL = 0, st_sumB = 0, st_sumC = 0, st_sumE = 0, st_n = 0, st_start = 0, st_end = 0, st_e0 = 0, st_e1 = 0;
    unsigned long long st_sumS = 0, st_sumW = 0, st_s0 = 0, st_s1 = 0, st_w0 = 0, st_rt0 = 0;      // tile set-up (entry, bias into the accumulators), wait in front of the epilogue
    VV_ST_DECL;
#endif
    int nk = nk_full;                                          // K-tiles of the CURRENT entry (the K-tile bodies read it by reference)
    // One K-tile = four phases.  The staging schedule is continuous across tiles: K-tile T of the current tile
    // stages Wn1/Am1 of K-tile T+1 and Wn0/Am0 of K-tile T+2, which roll over into the NEXT tile of this
    // block at the end (its first units land while this tile finishes and runs its epilogue).  Every phase
    // stages, so the counted wait is always vmcnt(8): everything staged more than four phases ago has landed.
    // The last tile of a block "prefetches" itself again (harmless: those slots are free), which keeps the
    // loop free of conditionals; the kernel drains with vmcnt(0) before it exits.
    auto ktile = [&](auto relaxed_c, int T, int par, int bm_c, int bn_c, int bm_n, int bn_n, int kb_c, int kb_n) __attribute__((always_inline)) {
        // relaxed: the 16 epilogue stores of the previous tile are still counted by vmcnt (stores and loads retire in issue
        // order); everything this K-tile reads was staged BEFORE them, so the counted wait may leave them in flight too.
        constexpr bool relaxed = decltype(relaxed_c)::value;
        const char* base = smem + par * (4 * UNIT);
        VV_STAMP(st_t0);
        const bool r1 = T + 1 >= nk, r2 = T + 2 >= nk;          // roll over into the next tile
        const int bm1 = r1 ? bm_n : bm_c, bn1 = r1 ? bn_n : bn_c, t1 = r1 ? kb_n + T + 1 - nk : kb_c + T + 1;      // absolute K-tile indices
        const int bm2 = r2 ? bm_n : bm_c, bn2 = r2 ? bn_n : bn_c, t2 = r2 ? kb_n + T + 2 - nk : kb_c + T + 2;
        // ---- phase 0: quadrant (m0, n0)
        read_w(0, base + 0 * UNIT);
        read_a(base + 1 * UNIT);
        stage(T2{}, bm1, bn1, t1, par ^ 1);
        VV_PHASE_WAIT(relaxed);
        VV_STAMP(st_t1); bar(); VV_STAMP(st_t2); VV_CLUSTER(0, 0, 0); VV_STAMP(st_t4); VV_ST_ACC(); VV_STAMP(st_t0);
        // ---- phase 1: quadrant (m0, n1)
        read_w(1, base + 2 * UNIT);
        stage(T3{}, bm1, bn1, t1, par ^ 1);
        VV_PHASE_WAIT(relaxed);
        VV_STAMP(st_t1); bar(); VV_STAMP(st_t2); VV_CLUSTER(0, 1, 1); VV_STAMP(st_t4); VV_ST_ACC(); VV_STAMP(st_t0);
        // ---- phase 2: quadrant (m1, n1)
        read_a(base + 3 * UNIT);
        stage(T0{}, bm2, bn2, t2, par);
        VV_PHASE_WAIT(relaxed);
        VV_STAMP(st_t1); bar(); VV_STAMP(st_t2); VV_CLUSTER(1, 1, 1); VV_STAMP(st_t4); VV_ST_ACC(); VV_STAMP(st_t0);
        // ---- phase 3: quadrant (m1, n0)   (both W fragment sets are still in registers)
        stage(T1{}, bm2, bn2, t2, par);
        VV_PHASE_WAIT(relaxed);
        VV_STAMP(st_t1); bar(); VV_STAMP(st_t2); VV_CLUSTER(1, 0, 0); VV_STAMP(st_t4); VV_ST_ACC(); VV_STAMP(st_t0);
    };


    auto ktiles = [&](auto relaxed_c, int& Gc, int bm_c, int bn_c, int bm_n, int bn_n, int kb_c, int kb_n) __attribute__((always_inline)) {
        ktile(relaxed_c, 0, Gc & 1, bm_c, bn_c, bm_n, bn_n, kb_c, kb_n);
        ++Gc;
        for (int T = 1; T < nk; ++T, ++Gc) ktile(std::false_type{}, T, Gc & 1, bm_c, bn_c, bm_n, bn_n, kb_c, kb_n);
    };

    // prologue = phases -6..-1 of the staging schedule for this block's first tile
    {
        int bm0, bn0, part0, nk0;
        entry(0, bm0, bn0, part0, nk0);
        const int kb0 = part0 * nk0;
        stage(T0{}, bm0, bn0, kb0, 0); stage(T1{}, bm0, bn0, kb0, 0); stage(T2{}, bm0, bn0, kb0, 0); stage(T3{}, bm0, bn0, kb0, 0);
        stage(T0{}, bm0, bn0, kb0 + 1, 1); stage(T1{}, bm0, bn0, kb0 + 1, 1);
    }
    float bias_nx = 0.f, gate_nx = 0.f;
    if constexpr (G1) {
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        const u32x4 z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < 16; ++i)       // past num_records: dropped by the hardware, but issued and counted (distinct offsets: not mergeable)
            __builtin_amdgcn_raw_buffer_store_b128(z4, rs_c, (int)(0x7ffffff0u - 16u * (unsigned)i), 0, 0);
        int bm0, bn0, part0, nk0;
        entry(0, bm0, bn0, part0, nk0);
        if (e.bias) bias_nx = e.bias[bn0 + wc * 64 + lane];              // compiler-counted: waited for before the loop
        if constexpr (MODE == MODE_GATE_STORE) gate_nx = e.gate[bn0 + wc * 64 + lane];
    }
    VV_WAITVM(8);
    bar();
    if (g == 1) bar();                                         // stagger group 1 by one segment

    int G = 0;                                                 // global K-tile counter (LDS parity)
    bool stores_pending = false;
#ifdef VV_GEMM_STAMP
    VV_STAMP(st_start);
    st_rt0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz: cycles / realtime ticks x 100 MHz = the clock this kernel ran at
#endif
    for (int it = 0; it < n_my; ++it) {
#ifdef VV_GEMM_STAMP
        VV_STAMP(st_s0);
#endif
        int bm, bn, part, bm_n, bn_n, part_n, nk_n;
        entry(it, bm, bn, part, nk);
        const bool last = it + 1 == n_my;
        entry(last ? it : it + 1, bm_n, bn_n, part_n, nk_n);
        const int kb = part * nk, kb_n = part_n * nk_n;
        // Tail entries (ks > 1): EVERY K part of a tail tile, part 0 included, goes to the fp32 partial buffer
        // [ks][M - row0][ldc] and the tail rows of C stay unwritten.  The consumer (vv_layernorm's delta tails) sums the parts
        // in fp32 and rounds the SUM to the operand dtype once -- where a row outside the tail is rounded in this epilogue -- so
        // a tail row differs from a row outside the tail by fp32 summation order only: a row's result does not depend on
        // where it sits in the launch (round 4; the round-2 form stored bf16-rounded parts: 24 LSB of PCM between batchings).
        const bool tail_entry = MODE == MODE_GATE_STORE && ks > 1 && it >= n_my_main;
        // accumulators start at the bias (feature-only), so the epilogue has no bias pass
        const float bias_cur = part == 0 ? bias_nx : 0.f;          // the bias belongs to K part 0 only
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (G1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) b4[j] = lane_get(bias_cur, nh * 32 + ni * 16 + cq * 4 + j);
                } else
                if (e.bias && part == 0) { const float4 t = *(const float4*)(e.bias + bn + wc * 64 + nh * 32 + ni * 16 + cq * 4); b4 = (f32x4){t.x, t.y, t.z, t.w}; }
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[mh][nh][mi][ni] = b4;
            }
        // first K-tile after a full bf16 tile store (every wave issued exactly 16 stores): leave those stores in flight
        // (compiled for the plain / activation store mode only: in the rope and gate modes the second K-tile body costs
        // registers -- 20 spilled VGPRs in the rope epilogue -- and measured neutral to -8 %; FF1 gains 7 %)
#ifdef VV_GEMM_STAMP
        VV_STAMP(st_s1);
        st_sumS += st_s1 - st_s0;
#endif
        if constexpr (G1) {
            ktiles(std::true_type{}, G, bm, bn, bm_n, bn_n, kb, kb_n);                   // 16 stores (real or dropped) always precede a tile
        } else if constexpr (MODE == MODE_STORE) {
            if (stores_pending) ktiles(std::true_type{}, G, bm, bn, bm_n, bn_n, kb, kb_n);
            else ktiles(std::false_type{}, G, bm, bn, bm_n, bn_n, kb, kb_n);
        } else {
            ktiles(std::false_type{}, G, bm, bn, bm_n, bn_n, kb, kb_n);
        }
#ifdef VV_GEMM_STAMP
        VV_STAMP(st_w0);
#endif
        stores_pending = sizeof(To) == 2 && MODE != MODE_GATE_RES && bm + 256 <= M && bn + 256 <= e.n_store;

        // E2: both groups run their epilogue in the SAME barrier interval.  With the plain one-segment stagger, group 0's epilogue
        // overlaps only group 1's last MFMA cluster and group 1's epilogue only group 0's first cluster of the next tile: the two
        // ~3 us epilogues of a tile run back to back with the matrix pipe idle.  Group 0 therefore waits out one interval here
        // (group 1 is in its last cluster), and group 1 re-establishes the stagger with one barrier after its epilogue.
        // (Letting group 0 run its pre-pass and first store pass inside that interval instead of idling was measured: QKV +1.4 %,
        // the rest flat -- profiles/r02/gemm_notes.md.)
        if (g == 0) bar();
#ifdef VV_GEMM_STAMP
        VV_STAMP(st_e0);
        st_sumW += st_e0 - st_w0;
#endif
        // ---- epilogue: lane owns features n0..n0+3 of token m;  D[n_local = cq*4 + j][m_local = r16]
        if (MODE == MODE_STORE && e.act != VV_ACT_NONE) {          // tanh-GELU or SiLU (erf-GELU is routed to the plain kernel)
            // x * sigmoid(2u) == 0.5 x (1 + tanh u): act_pair, two elements per instruction (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: both
            // groups run their epilogues in the same interval, no MFMA stream is issuing beside them); FF1 382 -> 368 us,
            // profiles/r02/gemm_ab_pkact*.txt
            float k1, k3;
            act_consts(e.act, k1, k3);
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                            for (int j = 0; j < 4; j += 2) {
                                const f32x2_t o = act_pair((f32x2_t){acc[mh][nh][mi][ni][j], acc[mh][nh][mi][ni][j + 1]}, k1, k3);
                                acc[mh][nh][mi][ni][j] = o.x; acc[mh][nh][mi][ni][j + 1] = o.y;
                            }
        }
        if constexpr (MODE == MODE_GATE_STORE) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    float4 gt;
                    if constexpr (G1) {
                        gt.x = lane_get(gate_nx, nh * 32 + ni * 16 + cq * 4 + 0); gt.y = lane_get(gate_nx, nh * 32 + ni * 16 + cq * 4 + 1);
                        gt.z = lane_get(gate_nx, nh * 32 + ni * 16 + cq * 4 + 2); gt.w = lane_get(gate_nx, nh * 32 + ni * 16 + cq * 4 + 3);
                    } else gt = *(const float4*)(e.gate + bn + wc * 64 + nh * 32 + ni * 16 + cq * 4);
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) {
                            f32x4& v = acc[mh][nh][mi][ni];
                            v[0] *= gt.x; v[1] *= gt.y; v[2] *= gt.z; v[3] *= gt.w;
                        }
                }
        }
        if (tail_entry) {
            if constexpr (MODE == MODE_GATE_STORE && G1) {
                // fp32 parts straight from the accumulators: a lane holds 4 consecutive features of a token (16 bytes), the 4 cq lanes of
                // a token make one 64-byte segment.  32 stores per wave instead of 16: every counted wait behind them only waits longer.
                const int row0 = mt_main * 256, tail_rows = M - row0;
                const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(e.c_part + (size_t)part * tail_rows * ldc * 4), 0, (int)min((size_t)tail_rows * ldc * 4, (size_t)0x7fffffff), 0x00020000);
                if (e.bias) bias_nx = load_async(e.bias + bn_n + wc * 64 + lane);
                gate_nx = load_async(e.gate + bn_n + wc * 64 + lane);
#pragma unroll
                for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) {
                        const int mrow = bm - row0 + g * 128 + mh * 64 + mi * 16 + r16;
#pragma unroll
                        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                            for (int ni = 0; ni < 2; ++ni) {
                                typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                                const u32x4 vv = __builtin_bit_cast(u32x4, acc[mh][nh][mi][ni]);
                                const unsigned off = ((unsigned)mrow * (unsigned)ldc + (unsigned)(bn + wc * 64 + nh * 32 + ni * 16 + cq * 4)) * 4u;
                                __builtin_amdgcn_raw_buffer_store_b128(vv, rs_f, (int)(mrow < tail_rows ? off : 0x7ffffff0u), 0, 2);
                            }
                    }
                asm volatile("s_waitcnt vmcnt(32)" : "+v"(bias_nx), "+v"(gate_nx)::"memory");
            }
        } else
        if constexpr (sizeof(To) == 2 && MODE != MODE_GATE_RES) {
            // bf16 output: transpose through a wave-private 4 KiB LDS region (32 tokens x 64 features per pass, 16-byte
            // chunk index XOR (row & 7)) so that global stores are whole 128-byte rows.  The unit slots are NOT
            // touched: the next tile's first units are landing there.
            char* stg = smem + 8 * UNIT + wave * 4096;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int mh = ps >> 1;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int mi = (ps & 1) * 2 + k;
                    const int lr = k * 16 + r16;
                    const int m = bm + g * 128 + ps * 32 + lr;
                    // RoPE: one 16-byte (cos, sin, cos, sin) load per 4 outputs from the compact table, all four issued first.
                    // With row-gathered tables (cs_by_row) the address depends on m alone: no position load, no second wait; the
                    // v columns (a third of the tiles) touch neither.
                    float4 cs4[2][2];
                    bool do_rope = false, rope_tile = false;
                    int pos = 0;
                    if constexpr (MODE == MODE_QKV_ROPE) {
                        rope_tile = bn + wc * 64 < 2 * e.rope_dim && bn + wc * 64 >= e.rope_lo;
                        do_rope = rope_tile && e.cs_q != nullptr && !(e.rope_k1 > 0.f);
                        if (rope_tile) {
                            const int mc = min(m, M - 1);
                            if (e.cs_by_row && !(e.rope_k1 > 0.f)) pos = mc;
                            else if (e.pos_tab) pos = e.pos_tab[mc];
                            else {                                     // uniform sequences: m mod seq_n by reciprocal multiply, no load
                                pos = mc - (int)__umulhi((unsigned)mc, e.seq_rcp) * e.seq_n;
                                if (pos >= e.seq_n) pos -= e.seq_n;
                            }
                        }
                        if (do_rope) {
                            const float* tab = (bn + wc * 64 >= e.rope_dim ? e.cs_k : e.cs_q) + (size_t)pos * 64;
#pragma unroll
                            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                                for (int ni = 0; ni < 2; ++ni) cs4[nh][ni] = *(const float4*)(tab + nh * 32 + ni * 16 + cq * 4);
                        }
                    }
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            f32x4 v = acc[mh][nh][mi][ni];
                            const int nl = nh * 32 + ni * 16 + cq * 4;               // feature inside the wave's 64
                            if constexpr (MODE == MODE_QKV_ROPE) {
                                if (do_rope) {
                                    const float4 t = cs4[nh][ni];          // (cos, sin) of the two pairs
                                    float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                                    rope_pair(a0, a1, t.x, t.y);
                                    rope_pair(a2, a3, t.z, t.w);
                                    v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3;
                                } else if (e.rope_k1 > 0.f) {
                                    if (rope_tile) {                    // computed angles: no load at all in this epilogue
                                        float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                                        rope_pair_computed(a0, a1, pos, nl >> 1, e.rope_k1, e.rope_k0);
                                        rope_pair_computed(a2, a3, pos, (nl >> 1) + 1, e.rope_k1, e.rope_k0);
                                        v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3;
                                    }
                                } else {
                                const int n0 = bn + wc * 64 + nl;
                                if (n0 >= e.rope_lo && n0 < 2 * e.rope_dim) {
                                    const bool is_k = n0 >= e.rope_dim;
                                    const int d = n0 & 63;
                                    const float4 c = *(const float4*)((is_k ? e.cos_k : e.cos_q) + (size_t)pos * 64 + d);
                                    const float4 sn = *(const float4*)((is_k ? e.sin_k : e.sin_q) + (size_t)pos * 64 + d);
                                    float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
                                    rope_pair(a0, a1, c.x, sn.x);
                                    rope_pair(a2, a3, c.z, sn.z);
                                    v[0] = a0; v[1] = a1; v[2] = a2; v[3] = a3;
                                }
                                }
                            }
                            const int chunk = (nl >> 3) ^ (lr & 7);
                            const bf16x4 pk = __builtin_convertvector(v, bf16x4);       // 2 x v_cvt_pk_bf16_f32 (element-wise casts cost 5 instructions)
                            *(bf16x4*)(stg + lr * 128 + chunk * 16 + (nl & 4) * 2) = pk;
                        }
                }
                // the same wave reads its region back: LDS ops of one wave execute in order, no barrier needed
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int lr = q * 8 + (lane >> 3);
                    const int ch = lane & 7;
                    const uint4 val = *(const uint4*)(stg + lr * 128 + ((ch ^ (lr & 7)) << 4));
                    const int m = bm + g * 128 + ps * 32 + lr;
                    const int n0 = bn + wc * 64 + ch * 8;
                    if constexpr (G1) {
                        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                        const u32x4 vv = {val.x, val.y, val.z, val.w};
                        // per-lane part of the address once per kernel (st_lane), the tile / pass / row-group part as the SCALAR offset of
                        // the instruction; validity is decided on the vector offset (the scalar one is not range-checked)
                        const int row_first = bm + g * 128 + ps * 32 + q * 8;                  // wave-uniform
                        const bool ok = (lane >> 3) < M - row_first && n0 < e.n_store;
                        const unsigned soff = ((unsigned)row_first * (unsigned)ldc + (unsigned)(bn + wc * 64)) * 2u;
                        __builtin_amdgcn_raw_buffer_store_b128(vv, rs_c, (int)(ok ? st_lane : 0x7ffffff0u), (int)soff, 2);   // nt; dropped when out of range
                    } else
                    if (m < M && n0 < e.n_store) {   // streamed once, read by the next kernel from HBM anyway
                        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
                        const u32x4 vv = {val.x, val.y, val.z, val.w};
                        __builtin_nontemporal_store(vv, (u32x4*)((bf16*)C + (size_t)m * ldc + n0));
                    }
                }
                if constexpr (G1) {
                    // next tile's bias / gate, issued behind the first pass's 4 stores (hipcc drains vmcnt before the first
                    // read-back of the staging region: the prefetch must not sit in front of that wait) and ahead of the other 12
                    if (ps == 0) {
                        if (e.bias) bias_nx = load_async(e.bias + bn_n + wc * 64 + lane);
                        if constexpr (MODE == MODE_GATE_STORE) gate_nx = load_async(e.gate + bn_n + wc * 64 + lane);
                    }
                }
            }
            // the two prefetch loads sit in front of the last 12 stores: after this wait they have landed (those stores stay in
            // flight), and naming the registers keeps every use / copy of them below it
            if constexpr (G1) asm volatile("s_waitcnt vmcnt(12)" : "+v"(bias_nx), "+v"(gate_nx)::"memory");
        } else {
#pragma unroll
            for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int m = bm + g * 128 + mh * 64 + mi * 16 + r16;
                    if (m >= M) continue;
                    const int pos = (MODE == MODE_QKV_ROPE) ? (e.pos_tab ? e.pos_tab[m] : m % e.seq_n) : 0;
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            const f32x4 v = acc[mh][nh][mi][ni];
                            const int n0 = bn + wc * 64 + nh * 32 + ni * 16 + cq * 4;
                            EpiArgs e2 = e; e2.bias = nullptr;                   // bias is already in the accumulators
                            if constexpr (MODE == MODE_GATE_STORE) store4<To>(C + (size_t)m * ldc + n0, v[0], v[1], v[2], v[3]);
                            else epi_store<MODE, To>(e2, C, ldc, m, pos, n0, v[0], v[1], v[2], v[3]);
                        }
                }
        }
#ifdef VV_GEMM_STAMP
        VV_STAMP(st_e1);
        st_sumE += st_e1 - st_e0;
#endif
        if (g == 1) bar();                                     // group 1 falls one segment behind again
    }
#ifdef VV_GEMM_STAMP
    VV_STAMP(st_end);
    if (lane == 0 && wc == 0 && e.c_part) {
        unsigned long long* d = (unsigned long long*)e.c_part + ((size_t)blockIdx.x * 2 + g) * 10;
        d[0] = st_sumL; d[1] = st_sumB; d[2] = st_sumC; d[3] = st_sumE; d[4] = st_n; d[5] = st_end - st_start; d[6] = (unsigned long long)n_my; d[7] = st_sumS + (st_sumW << 32); d[8] = __builtin_amdgcn_s_memrealtime() - st_rt0;
    }
#endif
    if (g == 0) bar();                                         // balance group 1's extra barrier
    VV_WAITVM(0);                                              // the self-prefetch of the last tile must land before the LDS is released
}

// One-time per-(kernel, device) setup: the dynamic-LDS opt-in is a per-device function attribute, and contexts on several
// GPUs (or several threads) may reach a launcher's first call at the same time.
struct KernelSetup {
    std::mutex mu;
    std::atomic<uint64_t> done{0};     // bit d = device d configured
    int n_cu[64] = {};
    hipError_t ensure(const void* kern, int lds_bytes, int* cu_out) {
        int dev = 0;
        hipError_t he = hipGetDevice(&dev);
        if (he != hipSuccess) return he;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!(done.load(std::memory_order_acquire) >> dev & 1)) {
            std::lock_guard<std::mutex> lk(mu);
            if (!(done.load(std::memory_order_relaxed) >> dev & 1)) {
                he = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                if (he != hipSuccess) return he;
                int cu = 0;
                if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8) cu = 256;
                n_cu[dev] = cu / 8 * 8;
                done.fetch_or(1ull << dev, std::memory_order_release);
            }
        }
        if (cu_out) *cu_out = n_cu[dev];
        return hipSuccess;
    }
};

// Split-K tail of the persistent kernel.  One workgroup per CU walks ceil(tiles / n_cu) tiles, so a tile count that is not a
// multiple of n_cu pays a whole round for the remainder (the flagship's 1600 tiles of the two N = 1024 GEMMs: 6.25 rounds of
// work in 7).  The plan cuts the row panels into a main part that fills whole rounds and a tail whose tiles are split `parts`
// ways along K so that tail_tiles * parts entries fill (at most) one more, 1/parts as long, round.  Every part of a tail row
// lands in an fp32 partial buffer [parts][M - row0][ldc]; the consumer sums them and rounds once (vv_ln_args.delta_tail).
// Returns parts = 0 when there is nothing to gain (no remainder, remainder too large, K too short).
// The persistent kernel addresses its operands through buffer resources with a 31-bit num_records: operands of 2 GiB or more
// take the plain-pointer kernels (64-bit addressing) instead.  The split-K tail exists in the persistent kernel only, so the
// planner applies the same condition -- planner and launcher agree for every shape (M = 300,000, K = 4096: no tail).
inline bool pp_fits(int M, int N, int K, int lda, int ldw, int ldc, size_t out_size, int act) {
    return K >= 128 && act != VV_ACT_GELU_ERF && (size_t)M * lda * 2 < ((size_t)1 << 31) && (size_t)N * ldw * 2 < ((size_t)1 << 31) &&
           (size_t)M * ldc * out_size < ((size_t)1 << 31);
}

void tail_plan(int M, int N, int K, int lda, int ldw, int ldc, int n_cu, int* row0, int* parts) {
    *row0 = 0; *parts = 0;
    if (N % 256 || K % 64 || M < 4096 || !pp_fits(M, N, K, lda, ldw, ldc, 2, VV_ACT_NONE)) return;
    const int m_tiles = (M + 255) / 256, n_tiles = N / 256, total = m_tiles * n_tiles;
    const int rounds = total / n_cu;
    if (rounds < 1 || total % n_cu == 0) return;
    // whole rounds must be whole panels, in multiples of the 8 XCD labels (each label walks every 8th panel)
    int main_panels = (int)((long long)rounds * n_cu / n_tiles) / 8 * 8;
    while (main_panels > 0 && ((long long)main_panels * n_tiles) % n_cu) main_panels -= 8;
    if (main_panels <= 0) return;
    const int tail_tiles = (m_tiles - main_panels) * n_tiles, nk = K >> 6;
    for (int ks = 4; ks >= 2; ks >>= 1)
        if (tail_tiles * ks <= n_cu && nk % ks == 0 && nk / ks >= 2) { *row0 = main_panels * 256; *parts = ks; return; }
}

int device_cus() {
    static std::atomic<int> cached{0};
    int cu = cached.load(std::memory_order_relaxed);
    if (cu) return cu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu < 8) cu = 256;
    cu = cu / 8 * 8;
    cached.store(cu, std::memory_order_relaxed);
    return cu;
}

template <int MODE, typename To>
hipError_t launch_pp(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const EpiArgs& e,
                     hipStream_t st) {
    constexpr int LDS = 2 * 4 * 128 * 128 + 8 * 4096;          // 160 KiB: the whole CU
    static KernelSetup setup;
    auto kern = gemm_pp_kernel<MODE, To>;
    int n_cu = 256;
    if (hipError_t he = setup.ensure((const void*)kern, LDS, &n_cu); he != hipSuccess) return he;
#ifdef VV_GEMM_EXP
    if (const char* v = getenv("VV_GEMM_GRID")) { const int gc = atoi(v) / 8 * 8; if (gc >= 8 && gc <= n_cu) n_cu = gc; }   // CU-masked stream probe
#endif
    const int m_tiles = (M + 255) / 256, n_tiles = N / 256;
    const int total = m_tiles * n_tiles;
    const int grid = std::min(n_cu, (total + 7) / 8 * 8);      // one persistent workgroup per CU, a multiple of the 8 XCD labels
    kern<<<grid, 512, LDS, st>>>((const bf16*)A, lda, (const bf16*)W, ldw, (To*)C, ldc, M, N, K, e, m_tiles, n_tiles);
    return hipGetLastError();
}

template <typename T, int MODE, typename To, int CFG>
hipError_t launch_t(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const EpiArgs& e,
                    hipStream_t st) {
    constexpr int BT = (CFG == 1 || CFG == 2) ? 256 : (CFG == 4 ? 64 : 128);
    constexpr int BTM = (CFG == 3 || CFG == 4) ? 64 : BT;
    constexpr int LDS = (CFG == 4 ? 3 : 2) * (BT + BTM) * 128;
    static KernelSetup setup;
    auto kern = gemm_kernel<T, MODE, To, CFG>;
    if (hipError_t he = setup.ensure((const void*)kern, LDS, nullptr); he != hipSuccess) return he;
    const int m_tiles = (M + BTM - 1) / BTM, n_tiles = N / BT;
    const int grid = ((m_tiles + 7) / 8) * 8 * n_tiles;
    kern<<<grid, CFG == 2 ? 1024 : (CFG == 1 ? 512 : 256), LDS, st>>>((const T*)A, lda, (const T*)W, ldw, (To*)C, ldc, M, N, K, e, m_tiles, n_tiles);
    return hipGetLastError();
}

template <typename T, int MODE, typename To>
hipError_t launch(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, const EpiArgs& e,
                  hipStream_t st, int force_tile, int chip_share) {
    bool big = force_tile == 256 || (force_tile == 0 && M >= 4096 && N % 256 == 0);
    if constexpr (sizeof(T) == 2) {
        // auto: below one full round of 256 x 256 tiles the persistent kernel leaves CUs idle for its whole length and the 128 x 128
        // kernel (four times the tiles, two workgroups per CU) is faster -- out-projection / FF2 up to M = 12,800, FF1 up to 6,400 --
        // except for the wide QKV shape, where the persistent kernel wins from M = 4,096 on (profiles/r04/gemm_tile_by_m.txt).  Both
        // kernels produce the same bits (one arithmetic, tests/test_kernels_gpu.py), so the choice is speed only.
        // A launch that shares the chip with another lane's launches (chip_share 2) is priced for its share: alone the 128 x 128 kernel wins
        // by reaching more CUs, beside another stream of kernels those CUs are busy anyway and the persistent kernel's lower cost per
        // flop decides (two lanes of 6,400 / 12,800 rows: -4.8 % / -4 % of the step with the threshold at CUs / 2, profiles/r04/lanes_notes.md).
        if (force_tile == 0 && big && N < 3072 && (long long)((M + 255) / 256) * (N / 256) < device_cus() / (chip_share == 2 ? 2 : 1)) big = false;
        // bf16 throughput path: ping-pong kernel.  Its buffer resources carry a 31-bit num_records, so operands of 2 GiB or
        // more take the plain-pointer kernels below (64-bit addressing) instead of reading zeros past the resource end.
        if (big && pp_fits(M, N, K, lda, ldw, ldc, sizeof(To), e.act))
            return launch_pp<MODE, To>(A, lda, W, ldw, C, ldc, M, N, K, e, st);
    }
    if (e.ks > 1) return hipErrorInvalidValue;                 // a split-K tail exists in the persistent kernel only (vvk_gemm checks)
    if (big) return launch_t<T, MODE, To, 2>(A, lda, W, ldw, C, ldc, M, N, K, e, st);
    if constexpr (sizeof(T) == 2) {
        // 64-token tiles (three workgroups per CU) when they need no more rounds of workgroups than the 128 x 128 tiles (two per CU) would:
        // launches that do not fill the chip -- every GEMM of a single utterance's CFG branch (M = 1,600: block sum 88 -> 73 us), the
        // N = 1024 shapes up to M = 4,800 -- run 8 - 29 % shorter; all 24 measured (shape, M) pairs agree with the rule
        // (profiles/r04/gemm_tile64_by_m.txt).  Same bits: only the token rows of a tile change.
        // 64 x 64 tiles on the three-stage ring (CFG 4), tile code 6464: explicit here; the path's rule is vv_api.hip's (option "ring_tiles":
        // bf16 GEMMs with N <= 1024 whose 64 x 128 tiles are fewer than the CUs -- the out-projection and FF2 of a single utterance's
        // CFG branch: 10.5 -> 9.7 us and 21.6 -> 17.5 us at M = 1,600, B = 1 113.1 -> 109.1 ms; profiles/r05/gemm_notes.md)
        if (force_tile == 6464 && (size_t)M * lda * sizeof(T) < ((size_t)1 << 31) && (size_t)N * ldw * sizeof(T) < ((size_t)1 << 31))
            return launch_t<T, MODE, To, 4>(A, lda, W, ldw, C, ldc, M, N, K, e, st);
        bool t64 = force_tile == 64;
        if (force_tile == 0) {
            // (a launch that shares the chip with another lane's launches is priced for its share: what 64-token tiles gain by reaching
            // idle CUs the other lane's kernels fill anyway, while their extra operand traffic stays -- M = 3,200 lanes: +3 % with the
            // whole-chip rule, -2 ... -5 % with this one, profiles/r04/lanes_small_shapes*.txt)
            const long long t = (long long)((M + 127) / 128) * (N / 128), cus = device_cus() / (chip_share == 2 ? 2 : 1);
            t64 = t <= 3 * cus && (2 * t + 3 * cus - 1) / (3 * cus) <= (t + 2 * cus - 1) / (2 * cus);
        }
        if (t64) return launch_t<T, MODE, To, 3>(A, lda, W, ldw, C, ldc, M, N, K, e, st);
    }
    return launch_t<T, MODE, To, 0>(A, lda, W, ldw, C, ldc, M, N, K, e, st);
}

}  // namespace

// The split-K tail the persistent bf16 gate-store GEMM of this shape would take on the current device (parts = 0: none).
void vvk_gemm_tail_plan(int M, int N, int K, int lda, int ldw, int ldc, int* row0, int* parts) {
    tail_plan(M, N, K, lda, ldw, ldc, device_cus(), row0, parts);
}

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
    if (g->tile != 0 && g->tile != 128 && g->tile != 256 && !((g->tile == 64 || g->tile == 6464) && g->dtype == VV_BF16)) { *err = "gemm: tile must be 0 (auto), 128, 256, or 64 / 6464 (bf16)"; return -22; }
    if (g->tile == 256 && g->N % 256) { *err = "gemm: the 256 tile needs N % 256 == 0"; return -22; }
    EpiArgs e{};                 // value-initialised: a field this function forgets is zero, never stack garbage
    e.cs_q = g->rope_cs_q; e.cs_k = g->rope_cs_k;
    e.bias = g->bias; e.gate = g->gate; e.cos_q = g->cos_q; e.sin_q = g->sin_q; e.cos_k = g->cos_k; e.sin_k = g->sin_k;
    e.act = g->act; e.n_store = g->n_store > 0 ? g->n_store : g->N; e.seq_n = g->seq_n > 0 ? g->seq_n : 1;
    e.seq_rcp = (unsigned)((((unsigned long long)1 << 32) + (unsigned)e.seq_n - 1) / (unsigned)e.seq_n);
    e.ks = 1; e.c_part = nullptr; e.tail_panel0 = 0;
#ifdef VV_GEMM_STAMP
    if (!g->tail_parts) e.c_part = (char*)g->C_tail;           // diagnostic build: the debug buffer of the stamps
#endif
    if (g->tail_parts) {
        int row0 = 0, parts = 0;
        if (g->mode == MODE_GATE_STORE && g->dtype == VV_BF16 && g->out_dtype == VV_BF16 && (g->tile == 0 || g->tile == 256))
            if (g->act != VV_ACT_GELU_ERF) tail_plan(g->M, g->N, g->K, g->lda, g->ldw, g->ldc, device_cus(), &row0, &parts);
        if (parts != g->tail_parts || row0 != g->tail_row0 || !g->C_tail || ((uintptr_t)g->C_tail % 16)) {
            *err = "gemm: split-K tail does not match vv_gemm_tail_plan for this shape"; return -22;
        }
        if ((size_t)(g->M - row0) * g->ldc * 4 >= ((size_t)1 << 31)) { *err = "gemm: split-K tail buffer of 2 GiB or more"; return -22; }
        e.c_part = (char*)g->C_tail; e.tail_panel0 = row0 / 256; e.ks = parts;
    }
    e.rope_dim = g->rope_dim; e.rope_lo = g->rope_skip_q ? g->rope_dim : 0; e.pos_tab = g->rope_pos;
    if (g->mode == MODE_QKV_ROPE && g->rope_theta > 0.f && g->dtype == VV_BF16) {
        e.rope_k1 = log2f(g->rope_theta) / 32.0f; e.rope_k0 = 2.6514961294723187f;        // log2(2 pi)
        if (!(e.rope_k1 > 0.f)) { *err = "gemm: rope_theta must be > 1"; return -22; }
    }
#ifdef VV_GEMM_EXP
    e.n_group = 0; e.a_nt = 0;
    // Diagnostic build ONLY (tools/build_variants.py vv_gemm exp=-DVV_GEMM_EXP): the walk / cache-hint experiment of profiles/r04/gemm_notes.md,
    // switched from the environment so that ONE library can be timed in all settings inside one process
    if (const char* v = getenv("VV_GEMM_NGROUP")) { const int ng = atoi(v); if (ng > 0 && (g->N / 256) % ng == 0 && g->N / 256 > ng && !g->tail_parts) e.n_group = ng; }
    if (const char* v = getenv("VV_GEMM_A_NT")) e.a_nt = atoi(v) != 0;
#endif
    e.cs_by_row = g->rope_by_row != 0 && g->rope_cs_q && g->rope_cs_k;
    if (g->mode == MODE_QKV_ROPE && !g->rope_pos && (unsigned long long)g->M * (unsigned long long)e.seq_n >= ((unsigned long long)1 << 32)) {
        *err = "gemm: rope without a position table needs M * seq_n < 2^32"; return -22;
    }
    if (g->mode == MODE_QKV_ROPE && (!g->cos_q || !g->sin_q || !g->cos_k || !g->sin_k || g->rope_dim % 64)) {
        *err = "gemm: rope epilogue needs the four tables and rope_dim % 64 == 0"; return -22;
    }
    if (g->mode == MODE_GATE_RES && g->out_dtype != VV_F32) { *err = "gemm: residual stream is fp32"; return -22; }
    const bool bf = g->dtype == VV_BF16, obf = g->out_dtype == VV_BF16;
    hipError_t he = hipSuccess;
#define GO(T, MODE, To) he = launch<T, MODE, To>(g->A, g->lda, g->W, g->ldw, g->C, g->ldc, g->M, g->N, g->K, e, st, g->tile, g->chip_share)
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
    } else if (g->mode == MODE_GATE_STORE) {
        if (!g->gate) { *err = "gemm: gated store needs a gate vector"; return -22; }
        if (bf && obf) GO(bf16, MODE_GATE_STORE, bf16);
        else if (!bf && !obf) GO(float, MODE_GATE_STORE, float);
        else { *err = "gemm: gated store writes the operand dtype"; return -22; }
    } else { *err = "gemm: unknown epilogue mode"; return -22; }
#undef GO
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}
