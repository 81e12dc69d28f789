// K8: flash-style attention forward for gfx950, head_dim 64, key-padding mask by length.
//
// Q/K/V live in the fused projection buffer [rows][3*D] (q | k | v), RoPE and the softmax scale
// already applied by the QKV GEMM epilogue.  One workgroup = 4 waves = 128 queries of one
// (sequence, head); each wave owns 32 queries and the whole 64-wide head.  K/V tiles of 64 keys
// stream into a double-buffered LDS ring by LDS-DMA (global_load_lds_dwordx4).
//
// The score product is computed swapped, S^T = K Q^T (MFMA A = K rows, B = Q rows), so the key
// index lands on the accumulator registers and the query on the lane: the online-softmax row
// max / row sum are in-lane (plus one cross-half exchange), the rescale of O is a per-lane
// scalar, and the exponentiated tile is already the B operand of O^T += V^T P^T with no LDS
// round trip.  bf16: V^T fragments come from the row-major V tile through the gfx950 transposed
// LDS read (ds_read_b64_tr_b16).  fp32: exact f32 MFMA (32x32x2), V read column-wise (ds_read_b32).
#include "vv_common.h"
#include "vv_kernels.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;
// bf16 kernel: the exponentials are taken against a row reference that is ZERO until a row needs one (round 4): p = 2^s' is a
// floating-point value either way -- bf16 / fp32 hold 2^64 as precisely as 1.0 -- so a reference only has to keep p inside the
// exponent range.  A half-row sum above 2^64 (or inf / nan) on any tile, or a whole-row sum below 2^-64 on the FIRST tile (nothing
// accumulated yet: every score far below zero), sends the tile to the careful path, which centres the row on its maximum.
constexpr float RESCALE_SUM = 1.8446744073709552e19f;      // 2^64
constexpr float TINY_SUM = 5.421010862427522e-20f;         // 2^-64

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// max / sum across the two 32-lane halves with the gfx950 half swap (one v_permlane32_swap instead of a ds_bpermute round trip)
__device__ __forceinline__ float half_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(__uint_as_float(r[0])), "v"(__uint_as_float(r[1])));   // no canonicalising pre-max
    return d;
}
__device__ __forceinline__ float bf16_round(float v) {           // round to nearest even onto the bf16 grid, result as f32
    const unsigned u = __float_as_uint(v);
    return __uint_as_float((u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
}
// fragment of the extra contraction step: value at k = 0 of lane half 0, zero elsewhere (v must be bf16-representable)
__device__ __forceinline__ bf16x8 make_q_ext(float v, int h) {
    bf16x8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (bf16)0.f;
    f[0] = (bf16)(h == 0 ? v : 0.f);
    return f;
}
__device__ __forceinline__ float half_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LDS-DMA as an asm statement (cdna guide, 'What hipcc does not do' item 1): hipcc treats the builtin form as a write to the LDS array
// and puts an s_waitcnt vmcnt(0) in front of the next ds_read of that array -- here in front of PV, right behind the next tile's
// staging loads.  The asm form is outside its bookkeeping: its completion is the explicit vmcnt(0) + barrier at the top of a tile.
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
__device__ __forceinline__ i32x4_t make_rsrc4(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void glds16_buf_asm(i32x4_t rs, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_dst) : "memory");
}

// Workgroup -> (query block, head, sequence), XCD-aware.  Workgroups are dealt round-robin over the 8 XCDs by linear id, and each
// XCD has its own L2: with a plain (q-block, head, seq) grid the 13 query blocks that share one (sequence, head)'s K/V land on
// 8 different L2s and the K/V stream is fetched from the fabric up to 8 times.  Here id % 8 labels the XCD group and every
// group walks whole (sequence, head) pairs, query blocks consecutively, so a pair's K/V is filled into ONE L2 once and re-read
// there by its other query blocks.  Placement is a speed matter only (any mapping is correct).
struct AttnBlock { int qb, head, seq; bool valid; };
__device__ __forceinline__ AttnBlock attn_block(int nqb, int heads, int n_seq) {
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int pair = (j / nqb) * 8 + xcd;              // (sequence, head) pairs are dealt to the XCD groups round-robin
    AttnBlock r;
    r.qb = j % nqb;
    r.valid = pair < heads * n_seq;
    r.head = pair % heads;
    r.seq = pair / heads;
    return r;
}

// Diagnostic builds ONLY (tools/build_variants.py vv_attention abl1=-DVV_ATTN_ABLATE=1 ...; never the shipped library; results are
// meaningless, only the time is read -- profiles/r04/attention_notes.md):
//   1  matrix pipe + LDS reads only: no v_exp, no sums, no row max / redo, no cvt (P = the raw score registers)
//   2  vector work only: exponentials, sums, the speculative check, cvt; the MFMAs and their LDS reads are gone (operands opaque)
//   3  staging only: the LDS-DMA stream, its waits and the barriers
#ifndef VV_ATTN_ABLATE
#define VV_ATTN_ABLATE 0
#endif

// ------------------------------------------------------------------------------------ bf16
__global__ __launch_bounds__(256, 2) void attn_bf16_kernel(const bf16* __restrict__ qkv, int ld, bf16* __restrict__ out,
                                                           int ldo, int seq_n, int D, const int* __restrict__ kv_len_arr,
                                                           const int* __restrict__ row_start, int total_rows, int heads, int n_seq,
                                                           const float* __restrict__ rope_cs_q, float q_mul) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 16384];   // per stage: K 8 KiB | V 8 KiB
    const AttnBlock blk = attn_block((seq_n + 127) / 128, heads, n_seq);
    if (!blk.valid) return;
    const int head = blk.head, seq = blk.seq, qblock = blk.qb;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    int kv_len = kv_len_arr ? kv_len_arr[seq] : seq_n;
    kv_len = max(1, min(kv_len, seq_n));
    // packed rows (row_start given): sequence `seq` owns rows [row_start[seq], +kv_len) only -- the rows after them are the
    // next sequence's, so queries stop at kv_len and every row index is clamped inside the sequence.
    const size_t row0 = row_start ? (size_t)row_start[seq] : (size_t)seq * seq_n;
    const int q_lim = row_start ? kv_len : seq_n;
    if (qblock * 128 >= q_lim) return;

    const bf16* Qp = qkv + row0 * ld + head * 64;
    const bf16* Kp = Qp + D;
    // bytes from Kp to the end of the qkv buffer (host-checked < 2 GiB): the bound of the K/V buffer resource
    const unsigned kv_bytes = (unsigned)(((size_t)total_rows - row0) * (size_t)ld * 2 - (size_t)(head * 64 + D) * 2);

    const int q0 = qblock * 128 + wave * 32;
    const int qrow = min(q0 + r32, q_lim - 1);
    bf16x8 qf[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) {
        qf[ds] = *(const bf16x8*)(Qp + (size_t)qrow * ld + ds * 16 + h * 8);
        if (rope_cs_q) {
            // Query-side RoPE here instead of in the QKV GEMM's epilogue (round 4): this lane's 8 dims are 4 interleaved pairs, the
            // position is the row inside the sequence, the (cos, sin) pairs carry the softmax scale.  Roped in fp32 from the bf16 the
            // GEMM stored, then scaled by log2(e) and rounded ONCE -- Q is rounded twice on its way into the MFMA either way (before:
            // after the rope in the GEMM, and here after the log2(e) scaling).  Once per workgroup: 32 bytes x 4 per lane from an
            // L2-resident table, against a third of the rope epilogue's table loads and in-order waits in 682 launches of the slowest GEMM.
            const float* tq = rope_cs_q + (size_t)qrow * 64 + ds * 16 + h * 8;
            const float4 t0 = *(const float4*)tq, t1 = *(const float4*)(tq + 4);
            const float cs[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float a = (float)qf[ds][j], b = (float)qf[ds][j + 1];
                const float na = __builtin_fmaf(a, cs[j], -(b * cs[j + 1])), nb = __builtin_fmaf(b, cs[j], a * cs[j + 1]);
                qf[ds][j] = (bf16)(na * LOG2E);
                qf[ds][j + 1] = (bf16)(nb * LOG2E);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[ds][j] = (bf16)((float)qf[ds][j] * q_mul);     // scores in the log2 domain: p = exp2(s); q_mul = log2(e) x q_scale
        }
    }

    // staging: K tile = 8 pieces of 8 rows x 128 B, V likewise; wave w issues pieces 2w, 2w+1 of each.  buffer_load ... lds (as asm
    // statements: glds16_buf_asm -- 689 -> 676 us at the bench shape against the builtin, bit-identical, profiles/r03/attn_ab_asm_stage.txt)
    // with a per-lane byte offset that advances by one add per piece and tile (the advance stays in the VGPR offset: the
    // SGPR offset of a buffer instruction is not range-checked).  Rows past the end of the buffer (last tile of the last
    // sequence) read as zero instead of needing a clamp; rows past the sequence but inside the buffer are the next
    // sequence's (finite data) and are masked like any key >= kv_len.
    const i32x4_t rs4 = make_rsrc4(Kp, kv_bytes);          // buffer resource of the K/V columns: base Kp, kv_bytes records (raw, range-checked)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned voff_k[2], voff_v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = (wave * 2 + u) * 8 + (lane >> 3);
        const int p = lane & 7;
        const int ck = p ^ ((row >> 1) & 7);              // swz128 (row reads, ds_read_b128)
        const int cv = p ^ (((row >> 1) & 1) << 2);       // V: conflict-free transposed reads
        voff_k[u] = (unsigned)row * (unsigned)ld * 2u + ck * 16;
        voff_v[u] = (unsigned)row * (unsigned)ld * 2u + (unsigned)D * 2u + cv * 16;      // V = K + D columns
    }
    const int tile_bytes = 64 * ld * 2;
    // INVARIANT (the asm form is outside hipcc's memory and waitcnt bookkeeping; nothing else guards this): stage(kt + 1, buf) must be
    // issued AFTER the barrier at the top of tile kt, and buf must be the buffer last read in tile kt - 1 -- every wave is past that
    // barrier only when it has finished tile kt - 1's PV reads.  The loads land under the explicit vmcnt(0) + barrier at the top of
    // tile kt + 1.  Moving the call above the barrier, or adding a third buffer, needs a new argument; the every-row headline-size test
    // (test_attention_headline_shape_every_row) is the regression guard.
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = wave * 2 + u;
            glds16_buf_asm(rs4, voff_k[u] + kt * tile_bytes, lds0 + buf * 16384 + q * 1024);
            glds16_buf_asm(rs4, voff_v[u] + kt * tile_bytes, lds0 + buf * 16384 + 8192 + q * 1024);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_eff = 0.f, l_run = 0.f;              // m_eff: row reference in the log2 domain, bf16-representable
    bool has_ref = false;                        // wave-uniform: some row of this wave has a non-zero reference (the scores then need the shift)
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    const bf16x8 k_one = make_q_ext(1.0f, h);    // K side of the 65th contraction element: 1.0 at k = 0 of lane half 0
    bf16x8 q_ext = make_q_ext(0.f, h);           // Q side: -m_eff

    // transposed-read lane constants: 16-lane group g reads a 4-key x 16-d block.  Per d-tile the byte offset of this
    // lane's 8-byte piece inside a 16-key slab is a constant; (key block, k-step, +8 rows) are immediates.
    const int tr_grp = (lane >> 4) & 1, tr_i = lane & 15, tr_q = tr_i >> 2, tr_p = tr_i & 3;
    int tr_off[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int c = dt * 4 + 2 * tr_grp + (tr_p >> 1);
        tr_off[dt] = (4 * h + tr_q) * 128 + ((c ^ (((tr_q >> 1) & 1) << 2)) << 4) + (tr_p & 1) * 8;
    }

    const int n_tiles = (kv_len + 63) / 64;
    stage(0, 0);
    for (int kt = 0; kt < n_tiles; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // The next tile's K/V pieces (4 LDS-DMA instructions per wave, 60-185 cycles of issue each) are issued BETWEEN the
        // exponentials and the PV block of this tile, not here in front of the QK^T block: 705 -> 683 us at the bench shape,
        // bit-identical (profiles/r02/attn_ab_stage.txt).
        const char* sK = smem + (kt & 1) * 16384;
        const char* sV = sK + 8192;

        // ---- S'^T = K Q^T - m : scores arrive in the log2 domain (Q fragments carry log2e) and ALREADY SHIFTED by the
        // row reference m_eff: the shift rides as a 65th contraction element (K side 1.0, Q side -m_eff, one extra MFMA
        // per key block) instead of 32 VALU fmas per tile -- the loop is VALU-bound (profiles/r01/attention_notes.md).
        // s[kb][reg] -> key kb*32 + (reg&3) + 8(reg>>2) + 4h, query r32
        f32x16 s[2];
        const int kbase = kt * 64;
        // The two MFMA blocks of a tile issue at raised wave priority (s_setprio 1): among the four waves of a SIMD (four different
        // workgroups) the one that is ready to feed the matrix pipe goes first, the ones in their softmax blocks fill in behind it.
        // 703 -> 694 us at the bench shape, bit-identical (profiles/r02/attn_ab_prio*.txt).
        auto scores = [&]() {
#if VV_ATTN_ABLATE == 2 || VV_ATTN_ABLATE == 3
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[kb][r] = -1.0f; asm volatile("" : "+v"(s[kb][r])); }       // opaque scores, no instruction
            return;
#endif
            __builtin_amdgcn_s_setprio(1);
            if (has_ref) {                       // rare: a row of this wave is centred on a reference: the shift rides as a 65th contraction element
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_one, q_ext, zero16, 0, 0, 0);
#pragma unroll
                    for (int ds = 0; ds < 4; ++ds) {
                        const bf16x8 kf = *(const bf16x8*)(sK + swz128(kb * 32 + r32, 2 * ds + h));
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], s[kb], 0, 0, 0);
                    }
                }
            } else {                             // the common case: no reference anywhere in the wave, 8 MFMAs instead of 10
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int ds = 0; ds < 4; ++ds) {
                        const bf16x8 kf = *(const bf16x8*)(sK + swz128(kb * 32 + r32, 2 * ds + h));
                        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], ds == 0 ? zero16 : s[kb], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (kbase + 64 > kv_len) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (key >= kv_len) s[kb][r] = NEG_BIG;
                    }
            }
        };
        // p = 2^s' in place, returns this half-wave's partial row sum.  Single f32 instructions on purpose -- packed f32 VALU
        // (v_pk_fma/add_f32) issues slowly beside MFMAs on gfx950 (measured 1011 -> 989 us when unpacked).
        auto exps = [&]() -> float {
            float ps[4] = {0.f, 0.f, 0.f, 0.f};          // four independent sum chains
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = fast_exp2(s[kb][r]);
                    ps[r & 3] += pv;
                    s[kb][r] = pv;
                }
            return (ps[0] + ps[1]) + (ps[2] + ps[3]);
        };
        // Online softmax with a SPECULATIVE tile: the exponentials are taken against the current reference m_eff (zero until a
        // row needs one) without looking for the row max first (21 VALU instructions per tile); the partial row sums then tell
        // whether that was safe -- an element above 2^64 (or an overflow to inf) puts its half-row sum above the bound.  Only
        // then the tile is redone the careful way: scores again, row max, reference moved (kept bf16-representable so that the
        // MFMA subtracts it exactly), O and l rescaled.  Softmax is invariant to the reference, so results do not depend on
        // which path ran, beyond the bf16 rounding of P.  (Rounds 1-3 centred every row on its first tile's maximum: one careful
        // pass per workgroup and 2 of 18 MFMAs per tile for a shift that random or trained logits never need.)
        float psum = 0.f;
        bool redo = false;
#if VV_ATTN_ABLATE == 1 || VV_ATTN_ABLATE == 3
        scores();
        if (kt + 1 < n_tiles) stage(kt + 1, (kt + 1) & 1);
#else
        scores();
        psum = exps();
        if (kt + 1 < n_tiles) stage(kt + 1, (kt + 1) & 1);
        redo = __any(!(psum <= RESCALE_SUM));
        if (kt == 0) redo = redo || __any(!(half_sum(psum) >= TINY_SUM));      // first tile: a row whose every weight underflowed needs its own reference
        if (redo) {
            scores();
            float mx = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            mx = half_max(mx);                        // the other half-wave holds the other 32 keys of the same query
            const float m_new = bf16_round(m_eff + (kt == 0 ? mx : fmaxf(mx, 0.f)));      // first tile: centre on the row max, up or down
            const float d = m_new - m_eff;                     // exact: both are bf16 values
            if (kt != 0) {                                     // O and l are zero on the first tile
                const float alpha = fast_exp2(-d);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] -= d;    // this tile was shifted by the old reference
            m_eff = m_new;
            q_ext = make_q_ext(-m_new, h);
            has_ref = true;
            psum = exps();
        }
#endif
        l_run += psum;
#if VV_ATTN_ABLATE == 3
        continue;
#endif

        // ---- O^T += V^T P^T : the accumulator registers 8st..8st+7 are the B fragment of k-step st
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const f32x8 pv = {s[kb][8 * st + 0], s[kb][8 * st + 1], s[kb][8 * st + 2], s[kb][8 * st + 3],
                                  s[kb][8 * st + 4], s[kb][8 * st + 5], s[kb][8 * st + 6], s[kb][8 * st + 7]};
#if VV_ATTN_ABLATE == 1
                typedef __attribute__((ext_vector_type(4))) float f32x4a;
                const bf16x8 pf = __builtin_bit_cast(bf16x8, (f32x4a){pv[0], pv[2], pv[4], pv[6]});      // no cvt: raw score bits as the operand
#else
                const bf16x8 pf = __builtin_convertvector(pv, bf16x8);       // 4 x v_cvt_pk_bf16_f32
#endif
#if VV_ATTN_ABLATE == 2
                asm volatile("" :: "v"(pf));                                  // the cvt results are consumed, the PV MFMAs and V reads are gone
                continue;
#endif
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const char* a0 = sV + tr_off[dt] + (kb * 32 + 16 * st) * 128;
                    const char* a1 = a0 + 8 * 128;
                    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a0);
                    const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)a1);
                    const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
    }
    const float l_tot = half_sum(l_run);
    const float inv = 1.0f / l_tot;
    const int q = q0 + r32;
    if (q < q_lim) {
        bf16* op = out + (row0 + q) * ldo + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = dt * 32 + 8 * g + 4 * h;
                store4<bf16>(op + d0, o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv,
                             o[dt][4 * g + 3] * inv);
            }
    }
}

// ------------------------------------------------------------------------------------ fp32
__global__ __launch_bounds__(256, 2) void attn_f32_kernel(const float* __restrict__ qkv, int ld, float* __restrict__ out,
                                                          int ldo, int seq_n, int D, const int* __restrict__ kv_len_arr,
                                                          const int* __restrict__ row_start, int heads, int n_seq) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 32768];   // per stage: K 16 KiB | V 16 KiB (256-B rows)
    const AttnBlock blk = attn_block((seq_n + 127) / 128, heads, n_seq);
    if (!blk.valid) return;
    const int head = blk.head, seq = blk.seq, qblock = blk.qb;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    int kv_len = kv_len_arr ? kv_len_arr[seq] : seq_n;
    kv_len = max(1, min(kv_len, seq_n));
    const size_t row0 = row_start ? (size_t)row_start[seq] : (size_t)seq * seq_n;       // packed rows: see the bf16 kernel
    const int q_lim = row_start ? kv_len : seq_n;
    if (qblock * 128 >= q_lim) return;

    const float* Qp = qkv + row0 * ld + head * 64;
    const float* Kp = Qp + D;
    const float* Vp = Qp + 2 * D;

    const int q0 = qblock * 128 + wave * 32;
    const int qrow = min(q0 + r32, q_lim - 1);
    f32x4 qf[8];   // lane half h takes d = 8kk + 4h + j: the same k pairing as the K fragment below
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = *(const f32x4*)(Qp + (size_t)qrow * ld + kk * 8 + h * 4);

    // staging: 16 pieces of 4 rows x 256 B per tile; wave w issues pieces 4w..4w+3 of K and V
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * 32768;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = wave * 4 + u;
            const int row = q * 4 + (lane >> 4);
            const int key = min(kt * 64 + row, q_lim - 1);
            const int p = lane & 15;
            const int ck = p ^ (row & 15);
            glds16(Kp + (size_t)key * ld + ck * 4, base + q * 1024);
            glds16(Vp + (size_t)key * ld + p * 4, base + 16384 + q * 1024);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = NEG_BIG, l_run = 0.f;

    const int n_tiles = (kv_len + 63) / 64;
    stage(0, 0);
    for (int kt = 0; kt < n_tiles; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < n_tiles) stage(kt + 1, (kt + 1) & 1);
        const char* sK = smem + (kt & 1) * 32768;
        const char* sV = sK + 16384;

        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
            const int row = kb * 32 + r32;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const f32x4 kf = *(const f32x4*)(sK + row * 256 + (((2 * kk + h) ^ (row & 15)) << 4));
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    s[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[j], qf[kk][j], s[kb], 0, 0, 0);
            }
        }
        const int kbase = kt * 64;
        if (kbase + 64 > kv_len) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kbase + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (key >= kv_len) s[kb][r] = NEG_BIG;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f((m_run - m_new) * LOG2E);
        m_run = m_new;
        const float msc = m_new * LOG2E;
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = exp2f(fmaf(s[kb][r], LOG2E, -msc));
                psum += p;
                s[kb][r] = p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;

        // O^T += V^T P^T, two keys per MFMA: lane half h contributes key (i&3)+8(i>>2)+4h of block kb
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const float vf = *(const float*)(sV + key * 256 + (dt * 32 + r32) * 4);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[kb][i], o[dt], 0, 0, 0);
                }
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + r32;
    if (q < q_lim) {
        float* op = out + (row0 + q) * ldo + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = dt * 32 + 8 * g + 4 * h;
                store4<float>(op + d0, o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv,
                              o[dt][4 * g + 3] * inv);
            }
    }
}

}  // namespace

int vvk_attention(const vv_attn_args* a, hipStream_t st, const char** err) {
    if (a->n_seq <= 0 || a->seq_n <= 0 || a->heads <= 0) { *err = "attention: empty shape"; return -22; }
    if (a->dim != a->heads * 64) { *err = "attention: head_dim must be 64"; return -22; }
    const int esz = a->dtype == VV_BF16 ? 2 : 4;
    if (((size_t)a->ld_qkv * esz) % 16 || ((size_t)a->ld_out * esz) % 8 || ((uintptr_t)a->qkv % 16) || ((uintptr_t)a->out % 16)) {
        *err = "attention: qkv/out must be 16-byte aligned"; return -22;
    }
    if (a->ld_qkv < 3 * a->dim || a->ld_out < a->dim) { *err = "attention: leading dimensions too small"; return -22; }
    if (a->row_start && !a->kv_len) { *err = "attention: packed rows need kv_len"; return -22; }
    if (a->rope_cs_q && (a->dtype != VV_BF16 || ((uintptr_t)a->rope_cs_q % 16))) { *err = "attention: the query-side rope table is taken by the bf16 kernel only, 16-byte aligned"; return -22; }
    // the last tile of the last packed sequence reads past its rows: total_rows is what bounds the K/V buffer resource there
    // (out-of-range rows read as zero), so with packed rows it cannot be defaulted
    if (a->row_start && a->total_rows <= 0) { *err = "attention: packed rows need total_rows (rows in the qkv buffer)"; return -22; }
    const int total_rows = a->total_rows > 0 ? a->total_rows : a->n_seq * a->seq_n;
    if (a->dtype == VV_BF16 && (size_t)total_rows * a->ld_qkv * 2 >= ((size_t)1 << 31)) { *err = "attention: qkv buffer must stay below 2 GiB"; return -22; }
    if (a->row_start == nullptr && total_rows < a->n_seq * a->seq_n) { *err = "attention: total_rows smaller than n_seq * seq_n"; return -22; }
    const int nqb = (a->seq_n + 127) / 128;
    const long long pairs8 = ((long long)a->heads * a->n_seq + 7) / 8;          // (sequence, head) pairs per XCD group
    if (pairs8 * nqb * 8 > 0x7fffffffLL) { *err = "attention: grid too large"; return -22; }
    const dim3 grid((unsigned)(pairs8 * nqb * 8));                               // 1-D: id % 8 = XCD group (attn_block)
    if (a->dtype == VV_BF16)
        attn_bf16_kernel<<<grid, 256, 0, st>>>((const bf16*)a->qkv, a->ld_qkv, (bf16*)a->out, a->ld_out, a->seq_n, a->dim, a->kv_len, a->row_start, total_rows,
                                               a->heads, a->n_seq, a->rope_cs_q, LOG2E * (a->q_scale > 0.f ? a->q_scale : 1.0f));
    else
        attn_f32_kernel<<<grid, 256, 0, st>>>((const float*)a->qkv, a->ld_qkv, (float*)a->out, a->ld_out, a->seq_n, a->dim, a->kv_len, a->row_start,
                                              a->heads, a->n_seq);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}
