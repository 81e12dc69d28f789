// K11x: the vocoder's conv1d / ConvTranspose1d implicit GEMM (vv_vocoder.hip, K11) on the bf16 matrix pipe with fp32
// fidelity -- "x3": every fp32 operand is cut into THREE bf16 pieces, a = h + m + l EXACTLY (24 significand bits = 3 x 8,
// each piece the truncated leading 8 bits of what is left; bf16 has fp32's exponent range, so nothing over- or underflows),
// and the product a * w is taken as the six piece products whose weight is >= 2^-16 of it,
//        h.l + l.h + m.m + h.m + m.h + h.h        (dropped: m.l + l.m + l.l <= 2^-23 |a w|),
// each an exact 16-bit product accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The error of a contraction is that of an
// fp32 FMA chain (one fp32 rounding per accumulation step, 2^-24) plus 2^-23 per term: the same class as the exact-f32 MFMA
// of K11, at 6 x 32 cycles per 32x32x16 block instead of 8 x 64 (v_mfma_f32_32x32x2_f32): 2.7x less matrix-pipe time in a
// kernel class that is matrix-pipe bound in fp32 (profiles/r02/vocoder_notes.md).  Parity against the float64 oracle is
// asserted at the fp32 path's own tolerance (tests/test_kernels_gpu.py, tests/test_fullsize_gpu.py).
//
// Layout.  Weights are split ONCE (vvk_conv_split_weights, at vv_finalize_weights) into
//        Wb[chunk = ci / 16][kw][piece 3][octet = (ci % 16) / 8][row (rows_pad)][ci % 8]   bf16,
// rows = co (conv) or co * up + phase (transposed, polyphase as in K11).  A workgroup owns VR = 32 RT rows x 256 time steps;
// the K loop walks 16 input channels at a time: the input window [q0 - left, q0 + 256 + span - left) of the 16 channels is
// loaded (LeakyReLU fused, zero outside [0, len)), split, and stored as xs[piece][octet][time][8 ch] -- an MFMA B fragment
// (column = time, 8 consecutive channels = one octet) is then ONE 16-byte LDS read, and the 32 lanes of a k-half read 32
// CONSECUTIVE 16-byte slots: ds_read_b128 serves fixed 16-lane groups, each of which then covers all 64 banks (an interleaved
// [time][16 ch] image, lanes 32 bytes apart, is 2-way conflicted: measured 4 % slower over the decode shapes) -- and the weight slab of a
// group of TG taps as ws[tap][piece][octet][row][8 ch] (A fragments likewise).  The epilogue (bias, residual, MRF scale / accumulate)
// is branch-free and batched: buffer loads / stores whose out-of-range elements carry an offset past num_records.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "vv_common.h"
#include "vv_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;      // a native vector (HIP's uint4 is a struct of unions: register arrays of it end up in scratch)

constexpr int XT = 256;       // time steps per workgroup

typedef __attribute__((ext_vector_type(2))) float f32x2;
// Pair form of lrelu + split3 for the staging loop (v_pk_mul_f32 / v_pk_add_f32: one instruction per two elements; the max as
// the bare instruction -- fmaxf would add a canonicalising multiply).  slope in (0, 1]: lrelu(x) = max(x, slope x).
__device__ __forceinline__ void lrelu_split3_pair(float a0, float a1, float slope, unsigned& h01, unsigned& m01, unsigned& l01) {
    const f32x2 a = {a0, a1};
    const f32x2 sa = a * slope;
    f32x2 v;
    asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(a.x), "v"(sa.x));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(a.y), "v"(sa.y));
    const unsigned h0 = __float_as_uint(v.x) & 0xffff0000u, h1 = __float_as_uint(v.y) & 0xffff0000u;
    const f32x2 hv = {__uint_as_float(h0), __uint_as_float(h1)};
    const f32x2 r1 = v - hv;
    const unsigned m0 = __float_as_uint(r1.x) & 0xffff0000u, m1 = __float_as_uint(r1.y) & 0xffff0000u;
    const f32x2 mv = {__uint_as_float(m0), __uint_as_float(m1)};
    const f32x2 r2 = r1 - mv;
    h01 = __builtin_amdgcn_perm(h1, h0, 0x07060302u);
    m01 = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    l01 = __builtin_amdgcn_perm(__float_as_uint(r2.y), __float_as_uint(r2.x), 0x07060302u);
}

// a = h + m + l exactly; the pieces are fp32 bit patterns whose low 16 bits are zero (bf16 in the high half)
__device__ __forceinline__ void split3(float a, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(a) & 0xffff0000u;
    const float r1 = a - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2);
}

// ---- weight split: Wt fp32 [Cin_pad][KW][rows_pad]  ->  Wb [chunks][KW][3][rows_pad][16]
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ Wt, uint16_t* __restrict__ Wb, int Cin_pad, int KW,
                                                            int rows_pad, int chunks) {
    const size_t n = (size_t)chunks * KW * rows_pad * 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int j = (int)(i & 15);
        const size_t t = i >> 4;
        const int row = (int)(t % rows_pad);
        const int kw = (int)((t / rows_pad) % KW);
        const int chunk = (int)(t / ((size_t)rows_pad * KW));
        const int c = chunk * 16 + j;
        const float w = c < Cin_pad ? Wt[((size_t)c * KW + kw) * rows_pad + row] : 0.f;
        unsigned h, m, l;
        split3(w, h, m, l);
        const size_t base = (((size_t)chunk * KW + kw) * 3) * 2 * rows_pad;           // [piece][octet][row][8]
        const size_t in_piece = ((size_t)(j >> 3) * rows_pad + row) * 8 + (j & 7);
        Wb[(base + 0 * 2 * (size_t)rows_pad) * 8 + in_piece] = (uint16_t)(h >> 16);
        Wb[(base + 1 * 2 * (size_t)rows_pad) * 8 + in_piece] = (uint16_t)(m >> 16);
        Wb[(base + 2 * 2 * (size_t)rows_pad) * 8 + in_piece] = (uint16_t)(l >> 16);
    }
}

// KW: taps.  TR: polyphase ConvTranspose (KW must be 2).  RT: 32-row MFMA tiles per wave (a workgroup = 4 waves owns
// 32 RT rows x 256 time steps, wave w the columns [64 w, 64 w + 64)).  TG: taps per weight phase.
//
// Pipeline.  The K loop is a sequence of phases (16-channel chunk, group of <= TG taps).  The weight slab of a phase is
// double-buffered in LDS: phase p loads the slab of phase p + 1 into registers BEFORE its MFMA block and stores it to the other
// buffer AFTER it (that buffer was last read in phase p - 1, which every wave left before the barrier that opened phase p), so
// the load has the whole MFMA block to land and a phase costs one barrier.  The input window of chunk c + 1 is loaded into
// registers before the MFMA block of chunk c's last phase and split / stored behind the next barrier (+ one barrier to publish).
// NWV: waves per workgroup, 4 (one row group) or 8 (two row groups of 32 RT rows sharing the staged window: the window is
// loaded and split once for 64 RT rows -- the stages with >= 128 rows).
template <int KW, bool TR, int RT, int TG, int NWV, int OCC>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void conv_x3_kernel(const float* __restrict__ in, const u32x4* __restrict__ Wb, const float* __restrict__ bias,
                                                         float* __restrict__ out, const float* __restrict__ resid, int Cin, int rows_total,
                                                         int rows_pad, int T_in, int T_out, int Cout, int dil, int up, float pre_slope,
                                                         float out_scale, int accumulate, const int* __restrict__ len_in, int n_tt, int n_rt, int n_win) {
    constexpr int NT = 64 * NWV;                       // threads
    constexpr int VR = (NWV / 4) * RT * 32;            // rows per workgroup
    constexpr int NIT = (2 * (XT + 50) + NT - 1) / NT; // window items (time, channel octet) per thread: 2 (256 + span), span <= 50
    constexpr int NPH = (KW + TG - 1) / TG;            // phases per chunk
    constexpr int NW = (TG * 3 * VR * 2 + NT - 1) / NT;  // 16-byte weight pieces per thread and phase
    constexpr int WSLAB = NW * NT;                     // pieces per weight buffer (rounded up: every thread loads and stores NW, unconditionally)
    const int left = TR ? 1 : dil * (KW - 1) / 2;
    const int span = TR ? 1 : dil * (KW - 1);
    const int xw = XT + span;                          // window index i <-> time q0 - left + i
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* ws = (u32x4*)smem;                          // [2 buffers][TG][3][2 octets][VR]
    u32x4* xs = ws + 2 * WSLAB;                        // [3][2 octets][xw]

    // Workgroup -> (window = (item, time tile), row tile), XCD-aware: the row tiles of one window all read the same input window, so
    // they are given ids 8 apart -- the same XCD (workgroups are dealt round-robin over the 8 XCDs by id), dispatched back to back --
    // and the window comes out of that XCD's L2 instead of being fetched from HBM once per row tile (measured before: 2x the
    // algorithmic reads at 2 row tiles, 5x at 4, up to 20x for the transposed convs: profiles/r02/voc_conv_pmc_traffic*.txt).
    const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
    const int win = (jj / n_rt) * 8 + xcd;
    if (win >= n_win) return;
    const int b = win / n_tt;
    const int r0 = (jj % n_rt) * VR;
    const int q0 = (win % n_tt) * XT;
    const int lane = threadIdx.x & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = wave_all & 3, rg = wave_all >> 2;  // column group (64 time steps), row group (32 RT rows)
    const int r32 = lane & 31, h = lane >> 5;
    const int lin = len_in ? min(len_in[b], T_in) : T_in;
    const float* inb = in + (size_t)b * Cin * T_in;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, 0, (int)min((size_t)Cin * T_in * 4, (size_t)0x7fffffff), 0x00020000);
    const int n_chunks = (Cin + 15) >> 4;

    f32x16 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float xreg[NIT][8];
    u32x4 wreg[NW];
    // ---- input window, item (o, t) = 8 channels c0 + 8 o .. at window column t (consecutive lanes = consecutive columns of one octet:
    // every load instruction reads one 256-byte run of a channel row, every store 1 KiB contiguous)
    auto load_x = [&](int ch) {
        const int c0 = ch * 16;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * NT;
            const int o = idx >= xw, t = idx - o * xw;
            const int time = q0 - left + t;
            const bool tin = t < xw && time >= 0 && time < lin;
            // no branches: an element outside the window / the row / the channels is a load past num_records, which returns zero
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = c0 + 8 * o + j;
                const unsigned off = (tin && c < Cin) ? (unsigned)(c * T_in + time) * 4u : 0x80000000u;
                xreg[it][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, (int)off, 0, 0));
            }
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = threadIdx.x + it * NT;
            const int o = idx >= xw, t = idx - o * xw;
            if (t < xw) {
                u32x4 wh, wm, wl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned hh, mm, ll;
                    lrelu_split3_pair(xreg[it][2 * j], xreg[it][2 * j + 1], pre_slope, hh, mm, ll);
                    wh[j] = hh; wm[j] = mm; wl[j] = ll;
                }
                xs[(0 * 2 + o) * xw + t] = wh;
                xs[(1 * 2 + o) * xw + t] = wm;
                xs[(2 * 2 + o) * xw + t] = wl;
            }
        }
    };
    // ---- weight slab of taps g0 .. g0 + tg of chunk ch: [tap][piece][octet][row] <- Wb[ch][kw][piece][octet][r0 + row]  (16-byte pieces)
    auto load_w = [&](int ch, int g0, int tg) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int idx = min((int)threadIdx.x + i * NT, tg * 3 * VR * 2 - 1);  // past the group: a harmless duplicate of its last element
            const int row = idx % VR, ko = idx / VR;                               // ko = (tap_local * 3 + piece) * 2 + octet
            wreg[i] = Wb[(((size_t)ch * KW + g0) * 3 * 2 + ko) * rows_pad + r0 + row];
        }
    };
    auto store_w = [&](int buf, int tg) {
#pragma unroll
        for (int i = 0; i < NW; ++i) ws[buf * WSLAB + threadIdx.x + i * NT] = wreg[i];
        (void)tg;
    };

    load_x(0);
    load_w(0, 0, KW < TG ? KW : TG);
    store_x();
    store_w(0, KW < TG ? KW : TG);
    int p = 0;                                         // phase counter (weight buffer parity)
    for (int ch = 0; ch < n_chunks; ++ch) {
#pragma unroll(OCC == 3 && KW > 7 ? 1 : NPH)
        for (int gi = 0; gi < NPH; ++gi, ++p) {
            const int g0 = gi * TG;
            const int tg = (KW - g0) < TG ? (KW - g0) : TG;                        // compile-time after unrolling
            const int g0n = gi + 1 < NPH ? g0 + TG : 0;
            const int tgn = (KW - g0n) < TG ? (KW - g0n) : TG;
            const int chn = gi + 1 < NPH ? ch : ch + 1;
            __syncthreads();                           // phase p - 1 is over everywhere: its weight buffer and (at a chunk boundary) xs are free
            if (gi == 0 && ch > 0) {
                store_x();                             // the window of this chunk, loaded during the previous chunk's last phase
                __syncthreads();
            }
            if (chn < n_chunks) load_w(chn, g0n, tgn);
            if (gi == NPH - 1 && ch + 1 < n_chunks) load_x(ch + 1);
            const u32x4* wb = ws + (p & 1) * WSLAB;
#pragma unroll(KW <= 3 && OCC < 3 ? 3 : 1)
            for (int lk = 0; lk < tg; ++lk) {
                const int kw = g0 + lk;
                const int off = TR ? (left - kw) : kw * dil;                       // window column = local time + off
                bf16x8 a[RT][3], x[2][3];
#pragma unroll
                for (int ri = 0; ri < RT; ++ri)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) a[ri][pc] = __builtin_bit_cast(bf16x8, wb[((lk * 3 + pc) * 2 + h) * VR + (rg * RT + ri) * 32 + r32]);
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) x[ti][pc] = __builtin_bit_cast(bf16x8, xs[(pc * 2 + h) * xw + wave * 64 + ti * 32 + r32 + off]);
                // six piece products, smallest first; the (row tile, time tile) accumulators alternate inside a term
#define VV_X3_TERM(pa, pb)                                                                                                   \
    _Pragma("unroll") for (int ri = 0; ri < RT; ++ri) _Pragma("unroll") for (int ti = 0; ti < 2; ++ti)                         \
        acc[ri][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ri][pa], x[ti][pb], acc[ri][ti], 0, 0, 0)
                VV_X3_TERM(0, 2); VV_X3_TERM(2, 0); VV_X3_TERM(1, 1); VV_X3_TERM(0, 1); VV_X3_TERM(1, 0); VV_X3_TERM(0, 0);
#undef VV_X3_TERM
            }
            if (chn < n_chunks) store_w((p + 1) & 1, tgn);
        }
    }

    // ---- epilogue: D[row_local = (reg&3) + 8(reg>>2) + 4h][time_local = r32].  Branch-free and batched: an element outside the
    // rows / the time range carries an offset past num_records (loads return zero, stores are dropped), so the 16 residual loads
    // and the 16 accumulate loads of a tile are all in flight together (the K11 form -- a bounds branch, a dependent load and a
    // full wait per element -- cost up to a third of the launch here, where no MFMA backlog hides it).
    const size_t out_elems = (size_t)Cout * T_out;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)b * out_elems), 0, (int)min(out_elems * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)((resid ? resid : out) + (size_t)b * out_elems), 0, (int)min(out_elems * 4, (size_t)0x7fffffff), 0x00020000);
#pragma unroll
    for (int ri = 0; ri < RT; ++ri) {
        float bv[16];
        int cor[16], tph[16];                          // output channel, and (transposed) the phase term of the output time
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + (rg * RT + ri) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            int co = row, ph = 0;
            if (TR) { co = row / up; ph = (row - co * up) - up / 2; }
            cor[r] = row < rows_total ? co : -1;
            tph[r] = ph;
            bv[r] = bias[min(co, Cout - 1)];
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int q = q0 + wave * 64 + ti * 32 + r32;
            unsigned off[16];
            float rv[16], ov[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = TR ? q * up + tph[r] : q;
                off[r] = (cor[r] >= 0 && t >= 0 && t < T_out) ? (unsigned)(cor[r] * T_out + t) * 4u : 0x80000000u;
            }
            if (resid) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_res, (int)off[r], 0, 0));
            }
            if (accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ov[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_out, (int)off[r], 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[ri][ti][r] + bv[r];
                if (resid) v += rv[r];
                v *= out_scale;
                if (accumulate) v += ov[r];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, (int)off[r], 0, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K11x, the x2 up-samplers (stages 2 and 3: 128 -> 64 and 64 -> 32 channels, kernel 4, stride 2) as a STREAM (round 5).  SURVEY 8(a)
// puts these two on the HBM roof (arithmetic intensity 64 / 32 flop per byte against a ridge of ~52 for the six-term form); the generic
// kernel above keeps ONE 16-channel window (16.5 KB) per workgroup in flight, behind a barrier, and its epilogue stores every other
// float (the two phases of a channel are different accumulator registers).  Here:
//   * no LDS and no barrier on the data path: the polyphase form has only two taps (input times q and q - 1), so a lane loads the 8
//     channels of its k-half at its own column directly into registers (the channel step rides in the SCALAR offset of the buffer
//     load: one vector offset per (column block, tap)), splits them in registers (LeakyReLU fused) and uses them as the MFMA B
//     fragment; the tap at q - 1 is loaded again (the same lines);
//   * a wave is an independent worker over (item, 64-column block) units of its workgroup's 64-row tile; the next chunk's loads are
//     issued before a chunk's MFMAs and the NEXT UNIT's first chunk before the epilogue (loads fly while a unit's stores drain);
//   * all weights of the row tile (49 / 98 KB split) are loaded into LDS once per workgroup (persistent: 256 workgroups walk all units);
//   * the two phases of an output channel sit in adjacent accumulator registers of one lane (rows co * 2 + phase): stored as ONE
//     8-byte store, a wave's store instruction writes two runs of 256 contiguous bytes.
// The contraction order (chunk, tap, the six piece products smallest first) is the generic kernel's: results are BIT-IDENTICAL to it
// (tests/test_kernels_gpu.py::test_up2_stream_equals_generic_x3).
// Measured (profiles/r05/vocoder_notes.md, B = 32, alternated with the generic kernel in one process): stage 3 0.62-0.64 ms against
// 0.68-0.70 (3.4-3.5 TB/s of algorithmic bytes = 0.43-0.44 of the HBM peak), stage 2 0.96 against 1.00-1.01.  Ablation builds of THIS
// kernel say where the rest is: without stores 0.48 / 0.81 ms, without loads 0.39 / 0.67, without split + MFMA 0.60 (stage 3) -- a plain
// copy of the same slabs in the same 128-byte / 256-byte runs takes 0.41 ms (tools/run_granularity.hip: 5.3 TB/s whatever the run
// length), so neither the bytes nor the run length is the bound; it is how many loads, stores and dependent waits one wave threads
// through one in-order vmcnt counter.  Tried and rejected: 32-column units with a ring of four register windows (three chunks ahead:
// MORE waiting, 0.75 ms), three waves per SIMD (168 VGPRs: 25-38 spilled dwords), non-temporal loads / stores (+14 ... +34 %),
// 8-byte-aligned pair stores (-4 %, needs a cross-lane exchange) and no second tap load (-5 %).
// ABL (diagnostic builds only, -DVV_UP2_DIAG + VV_UP2_ABLATE in the environment): 1 = no stores, 2 = no loads, 3 = no split / MFMA,
// 4 = non-temporal loads, 5 = non-temporal stores, 6 = both, 7 = pair stores moved to an 8-byte boundary (wrong data: timing only),
// 8 = the tap at q - 1 not loaded (copied from the tap at q: wrong data, timing only), 9 = 7 + 8
template <int NCH, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void up2_stream_x3_kernel(
    const float* __restrict__ in, const u32x4* __restrict__ Wb, const float* __restrict__ bias, float* __restrict__ out, int rows_pad, int T_in,
    int T_out, int Cout, float pre_slope, float out_scale, const int* __restrict__ len_in, int n_qb, int n_rt, int n_units) {
    constexpr int Cin = NCH * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* wsl = (u32x4*)smem;                         // [NCH][2 taps][3 pieces][2 octets][64 rows]
    const int id = blockIdx.x;
    const int rt = (id >> 3) % n_rt;                   // workgroups 8 apart share an XCD: the row tiles of one window walk it together
    const int slot = ((id >> 3) / n_rt) * 8 + (id & 7);
    const int n_slots = gridDim.x / n_rt;
    const int r0 = rt * 64;
    for (int i = threadIdx.x; i < NCH * 768; i += 512) wsl[i] = Wb[(size_t)(i >> 6) * rows_pad + r0 + (i & 63)];
    __syncthreads();                                   // the only barrier: from here on every wave runs by itself
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int stride = n_slots * 8;
    int u = slot * 8 + wave;
    if (u >= n_units) return;

    typedef float xbuf_t[2][2][8];                     // [time tile][tap][channel of the k-half]
    xbuf_t xa, xb;
    auto issue = [&](xbuf_t& x, int unit, int ch) {
        const int b = unit / n_qb, q0 = (unit - b * n_qb) * 64;
        const int lin = len_in ? min(len_in[b], T_in) : T_in;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(in + (size_t)b * Cin * T_in), 0, Cin * T_in * 4, 0x00020000);
        const int c0 = ch * 16 + 8 * h;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tap = 0; tap < 2; ++tap) {
                if ((ABL == 8 || ABL == 9) && tap == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[ti][1][j] = x[ti][0][j];
                    continue;
                }
                const int time = q0 + ti * 32 + r32 - tap;
                // one vector offset per (time tile, tap); the channel step j * T_in goes into the SCALAR offset, which the range check
                // ignores: an invalid column keeps its offset past num_records for all 8 loads and reads zero
                const unsigned off = (time >= 0 && time < lin) ? (unsigned)(c0 * T_in + time) * 4u : 0x80000000u;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    x[ti][tap][j] = ABL == 2 ? __uint_as_float(off + j)
                                             : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, j * T_in * 4, (ABL == 4 || ABL == 6) ? 2 : 0));
            }
    };
    f32x16 acc[2][2];
    auto compute = [&](const xbuf_t& x, int ch) {
        if (ABL == 3) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int kw = 0; kw < 2; ++kw)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[kw][ti][j] += x[ti][kw][j];
            return;
        }
#pragma unroll
        for (int kw = 0; kw < 2; ++kw) {
            bf16x8 a[2][3], xf[2][3];
#pragma unroll
            for (int ri = 0; ri < 2; ++ri)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) a[ri][pc] = __builtin_bit_cast(bf16x8, wsl[(((ch * 2 + kw) * 3 + pc) * 2 + h) * 64 + ri * 32 + r32]);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                u32x4 wh, wm, wl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned hh, mm, ll;
                    lrelu_split3_pair(x[ti][kw][2 * j], x[ti][kw][2 * j + 1], pre_slope, hh, mm, ll);
                    wh[j] = hh; wm[j] = mm; wl[j] = ll;
                }
                xf[ti][0] = __builtin_bit_cast(bf16x8, wh); xf[ti][1] = __builtin_bit_cast(bf16x8, wm); xf[ti][2] = __builtin_bit_cast(bf16x8, wl);
            }
#define VV_UP2_TERM(pa, pb)                                                                                                 \
    _Pragma("unroll") for (int ri = 0; ri < 2; ++ri) _Pragma("unroll") for (int ti = 0; ti < 2; ++ti)                         \
        acc[ri][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ri][pa], xf[ti][pb], acc[ri][ti], 0, 0, 0)
            VV_UP2_TERM(0, 2); VV_UP2_TERM(2, 0); VV_UP2_TERM(1, 1); VV_UP2_TERM(0, 1); VV_UP2_TERM(1, 0); VV_UP2_TERM(0, 0);
#undef VV_UP2_TERM
        }
    };

    issue(xa, u, 0);
    while (true) {
        const int un = u + stride;                     // wave-uniform
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ri][ti][r] = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ch += 2) {
            issue(xb, u, ch + 1);
            compute(xa, ch);
            if (ch + 2 < NCH) issue(xa, u, ch + 2);
            else if (un < n_units) issue(xa, un, 0);   // the next unit's first window is in flight while this unit's stores drain
            compute(xb, ch + 1);
        }
        // ---- epilogue: D[row_local = (r & 3) + 8 (r >> 2) + 4 h][column r32]; rows (2 co, 2 co + 1) = the two phases of channel co,
        // output times 2 q - 1 and 2 q: registers (r, r + 1), r even, are adjacent samples of one channel
        const int b = u / n_qb, q0 = (u - b * n_qb) * 64;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)b * Cout * T_out), 0, Cout * T_out * 4, 0x00020000);
#pragma unroll
        for (int ri = 0; ri < 2; ++ri)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const int q = q0 + ti * 32 + r32;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int co = (r0 + ri * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) >> 1;
                    const float bv = bias[co];
                    float v0 = acc[ri][ti][r] + bv, v1 = acc[ri][ti][r + 1] + bv;
                    v0 *= out_scale; v1 *= out_scale;
                    const unsigned e = (unsigned)(co * T_out + 2 * q - 1);
                    if (ABL == 1) {
                        if (v0 == 123.456f && v1 == 654.321f) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rs_out, (int)(e * 4u), 0, 0);
                    } else if (q >= 1 && q < T_in) {
                        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                        const u32x2 pv = {__float_as_uint(v0), __float_as_uint(v1)};
                        __builtin_amdgcn_raw_buffer_store_b64(pv, rs_out, (int)((e + ((ABL == 7 || ABL == 9) ? 1u : 0u)) * 4u), 0, (ABL == 5 || ABL == 6) ? 2 : 0);
                    } else if (q == 0) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), rs_out, (int)((e + 1u) * 4u), 0, 0);
                    } else if (q == T_in) {
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), rs_out, (int)(e * 4u), 0, 0);
                    }
                }
            }
        if (un >= n_units) break;
        u = un;
    }
}

struct X3Setup {                                       // dynamic LDS above 64 KiB needs the attribute once per kernel and device
    std::mutex mu;
    std::atomic<unsigned long long> done{0};
    hipError_t ensure(const void* kern, int lds) {
        int dev = 0;
        if (hipError_t he = hipGetDevice(&dev); he != hipSuccess) return he;
        if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
        if (!(done.load(std::memory_order_acquire) >> dev & 1)) {
            std::lock_guard<std::mutex> lk(mu);
            if (!(done.load(std::memory_order_relaxed) >> dev & 1)) {
                if (hipError_t he = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds); he != hipSuccess) return he;
                done.fetch_or(1ull << dev, std::memory_order_release);
            }
        }
        return hipSuccess;
    }
};

template <int KW, bool TR, int RT, int TG, int NWV, int OCC>
hipError_t launch_x3_t(const vv_conv_args* a, hipStream_t st) {
    constexpr int NT = 64 * NWV, VR = (NWV / 4) * RT * 32;
    constexpr int WSLAB = (TG * 3 * VR * 2 + NT - 1) / NT * NT;
    const size_t lds_max = (size_t)(3 * (XT + (TR ? 1 : 5 * (KW - 1))) * 2 + 2 * WSLAB) * 16;      // the attribute is set for the largest dilation
    const size_t lds = (size_t)(3 * (XT + (TR ? 1 : a->dil * (KW - 1))) * 2 + 2 * WSLAB) * 16;
    static X3Setup setup;
    auto kern = conv_x3_kernel<KW, TR, RT, TG, NWV, OCC>;
    if (hipError_t he = setup.ensure((const void*)kern, (int)lds_max); he != hipSuccess) return he;
    const int q_total = TR ? a->T_in + 1 : a->T_out;
    const int n_tt = (q_total + XT - 1) / XT, n_rt = (a->rows_total + VR - 1) / VR;
    const long long wins8 = ((long long)n_tt * a->B + 7) / 8;
    if (wins8 * 8 * n_rt > 0x7fffffffLL) return hipErrorInvalidValue;
    kern<<<dim3((unsigned)(wins8 * 8 * n_rt)), NT, lds, st>>>(a->in, (const u32x4*)a->W_x3, a->bias, a->out, a->resid, a->Cin, a->rows_total, a->rows_pad,   // 1-D walk, id % 8 = XCD group
                                                              a->T_in, a->T_out, a->Cout, a->dil, a->up, a->pre_slope, a->out_scale, a->accumulate, a->len_in,
                                                              n_tt, n_rt, n_tt * a->B);
    return hipGetLastError();
}
template <int NCH, int ABL = 0>
hipError_t launch_up2_stream(const vv_conv_args* a, hipStream_t st) {
#ifdef VV_UP2_DIAG
    if (ABL == 0) {
        const char* e = getenv("VV_UP2_ABLATE");
        switch (e ? atoi(e) : 0) {
            case 1: return launch_up2_stream<NCH, 1>(a, st);
            case 2: return launch_up2_stream<NCH, 2>(a, st);
            case 3: return launch_up2_stream<NCH, 3>(a, st);
            case 4: return launch_up2_stream<NCH, 4>(a, st);
            case 5: return launch_up2_stream<NCH, 5>(a, st);
            case 6: return launch_up2_stream<NCH, 6>(a, st);
            case 7: return launch_up2_stream<NCH, 7>(a, st);
            case 8: return launch_up2_stream<NCH, 8>(a, st);
            case 9: return launch_up2_stream<NCH, 9>(a, st);
            default: break;
        }
    }
#endif
    static X3Setup setup;
    auto kern = up2_stream_x3_kernel<NCH, ABL>;
    const int lds = NCH * 768 * 16;
    if (hipError_t he = setup.ensure((const void*)kern, lds); he != hipSuccess) return he;
    const int n_qb = (a->T_in + 1 + 63) / 64, n_rt = a->rows_total / 64;
    const long long units = (long long)n_qb * a->B;
    if (units > 0x7fffffffLL) return hipErrorInvalidValue;
    int grid = 256 / (8 * n_rt) * (8 * n_rt);          // one persistent workgroup per CU, a whole number of (XCD group, row tile) sets
    while (grid > 8 * n_rt && (long long)(grid / n_rt - 8) * 8 >= units) grid -= 8 * n_rt;      // short inputs: no idle workgroups loading weights
    kern<<<dim3((unsigned)grid), 512, lds, st>>>(a->in, (const u32x4*)a->W_x3, a->bias, a->out, a->rows_pad, a->T_in, a->T_out, a->Cout, a->pre_slope,
                                                 a->out_scale, a->len_in, n_qb, n_rt, (int)units);
    return hipGetLastError();
}

// the streaming form serves the x2 up-samplers (no residual, no accumulate) with 64 or 128 input channels
static bool up2_stream_fits(const vv_conv_args* a) {
    return a->transposed && a->up == 2 && a->KW == 2 && !a->resid && !a->accumulate && (a->Cin == 64 || a->Cin == 128) && a->rows_total % 64 == 0 &&
           a->rows_total >= 64 && a->rows_total <= 128 && a->wg_rows != -1;
}

template <int KW, bool TR>
hipError_t launch_x3(const vv_conv_args* a, hipStream_t st) {
    if (TR && up2_stream_fits(a)) return a->Cin == 64 ? launch_up2_stream<4>(a, st) : launch_up2_stream<8>(a, st);
    if (a->rows_total <= 32) return launch_x3_t<KW, TR, 1, 2, 4, 3>(a, st);      // narrow stage: no zero-padded MFMA rows; 42 KiB of LDS: 3 workgroups per CU
    // 64-row workgroups of 4 waves by default: 128-row workgroups of 8 waves (the window split once for twice the rows) are 1-3 %
    // faster per launch on the k = 3 shapes but 1.2 % slower in the decode (one resident workgroup per CU: nothing overlaps its epilogue)
    // 3 workgroups per CU (2 taps per phase, <= 168 VGPRs, taps not unrolled) where the kernel fits with at most a few spilled dwords:
    // k = 3, k = 7 and the transposed form are 4-11 % faster per launch; k = 11 keeps 2 workgroups per CU and 4 taps per phase (at 168
    // VGPRs it spills 21-65 dwords and is 10 % slower): profiles/r02/voc_x3_conv_shapes_occ3*.txt
    // the 128-row form loads weight rows up to n_rt * 128 - 1 of each [ko] slab: only when the padded slab has them (rows_pad is a
    // multiple of 64, not necessarily of 128 -- e.g. 192 rows: the 64-row form is taken)
    const bool wide = a->wg_rows == 128 && a->rows_pad % 128 == 0;
    if (!wide && (KW <= 7 || TR)) return launch_x3_t<KW, TR, 2, 2, 4, 3>(a, st);
    if (a->rows_total <= 64 || !wide) return launch_x3_t<KW, TR, 2, (4 < KW ? 4 : KW), 4, 2>(a, st);
    return launch_x3_t<KW, TR, 2, 2, 8, 2>(a, st);
}

}  // namespace

size_t vvk_conv_split_bytes(int Cin_pad, int KW, int rows_pad) { return (size_t)((Cin_pad + 15) / 16) * KW * 3 * rows_pad * 16 * 2; }

int vvk_conv_split_weights(const float* Wt, int Cin_pad, int KW, int rows_pad, void* Wb, hipStream_t st, const char** err) {
    if (!Wt || !Wb || Cin_pad < 1 || KW < 1 || rows_pad < 1 || rows_pad % 64 || ((uintptr_t)Wb % 16)) { *err = "conv_split_weights: bad arguments"; return -22; }
    const int chunks = (Cin_pad + 15) / 16;
    const size_t n = (size_t)chunks * KW * rows_pad * 16;
    const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
    split_weights_kernel<<<grid, 256, 0, st>>>(Wt, (uint16_t*)Wb, Cin_pad, KW, rows_pad, chunks);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}

// Same contract as vvk_conv (shapes already validated there); a->W_x3 = the split slab of a->W.
int vvk_conv_x3(const vv_conv_args* a, hipStream_t st, const char** err) {
    if (!a->W_x3 || ((uintptr_t)a->W_x3 % 16)) { *err = "conv: split weight slab missing or misaligned"; return -22; }
    hipError_t he = hipSuccess;
    if (a->transposed) {
        he = launch_x3<2, true>(a, st);
    } else {
        switch (a->KW) {
            case 3: he = launch_x3<3, false>(a, st); break;
            case 7: he = launch_x3<7, false>(a, st); break;
            case 11: he = launch_x3<11, false>(a, st); break;
            default: *err = "conv: kernel width must be 3, 7 or 11"; return -22;
        }
    }
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}
