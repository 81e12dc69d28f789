// K10-K13: vocoder kernels (fp32), channel-major activations [B][C][T] (T contiguous).
//
// conv1d / ConvTranspose1d are ONE implicit-GEMM kernel on the exact-f32 MFMA
// (v_mfma_f32_32x32x2_f32):  D[row][col = time] = sum_k Wt[k][row] * X[k][time + off(k)].
//   * Conv1d (KW taps, dilation d):   row = co,           off(kw) = kw*d - d*(KW-1)/2
//   * ConvTranspose1d (stride u, kernel 2u, pad u/2) in polyphase form, no zero-stuffing:
//        row = co*u + p,  taps j in {0,1}: in[q - j] * W[ci][co][p + j*u],  out time = q*u + p - u/2
// A workgroup owns 64 rows x 256 time steps; the K loop walks input channels 8 at a time, staging
// the 8-channel input window (with the fused input LeakyReLU and zero fill outside [0,len)) and
// the matching weight slab in LDS.  MFMA lane half h takes channel 2j+h, so both halves read
// conflict-free consecutive-time / consecutive-row LDS words.  Epilogue fuses bias, residual add,
// the 1/3 MRF average and accumulation across the three resblocks; stores are 128-byte segments.
//
// conv_post (K13): 32->1 channel k=7 conv + tanh + int16 quantisation, HBM-bound, LDS window.
#include "vv_common.h"
#include "vv_kernels.h"

namespace {

constexpr int VT = 256;       // time steps per workgroup

// LeakyReLU (a two-instruction max form and an interior staging fast path measured slower: profiles/r02/vocoder_notes.md)
__device__ __forceinline__ float lrelu(float x, float slope) { return x >= 0.f ? x : x * slope; }

// KW: taps.  TRANSPOSED: polyphase ConvTranspose (KW must be 2).  VCI: input channels per K chunk.
// RT: 32-row MFMA tiles per wave (2 -> 64 rows per workgroup, 1 -> 32 rows for the narrow last stage).
template <int KW, bool TRANSPOSED, int VCI, int RT>
__global__ __launch_bounds__(256, 4) void conv_mfma_kernel(const float* __restrict__ in, const float* __restrict__ Wt /*[Cin_pad][KW][rows_pad]*/,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           const float* __restrict__ resid, int Cin, int rows_total,
                                                           int rows_pad, int T_in, int T_out, int Cout, int dil, int up,
                                                           float pre_slope, float out_scale, int accumulate,
                                                           const int* __restrict__ len_in) {
#pragma clang fp contract(off)      // the epilogue's add / multiply sequence is part of the bit-exact contract with mrf_pair_kernel
    constexpr int VR = RT * 32;
    // halo: conv reads time + kw*dil - left; transposed reads q and q-1.  The staged window starts at the 16-byte aligned
    // time q0 - L4 (L4 = left rounded up to 4), so whole float4s are loaded and the reads shift by L4 - left.
    const int left = TRANSPOSED ? 1 : dil * (KW - 1) / 2;
    const int span = TRANSPOSED ? 1 : dil * (KW - 1);
    const int L4 = (left + 3) & ~3;
    const int shift = L4 - left;
    const int xw4 = (VT + span + shift + 3) >> 2;     // float4s per channel
    const int xw_pad = xw4 * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;                         // [VCI][xw_pad]
    float* ws = xs + VCI * xw_pad;                    // [VCI][KW][VR]

    const int b = blockIdx.z;
    const int r0 = blockIdx.y * VR;
    const int q0 = blockIdx.x * VT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int lin = len_in ? max(0, min(len_in[b], T_in)) : T_in;
    const float* inb = in + (size_t)b * Cin * T_in;
    const bool vec_ok = (T_in & 3) == 0 && ((uintptr_t)in & 15) == 0;

    f32x16 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Latency hiding is by occupancy: small channel chunks (VCI 4 for k = 7 / 11, 8 for k = 3, 16 for the transposed form,
    // measured best of 2..32) keep the LDS footprint at 8-16 KiB so up to 8 workgroups share a CU; a register-prefetch
    // pipeline was measured slower because its staging maps halve the occupancy.
    for (int c0 = 0; c0 < Cin; c0 += VCI) {
        __syncthreads();
        // ---- stage the input window: xs[c][i] = lrelu(in[c0+c][q0 - L4 + i]), zero outside [0, lin); wave w takes
        // channels w, w+4, ...; aligned float4 loads wherever the four samples are inside the row
        for (int c = wave; c < VCI; c += 4) {
            const bool live = c0 + c < Cin;
            const float* row = inb + (size_t)(c0 + c) * T_in;
            for (int i4 = lane; i4 < xw4; i4 += 64) {
                const int pos = q0 - L4 + i4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live) {
                    if (vec_ok && pos >= 0 && pos + 3 < lin) {
                        v = *(const float4*)(row + pos);
                    } else {
                        if (pos >= 0 && pos < lin) v.x = row[pos];
                        if (pos + 1 >= 0 && pos + 1 < lin) v.y = row[pos + 1];
                        if (pos + 2 >= 0 && pos + 2 < lin) v.z = row[pos + 2];
                        if (pos + 3 >= 0 && pos + 3 < lin) v.w = row[pos + 3];
                    }
                    v.x = lrelu(v.x, pre_slope); v.y = lrelu(v.y, pre_slope); v.z = lrelu(v.z, pre_slope); v.w = lrelu(v.w, pre_slope);
                }
                *(float4*)(xs + c * xw_pad + i4 * 4) = v;
            }
        }
        // ---- stage the weight slab: ws[c][kw][r] = Wt[c0+c][kw][r0+r]   (padded, no masks)
        for (int i = threadIdx.x; i < VCI * KW * (VR / 4); i += 256) {
            const int r4 = i % (VR / 4), ck = i / (VR / 4);
            *(float4*)(ws + ck * VR + r4 * 4) = *(const float4*)(Wt + ((size_t)c0 * KW + ck) * rows_pad + r0 + r4 * 4);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < VCI / 2; ++j) {
            const int c = 2 * j + h;                   // lane half h takes channel 2j + h
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const int off = (TRANSPOSED ? (left - kw) : kw * dil) + shift;          // window index = local time + off
                float a[RT], x[2];
#pragma unroll
                for (int i = 0; i < RT; ++i) a[i] = ws[(c * KW + kw) * VR + i * 32 + r32];
#pragma unroll
                for (int i = 0; i < 2; ++i) x[i] = xs[c * xw_pad + wave * 64 + i * 32 + r32 + off];
#pragma unroll
                for (int ri = 0; ri < RT; ++ri)
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti)
                        acc[ri][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ri], x[ti], acc[ri][ti], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: D[row_local = (reg&3) + 8(reg>>2) + 4h][time_local = r32].  Branch-free and batched (as in vv_vocoder_x3.hip):
    // an element outside the rows / the time range carries an offset past num_records (loads return zero, stores are dropped), so
    // the residual and accumulate loads of a tile are all in flight together.  The arithmetic -- (acc + bias [+ resid]) * scale
    // [+ out], separate roundings -- is unchanged: the bit-exact contract with mrf_pair_kernel holds.
    const size_t out_elems = (size_t)Cout * T_out;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)b * out_elems), 0, (int)min(out_elems * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)((resid ? resid : out) + (size_t)b * out_elems), 0, (int)min(out_elems * 4, (size_t)0x7fffffff), 0x00020000);
    if constexpr (TRANSPOSED) {                        // 4 launches per decode, and the phase arithmetic of a batch spills here: element-wise
        float* outb = out + (size_t)b * Cout * T_out;
#pragma unroll
        for (int ri = 0; ri < RT; ++ri)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const int q = q0 + wave * 64 + ti * 32 + r32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + ri * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row >= rows_total) continue;
                    const int co = row / up, t = q * up + (row - co * up) - up / 2;
                    if (t < 0 || t >= T_out) continue;
                    float v = acc[ri][ti][r] + bias[co];
                    const size_t o = (size_t)co * T_out + t;
                    if (resid) v += resid[(size_t)b * out_elems + o];
                    v *= out_scale;
                    if (accumulate) v += outb[o];
                    outb[o] = v;
                }
            }
        return;
    }
#pragma unroll
    for (int ri = 0; ri < RT; ++ri)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int q = q0 + wave * 64 + ti * 32 + r32;
#pragma unroll
            for (int rh = 0; rh < 16; rh += 4) {       // four accumulator registers at a time: this kernel hides latency by occupancy (<= 106 VGPRs)
                unsigned off[4];
                float bv[4], rv[4], ov[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = rh + k;
                    const int row = r0 + ri * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    int co = row, t = q;
                    if (TRANSPOSED) { co = row / up; t = q * up + (row - co * up) - up / 2; }
                    off[k] = (row < rows_total && t >= 0 && t < T_out) ? (unsigned)(co * T_out + t) * 4u : 0x80000000u;
                    bv[k] = bias[min(co, Cout - 1)];
                }
                if (resid) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) rv[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_res, (int)off[k], 0, 0));
                }
                if (accumulate) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) ov[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_out, (int)off[k], 0, 0));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = acc[ri][ti][rh + k] + bv[k];
                    if (resid) v += rv[k];
                    v *= out_scale;
                    if (accumulate) v += ov[k];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, (int)off[k], 0, 0);
                }
            }
        }
}

// ---- K12 fused: one (kernel, dilation) pair of an MRF resblock in ONE launch,
//        out = [accumulate ? out : 0] + scale * ( conv2(lrelu(conv1(lrelu(y)))) + y ),
// conv1 dilated by `dil`, conv2 undilated, both C -> C with C = 32 or 64 (the whole channel dimension is one row tile, so the
// intermediate never leaves the CU): a workgroup computes a 128-column window of the intermediate t1 into LDS (bias added,
// second LeakyReLU applied, zero outside [0, len) exactly as the unfused conv2 zero-fills its staged window), then runs conv2
// straight out of that LDS image and writes the central 128 - E columns (E = KW - 1 rounded up to 4).  Both products keep the
// unfused kernel's contraction order (channel pairs ascending, taps ascending inside a pair) and epilogue arithmetic, so the
// result is BIT-IDENTICAL to vv_conv1d(conv1) followed by vv_conv1d(conv2, resid = y).  HBM traffic per pair: y read once
// (+ halo), out written once -- 2 tensor passes instead of 5.
// CT = 32-column MFMA tiles per wave: the window is FT = 128 * CT columns (C = 32 takes CT = 2: same accumulator registers as
// C = 64 with CT = 1, half the halo recompute).
template <int KW, int RT, int VCI, int CT>
__global__ __launch_bounds__(256, 2) void mrf_pair_kernel(const float* __restrict__ y, const float* __restrict__ W1, const float* __restrict__ b1,
                                                          const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ out,
                                                          int C, int rows_pad, int T, int dil, float slope, float out_scale, int accumulate,
                                                          const int* __restrict__ len_in) {
#pragma clang fp contract(off)      // same separate add / multiply roundings as conv_mfma_kernel's epilogue
    constexpr int VR = RT * 32;
    constexpr int FT = 128 * CT;                   // intermediate columns per workgroup
    constexpr int half = (KW - 1) / 2;
    constexpr int E = (KW - 1 + 3) & ~3;           // columns of the window that are halo only
    constexpr int VT2 = FT - E;                    // output columns per workgroup
    constexpr int P = E - half;                    // the t1 window starts P columns before the first output column
    constexpr int T1P = FT + 4;                    // LDS row pitch of the intermediate
    const int left1 = dil * (KW - 1) / 2, span1 = dil * (KW - 1);
    const int A = (P + left1 + 3) & ~3;            // staged input window starts at the 16-byte aligned time q0 - A
    const int shift = A - (P + left1);
    const int xw4 = (FT + span1 + shift + 3) >> 2;
    const int xw_pad = xw4 * 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* t1s = (float*)smem;                     // [VR][T1P]
    float* xs = t1s + VR * T1P;                    // [VCI][xw_pad]
    float* ws = xs + VCI * xw_pad;                 // [VCI][KW][VR]

    const int b = blockIdx.z;
    const int q0 = blockIdx.x * VT2;               // first output column (multiple of 4)
    const int w0 = q0 - P;                         // time of t1 window column 0
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int lin = len_in ? max(0, min(len_in[b], T)) : T;
    const float* yb = y + (size_t)b * C * T;
    const bool vec_ok = (T & 3) == 0 && ((uintptr_t)y & 15) == 0;

    f32x16 acc[RT][CT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    };
    auto stage_w = [&](const float* Wt, int c0) {
        for (int i = threadIdx.x; i < VCI * KW * (VR / 4); i += 256) {
            const int r4 = i % (VR / 4), ck = i / (VR / 4);
            *(float4*)(ws + ck * VR + r4 * 4) = *(const float4*)(Wt + ((size_t)c0 * KW + ck) * rows_pad + r4 * 4);
        }
    };

    // ---------------- conv1 (dilated): t1 window column i <-> time w0 + i
    zero_acc();
    for (int c0 = 0; c0 < C; c0 += VCI) {
        __syncthreads();
        for (int c = wave; c < VCI; c += 4) {
            const float* row = yb + (size_t)(c0 + c) * T;
            for (int i4 = lane; i4 < xw4; i4 += 64) {
                const int pos = q0 - A + i4 * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (vec_ok && pos >= 0 && pos + 3 < lin) {
                    v = *(const float4*)(row + pos);
                } else {
                    if (pos >= 0 && pos < lin) v.x = row[pos];
                    if (pos + 1 >= 0 && pos + 1 < lin) v.y = row[pos + 1];
                    if (pos + 2 >= 0 && pos + 2 < lin) v.z = row[pos + 2];
                    if (pos + 3 >= 0 && pos + 3 < lin) v.w = row[pos + 3];
                }
                v.x = lrelu(v.x, slope); v.y = lrelu(v.y, slope); v.z = lrelu(v.z, slope); v.w = lrelu(v.w, slope);
                *(float4*)(xs + c * xw_pad + i4 * 4) = v;
            }
        }
        stage_w(W1, c0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < VCI / 2; ++j) {
            const int c = 2 * j + h;
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                float a[RT], x[CT];
#pragma unroll
                for (int ri = 0; ri < RT; ++ri) a[ri] = ws[(c * KW + kw) * VR + ri * 32 + r32];
#pragma unroll
                for (int ti = 0; ti < CT; ++ti) x[ti] = xs[c * xw_pad + (wave * CT + ti) * 32 + r32 + kw * dil + shift];
#pragma unroll
                for (int ri = 0; ri < RT; ++ri)
#pragma unroll
                    for (int ti = 0; ti < CT; ++ti) acc[ri][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ri], x[ti], acc[ri][ti], 0, 0, 0);
            }
        }
    }
    // t1 = conv1 + bias (the unfused conv1 epilogue: no residual, scale 1), then what the unfused conv2 does while staging
    // its window: zero outside [0, len), LeakyReLU.  D[row = (reg&3) + 8(reg>>2) + 4h][column = r32]
#pragma unroll
    for (int ti = 0; ti < CT; ++ti) {
        const int col = (wave * CT + ti) * 32 + r32;
        const int t = w0 + col;
        const bool inside = t >= 0 && t < lin;
#pragma unroll
        for (int ri = 0; ri < RT; ++ri)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = ri * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = 0.f;
                if (inside && row < C) { v = acc[ri][ti][r] + b1[row]; v *= 1.0f; v = lrelu(v, slope); }
                t1s[row * T1P + col] = v;
            }
    }
    // ---------------- conv2 (undilated) out of the LDS image: output column c <-> time w0 + half + c, reads t1 columns c .. c + KW - 1
    zero_acc();
    for (int c0 = 0; c0 < C; c0 += VCI) {
        __syncthreads();                            // first pass: t1s complete; later passes: ws free again
        stage_w(W2, c0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < VCI / 2; ++j) {
            const int c = c0 + 2 * j + h;
            const int cl = 2 * j + h;
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                float a[RT], x[CT];
#pragma unroll
                for (int ri = 0; ri < RT; ++ri) a[ri] = ws[(cl * KW + kw) * VR + ri * 32 + r32];
#pragma unroll
                for (int ti = 0; ti < CT; ++ti) x[ti] = t1s[c * T1P + min((wave * CT + ti) * 32 + r32 + kw, FT - 1)];   // past the window: discarded outputs only
#pragma unroll
                for (int ri = 0; ri < RT; ++ri)
#pragma unroll
                    for (int ti = 0; ti < CT; ++ti) acc[ri][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ri], x[ti], acc[ri][ti], 0, 0, 0);
            }
        }
    }
    // epilogue as in conv_mfma_kernel: branch-free, batched buffer loads / stores (an element outside the rows or the window's
    // central columns carries an offset past num_records); same arithmetic, same order: still bit-identical to the two launches
    const size_t slab = (size_t)C * T;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)b * slab), 0, (int)min(slab * 4, (size_t)0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)yb, 0, (int)min(slab * 4, (size_t)0x7fffffff), 0x00020000);
#pragma unroll
    for (int ti = 0; ti < CT; ++ti) {
        const int cc = (wave * CT + ti) * 32 + r32;  // conv2 output column
        const int t = w0 + half + cc;
        const bool col_ok = cc >= P - half && cc < P - half + VT2 && t < T;
#pragma unroll
        for (int ri = 0; ri < RT; ++ri)
#pragma unroll
            for (int rh = 0; rh < 16; rh += 8) {
                unsigned off[8];
                float bv[8], yv[8], ov[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int r = rh + k;
                    const int row = ri * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    off[k] = (col_ok && row < C) ? (unsigned)(row * T + t) * 4u : 0x80000000u;
                    bv[k] = b2[min(row, C - 1)];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) yv[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_y, (int)off[k], 0, 0));
                if (accumulate) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) ov[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_out, (int)off[k], 0, 0));
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float v = acc[ri][ti][rh + k] + bv[k];
                    v += yv[k];
                    v *= out_scale;
                    if (accumulate) v += ov[k];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, (int)off[k], 0, 0);
                }
            }
    }
}

// ---- K13: conv_post (C -> 1, k = 7) on lrelu(x), tanh, *32767, clip, truncate to int16.  HBM-bound: 4*C bytes read + 2 written
// per sample, and the kernel is written as a STREAM: no LDS, no barrier.  A lane owns 4 consecutive output samples and loads
// exactly its own aligned float4 of every channel (one coalesced 1 KiB load per wave and channel, 8 channels = 8 loads in
// flight per lane before the first use); the 3 + 3 halo samples of the k = 7 window come from the neighbouring lanes by DPP
// whole-wave shifts (wave_shr:1 / wave_shl:1), so lanes 0 and 63 of a wave are halo lanes that only load -- a wave produces
// 62 x 4 = 248 samples and the waves of a row overlap by two float4 (3 % of the bytes, served by L2).  Weights are wave-uniform
// scalar loads.  The FMA order (channel-major, taps 0..6) is the one of the round-1 LDS-window kernel: PCM is bit-identical.
// (That kernel staged 8 channels x 1,032 samples through LDS with two barriers per pass: one pass of loads in flight per
// workgroup, 0.34-0.40 ms at the headline batch = 0.34-0.41 of HBM; profiles/r03/vocoder_notes.md.)
__device__ __forceinline__ float wave_prev(float v) {      // lane i <- lane i - 1 (lane 0: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_next(float v) {      // lane i <- lane i + 1 (lane 63: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
constexpr int POST_WAVE_OUT = 62 * 4;       // samples produced per wave
// VEC: rows are 16-byte aligned (T % 4 == 0): one buffer_load_dwordx4 per lane and channel; otherwise four dword loads.  Loads go
// through a buffer resource of the row's VALID bytes (len_in samples): the left halo of a row's first wave (negative offset) and
// everything past the valid length read as zero from the hardware range check -- the main path has no branch and no exec mask.
template <int KW, bool VEC, int CB /* channels loaded ahead of their use; divides C */>
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ in, const float* __restrict__ w /*[C][KW]*/,
                                                        float bias, int16_t* __restrict__ pcm, int ld_pcm, float* __restrict__ wave_f32,
                                                        int C, int T, float pre_slope, const int* __restrict__ len_in) {
    static_assert(KW == 7, "the window is 3 + 4 + 3 samples");
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t = (blockIdx.x * 4 + wave) * POST_WAVE_OUT + (lane - 1) * 4;       // first of this lane's 4 samples (lane 0: the left halo)
    const int lin = len_in ? max(0, min(len_in[b], T)) : T;
    const float* row = in + (size_t)b * C * T;
    // a float4 that straddles the valid length is cut per element (the range check of a 16-byte load is not relied on for that)
    const bool m1 = t + 1 < lin, m2 = t + 2 < lin, m3 = t + 3 < lin;
    float acc[4] = {bias, bias, bias, bias};
    for (int c0 = 0; c0 < C; c0 += CB) {         // CB = 8: 8 KiB in flight per wave
        float4 v[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(row + (size_t)(c0 + c) * T), 0, lin * 4, 0x00020000);
            if constexpr (VEC) {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, t * 4, 0, 2);      // nt: read once
                v[c].x = __uint_as_float(q[0]);                                    // t >= lin: the whole load is out of range (zeros)
                v[c].y = m1 ? __uint_as_float(q[1]) : 0.f;
                v[c].z = m2 ? __uint_as_float(q[2]) : 0.f;
                v[c].w = m3 ? __uint_as_float(q[3]) : 0.f;
            } else {
                v[c].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, t * 4, 0, 2));
                v[c].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, t * 4 + 4, 0, 2));
                v[c].z = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, t * 4 + 8, 0, 2));
                v[c].w = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, t * 4 + 12, 0, 2));
            }
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const float x4 = lrelu(v[c].x, pre_slope), x5 = lrelu(v[c].y, pre_slope), x6 = lrelu(v[c].z, pre_slope), x7 = lrelu(v[c].w, pre_slope);
            // window xv[1..10] = samples t - 3 .. t + 6 (xv[o + k + 1] multiplies tap k of output o, as in the LDS-window kernel)
            const float xv[12] = {0.f, wave_prev(x5), wave_prev(x6), wave_prev(x7), x4, x5, x6, x7, wave_next(x4), wave_next(x5), wave_next(x6), 0.f};
            const float* wc = w + (c0 + c) * KW;              // wave-uniform: scalar loads
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const float wk = wc[k];
#pragma unroll
                for (int o = 0; o < 4; ++o) acc[o] = fmaf(wk, xv[o + k + 4 - KW / 2], acc[o]);
            }
        }
    }
    if (lane == 0 || lane == 63 || t >= T) return;          // halo lanes, and lanes past the row
    float ys[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) ys[o] = tanhf(acc[o]);
    short4 pk;
    pk.x = (short)fminf(fmaxf(ys[0] * 32767.0f, -32768.0f), 32767.0f);     // truncation toward zero, as an ONNX Cast does
    pk.y = (short)fminf(fmaxf(ys[1] * 32767.0f, -32768.0f), 32767.0f);
    pk.z = (short)fminf(fmaxf(ys[2] * 32767.0f, -32768.0f), 32767.0f);
    pk.w = (short)fminf(fmaxf(ys[3] * 32767.0f, -32768.0f), 32767.0f);
    if (VEC && t + 3 < T && (ld_pcm & 3) == 0) {
        *(short4*)(pcm + (size_t)b * ld_pcm + t) = pk;
        if (wave_f32) *(float4*)(wave_f32 + (size_t)b * T + t) = make_float4(ys[0], ys[1], ys[2], ys[3]);
    } else {
        const short ps[4] = {pk.x, pk.y, pk.z, pk.w};
        for (int o = 0; o < 4 && t + o < T; ++o) {
            pcm[(size_t)b * ld_pcm + t + o] = ps[o];
            if (wave_f32) wave_f32[(size_t)b * T + t + o] = ys[o];
        }
    }
}

// ---- K10: slice generated frames and transpose to channel-major: out[b][c][t] = x[b][ref_len+t][c]
__global__ __launch_bounds__(256) void mel_slice_kernel(const float* __restrict__ x, int N, int n_mel,
                                                        const int* __restrict__ ref_len, const int* __restrict__ seq_len,
                                                        float* __restrict__ out, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int rl = ref_len[b];
    const int tg = max(min(seq_len[b], N) - rl, 0);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        float v = 0.f;
        if (t < tg && c < n_mel) v = x[((size_t)b * N + rl + t) * n_mel + c];
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (c < n_mel && t < T) out[((size_t)b * n_mel + c) * T + t] = tile[tx][i];
    }
}

template <int KW, bool TR, int VCI, int RT>
int launch_conv_t(const vv_conv_args* a, hipStream_t st) {
    constexpr int VR = RT * 32;
    const int span = TR ? 1 : a->dil * (KW - 1);
    const int left = TR ? 1 : a->dil * (KW - 1) / 2;
    const int xw_pad = ((VT + span + (((left + 3) & ~3) - left) + 3) >> 2) * 4;
    const size_t lds = (size_t)(VCI * xw_pad + VCI * KW * VR) * sizeof(float);
    const int q_total = TR ? a->T_in + 1 : a->T_out;
    dim3 grid((q_total + VT - 1) / VT, (a->rows_total + VR - 1) / VR, a->B);
    conv_mfma_kernel<KW, TR, VCI, RT><<<grid, 256, lds, st>>>(a->in, a->W, a->bias, a->out, a->resid, a->Cin, a->rows_total, a->rows_pad,
                                                             a->T_in, a->T_out, a->Cout, a->dil, a->up, a->pre_slope, a->out_scale,
                                                             a->accumulate, a->len_in);
    return 0;
}
template <int KW, bool TR, int VCI>
int launch_conv(const vv_conv_args* a, hipStream_t st) {
    if (a->rows_total <= 32) return launch_conv_t<KW, TR, VCI, 1>(a, st);      // narrow stage: no zero-padded MFMA rows
    return launch_conv_t<KW, TR, VCI, 2>(a, st);
}

}  // namespace

int vvk_conv(const vv_conv_args* a, hipStream_t st, const char** err) {
    if (a->B <= 0 || a->Cin <= 0 || a->Cout <= 0 || a->T_in <= 0 || a->T_out <= 0) { *err = "conv: empty shape"; return -22; }
    if (!(a->pre_slope > 0.f && a->pre_slope <= 1.f)) { *err = "conv: pre_slope must be in (0, 1] (1 = no activation)"; return -22; }
    // input windows and output tiles are addressed through buffer resources with 32-bit byte offsets (out-of-range = dropped)
    if ((size_t)a->Cin * a->T_in * 4 >= ((size_t)1 << 31) || (size_t)a->Cout * a->T_out * 4 >= ((size_t)1 << 31)) {
        *err = "conv: a [C][T] slab of one item reaches 2 GiB; decode fewer frames per call"; return -22;
    }
    if (a->rows_pad % 64 || a->rows_pad < a->rows_total || ((uintptr_t)a->W % 16)) { *err = "conv: weight slab must be padded to 64 rows and 16-byte aligned"; return -22; }
    if (a->transposed) {
        if (a->KW != 2 || a->up < 2 || (a->up & 1) || a->rows_total != a->Cout * a->up || a->T_out != a->T_in * a->up) {
            *err = "conv: transposed form needs kernel = 2*stride, even stride"; return -22;
        }
        if (a->W_x3) return vvk_conv_x3(a, st, err);            // split-bf16 products (vv_vocoder_x3.hip), same contract
        launch_conv<2, true, 16>(a, st);
    } else {
        if (a->rows_total != a->Cout || a->T_out != a->T_in || a->dil < 1 || a->dil > 5) { *err = "conv: bad conv shape (dilation 1..5)"; return -22; }
        if (a->W_x3) return vvk_conv_x3(a, st, err);
        switch (a->KW) {
            case 3: launch_conv<3, false, 8>(a, st); break;
            case 7: launch_conv<7, false, 4>(a, st); break;
            case 11: launch_conv<11, false, 4>(a, st); break;
            default: *err = "conv: kernel width must be 3, 7 or 11"; return -22;
        }
    }
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}

namespace {
template <int KW, int RT, int VCI, int CT>
void launch_mrf(const vv_mrf_args* a, hipStream_t st) {
    constexpr int FT = 128 * CT, E = (KW - 1 + 3) & ~3, VT2 = FT - E, half = (KW - 1) / 2, P = E - half;
    const int left1 = a->dil * (KW - 1) / 2, span1 = a->dil * (KW - 1);
    const int Aw = (P + left1 + 3) & ~3;
    const int xw_pad = ((FT + span1 + (Aw - (P + left1)) + 3) >> 2) * 4;
    const size_t lds = (size_t)(RT * 32 * (FT + 4) + VCI * xw_pad + VCI * KW * RT * 32) * sizeof(float);
    dim3 grid((a->T + VT2 - 1) / VT2, 1, a->B);
    mrf_pair_kernel<KW, RT, VCI, CT><<<grid, 256, lds, st>>>(a->y, a->W1, a->b1, a->W2, a->b2, a->out, a->C, a->rows_pad, a->T, a->dil, a->slope,
                                                       a->out_scale, a->accumulate, a->len_in);
}
}  // namespace

int vvk_mrf_pair(const vv_mrf_args* a, hipStream_t st, const char** err) {
    if (a->B <= 0 || a->T <= 0 || (a->C != 32 && a->C != 64)) { *err = "mrf_resblock: fused form needs C = 32 or 64 (the intermediate tile must fit the CU)"; return -22; }
    if (a->rows_pad != 64 || ((uintptr_t)a->W1 % 16) || ((uintptr_t)a->W2 % 16)) { *err = "mrf_resblock: weight slabs [C_pad8][KW][64], 16-byte aligned"; return -22; }
    if (a->dil < 1 || a->dil > 5) { *err = "mrf_resblock: dilation 1..5"; return -22; }
    if (!(a->slope > 0.f && a->slope <= 1.f)) { *err = "mrf_resblock: slope must be in (0, 1]"; return -22; }
    if (a->out == a->y) { *err = "mrf_resblock: out must not alias y (neighbouring workgroups read y's halo)"; return -22; }
    if ((size_t)a->C * a->T * 4 >= ((size_t)1 << 31)) { *err = "mrf_resblock: a [C][T] slab of one item reaches 2 GiB"; return -22; }
    const bool wide = a->C == 64;
    switch (a->KW) {
        // 128-column windows for both widths: a 256-column window for C = 32 (half the halo recompute) measured SLOWER at the
        // headline batch (conv class 199 vs 193 ms, profiles/r02/vocoder_notes.md): occupancy beats the halo
        case 3: wide ? launch_mrf<3, 2, 8, 1>(a, st) : launch_mrf<3, 1, 8, 1>(a, st); break;
        case 7: wide ? launch_mrf<7, 2, 4, 1>(a, st) : launch_mrf<7, 1, 4, 1>(a, st); break;
        case 11: wide ? launch_mrf<11, 2, 4, 1>(a, st) : launch_mrf<11, 1, 4, 1>(a, st); break;
        default: *err = "mrf_resblock: kernel width must be 3, 7 or 11"; return -22;
    }
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}

int vvk_conv_post(const float* in, const float* w, float bias, int16_t* pcm, int ld_pcm, float* wave_f32, int B, int C, int T, int KW,
                  float pre_slope, const int* len_in, hipStream_t st, const char** err) {
    if (KW != 7 || C > 64 || C < 1) { *err = "conv_post: k=7 and C<=64 expected"; return -22; }
    if (!(pre_slope > 0.f && pre_slope <= 1.f)) { *err = "conv_post: pre_slope must be in (0, 1]"; return -22; }
    dim3 grid((T + 4 * POST_WAVE_OUT - 1) / (4 * POST_WAVE_OUT), B);          // 4 waves x 248 samples per workgroup
    if ((size_t)T * 4 >= ((size_t)1 << 31)) { *err = "conv_post: rows of 2 GiB or more"; return -22; }
    const bool vec = (T & 3) == 0 && ((uintptr_t)in % 16) == 0;
#define POST_GO(VEC, CB) conv_post_kernel<7, VEC, CB><<<grid, 256, 0, st>>>(in, w, bias, pcm, ld_pcm, wave_f32, C, T, pre_slope, len_in)
    if (vec && C % 8 == 0) POST_GO(true, 8);
    else if (vec && C % 4 == 0) POST_GO(true, 4);
    else if (vec) POST_GO(true, 1);
    else POST_GO(false, 1);
#undef POST_GO
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}

int vvk_mel_slice(const float* x, int B, int N, int n_mel, const int* ref_len, const int* seq_len, float* out, int T,
                  hipStream_t st, const char** err) {
    if (T <= 0) { *err = "mel_slice: empty"; return -22; }
    dim3 grid((T + 31) / 32, (n_mel + 31) / 32, B);
    mel_slice_kernel<<<grid, 256, 0, st>>>(x, N, n_mel, ref_len, seq_len, out, T);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}
