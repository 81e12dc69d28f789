// HBM-bound kernels of the acoustic path: K4 LayerNorm/AdaLN-modulate, K2 conditioning pack,
// K9 CFG+Euler, text-side embedding gather / depthwise conv / GRN, and the tiny time-grid helpers.
// All are wave64-native: 16-byte vector loads, wave shuffles for row reductions, no LDS unless a
// cross-wave reduction is needed.
#include "vv_common.h"
#include "vv_kernels.h"

namespace {

// fp32 -> T -> fp32, element-wise (round to nearest even: what store4<T> / the GEMM epilogue's v_cvt_pk_bf16_f32 do)
template <typename T> __device__ __forceinline__ float4 round4_as(float4 a) {
    if constexpr (sizeof(T) == 2) {
        typedef __attribute__((ext_vector_type(4))) float f4v;
        const f4v r = __builtin_convertvector(__builtin_convertvector((f4v){a.x, a.y, a.z, a.w}, bf16x4), f4v);
        return make_float4(r[0], r[1], r[2], r[3]);
    } else {
        return a;
    }
}

// ---------------------------------------------------------------- K4: [x += delta;] y = LN(x) * (w [+1]) + b
// One wave per row, the row cached in registers (D <= 1024): one HBM read, one write.  When a
// delta is given (the gated branch output of the previous GEMM) the residual add is fused here:
// the fp32 stream is read once, updated, written back and normalised in the same pass.
template <typename To, typename Td>
__global__ __launch_bounds__(256) void ln_mod_kernel(float* __restrict__ x, int ldx, To* __restrict__ y, int ldy,
                                                     int R, int D, const float* __restrict__ w,
                                                     const float* __restrict__ b, int add_one, float eps,
                                                     const Td* __restrict__ delta, int ldd, const Td* __restrict__ delta2,
                                                     int keep_x, int tail_row0, int tp1, const float* __restrict__ dt1, int tp2,
                                                     const float* __restrict__ dt2) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const int n4 = D >> 2;
    float* xr = x + (size_t)row * ldx;
    float4 v[4];
    float s = 0.f;
    if (delta) {
        // all loads of the row first (x, delta, delta2: up to 12 independent 16-byte loads in flight), then the adds
        float4 d[4], d2[4];
        // rows of a split-K tail (wave-uniform, a few per cent of the rows): the GEMM left the row's K parts in fp32 and did NOT
        // write the row of the delta itself; the parts are summed here in part order and the SUM is rounded to the delta's dtype
        // once -- the rounding a row outside the tail got in the GEMM epilogue -- so the stream sees the same kind of delta on
        // every row, whatever the row's position in the launch.
        const bool tl1 = tp1 > 0 && row >= tail_row0, tl2 = tp2 > 0 && row >= tail_row0;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + i * 64;
            if (c < n4) {
                v[i] = load4_nt<float>(xr + c * 4);             // the fp32 stream is far larger than the caches: stream it
                d[i] = tl1 ? z4 : load4_nt<Td>(delta + (size_t)row * ldd + c * 4);      // read once: keep it out of the caches
                d2[i] = (delta2 && !tl2) ? load4_nt<Td>(delta2 + (size_t)row * ldd + c * 4) : z4;
            } else {
                v[i] = d[i] = d2[i] = z4;
            }
        }
        if (row >= tail_row0) {
            const size_t tr = (size_t)(R - tail_row0), r = (size_t)(row - tail_row0);
            if (tl1) {
                for (int p = 0; p < tp1; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = lane + i * 64;
                        if (c < n4) {
                            const float4 t = load4_nt<float>(dt1 + (p * tr + r) * ldd + c * 4);
                            d[i].x += t.x; d[i].y += t.y; d[i].z += t.z; d[i].w += t.w;
                        }
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) d[i] = round4_as<Td>(d[i]);
            }
            if (tl2) {
                for (int p = 0; p < tp2; ++p)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int c = lane + i * 64;
                        if (c < n4) {
                            const float4 t = load4_nt<float>(dt2 + (p * tr + r) * ldd + c * 4);
                            d2[i].x += t.x; d2[i].y += t.y; d2[i].z += t.z; d2[i].w += t.w;
                        }
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) d2[i] = round4_as<Td>(d2[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + i * 64;
            if (c < n4) {
                v[i].x += d[i].x; v[i].y += d[i].y; v[i].z += d[i].z; v[i].w += d[i].w;
                if (delta2) {                                // (x + delta) + delta2: the order the one-delta-at-a-time protocol adds in
                    v[i].x += d2[i].x; v[i].y += d2[i].y; v[i].z += d2[i].z; v[i].w += d2[i].w;
                }
                if (!keep_x) {
                    typedef __attribute__((ext_vector_type(4))) float f4;
                    const f4 w4 = {v[i].x, v[i].y, v[i].z, v[i].w};
                    __builtin_nontemporal_store(w4, (f4*)(xr + c * 4));
                }
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = lane + i * 64;
            if (c < n4) {
                v[i] = load4_nt<float>(xr + c * 4);
                s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
            } else {
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < n4) {
            const float a = v[i].x - mean, bb = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
            q += (a * a + bb * bb) + (cc * cc + dd * dd);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    const float one = add_one ? 1.0f : 0.0f;
    To* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < n4) {
            float4 ww = w ? *(const float4*)(w + c * 4) : make_float4(1.f - one, 1.f - one, 1.f - one, 1.f - one);
            float4 bv = b ? *(const float4*)(b + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            store4<To>(yr + c * 4, (v[i].x - mean) * rstd * (ww.x + one) + bv.x, (v[i].y - mean) * rstd * (ww.y + one) + bv.y,
                       (v[i].z - mean) * rstd * (ww.z + one) + bv.z, (v[i].w - mean) * rstd * (ww.w + one) + bv.w);
        }
    }
}

// ---------------------------------------------------------------- K2: pack [x | cond | text | 0-pad] rows
// Rows are [branch][b][t]; branch 0 reads cat, branch 1 reads cat_drop.  only_x rewrites just the
// n_mel state columns (the per-step part); the conditioning columns are written once per call.
template <typename T>
__global__ __launch_bounds__(256) void pack_cat_kernel(const float* __restrict__ x, const float* __restrict__ cat,
                                                       const float* __restrict__ cat_drop, T* __restrict__ out, int ldo,
                                                       int BN, int n_mel, int cond_dim, int only_x,
                                                       const int* __restrict__ row_src) {
    const int cols4 = (only_x ? n_mel : ldo) >> 2;
    const size_t total = (size_t)2 * BN * cols4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % cols4) * 4;
        const size_t row = i / cols4;
        size_t bt = row >= (size_t)BN ? row - BN : row;
        if (row_src) bt = (size_t)row_src[bt];               // packed rows: bt indexes the padded [B][N] inputs
        float4 v;
        if (c < n_mel) v = *(const float4*)(x + bt * n_mel + c);
        else if (c < n_mel + cond_dim) v = *(const float4*)((row >= (size_t)BN ? cat_drop : cat) + bt * cond_dim + (c - n_mel));
        else v = make_float4(0.f, 0.f, 0.f, 0.f);
        store4<T>(out + row * ldo + c, v.x, v.y, v.z, v.w);
    }
}

// ---------------------------------------------------------------- K9: x += dt * (pc + cfg (pc - pu))
__global__ __launch_bounds__(256) void cfg_euler_kernel(float* __restrict__ x, const float* __restrict__ pred, int ldp,
                                                        int BN, int n_mel, float cfg, float dt, const int* __restrict__ row_src) {
    const int c4 = n_mel >> 2;
    const size_t total = (size_t)BN * c4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % c4) * 4;
        const size_t row = i / c4;
        const float4 pc = *(const float4*)(pred + row * ldp + c);
        const float4 pu = *(const float4*)(pred + (row + BN) * ldp + c);
        const size_t xr = row_src ? (size_t)row_src[row] : row;
        float4 xv = *(float4*)(x + xr * n_mel + c);
        xv.x += dt * (pc.x + (pc.x - pu.x) * cfg);
        xv.y += dt * (pc.y + (pc.y - pu.y) * cfg);
        xv.z += dt * (pc.z + (pc.z - pu.z) * cfg);
        xv.w += dt * (pc.w + (pc.w - pu.w) * cfg);
        *(float4*)(x + xr * n_mel + c) = xv;
    }
}

// ---------------------------------------------------------------- text: embed gather + position table
// Sequence s in [0, 2B): b = s % B, drop branch when s >= B (all filler ids).  Token t of the
// text is id+1; beyond the text (or the sequence) the filler id 0.
__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ ids, int ld_ids,
                                                         const int* __restrict__ text_len, const float* __restrict__ emb,
                                                         const float* __restrict__ pos, float* __restrict__ out, int B,
                                                         int N, int Dt, int vocab_rows) {
    const int c4 = Dt >> 2;
    const size_t total = (size_t)2 * B * N * c4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % c4) * 4;
        const size_t row = i / c4;
        const int t = (int)(row % N);
        const int s = (int)(row / N);
        const int b = s % B;
        int id = 0;
        if (s < B && t < text_len[b] && t < ld_ids) id = ids[(size_t)b * ld_ids + t] + 1;
        id = min(max(id, 0), vocab_rows - 1);
        const float4 e = *(const float4*)(emb + (size_t)id * Dt + c);
        const float4 p = *(const float4*)(pos + (size_t)t * Dt + c);
        *(float4*)(out + row * Dt + c) = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    }
}

// depthwise conv k=KW along tokens, token-major [seq][t][c]; zero beyond [0, len)
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                     const float* __restrict__ w /*[C][KW]*/, const float* __restrict__ bias,
                                                     const int* __restrict__ seq_len, int B, int n_seq, int N, int C, int KW) {
    const int c4 = C >> 2;
    const size_t total = (size_t)n_seq * N * c4;
    const int pad = KW / 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % c4) * 4;
        const size_t row = i / c4;
        const int t = (int)(row % N);
        const int s = (int)(row / N);
        const int len = seq_len ? min(seq_len[s % B], N) : N;
        float4 acc = *(const float4*)(bias + c);
        for (int k = 0; k < KW; ++k) {
            const int tt = t + k - pad;
            if (tt < 0 || tt >= len) continue;
            const float4 v = *(const float4*)(in + ((size_t)s * N + tt) * C + c);
            acc.x += w[(c + 0) * KW + k] * v.x;
            acc.y += w[(c + 1) * KW + k] * v.y;
            acc.z += w[(c + 2) * KW + k] * v.z;
            acc.w += w[(c + 3) * KW + k] * v.w;
        }
        *(float4*)(out + row * C + c) = acc;
    }
}

// GRN statistics: sumsq[s][c] over valid tokens.  grid (C/64, n_seq), block 256 = 4 token phases x 64 channels
template <typename T>
__global__ __launch_bounds__(256) void grn_stats_kernel(const T* __restrict__ x, float* __restrict__ sumsq,
                                                        const int* __restrict__ seq_len, int B, int N, int C) {
    __shared__ float red[256];
    const int s = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ph = threadIdx.x >> 6;
    const int len = seq_len ? min(seq_len[s % B], N) : N;
    float acc = 0.f;
    for (int t = ph; t < len; t += 4) {
        const float v = to_f32<T>(x[((size_t)s * N + t) * C + c]);
        acc += v * v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (ph == 0) sumsq[(size_t)s * C + c] = (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// GRN apply: y = gamma * (x * nx) + beta + x, nx = g / (mean_c g + 1e-6), g = sqrt(sumsq).  In place.
template <typename T>
__global__ __launch_bounds__(256) void grn_apply_kernel(T* __restrict__ x, const float* __restrict__ sumsq,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int N, int C, int rows_per_block) {
    extern __shared__ float sc[];      // [C] scale = gamma*nx + 1
    __shared__ float red[4];
    const int s = blockIdx.y;
    float part = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) part += sqrtf(sumsq[(size_t)s * C + c]);
    part = wave_sum(part);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C;
    for (int c = threadIdx.x; c < C; c += 256) sc[c] = gamma[c] * (sqrtf(sumsq[(size_t)s * C + c]) / (mean + 1e-6f)) + 1.0f;
    __syncthreads();
    const int t0 = blockIdx.x * rows_per_block;
    const int c4 = C >> 2;
    for (int i = threadIdx.x; i < rows_per_block * c4; i += 256) {
        const int t = t0 + i / c4;
        if (t >= N) break;
        const int c = (i % c4) * 4;
        T* p = x + ((size_t)s * N + t) * C + c;
        const float4 v = load4<T>(p);
        const float4 bb = *(const float4*)(beta + c);
        store4<T>(p, v.x * sc[c] + bb.x, v.y * sc[c + 1] + bb.y, v.z * sc[c + 2] + bb.z, v.w * sc[c + 3] + bb.w);
    }
}

// conditioning: cat[b][t] = [mel (t < ref_len) | text(b)], cat_drop[b][t] = [0 | text(B+b)]
__global__ __launch_bounds__(256) void build_cat_kernel(const float* __restrict__ mel, int F_max,
                                                        const int* __restrict__ ref_len, const float* __restrict__ text,
                                                        float* __restrict__ cat, float* __restrict__ cat_drop, int B, int N,
                                                        int n_mel, int Dt) {
    const int cd = n_mel + Dt;
    const int c4 = cd >> 2;
    const size_t total = (size_t)B * N * c4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % c4) * 4;
        const size_t row = i / c4;
        const int t = (int)(row % N);
        const int b = (int)(row / N);
        float4 v, vd;
        if (c < n_mel) {
            v = (t < ref_len[b] && t < F_max) ? *(const float4*)(mel + ((size_t)b * F_max + t) * n_mel + c)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
            vd = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            v = *(const float4*)(text + row * Dt + (c - n_mel));
            vd = *(const float4*)(text + ((size_t)B * N + row) * Dt + (c - n_mel));
        }
        *(float4*)(cat + row * cd + c) = v;
        *(float4*)(cat_drop + row * cd + c) = vd;
    }
}

// small integer helpers (lengths live on the device so that batches need no host round trip)
__global__ void ref_len_kernel(const int* __restrict__ audio_len, int* __restrict__ ref_len, int B, int hop) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b < B) ref_len[b] = audio_len[b] / hop + 1;      // reference core/tts_engine.py:55
}
__global__ void decode_len_kernel(const int* __restrict__ seq_len, const int* __restrict__ ref_len, int* __restrict__ lens,
                                  int B, int n_levels, const int* __restrict__ mult) {
    // lens[level][b] = max(seq_len - ref_len, 0) * mult[level]
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int tg = max(seq_len[b] - ref_len[b], 0);
    for (int l = 0; l < n_levels; ++l) lens[l * B + b] = tg * mult[l];
}
__global__ void dup_len_kernel(const int* __restrict__ seq_len, int* __restrict__ out, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b < B) { out[b] = seq_len[b]; out[B + b] = seq_len[b]; }
}

// Packed-row tables of one vv_transformer_steps call, built on the device from the per-item lengths (one workgroup per item):
//   row_start[2B] (conditional branch first), row_src[Rc] = b * N + t, row_pos[2 Rc] = t.
__global__ __launch_bounds__(256) void row_tables_kernel(const int* __restrict__ seq_len, int B, int N, int Rc, int* __restrict__ row_start,
                                                         int* __restrict__ row_src, int* __restrict__ row_pos) {
    // The table sizes (Rc, 2 Rc) and every launch shape come from the HOST copy of the lengths; the device copy is clamped to
    // [0, N] and to the Rc rows that exist, so a device array that disagrees with the host one (stale tensor, wrong batch) cannot
    // write past the tables -- it produces wrong audio for that call, never a stray store: when the device lengths sum to FEWER
    // rows than the host's Rc, the last workgroup maps the rows nobody owns onto the PADDING rows of the last item (the rows of
    // x / cat right after its device length, clamped to N - 1), so the gathers and the read-modify-write of cfg_euler through
    // row_src never see an unwritten index and never touch a valid row of any item.
    const int b = blockIdx.x;
    int r0 = 0;
    for (int i = 0; i < b; ++i) r0 += min(max(seq_len[i], 0), N);   // B is at most a few hundred: a serial prefix per workgroup is cheaper than a scan
    r0 = min(r0, Rc);
    const int len = min(min(max(seq_len[b], 0), N), Rc - r0);
    if (threadIdx.x == 0) { row_start[b] = r0; row_start[B + b] = Rc + r0; }
    for (int t = threadIdx.x; t < len; t += 256) {
        row_src[r0 + t] = b * N + t;
        row_pos[r0 + t] = t;
        row_pos[Rc + r0 + t] = t;
    }
    if (b == B - 1)
        for (int t = r0 + len + (int)threadIdx.x; t < Rc; t += 256) {
            const int p = min(len + (t - (r0 + len)), N - 1);
            row_src[t] = b * N + p;
            row_pos[t] = p;
            row_pos[Rc + t] = p;
        }
}

// ---------------------------------------------------------------- K5: GroupNorm over channel-major slabs [B][C][T]
// One workgroup per (group, batch item) slab of (C / G) x T contiguous floats.  Two passes, 2 reads + 1 write per element:
// (1) sum and sum of squares of (x - p), p = the slab's first element (a shifted one-pass variance: no cancellation for data with
// a large mean), 16-byte loads, wave-shuffle + LDS reductions; (2) normalise with the per-channel affine and the optional fused
// activation, 16-byte loads and stores.  A slab that is not 16-byte aligned or whose T is not a multiple of 4 takes scalar accesses.
__global__ __launch_bounds__(256) void groupnorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int C, int T, int G, float eps, int act) {
    __shared__ float red[2][4];
    const int b = blockIdx.y, g = blockIdx.x;
    const int cpg = C / G;
    const size_t n = (size_t)cpg * T;
    const float* xs = x + ((size_t)b * C + (size_t)g * cpg) * T;
    float* ys = y + ((size_t)b * C + (size_t)g * cpg) * T;
    const bool vec = (T & 3) == 0 && (((uintptr_t)xs | (uintptr_t)ys) & 15) == 0;
    const float p = xs[0];
    float s = 0.f, q = 0.f;
    if (vec) {
        const size_t n4 = n >> 2;
        for (size_t i = threadIdx.x; i < n4; i += 256) {
            const float4 v = *(const float4*)(xs + i * 4);
            const float a0 = v.x - p, a1 = v.y - p, a2 = v.z - p, a3 = v.w - p;
            s += (a0 + a1) + (a2 + a3);
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    } else {
        for (size_t i = threadIdx.x; i < n; i += 256) { const float d = xs[i] - p; s += d; q += d * d; }
    }
    s = wave_sum(s); q = wave_sum(q);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = q; }
    __syncthreads();
    const float sd = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)n;      // mean of x - p
    const float var = fmaxf(((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)n - sd * sd, 0.f);
    const float mean = p + sd, rstd = rsqrtf(var + eps);
    if (vec) {
        const int t4 = T >> 2;
        const size_t n4 = n >> 2;
        for (size_t i = threadIdx.x; i < n4; i += 256) {
            const int c = g * cpg + (int)(i / t4);
            const float ga = (gamma ? gamma[c] : 1.f) * rstd, be = beta ? beta[c] : 0.f;
            const float4 v = *(const float4*)(xs + i * 4);
            float4 o;
            o.x = act_apply((v.x - mean) * ga + be, act); o.y = act_apply((v.y - mean) * ga + be, act);
            o.z = act_apply((v.z - mean) * ga + be, act); o.w = act_apply((v.w - mean) * ga + be, act);
            *(float4*)(ys + i * 4) = o;
        }
    } else {
        for (size_t i = threadIdx.x; i < n; i += 256) {
            const int c = g * cpg + (int)(i / T);
            ys[i] = act_apply((xs[i] - mean) * ((gamma ? gamma[c] : 1.f) * rstd) + (beta ? beta[c] : 0.f), act);
        }
    }
}

__global__ __launch_bounds__(256) void rope_compact_kernel(const float* __restrict__ c, const float* __restrict__ s, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;      // one (pos, pair)
    if (i >= n * 32) return;
    const int pos = i >> 5, pr = i & 31;
    out[(size_t)pos * 64 + 2 * pr] = c[(size_t)pos * 64 + 2 * pr];
    out[(size_t)pos * 64 + 2 * pr + 1] = s[(size_t)pos * 64 + 2 * pr];
}

__global__ __launch_bounds__(256) void rope_rows_kernel(const float* __restrict__ cs, const int* __restrict__ pos, float* __restrict__ out, int rows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;      // one float4 of one row
    if (i >= (size_t)rows * 16) return;
    const size_t r = i >> 4;
    const int q = (int)(i & 15);
    *(float4*)(out + r * 64 + q * 4) = *(const float4*)(cs + (size_t)pos[r] * 64 + q * 4);
}

__global__ __launch_bounds__(256) void silu_kernel(float* __restrict__ x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = x[i];
        x[i] = v / (1.0f + expf(-v));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ in, T* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = *(const float4*)(in + i * 4);
        store4<T>(out + i * 4, v.x, v.y, v.z, v.w);
    }
}

inline int grid_for(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 256 * 8); }

}  // namespace

#define VVK_CHECK_LAUNCH()                                               \
    do {                                                                 \
        hipError_t he__ = hipGetLastError();                             \
        if (he__ != hipSuccess) { *err = hipGetErrorString(he__); return -5; } \
    } while (0)

int vvk_ln_mod(const vv_ln_args* a, hipStream_t st, const char** err) {
    if (a->R <= 0) { *err = "ln: empty"; return -22; }
    if (a->D % 4 || a->D > 1024 || a->ldx % 4 || a->ldy % 4) { *err = "ln: D must be a multiple of 4 and <= 1024"; return -22; }
    if (a->delta && (a->ld_delta % 4 || a->ld_delta < a->D)) { *err = "ln: bad delta leading dimension"; return -22; }
    const int grid = (a->R + 3) / 4;
    float* x = const_cast<float*>(a->x);
    const bool ob = a->out_dtype == VV_BF16, db = a->delta_dtype == VV_BF16;
    const int tp1 = a->delta && a->delta_tail_parts > 1 ? a->delta_tail_parts : 0, tp2 = a->delta2 && a->delta2_tail_parts > 1 ? a->delta2_tail_parts : 0;
    if ((tp1 || tp2) && (((uintptr_t)a->delta_tail | (uintptr_t)a->delta2_tail) % 16)) { *err = "ln: split-K tail buffers must be 16-byte aligned"; return -22; }
    if ((a->delta_tail_parts > 1 && !a->delta) || (a->delta2_tail_parts > 1 && !a->delta2)) { *err = "ln: a delta tail without its delta"; return -22; }
    if ((tp1 || tp2) && (a->tail_row0 < 0 || a->tail_row0 >= a->R || (tp1 && !a->delta_tail) || (tp2 && !a->delta2_tail) || tp1 > 8 || tp2 > 8)) {
        *err = "ln: bad split-K tail (row0 within [0, R), buffers given, at most 8 parts)"; return -22;
    }
    const int tail_row0 = (tp1 || tp2) ? a->tail_row0 : a->R;
#define LN_GO(To, Td) ln_mod_kernel<To, Td><<<grid, 256, 0, st>>>(x, a->ldx, (To*)a->y, a->ldy, a->R, a->D, a->w, a->b, a->add_one, a->eps, (const Td*)a->delta, a->ld_delta, (const Td*)a->delta2, a->keep_x, tail_row0, tp1, (const float*)a->delta_tail, tp2, (const float*)a->delta2_tail)
    if (ob && db) LN_GO(bf16, bf16);
    else if (ob) LN_GO(bf16, float);
    else if (db) LN_GO(float, bf16);
    else LN_GO(float, float);
#undef LN_GO
    VVK_CHECK_LAUNCH();
    return 0;
}

int vvk_pack_cat(int dtype, const float* x, const float* cat, const float* cat_drop, void* out, int ldo, int BN, int n_mel,
                 int cond_dim, int only_x, const int* row_src, hipStream_t st, const char** err) {
    if (n_mel % 4 || cond_dim % 4 || ldo % 4 || ldo < n_mel + cond_dim) { *err = "pack_cat: bad widths"; return -22; }
    const size_t total = (size_t)2 * BN * ((only_x ? n_mel : ldo) / 4);
    if (dtype == VV_BF16)
        pack_cat_kernel<bf16><<<grid_for(total), 256, 0, st>>>(x, cat, cat_drop, (bf16*)out, ldo, BN, n_mel, cond_dim, only_x, row_src);
    else
        pack_cat_kernel<float><<<grid_for(total), 256, 0, st>>>(x, cat, cat_drop, (float*)out, ldo, BN, n_mel, cond_dim, only_x, row_src);
    VVK_CHECK_LAUNCH();
    return 0;
}

int vvk_cfg_euler(float* x, const float* pred, int ldp, int BN, int n_mel, float cfg, float dt, const int* row_src, hipStream_t st,
                  const char** err) {
    if (n_mel % 4 || ldp % 4) { *err = "cfg_euler: widths must be multiples of 4"; return -22; }
    cfg_euler_kernel<<<grid_for((size_t)BN * n_mel / 4), 256, 0, st>>>(x, pred, ldp, BN, n_mel, cfg, dt, row_src);
    VVK_CHECK_LAUNCH();
    return 0;
}

int vvk_text_embed(const int* ids, int ld_ids, const int* text_len, const float* emb, const float* pos, float* out, int B,
                   int N, int Dt, int vocab_rows, hipStream_t st, const char** err) {
    if (Dt % 4) { *err = "text_embed: Dt % 4"; return -22; }
    text_embed_kernel<<<grid_for((size_t)2 * B * N * Dt / 4), 256, 0, st>>>(ids, ld_ids, text_len, emb, pos, out, B, N, Dt, vocab_rows);
    VVK_CHECK_LAUNCH();
    return 0;
}

int vvk_dwconv(const float* in, float* out, const float* w, const float* bias, const int* seq_len, int B, int n_seq, int N,
               int C, int KW, hipStream_t st, const char** err) {
    if (C % 4) { *err = "dwconv: C % 4"; return -22; }
    dwconv_kernel<<<grid_for((size_t)n_seq * N * C / 4), 256, 0, st>>>(in, out, w, bias, seq_len, B, n_seq, N, C, KW);
    VVK_CHECK_LAUNCH();
    return 0;
}

int vvk_grn(int dtype, void* x, float* sumsq, const float* gamma, const float* beta, const int* seq_len, int B, int n_seq,
            int N, int C, hipStream_t st, const char** err) {
    if (C % 64 || C > 8192) { *err = "grn: C must be a multiple of 64"; return -22; }
    dim3 g1(C / 64, n_seq);
    const int rpb = 16;
    dim3 g2((N + rpb - 1) / rpb, n_seq);
    if (dtype == VV_BF16) {
        grn_stats_kernel<bf16><<<g1, 256, 0, st>>>((const bf16*)x, sumsq, seq_len, B, N, C);
        grn_apply_kernel<bf16><<<g2, 256, C * sizeof(float), st>>>((bf16*)x, sumsq, gamma, beta, N, C, rpb);
    } else {
        grn_stats_kernel<float><<<g1, 256, 0, st>>>((const float*)x, sumsq, seq_len, B, N, C);
        grn_apply_kernel<float><<<g2, 256, C * sizeof(float), st>>>((float*)x, sumsq, gamma, beta, N, C, rpb);
    }
    VVK_CHECK_LAUNCH();
    return 0;
}

int vvk_build_cat(const float* mel, int F_max, const int* ref_len, const float* text, float* cat, float* cat_drop, int B, int N,
                  int n_mel, int Dt, hipStream_t st, const char** err) {
    if (n_mel % 4 || Dt % 4) { *err = "build_cat: widths % 4"; return -22; }
    build_cat_kernel<<<grid_for((size_t)B * N * (n_mel + Dt) / 4), 256, 0, st>>>(mel, F_max, ref_len, text, cat, cat_drop, B, N, n_mel, Dt);
    VVK_CHECK_LAUNCH();
    return 0;
}

int vvk_ref_len(const int* audio_len, int* ref_len, int B, int hop, hipStream_t st, const char** err) {
    ref_len_kernel<<<(B + 63) / 64, 64, 0, st>>>(audio_len, ref_len, B, hop);
    VVK_CHECK_LAUNCH();
    return 0;
}
int vvk_decode_len(const int* seq_len, const int* ref_len, int* lens, int B, int n_levels, const int* mult, hipStream_t st, const char** err) {
    decode_len_kernel<<<(B + 63) / 64, 64, 0, st>>>(seq_len, ref_len, lens, B, n_levels, mult);
    VVK_CHECK_LAUNCH();
    return 0;
}
int vvk_row_tables(const int* seq_len, int B, int N, int Rc, int* row_start, int* row_src, int* row_pos, hipStream_t st, const char** err) {
    if (B <= 0 || N <= 0 || Rc <= 0) { *err = "row_tables: empty"; return -22; }
    row_tables_kernel<<<B, 256, 0, st>>>(seq_len, B, N, Rc, row_start, row_src, row_pos);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}

int vvk_dup_len(const int* seq_len, int* out, int B, hipStream_t st, const char** err) {
    dup_len_kernel<<<(B + 63) / 64, 64, 0, st>>>(seq_len, out, B);
    VVK_CHECK_LAUNCH();
    return 0;
}
int vvk_groupnorm(const float* x, float* y, const float* gamma, const float* beta, int B, int C, int T, int G, float eps, int act,
                  hipStream_t st, const char** err) {
    if (B <= 0 || C <= 0 || T <= 0 || G <= 0 || C % G) { *err = "groupnorm: C must be a multiple of G"; return -22; }
    dim3 grid(G, B);
    groupnorm_kernel<<<grid, 256, 0, st>>>(x, y, gamma, beta, C, T, G, eps, act);
    VVK_CHECK_LAUNCH();
    return 0;
}
int vvk_rope_rows(const float* cs, const int* pos, float* out, int rows, hipStream_t st, const char** err) {
    if (rows <= 0 || !cs || !pos || !out) { *err = "rope_rows: bad arguments"; return -22; }
    rope_rows_kernel<<<(unsigned)(((size_t)rows * 16 + 255) / 256), 256, 0, st>>>(cs, pos, out, rows);
    hipError_t he = hipGetLastError();
    if (he != hipSuccess) { *err = hipGetErrorString(he); return -5; }
    return 0;
}

int vvk_rope_compact(const float* c, const float* s, float* out, int n, hipStream_t st, const char** err) {
    if (n <= 0) { *err = "rope_compact: empty"; return -22; }
    rope_compact_kernel<<<(n * 32 + 255) / 256, 256, 0, st>>>(c, s, out, n);
    VVK_CHECK_LAUNCH();
    return 0;
}
int vvk_silu(float* x, size_t n, hipStream_t st, const char** err) {
    silu_kernel<<<grid_for(n), 256, 0, st>>>(x, n);
    VVK_CHECK_LAUNCH();
    return 0;
}
int vvk_cast(int dtype, const float* in, void* out, size_t n, hipStream_t st, const char** err) {
    if (n % 4) { *err = "cast: n % 4"; return -22; }
    if (dtype == VV_BF16) cast_kernel<bf16><<<grid_for(n / 4), 256, 0, st>>>(in, (bf16*)out, n / 4);
    else cast_kernel<float><<<grid_for(n / 4), 256, 0, st>>>(in, (float*)out, n / 4);
    VVK_CHECK_LAUNCH();
    return 0;
}
