/* vvtts.h -- C ABI of the MI355X-native VietVoice-TTS synthesis hot path (libvvtts_hip.so).
 *
 * The reference has no FFI: its hot path is three onnxruntime sessions driven from Python
 *   preprocess : vietvoicetts/core/tts_engine.py:133-146  (sessions['preprocess'].run)
 *   transformer: vietvoicetts/core/tts_engine.py:148-174  (sessions['transformer'].run, 31 calls)
 *   decode     : vietvoicetts/core/tts_engine.py:176-187  (sessions['decode'].run)
 * created in vietvoicetts/core/model.py:65-129.  This header is what a binding for that path
 * would bind instead (INTEGRATION.md shows the ctypes stub): plain pointers and sizes, no torch
 * types, every call returns 0 or a negative errno-style code, the message is read with
 * vv_last_error(); nothing aborts or throws across the ABI.  All pointers named *device* are
 * HBM addresses on the context's GPU; `stream` is a hipStream_t passed as void*.
 * One context per GPU; calls on one context must be serialised by the caller
 * (reference threading: vietvoicetts/api/tts_engine.py:64-87).
 */
#ifndef VVTTS_H
#define VVTTS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* exported entry points: the library is built with -fvisibility=hidden, only these symbols leave it */
#if defined(__GNUC__)
#define VV_API __attribute__((visibility("default")))
#else
#define VV_API
#endif

#define VV_DTYPE_F32 0
#define VV_DTYPE_BF16 1
#define VV_MAX_UP 8
#define VV_MAX_RES 4

/* activation codes (gemm epilogue) */
#define VV_ACT_NONE_ 0
#define VV_ACT_GELU_TANH_ 1
#define VV_ACT_GELU_ERF_ 2
#define VV_ACT_SILU_ 3
#define VV_ACT_MISH_ 4
/* gemm epilogue modes */
#define VV_EPI_STORE 0     /* C = act(A W^T + bias)                         */
#define VV_EPI_QKV_ROPE 1  /* C = rope(A W^T + bias) on the q and k columns */
#define VV_EPI_GATE_RES 2  /* C += gate * (A W^T + bias)   (fp32 residual)  */
#define VV_EPI_GATE_STORE 3 /* C = gate * (A W^T + bias); the add is fused into the next vv_layernorm (delta) */

typedef struct vv_ctx vv_ctx;

/* Architecture constants (the reference hides them in the ONNX graphs). */
typedef struct vv_model_cfg {
    int32_t n_mel, n_fft, win_length, hop_length;
    int32_t dim, depth, heads, head_dim, ff_mult;
    int32_t text_dim, text_layers, text_conv_k, text_ff_mult, vocab_rows;
    int32_t pos_conv_k, pos_conv_groups, time_freq_dim;
    float cfg_strength;
    int32_t voc_pre_ch, voc_pre_k, voc_post_k;
    int32_t voc_n_up, voc_up_rates[VV_MAX_UP], voc_up_kernels[VV_MAX_UP];
    int32_t voc_n_res, voc_res_kernels[VV_MAX_RES];
    int32_t voc_n_dil, voc_res_dilations[VV_MAX_RES];
    float voc_lrelu;
    int32_t max_pos; /* rows of the rope / text position tables */
} vv_model_cfg;

/* ---- context ------------------------------------------------------------------------------ */
/* replaces onnxruntime.InferenceSession creation, core/model.py:96-102 */
VV_API int vv_create(vv_ctx** out, int device, const vv_model_cfg* cfg, int acoustic_dtype);
VV_API void vv_destroy(vv_ctx* ctx);
VV_API const char* vv_last_error(const vv_ctx* ctx);   /* ctx may be NULL: last create error */
VV_API const char* vv_version(void);

/* Bind one named tensor of the (already uploaded) flat weight buffer.  Layouts: DESIGN.md 3. */
VV_API int vv_bind_weight(vv_ctx* ctx, const char* name, const void* device_ptr, uint64_t bytes);
/* Verify every tensor the three stages need is bound (names listed in the error if not). */
VV_API int vv_finalize_weights(vv_ctx* ctx);

/* ODE time grid: sinus[n_steps][time_freq_dim] (host), dt[n_steps] (host).  Runs the time MLP and
 * every block's AdaLN projection once on the GPU and keeps the modulation tables in HBM. */
VV_API int vv_set_time_grid(vv_ctx* ctx, const float* sinus_host, const float* dt_host, int n_steps, void* stream);

/* ---- the three stages (device-resident, batched) ------------------------------------------ */
/* replaces sessions['preprocess'].run, core/tts_engine.py:133-146.
 * audio [B][ld_audio] int16, audio_len[B], text_ids [B][ld_text] int32, text_len[B], seq_len[B]
 * (= max_duration per item, frames), N = padded frame count (>= every seq_len).
 * Outputs: cat_mel_text, cat_mel_text_drop [B][N][n_mel+text_dim] f32, ref_signal_len[B] int32.
 * (noise is supplied by the caller; the rope tables are slices of the bound constant tables.) */
VV_API int vv_preprocess(vv_ctx* ctx, int B, int N, const int16_t* audio, int ld_audio, int max_audio_len,
                  const int32_t* audio_len, const int32_t* text_ids, int ld_text, const int32_t* text_len,
                  const int32_t* seq_len, float* cat_mel_text, float* cat_mel_text_drop,
                  int32_t* ref_signal_len, void* stream);

/* The same call with the clip lengths also handed over on the HOST (same values as the device array).  A centred STFT reflects
 * n_fft / 2 samples at both ends of a clip, which is defined only for clips of more than n_fft / 2 samples (the reference admits
 * any clip, core/audio_processor.py:15-26, core/tts_engine.py:46-56; torch.stft refuses a shorter one): this form returns -22
 * for such an item before anything is launched.  Without host lengths (vv_preprocess) the mel kernel clamps the doubly reflected
 * index -- finite, deterministic, defined by no reference. */
VV_API int vv_preprocess_h(vv_ctx* ctx, int B, int N, const int16_t* audio, int ld_audio, int max_audio_len,
                    const int32_t* audio_len, const int32_t* audio_len_host, const int32_t* text_ids, int ld_text,
                    const int32_t* text_len, const int32_t* seq_len, float* cat_mel_text, float* cat_mel_text_drop,
                    int32_t* ref_signal_len, void* stream);

/* replaces the loop over sessions['transformer'].run, core/tts_engine.py:148-174: n_steps Euler
 * steps of the flow ODE starting at step index step0, state x [B][N][n_mel] f32 updated in HBM.
 * rope tables are [>=N][head_dim] f32 (q tables carry the softmax scale). */
VV_API int vv_transformer_steps(vv_ctx* ctx, int B, int N, const int32_t* seq_len, float* x, const float* cat_mel_text,
                         const float* cat_mel_text_drop, const float* rope_cos_q, const float* rope_sin_q,
                         const float* rope_cos_k, const float* rope_sin_k, int step0, int n_steps, void* stream);

/* The same call with the per-item lengths also handed over on the HOST (same values as the device array): the host needs them
 * for the launch shapes, so vv_transformer_steps reads them back (one 4*B-byte copy + stream synchronisation per call); this form
 * has no synchronisation at all and can be captured into a hipGraph once the context arena is large enough. */
VV_API int vv_transformer_steps_h(vv_ctx* ctx, int B, int N, const int32_t* seq_len, const int32_t* seq_len_host, float* x,
                           const float* cat_mel_text, const float* cat_mel_text_drop, const float* rope_cos_q, const float* rope_sin_q,
                           const float* rope_cos_k, const float* rope_sin_k, int step0, int n_steps, void* stream);

/* replaces sessions['decode'].run, core/tts_engine.py:176-187: frames [ref_len, seq_len) of x ->
 * vocoder -> int16 PCM.  pcm [B][ld_pcm], ld_pcm >= t_gen_max*hop; pcm_len[B] = samples per item.
 * wave_f32 (optional, [B][t_gen_max*hop]) receives the pre-quantisation waveform. */
VV_API int vv_decode(vv_ctx* ctx, int B, int N, const float* x, const int32_t* ref_signal_len, const int32_t* seq_len,
              int t_gen_max, int16_t* pcm, int ld_pcm, int32_t* pcm_len, float* wave_f32, void* stream);

/* vv_transformer_steps_h with every intermediate carved from a CALLER-OWNED device block `ws` (256-byte aligned, >=
 * vv_transformer_ws_bytes for the same B, N and host lengths) instead of the context arena: no allocation, no synchronisation and
 * nothing that can move -- the form to capture into a hipGraph (all Euler steps of an utterance + vv_decode_into as ONE graph
 * launch: the single-utterance latency path).  No reference counterpart (the reference pays a host round trip per step,
 * core/tts_engine.py:157-172). */
VV_API int vv_transformer_ws_bytes(vv_ctx* ctx, int B, int N, const int32_t* seq_len_host, uint64_t* bytes);
VV_API int vv_transformer_steps_into(vv_ctx* ctx, int B, int N, const int32_t* seq_len, const int32_t* seq_len_host, float* x,
                              const float* cat_mel_text, const float* cat_mel_text_drop, const float* rope_cos_q,
                              const float* rope_sin_q, const float* rope_cos_k, const float* rope_sin_k, int step0,
                              int n_steps, void* ws, uint64_t ws_bytes, void* stream);

/* The same decode stage with every intermediate carved from a CALLER-OWNED device block `ws` (256-byte aligned,
 * >= vv_decode_ws_bytes bytes) instead of the context arena.  The context arena may be reallocated by any later call
 * that needs more bytes (vv_ws_generation counts those moves); a launch sequence captured into a hipGraph
 * (BASELINE.json configs[4], "hipGraph-captured vocoder step") must therefore run through vv_decode_into so that
 * nothing it points at can move while the graph lives.  No reference counterpart (the reference never captures). */
VV_API int vv_decode_ws_bytes(vv_ctx* ctx, int B, int t_gen_max, uint64_t* bytes);
VV_API int vv_decode_into(vv_ctx* ctx, int B, int N, const float* x, const int32_t* ref_signal_len, const int32_t* seq_len,
                   int t_gen_max, int16_t* pcm, int ld_pcm, int32_t* pcm_len, float* wave_f32, void* ws, uint64_t ws_bytes,
                   void* stream);
VV_API uint64_t vv_ws_generation(const vv_ctx* ctx);   /* number of times the context arena has been (re)allocated */

/* Declares that the rope tables handed to vv_transformer_steps are the STANDARD RoPE tables of base `theta` (angle = pos *
 * theta^(-2i/head_dim); the q tables times head_dim^-0.5) -- which is what vv_preprocess's contract produces.  The bf16 model then
 * computes the angles in the QKV GEMM's epilogue instead of reading the tables (vv_gemm_args.rope_theta) and hands the softmax scale to
 * the attention kernel (vv_attn_args.q_scale).  theta = 0 (the default of a new context): the tables are read.  The fp32 model always
 * reads them. */
VV_API int vv_set_rope_theta(vv_ctx* ctx, float theta);

/* Context switches (explicit API, never the environment).  "fuse_mrf": run the MRF resblock pairs of the C <= 64 vocoder
 * stages through vv_mrf_resblock's fused kernel -- 0 never (two vv_conv1d launches per pair), 1 always, 2 (default) for
 * decodes of <= 8 items, where the stage is launch-bound.  Results are bit-identical either way.
 * "split_k_tail" (bf16 acoustic model): lets vv_transformer_steps take vv_gemm_tail_plan's split-K tail -- 0 (default) never,
 * 1 for the out-projection and FF2 GEMMs, 2 for FF2 only.  The K parts of a tail row are kept in fp32, summed by the consuming
 * norm and the SUM rounded to bf16 once, where a row outside the tail is rounded in the GEMM epilogue: a tail row differs from
 * the plain launch by fp32 summation order -- which later bf16 roundings amplify to bf16-level noise, so with the tail on a row's
 * result depends (inside the bf16 tolerance class) on its position in the launch.  Off, it does not; the fp32 path never splits.
 * "rope_q_attn" (bf16): 1 (default) = the query side of the rope is applied by the attention kernel while it loads Q, the QKV GEMM
 * ropes the k columns only; 0 = all of it in the GEMM epilogue (the fp32 model always does).
 * "rope_rows": 1 gathers the compact rope tables per packed row once per call (vv_rope_rows); 0 (default) looks positions up.
 * "lanes": vv_transformer_steps* runs a batch of >= 2 items as TWO half batches on two HIP streams -- lane 0 on the caller's stream,
 * lane 1 on a context-owned stream forked from it at the start of the call and joined to it before the call returns (so the call
 * keeps its stream semantics, and can be captured into a hipGraph) -- so that one lane's kernel tails and launch gaps are filled by
 * the other's kernels.  A single item runs its two CFG branches (conditional / unconditional rows) as the lanes, forked and joined once
 * per Euler step.  0 (default) = for the bf16 model from 1,024 packed rows (2 x sum of the lengths) on, 1 = never, 2 = always.
 * Results are bit-identical: every row's arithmetic is independent of what shares its launch.
 * "ring_tiles": 1 (default) = bf16 GEMMs of the path with N <= 1024 whose 64-token x 128-feature tiles are fewer than the CUs (the
 * out-projection and FF2 of a single utterance's CFG branch) take 64 x 64 tiles on a three-stage LDS ring (vv_gemm tile 6464); 0 = never;
 * n > 1 = the same with n as the tile-count bound.  Same bits either way.
 * "pp_min_tiles": -1 (default) = vv_gemm's own choice between its persistent 256 x 256 kernel and the 128 x 128 one; n >= 0 = the
 * persistent kernel for every bf16 GEMM of the path with M >= 4096, N % 256 == 0 and >= n 256-tiles.  Same bits either way. */
VV_API int vv_set_option(vv_ctx* ctx, const char* name, int value);

/* ---- profiling (HIP events on the launch stream, per kernel class) ------------------------- */
#define VV_PROF_GEMM 0
#define VV_PROF_ATTN 1
#define VV_PROF_NORM 2
#define VV_PROF_POSCONV 3
#define VV_PROF_ELEMWISE 4
#define VV_PROF_VOC_CONV 5
#define VV_PROF_VOC_POST 6
#define VV_PROF_MEL 7
#define VV_PROF_TEXT 8
/* the vocoder convs once more, by stage (each launch is counted in VV_PROF_VOC_CONV and in exactly one of these): conv_pre, the four
 * transposed-conv upsamplers (K11), the four MRF stacks (K12) */
#define VV_PROF_VOC_PRE 9
#define VV_PROF_VOC_UP0 10      /* + stage, stages 0..3 */
#define VV_PROF_VOC_MRF0 14     /* + stage, stages 0..3 */
#define VV_PROF_NCLASS 18
VV_API int vv_prof_enable(vv_ctx* ctx, int on);
/* Synchronises, then fills per class: launches, total ms, algorithmic flops, algorithmic bytes. */
VV_API int vv_prof_collect(vv_ctx* ctx, int64_t* launches, double* ms, double* flops, double* bytes);

/* ---- single-kernel entry points (unit parity tests call these through the ABI) ------------- */
typedef struct vv_gemm_args {
    int32_t dtype, out_dtype, mode, act;
    const void* A; int32_t lda;
    const void* W; int32_t ldw;
    void* C; int32_t ldc;
    int32_t M, N, K;
    const float *bias, *gate, *cos_q, *sin_q, *cos_k, *sin_k;
    int32_t n_store, seq_n, rope_dim;
    const float *rope_cs_q, *rope_cs_k;   /* optional compact [pos][64] (cos,sin) pair tables, see vv_rope_compact */
    int32_t tile;   /* 0 = auto (bf16: the persistent 256x256 kernel when M >= 4096, N % 256 == 0 and the shape has at least one round of
                       256-tiles for the chip's CUs or N >= 3072, below that 128x128 tiles or, for launches that do not fill the chip, 64-token x 128-feature
                       tiles; fp32: 256x256 when M >= 4096 and N % 256 == 0), 128 or 256 to force (bf16: also 64, and 6464 = 64 x 64 tiles on a three-stage
                       LDS ring).  Every bf16 choice gives the same bits */
    const int32_t* rope_pos;   /* optional [M]: rope position of each row (packed ragged rows); default row % seq_n */
    int32_t rope_by_row;       /* 1: rope_cs_q / rope_cs_k are [M][64] tables gathered per row by vv_rope_rows (the persistent kernel then
                                  needs no position lookup); the cos/sin tables + rope_pos still serve the other kernels */
    int32_t tail_parts, tail_row0;   /* split-K tail (VV_EPI_GATE_STORE, bf16): both exactly as vv_gemm_tail_plan returns them, 0 = off */
    void* C_tail;              /* tail_parts > 1: fp32 [tail_parts][M - tail_row0][ldc] gated products of the K parts of rows >= tail_row0
                                  (part 0 carries the bias); those rows of C are NOT written: the consumer sums the parts and rounds
                                  the sum to the output dtype once (vv_ln_args.delta_tail) */
    float rope_theta;          /* VV_EPI_QKV_ROPE, bf16: > 0 = the tables are the standard RoPE tables of this base (angle = pos * theta^(-2i/64)):
                                  the epilogue COMPUTES cos / sin of the q and k columns (v_exp / v_fract / v_sin / v_cos on the position: no
                                  table load behind the stores, the costliest part of the rope epilogue) and applies NO softmax scale to q
                                  (vv_attn_args.q_scale carries it); angle accurate to ~3e-4 rad at position 4096 -- a tenth of a bf16
                                  rounding of the roped value.  0 = read the tables (always in fp32) */
    int32_t rope_skip_q;       /* VV_EPI_QKV_ROPE: 1 = leave the q columns [0, rope_dim) un-roped (plain bias + store); the attention
                                  kernel ropes them while it loads Q (vv_attn_args.rope_cs_q).  The k columns are roped as always */
    int32_t chip_share;        /* 0 / 1 = the launch has the chip to itself; 2 = it shares the chip with another stream of launches (the
                                  two lanes of vv_transformer_steps): tile = 0 then prices its tilings for half of the CUs.  Speed only */
} vv_gemm_args;
VV_API int vv_gemm(vv_ctx* ctx, const vv_gemm_args* args, void* stream);
/* The persistent bf16 GEMM walks ceil(tiles / CUs) rounds of 256x256 tiles; when the tile count leaves a partial last round, the
 * gate-store form can split the K range of the last row panels `parts` ways so that the remainder costs 1/parts of a round.
 * Returns the plan for this shape (contiguous operands: lda = ldw = K, ldc = N) on the context's device: rows >= row0 are split
 * `parts` ways; parts = 0: nothing to gain, or an operand of 2 GiB or more (those take the plain-pointer kernel, which has no
 * tail).  ctx may be NULL (the process's current device, 256 CUs when there is none). */
VV_API int vv_gemm_tail_plan(vv_ctx* ctx, int32_t M, int32_t N, int32_t K, int32_t* row0, int32_t* parts);

typedef struct vv_attn_args {
    int32_t dtype;
    const void* qkv; int32_t ld_qkv;
    void* out; int32_t ld_out;
    int32_t n_seq, seq_n, heads, dim;
    const int32_t* kv_len;
    const int32_t* row_start;  /* optional [n_seq]: packed ragged rows -- sequence s owns rows [row_start[s], +kv_len[s]);
                                  default s * seq_n (padded layout, rows beyond kv_len are computed and ignored) */
    int32_t total_rows;        /* rows in the qkv / out buffers (bounds the K/V buffer resource; reads past it return zero).
                                  Required (> 0) with row_start; 0 = n_seq * seq_n in the padded layout */
    float q_scale;             /* bf16 kernel: factor applied to q while it is loaded (the softmax scale when the projection did not carry
                                  it: vv_gemm_args.rope_theta); 0 = 1.0 */
    const float* rope_cs_q;    /* optional (bf16 kernel): compact [seq_n][64] (cos, sin) pair table of the QUERY side (vv_rope_compact of
                                  the q tables, which carry the softmax scale): the q columns arrive un-roped (vv_gemm_args.rope_skip_q)
                                  and are roped here, position = row inside the sequence, in fp32 before the one rounding to bf16 the
                                  kernel applies to Q anyway.  NULL = q is already roped */
} vv_attn_args;
VV_API int vv_attention(vv_ctx* ctx, const vv_attn_args* args, void* stream);

typedef struct vv_ln_args {
    int32_t out_dtype;
    const float* x; int32_t ldx;
    void* y; int32_t ldy;
    int32_t R, D;
    const float *w, *b;
    int32_t add_one;
    float eps;
    const void* delta;      /* optional [R][ld_delta]: x += delta first (x is then updated in place) */
    int32_t delta_dtype, ld_delta;
    const void* delta2;     /* optional second delta (same dtype / ld): y = LN((x + delta) + delta2) */
    int32_t keep_x;         /* 1: normalise x + delta but leave x as it is (the caller adds this delta again later, with delta2) */
    int32_t tail_row0;      /* split-K tails of the deltas (vv_gemm_args.C_tail): for rows >= tail_row0 the delta is NOT read;    */
    int32_t delta_tail_parts, delta2_tail_parts;   /*   it is the sum, in part order, of fp32 delta_tail [parts][R - tail_row0][ld_delta] */
    const void *delta_tail, *delta2_tail;          /*   rounded once to delta_dtype (0 / 1 parts = none)                       */
} vv_ln_args;
VV_API int vv_layernorm(vv_ctx* ctx, const vv_ln_args* args, void* stream);

typedef struct vv_posconv_args {
    int32_t dtype, out_dtype;
    const void* in; int32_t ld_in;
    const void* W;            /* bf16: [G][KW][64 co][64 ci]   f32: [G][KW][64 ci][64 co] */
    const float* bias;
    void* out; int32_t ld_out;
    const void* resid; int32_t ld_resid;   /* optional, operand dtype */
    int32_t n_seq, seq_n, groups, KW, B;
    const int32_t* seq_len;
    const int32_t* row_start;  /* optional [n_seq]: packed ragged rows, as in vv_attn_args */
} vv_posconv_args;
VV_API int vv_posconv(vv_ctx* ctx, const vv_posconv_args* args, void* stream);

typedef struct vv_conv_args {
    const float* in;          /* [B][Cin][T_in]  */
    const float* W;           /* [Cin_pad8][KW][rows_pad64], rows = co (conv) or co*up + phase (transposed) */
    const float* bias;        /* [Cout] */
    float* out;               /* [B][Cout][T_out] */
    const float* resid;       /* optional, like out */
    int32_t B, Cin, Cout, T_in, T_out, KW, dil, transposed, up, rows_total, rows_pad, accumulate;
    float pre_slope, out_scale;
    const int32_t* len_in;    /* optional per-item valid input length */
    const void* W_x3;         /* optional: W split by vv_conv_split_weights; when given, the products run as exact 3-way bf16
                                 splits on the bf16 matrix pipe (six piece products, fp32 accumulate: fp32 fidelity, ~2.7x less
                                 matrix time) instead of v_mfma_f32_32x32x2_f32.  Same result class, not bit-identical. */
    int32_t wg_rows;          /* x3 only: 0 = default (64-row workgroups of 4 waves; the x2 up-samplers with 64 / 128 input channels take the
                                 streaming kernel), 128 = 128-row workgroups of 8 waves when rows_total > 64, -1 = the generic kernel
                                 also where the streaming one would be taken (its bit-identical twin: tests, A/B) */
} vv_conv_args;
VV_API int vv_conv1d(vv_ctx* ctx, const vv_conv_args* args, void* stream);
/* W fp32 [Cin_pad][KW][rows_pad] -> out [ceil(Cin_pad / 16)][KW][3 pieces][2 octets][rows_pad][8] bf16 (w = h + m + l exactly, each piece the
 * truncated leading 8 significand bits of the remainder).  vv_conv_split_bytes gives the size of `out` (16-byte aligned). */
VV_API uint64_t vv_conv_split_bytes(int32_t Cin_pad, int32_t KW, int32_t rows_pad);
VV_API int vv_conv_split_weights(vv_ctx* ctx, const float* W, int32_t Cin_pad, int32_t KW, int32_t rows_pad, void* out, void* stream);

/* K12, one (kernel, dilation) pair of an MRF resblock fused through LDS (SURVEY 8(a) K12 / 8(b) vv_mrf_resblock):
 *   out = [accumulate ? out : 0] + out_scale * ( conv2(lrelu(conv1(lrelu(y)))) + y ),  conv1 dilated, conv2 undilated, C -> C.
 * Fused form for C = 32 / 64 (the intermediate tile stays in the CU); bit-identical to vv_conv1d(conv1) + vv_conv1d(conv2, resid = y).
 * replaces two nodes of decode.onnx's HiFi-GAN resblock (reference core/tts_engine.py:176-187 runs the graph; no source). */
typedef struct vv_mrf_args {
    const float* y;           /* [B][C][T] resblock input (also the residual) */
    const float *W1, *b1;     /* conv1: [C_pad8][KW][64], [C] */
    const float *W2, *b2;     /* conv2 */
    float* out;               /* [B][C][T], must not alias y */
    int32_t B, C, T, KW, dil, rows_pad, accumulate;
    float slope, out_scale;
    const int32_t* len_in;    /* optional per-item valid length */
} vv_mrf_args;
VV_API int vv_mrf_resblock(vv_ctx* ctx, const vv_mrf_args* args, void* stream);

VV_API int vv_conv_post(vv_ctx* ctx, const float* in, const float* w, float bias, int16_t* pcm, int ld_pcm, float* wave_f32,
                 int B, int C, int T, int KW, float pre_slope, const int32_t* len_in, void* stream);
VV_API int vv_mel(vv_ctx* ctx, const int16_t* audio, int ld_audio, const int32_t* audio_len, float* mel, int B, int F_max,
           void* stream);
/* K5 GroupNorm over channel-major [B][C][T] fp32 (G groups), optional per-channel affine and fused activation code */
VV_API int vv_groupnorm(vv_ctx* ctx, const float* x, float* y, const float* gamma, const float* beta, int B, int C, int T, int G,
                 float eps, int act, void* stream);
/* out[pos][2i] = cos[pos][2i], out[pos][2i+1] = sin[pos][2i]  (tables with duplicated pairs, n rows x 64) */
VV_API int vv_rope_compact(vv_ctx* ctx, const float* cos_t, const float* sin_t, float* out, int n, void* stream);
/* out[r][0..63] = compact[pos[r]][0..63]: the compact table gathered once per call for every packed row */
VV_API int vv_rope_rows(vv_ctx* ctx, const float* compact, const int32_t* pos, float* out, int rows, void* stream);
VV_API int vv_cfg_euler(vv_ctx* ctx, float* x, const float* pred, int ldp, int BN, int n_mel, float cfg, float dt, void* stream);

/* ---- reference-clip ingest on the device (a8 + SURVEY 8(f) N3).  Together they replace the arithmetic of
 * AudioProcessor.load_audio after RIFF parsing (reference core/audio_processor.py:15-44): pydub's set_channels(1) =
 * audioop.tomono(.., 0.5, 0.5), set_frame_rate = audioop.ratecv(.., None), float32 conversion, then normalize_to_int16 =
 * remove DC (numpy's float32 mean, in numpy's summation order), peak -> 29491, truncate.  Bit-exact to stdlib audioop + numpy. */
/* pcm = the clips' interleaved little-endian signed PCM bytes, back to back (each clip 4-byte aligned); desc = n_clips x 8 int64
 * (device): {byte offset, sample width 1|2|4, channels, n_frames, src_rate / g, dst_rate / g, out offset (floats), n_out} with
 * g = gcd(src_rate, dst_rate) and n_out = (n_frames - 1) * (dst_rate / g) / (src_rate / g) + 1 (= n_frames at equal rates).
 * out[out offset + m] = float32 of the mono sample m at the destination rate.  max_out = the largest n_out.  The rows live in device
 * memory, so the library cannot check them: the CALLER validates them against the two buffers before the call (runtime.HipSynth.ingest_pcm does). */
VV_API int vv_ingest_pcm(vv_ctx* ctx, const void* pcm, const int64_t* desc, int n_clips, int64_t max_out, float* out, void* stream);
/* opt-in, NOT the reference's arithmetic: polyphase FIR resampler y[n] = sum_i x[i] * taps[(n + skip) * down - i * up],
 * f64 accumulate, f32 out.  taps = host-designed low-pass already scaled by `up` (f64, device). */
VV_API int vv_resample_poly(vv_ctx* ctx, const float* x, int n_in, const double* taps, int n_taps, int up, int down, int skip,
                     float* y, int n_out, void* stream);
/* n_clips mono f32 clips stored back to back, clip i = [offsets[i], offsets[i+1]); out has the same offsets.
 * scratch = vv_normalize_scratch_bytes(n_clips, offsets[n_clips]) bytes of device memory; clips shorter than 2^24 samples. */
VV_API size_t vv_normalize_scratch_bytes(int n_clips, int64_t total_len);
VV_API int vv_normalize_clips(vv_ctx* ctx, const float* x, const int64_t* offsets, int n_clips, int64_t max_len, void* scratch,
                       int16_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
