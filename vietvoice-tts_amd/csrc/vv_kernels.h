// Internal launcher interface between the kernel translation units and vv_api.cpp.
// Every launcher validates operand shapes on the host BEFORE launching (a faulting kernel can
// reset the whole box) and returns 0 / negative errno with a static message in *err.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include "../../include/vvtts.h"

typedef vv_gemm_args vvk_gemm_args;

int vvk_gemm(const vv_gemm_args* g, hipStream_t st, const char** err);
void vvk_gemm_tail_plan(int M, int N, int K, int lda, int ldw, int ldc, int* row0, int* parts);
int vvk_attention(const vv_attn_args* a, hipStream_t st, const char** err);
int vvk_ln_mod(const vv_ln_args* a, hipStream_t st, const char** err);
int vvk_posconv(const vv_posconv_args* a, hipStream_t st, const char** err);
int vvk_conv(const vv_conv_args* a, hipStream_t st, const char** err);
int vvk_conv_x3(const vv_conv_args* a, hipStream_t st, const char** err);
size_t vvk_conv_split_bytes(int Cin_pad, int KW, int rows_pad);
int vvk_conv_split_weights(const float* Wt, int Cin_pad, int KW, int rows_pad, void* Wb, hipStream_t st, const char** err);
int vvk_mrf_pair(const vv_mrf_args* a, hipStream_t st, const char** err);
int vvk_conv_post(const float* in, const float* w, float bias, int16_t* pcm, int ld_pcm, float* wave_f32, int B, int C, int T,
                  int KW, float pre_slope, const int* len_in, hipStream_t st, const char** err);
int vvk_mel_slice(const float* x, int B, int N, int n_mel, const int* ref_len, const int* seq_len, float* out, int T,
                  hipStream_t st, const char** err);
int vvk_mel(const int16_t* audio, int ld_audio, const int* audio_len, const float* window, const float* tw_cos,
            const float* tw_sin, const float* fb, float* mel, int B, int F_max, int n_fft, int hop, int n_mel, hipStream_t st,
            const char** err);
int vvk_pack_cat(int dtype, const float* x, const float* cat, const float* cat_drop, void* out, int ldo, int BN, int n_mel,
                 int cond_dim, int only_x, const int* row_src, hipStream_t st, const char** err);
int vvk_cfg_euler(float* x, const float* pred, int ldp, int BN, int n_mel, float cfg, float dt, const int* row_src, hipStream_t st,
                  const char** err);
int vvk_text_embed(const int* ids, int ld_ids, const int* text_len, const float* emb, const float* pos, float* out, int B, int N,
                   int Dt, int vocab_rows, hipStream_t st, const char** err);
int vvk_dwconv(const float* in, float* out, const float* w, const float* bias, const int* seq_len, int B, int n_seq, int N, int C,
               int KW, hipStream_t st, const char** err);
int vvk_grn(int dtype, void* x, float* sumsq, const float* gamma, const float* beta, const int* seq_len, int B, int n_seq, int N,
            int C, hipStream_t st, const char** err);
int vvk_build_cat(const float* mel, int F_max, const int* ref_len, const float* text, float* cat, float* cat_drop, int B, int N,
                  int n_mel, int Dt, hipStream_t st, const char** err);
int vvk_ref_len(const int* audio_len, int* ref_len, int B, int hop, hipStream_t st, const char** err);
int vvk_decode_len(const int* seq_len, const int* ref_len, int* lens, int B, int n_levels, const int* mult, hipStream_t st,
                   const char** err);
int vvk_dup_len(const int* seq_len, int* out, int B, hipStream_t st, const char** err);
int vvk_row_tables(const int* seq_len, int B, int N, int Rc, int* row_start, int* row_src, int* row_pos, hipStream_t st, const char** err);
int vvk_silu(float* x, size_t n, hipStream_t st, const char** err);
int vvk_resample_poly(const float* x, int n_in, const double* h, int n_taps, int up, int down, int skip, float* y, int n_out,
                      hipStream_t st, const char** err);
int vvk_ingest_pcm(const void* pcm, const long long* desc, int n_clips, long long max_out, float* out, hipStream_t st, const char** err);
size_t vvk_normalize_scratch_bytes(int n_clips, long long total_len);
int vvk_normalize_clips(const float* x, const long long* off, int n_clips, long long max_len, void* scratch, int16_t* out,
                        hipStream_t st, const char** err);
int vvk_cast(int dtype, const float* in, void* out, size_t n, hipStream_t st, const char** err);
int vvk_rope_compact(const float* c, const float* s, float* out, int n, hipStream_t st, const char** err);
int vvk_rope_rows(const float* cs, const int* pos, float* out, int rows, hipStream_t st, const char** err);
int vvk_groupnorm(const float* x, float* y, const float* gamma, const float* beta, int B, int C, int T, int G, float eps, int act,
                  hipStream_t st, const char** err);
