/* e2e_asr_hip.h -- C ABI of the MI355X (gfx950) hot-path library libe2e_asr_hip.so.
 *
 * The reference (shtoshni/e2e_asr) is pure Python/TensorFlow-1.x and has NO native
 * interface to mirror; what each entry point replaces is the TensorFlow op sequence the
 * cited reference lines execute inside sess.run (train.py:297-299).  The Python package
 * e2e_asr_amd binds these with ctypes (e2e_asr_amd/_lib.py); INTEGRATION.md shows the stub.
 *
 * Conventions: plain C, no torch types.  Every pointer is a DEVICE pointer unless the
 * parameter is documented as host.  `stream` is a hipStream_t passed as void*.  All calls
 * are asynchronous on `stream`, allocate nothing, never synchronise (graph-capturable);
 * the caller owns every buffer including workspaces.  Return: 0 ok, -1 invalid argument,
 * -2 launch failure, -3 unsupported shape.  float = IEEE binary32, row-major, weights in
 * the TensorFlow layout the reference checkpoints use (LSTM kernel [in+H, 4H], gate
 * order i,j,f,o, forget bias +1 added at run time -- basic_lstm.py:14-23).
 */
#ifndef E2E_ASR_HIP_H
#define E2E_ASR_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* C[M,N] (+)= op(A).op(B) + bias[N]; fp32 MFMA.  transA: A is [K,M]; transB: B is [N,K].
 * Replaces tf.matmul / conv2d-1x1 call sites: encoder.py:78-81 (input half of the LSTM
 * kernel, hoisted over all timesteps), attn_decoder.py:73 (AttnW). */
int asr_gemm_f32(void* stream, int transA, int transB, int M, int N, int K,
                 const float* A, int lda, const float* B, int ldb,
                 float* C, int ldc, const float* bias, int accumulate);

/* Batched form: batch b uses A + b*strideA, B + b*strideB, C + b*strideC (strides in elements). */
int asr_gemm_f32_batched(void* stream, int transA, int transB, int M, int N, int K,
                         const float* A, int lda, long long strideA, const float* B, int ldb, long long strideB,
                         float* C, int ldc, long long strideC, const float* bias, int accumulate, int batch);

/* Precision of the MFMA GEMMs.  0 (default): fp32 (asr_set_gemm_split says how).  1 (BASELINE config 3, "same model bf16"):
 * products made of whole 128x128 tiles round their fp32 operands to bf16 (round-to-nearest-even) while staging them into LDS
 * and run ONE v_mfma_f32_32x32x16_bf16 product.  2 ("bf16x2"): two bf16 planes per operand value (16 significand bits), three
 * products -- ~2^-16 relative, for callers that want the bf16 pipe's speed within the fp32 north-star tolerance.  In every
 * mode accumulation, outputs, recurrent state, attention, loss and the optimizer stay fp32.  Process-wide; set between steps. */
int asr_set_gemm_precision(int mode);
int asr_get_gemm_precision(void);
/* How the fp32 products of whole 128x128x16 tiles are evaluated (mode 0 of asr_set_gemm_precision).  on (default; the
 * environment variable ASR_GEMM_SPLIT=0 turns it off): every fp32 operand value is split EXACTLY into three bf16 terms
 * while the tile is staged into LDS and the six leading cross products run on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation -- the error of an fp32 GEMM in another summation order (dropped terms < 2^-24 |a||b|), at up to 2.7x
 * the rate of v_mfma_f32_32x32x2_f32.  off: v_mfma_f32_32x32x2_f32 everywhere (bitwise an fmaf chain).  Products with
 * partial tiles or few output tiles always use the latter.  Process-wide; set between steps. */
int asr_set_gemm_split(int on);
int asr_get_gemm_split(void);

/* ---- GEMMs on operands pre-split into bf16 planes ("P3" operands; csrc/gemm_p3.hip) -------------------------------------
 * Same products as asr_gemm_f32 (encoder.py:78-81 input projections, seq2seq_model.py:148 tf.gradients products), with the
 * fp32 -> bf16-plane split taken OUT of the GEMM's k-loop: the producer of an operand writes it as planes.
 * P3 image of a logical matrix X[rows][cols] (cols % 8 == 0) with np planes (3: x = h1 + h2 + h3 exactly = fp32-accurate
 * products; 2: 16 significand bits; 1: plain bf16): 16-byte pieces, piece(r, c8, p) = the 8 bf16 values of plane p for
 * X[r][8 c8 .. 8 c8 + 7] at byte  r * ld8 * 16 np + (c8 * np + p) * 16  (ld8 >= cols / 8: row pitch in 8-element chunks). */
size_t asr_p3_bytes(int rows, int cols, int np);
/* fp32 src[rows][cols] (leading dimension ld) -> P3 image of src (transpose = 0) or of src^T (transpose = 1), tightly packed. */
int asr_p3_split_f32(void* stream, const float* src, int rows, int cols, int ld, void* dst, int np, int transpose);
/* KK form: C[M,N] (+)= A . B^T + bias[N], A = P3[M][K], B = P3[N][K] (both contiguous along the contraction).
 * M % 128 == 0, N % 256 == 0, K % 16 == 0 (ASR_EUNSUPPORTED otherwise).  splits > 1: K split over that many workgroups
 * per tile, float-atomic epilogue. */
int asr_gemm_p3_kk(void* stream, int M, int N, int K, const void* A, int lda8, const void* B, int ldb8, int np,
                   float* C, int ldc, const float* bias, int accumulate, int splits);

/* General form of asr_p3_split_f32: dst_cols (0 = the logical column count rounded up to 8): columns of the image, zero past the
 * logical ones (a tile-aligned image of a narrower matrix); unit_major_h > 0 (transpose = 0): image column d*4h + 4u + g holds
 * source column d*4h + g*h + u -- the gate-major columns of a TF LSTM kernel (beam_search.py:56-59 layout) in the unit-major order
 * the BPTT writes dG in. */
int asr_p3_split_ex(void* stream, const float* src, int rows, int cols, int ld, void* dst, int np, int transpose,
                    int dst_cols, int unit_major_h);
/* Up to 8 asr_p3_split_ex jobs in ONE launch (the per-step weight images of all encoder layers). */
typedef struct { const float* src; int rows, cols, ld; void* dst; int np, transpose, dst_cols, unit_major_h; } asr_p3_split_job;
int asr_p3_split_multi(void* stream, int njobs, const asr_p3_split_job* jobs);
/* RR form: C[M,N] (+)= A^T . B, A = P3[K][M], B = P3[K][N] (the contraction runs over the ROWS of both: the B*T frames of a weight
 * gradient X^T . dG, seq2seq_model.py:148).  M % 128 == 0, N % 256 == 0, K % 16 == 0.  splits: K slices per output tile (0: fill
 * the chip); slices meet in C through float atomics (C zeroed first unless accumulate).  colmap (device int[N] or NULL): product
 * column n goes to column colmap[n] of C. */
int asr_gemm_p3_rr(void* stream, int M, int N, int K, const void* A, int lda8, const void* B, int ldb8, int np,
                   float* C, int ldc, int accumulate, int splits, const int* colmap);

/* Workspace sizes (hx_bytes arguments) may carry this flag: the caller has ALREADY zeroed the workspace (e.g. one fill per train
 * step over an arena of all the step's exchange workspaces) and the library skips its memset launch in front of the persistent
 * kernel.  Each call needs its own zeroed region: a persistent launch leaves its workspace dirty. */
#define ASR_WS_PREZEROED ((size_t)1 << 62)

/* One (Bi)LSTM encoder layer over a whole padded batch -- encoder.py:55-91
 * (bidirectional_dynamic_rnn / dynamic_rnn over BasicLSTMCell with sequence_length).
 * x [B,T,in] batch-major (row stride ldx); out [B,Tout,ndir*H] with fw in [:H], bw in
 * [H:]; rows t >= len[b] (and the pad frames T..Tout-1 of the pyramid, encoder.py:104-110)
 * are written as exact zeros.  gates [B,T,ndir,4H] is workspace (x.K_x + b).  For training pass
 * act [B,T,ndir,H,8] (records {i,j,f,o | c, c_prev, -, -}) and hprev [B,T,ndir,H] (undropped
 * previous hidden states); both NULL for inference.
 * keep_prob < 1 applies DropoutWrapper(output_keep_prob) (encoder.py:49-52) to `out`.
 * err_flag: device int, set non-zero if an inter-workgroup wait timed out (the value names the waiting site: 11-13 recurrent
 * polls, 21 / 51-55 decoder chains, 41-42 granule polls, 31 + 100 * k the XCD agreement at the start of persistent kernel k). H in {64,128,256,512}.
 * kx_cat / bias_cat (optional, ndir = 2; NULL = one product per direction): the input rows of the two kernels side by side,
 * [in, 8H] = [kernel_fw[:in] | kernel_bw[:in]], and [bias_fw | bias_bw] -- the input projection of both directions then runs
 * as one product with N = 8H (asr_lstm_layer_bwd takes the same array for dX as one product with K = 8H). */
size_t asr_lstm_ws_bytes(int B, int H, int ndir);
int asr_lstm_layer_fwd(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                       const int* len, int H, int ndir,
                       const float* kernel_fw, const float* bias_fw,
                       const float* kernel_bw, const float* bias_bw,
                       float* out, int Tout, float* gates, float* act, float* hprev,
                       void* hx_ws, size_t hx_bytes, int* err_flag,
                       float keep_prob, unsigned seed, const float* kx_cat, const float* bias_cat);

/* Plane operands of one (Bi)LSTM layer (asr_lstm_layer_fwd_p3 / asr_lstm_layer_bwd_p3; shapes: asr_lstm_p3_supported).  Every
 * pointer may be NULL = that product keeps its fp32 operands.  The groups-of-four recurrent kernels (H = 256) write the planes. */
typedef struct {
    int np;                 /* planes per value: 3 (fp32-accurate products), 2, 1 (bf16) */
    const void* x_p3;       /* P3 image of the layer input x [B*T][x_cols]: forward projection, weight gradient dK_x */
    int x_cols;             /* >= in_dim, zero columns past it; a multiple of 128 for the weight gradient */
    const void* kxT_p3;     /* forward: P3 image of kx_cat^T [ndir*4H][in_dim] (asr_p3_split_f32, transpose = 1) */
    void* out_p3;           /* forward, WRITTEN: P3 image of out [B*Tout][ndir*H] (= the next layer's x_p3) */
    void* hprev_p3;         /* forward, WRITTEN: P3 image of hprev [B*T][ndir*H] (then `hprev` is not written); read by the backward */
    void* dg_p3;            /* backward, WRITTEN: P3 image of dG [B*T][ndir*4H], unit-major columns inside a direction */
    const void* kxu_p3;     /* backward: P3 image of kx_cat [in_dim][ndir*4H] with unit-major columns (asr_p3_split_ex) */
    const int* colmap;      /* backward: device int[4H], unit-major column 4u+g -> gate-major column g*H+u */
} asr_lstm_p3;
int asr_lstm_p3_supported(int B, int T, int in_dim, int H, int ndir);
int asr_lstm_layer_fwd_p3(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                          const int* len, int H, int ndir,
                          const float* kernel_fw, const float* bias_fw,
                          const float* kernel_bw, const float* bias_bw,
                          float* out, int Tout, float* gates, float* act, float* hprev,
                          void* hx_ws, size_t hx_bytes, int* err_flag,
                          float keep_prob, unsigned seed, const float* kx_cat, const float* bias_cat, const asr_lstm_p3* p3);
int asr_lstm_layer_bwd_p3(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                          const int* len, int H, int ndir,
                          const float* kernel_fw, const float* kernel_bw,
                          const float* dout, int Tout, float* gates, const float* act,
                          const float* hprev, float* dx,
                          float* dkernel_fw, float* dbias_fw, float* dkernel_bw, float* dbias_bw,
                          void* hx_ws, size_t hx_bytes, int* err_flag, float keep_prob, unsigned seed,
                          const float* kx_cat, const asr_lstm_p3* p3);

/* Backward of asr_lstm_layer_fwd (tf.gradients through encoder.py:55-91): persistent BPTT
 * kernel (dG overwrites `gates`), then dX = dG.K_x^T, dK_x = X^T.dG, dK_h = Hprev^T.dG and
 * dbias = colsum(dG).  dkernel and dbias arguments are ACCUMULATED into; dx may be NULL. */
size_t asr_lstm_bwd_ws_bytes(int B, int H, int ndir);
int asr_lstm_layer_bwd(void* stream, const float* x, int B, int T, int in_dim, int ldx,
                       const int* len, int H, int ndir,
                       const float* kernel_fw, const float* kernel_bw,
                       const float* dout, int Tout, float* gates, const float* act,
                       const float* hprev, float* dx,
                       float* dkernel_fw, float* dbias_fw, float* dkernel_bw, float* dbias_bw,
                       void* hx_ws, size_t hx_bytes, int* err_flag, float keep_prob, unsigned seed,
                       const float* kx_cat);

/* out[M,N] (+)= X.Wt^T with Wt given [N,K]: data-gradient products of the decoder backward. */
int asr_linear_wt_fwd(void* stream, const float* x, int ldx, int K, const float* Wt, int ldw,
                      float* out, int ldo, int M, int N, int accumulate);

/* out[N] (+)= column sums of x[M,N] (bias gradients); fixed-order, reproducible. */
/* Pyramid time reduction, encoder.py:94-119 (_get_pyramid_input): x [B,T,F] -> y [B,ceil(T/skip),skip*F], frames past T
 * are zeros (the reference pads when max(len) % skip != 0: pass T = max(len)); len_out = ceil(len_in / skip).  Inside
 * Encoder this is a view (the recurrent kernel writes fw|bw and the zero pad frame into the fused layout); the
 * standalone coalesced copy serves callers with their own layouts.  _bwd scatters dy back to the T un-padded frames. */
int asr_pyramid_reduce_fwd(void* stream, const float* x, const int* len_in, float* y, int* len_out,
                           int B, int T, int F, int skip);
int asr_pyramid_reduce_bwd(void* stream, const float* dy, float* dx, int B, int T, int F, int skip);
/* num_utils.py:6-14: y = 1/(1+exp(-x)) elementwise; y = softmax(x) over a 1-D vector (max-shifted). */
int asr_sigmoid_f32(void* stream, const float* x, float* y, size_t n);
int asr_softmax_f32(void* stream, const float* x, float* y, int n);
int asr_colsum_f32(void* stream, const float* x, int ldx, int M, int N, float* out, int accumulate);
/* out[r,:] = table[idx[r],:] (embedding_lookup) and table_grad[idx[r],:] += g[r,:] (its gradient). */
int asr_gather_rows(void* stream, const float* table, const int* idx, float* out, int rows, int width);
/* n <= 8 row-wise concatenations in ONE launch: dst[i] [rows[i], wa[i]+wb[i]] = [a[i] | b[i]] (row strides lda[i], ldb[i]; host
 * arrays of device pointers / ints).  The encoder builds its per-layer [in,8H] kernel and [8H] bias concatenations of a step
 * with one call (new here: the reference's graph holds the per-direction kernels, encoder.py:55-91). */
int asr_concat2_multi(void* stream, int n, const float* const* a, const float* const* b, float* const* dst,
                      const int* rows, const int* wa, const int* wb, const int* lda, const int* ldb);
int asr_scatter_add_rows(void* stream, float* table_grad, const int* idx, const float* g, int rows, int width);

/* tf.clip_by_global_norm + tf.train.AdamOptimizer over flat buffers (seq2seq_model.py:137-155).
 * asr_sumsq: out[0] = sum(x^2) (two-stage, fixed order; ws >= 1024 floats).
 * asr_clip_adam: g' = g * grad_scale * clip/max(sqrt(sumsq*grad_scale^2), clip);
 *   m,v EMA; p -= lr_t * m/(sqrt(v)+eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller. */
int asr_sumsq_f32(void* stream, const float* x, size_t n, float* ws, float* out);
int asr_clip_adam_f32(void* stream, float* p, float* m, float* v, const float* g, size_t n,
                      const float* sumsq, float grad_scale, float clip_norm, float lr_t,
                      float beta1, float beta2, float eps);

/* out[M,N] = [X1 | X2].W + bias -- `_linear` of attn_decoder.py:117,122,125,151,158.
 * gather1 != NULL: row b of X1 is x1[gather1[b]] (embedding_lookup, decoder.py:101).
 * zero_from != NULL: rows with zero_t >= zero_from[b] are emitted as zeros (raw_rnn). */
int asr_linear_fwd(void* stream, const float* x1, int ld1, int K1, const int* gather1,
                   const float* x2, int ld2, int K2, const float* W, int ldw,
                   const float* bias, float* out, int ldo, int M, int N,
                   const int* zero_from, int zero_t);

/* One BasicLSTMCell step for M rows -- basic_lstm.py:14-23 / attn_decoder.py:148,166. */
int asr_lstm_cell_fwd(void* stream, const float* x1, int ld1, int K1, const int* gather1,
                      const float* h_prev, const float* c_prev, const float* kernel,
                      const float* bias, int H, int M, float* c_out, float* h_out,
                      float* hdrop_out, float* gates_out, float keep, unsigned seed, unsigned step);

/* Fused attention: query projection + score + masked softmax + context --
 * attn_decoder.py:77-93, beam_search.py:150-159. */
size_t asr_attention_lds_bytes(int Te, int H, int A);
int asr_attention_fwd(void* stream, const float* q, int ldq, const float* w_att,
                      const float* b_att, const float* v, const float* hf,
                      const float* enc, const int* enc_len, float* alpha, float* ctx,
                      int B, int Te, int H, int A, int D);

/* Same, with ONE utterance (hf [Te,A], enc [Te,D], enc_len[0]) attended by all B query rows --
 * the hypotheses of a beam (beam_search.py:137-161). */
int asr_attention_shared_fwd(void* stream, const float* q, int ldq, const float* w_att,
                             const float* b_att, const float* v, const float* hf,
                             const float* enc, const int* enc_len, float* alpha, float* ctx,
                             int B, int Te, int H, int A, int D, int shared);

/* asr_masked_ce_fwd and asr_masked_ce_bwd in ONE pass over the logits (the training step knows the gradient scale when it forms
 * the loss): same loss, lse and dlogits bit for bit, one launch less and the logits read once (losses.py:20-35). */
int asr_masked_ce_fwd_bwd(void* stream, const float* logits, const int* targets, const int* len, const float* grad_scale,
                          float* nll_ws, float* lse_ws, float* loss, float* dlogits, int T, int B, int V);
/* losses.py:7-35.  logits [T*B,V] time-major; targets [T,B]; nll_ws,lse_ws [T*B]. */
int asr_masked_ce_fwd(void* stream, const float* logits, const int* targets, const int* len,
                      float* nll_ws, float* lse_ws, float* loss, int T, int B, int V);
int asr_masked_ce_bwd(void* stream, const float* logits, const int* targets, const float* lse_ws,
                      const int* len, const float* grad_scale, float* dlogits, int T, int B, int V);

/* decoder.py:149-150 (argmax feedback) / :176-177 (multinomial feedback, Gumbel-max). */
int asr_next_token(void* stream, const float* logits, int B, int V, int ldl, int* tok_out,
                   int sample, unsigned seed, unsigned step);

/* ---- whole attention-decoder forward (attn_decoder.py:37-172 under tf.nn.raw_rnn) ---- */
typedef struct {
    const float* embedding;   /* [V,E]        model/rnn_decoder_<task>/decoder/embedding */
    const float* attn_enc_w;  /* [D,A]        .../AttnW (squeezed) */
    const float* attn_v;      /* [A]          .../AttnV */
    const float* attn_w;      /* [H,A]        .../rnn/Attention/kernel */
    const float* attn_b;      /* [A] */
    const float* lm_kernel;   /* [E+lmH,4lmH] .../rnn/basic_lstm_cell/kernel */
    const float* lm_bias;
    const float* dec_kernel;  /* [E+H,4H]     .../rnn/basic_lstm_cell_1/kernel */
    const float* dec_bias;
    const float* inp_w;       /* [P+D,E]      .../rnn/InputProjection/kernel, P = H if simple else lmH */
    const float* inp_b;
    const float* ap_w;        /* [H+D,H]      .../rnn/AttnProjection/kernel */
    const float* ap_b;
    const float* out_w;       /* [H,V]        .../rnn/OutputProjection/kernel */
    const float* out_b;
    const float* simple_w;    /* [lmH,H] or NULL  .../rnn/SimpleProjection/kernel */
    const float* simple_b;
} asr_dec_weights;

typedef struct { int B, Te, D, A, H, lmH, E, V, T_out; } asr_dec_dims;

typedef struct {              /* activations: outputs of the forward, inputs of the backward */
    float* hf;                /* [B,Te,A] */
    int*   tok;               /* [T_out,B] token fed at each step; caller pre-fills with dec_inp[:T_out] */
    float* lm_gates;          /* [T_out,B,4lmH] */
    float* lm_c;              /* [T_out,B,lmH] */
    float* lm_h;              /* [T_out,B,lmH] */
    float* lm_hd;             /* [T_out,B,lmH] dropped lm output (NULL when keep_lm >= 1) */
    float* sp;                /* [T_out,B,H] SimpleProjection output (NULL unless simple_w) */
    float* x;                 /* [T_out,B,E] */
    float* dec_gates;         /* [T_out,B,4H] */
    float* dec_c;             /* [T_out,B,H]  (= attention query, decoder.py:79-80) */
    float* dec_h;             /* [T_out,B,H] */
    float* alpha;             /* [T_out,B,Te] */
    float* ctx;               /* [T_out,B,D] */
    float* p;                 /* [T_out,B,H] */
    float* zeros;             /* [B*max(H,lmH,D)] zeros */
    float* y;                 /* [T_out,B,A] attention query projection (saved for the backward; may be NULL) */
    float* w2k;               /* [(P+D+1),4H] scratch: W_inp.K_x and the composed bias (persistent chain path; NULL disables it) */
    void*  chain_ws;          /* asr_decoder_chain_ws_bytes() granule workspace (NULL disables the chain path) */
    int*   err;               /* device int set on an exchange timeout */
    /* persistent LM cell chain (asr_decoder_lm_chain_supported; all NULL = per-step LM cells) */
    float* lm_act;            /* [T_out,B,lmH,8] records {i,j,f,o | c, c_prev, -, -} (format of asr_lstm_layer_fwd) */
    float* lm_hprev;          /* [T_out,B,lmH]   undropped h_{t-1} */
    float* lm_state;          /* [2,2,B,lmH]     (h,c) hand-over between scheduled-sampling segments */
    int*   lm_len;            /* [B] ints, each >= T_out */
    void*  lm_hx;             /* asr_lstm_ws_bytes(B, lmH, 1) bytes */
    void*  greedy_ws;         /* asr_decoder_greedy_ws_bytes() bytes: mode 1 runs as ONE persistent launch (needs w2k, err); NULL = per-step launches */
} asr_dec_ws;

/* mode 0: teacher forcing; 1: greedy (eval, decoder.py:139-154); 2: scheduled sampling
 * with host coins coin[t] (attn_decoder.py:131-139).  logits [T_out*B,V] time-major. */
int asr_attn_decoder_fwd(void* stream, const asr_dec_weights* w, const asr_dec_dims* d,
                         const asr_dec_ws* ws, const float* enc, const int* enc_len,
                         const int* seq_len, int mode, const float* coin_host, float samp_prob,
                         float keep_lm, unsigned seed, float* logits);

typedef struct {              /* backward scratch (device), sized by the caller */
    float* dP;                /* [T_out,B,H]      dLogits . W_out^T */
    float* dQC;               /* [T_out,B,H+D]    dP . W_ap^T = [dq | dctx] */
    float* dY;                /* [T_out,B,A]      grad of the attention query projection */
    float* dXH;               /* [T_out,B,E+H]    [dx | dh_prev] of the outer cell */
    float* dLC;               /* [T_out,B,P+D]    [dlm_out | dctx_prev] */
    float* dlm;               /* [T_out,B,lmH]    (SimpleProjection only, else NULL) */
    float* dEH;               /* [T_out,B,E+lmH]  [demb | dlm_h_prev] */
    float* dc_dec;            /* [B,H]  carry */
    float* dc_lm;             /* [B,lmH] carry */
    float* dhf;               /* [B,Te,A] */
    float* dv_part;           /* [16*B,A] (one partial per workgroup of the backward chain) */
    float* dctx;              /* [T_out,B,D] total gradient w.r.t. each step's context */
    float* emb_all;           /* [T_out,B,E] gathered embeddings */
    void*  chain_ws;          /* asr_decoder_chain_bwd_ws_bytes() bytes, or NULL: per-step launches */
    float* wc;                /* [D,4H] W_inp[P:] . K_x (persistent chain only, else NULL) */
    void*  lm_hx;             /* asr_lstm_bwd_ws_bytes(B, lmH, 1) bytes (persistent LM chain only, else NULL) */
    int    lm_deferred;       /* 1: asr_attn_decoder_bwd leaves the LM cell chain's backward to a later asr_attn_decoder_bwd_lm */
    int    side_busy;         /* 1: the library's side stream still holds an earlier decoder's weight gradients (second task of a
                               * multitask step): the chain's small preparations stay on the caller's stream instead of queueing
                               * behind them */
} asr_dec_bwd_ws;

/* Backward of asr_attn_decoder_fwd.  Weight gradients are ACCUMULATED into `g` (same field
 * layout as the weights, pointing into the flat gradient buffer); denc [B,Te,D] is
 * accumulated into; ws->dec_gates / ws->lm_gates are overwritten. */
int asr_attn_decoder_bwd(void* stream, const asr_dec_weights* w, const asr_dec_weights* g,
                         const asr_dec_dims* d, const asr_dec_ws* ws, const asr_dec_bwd_ws* bw,
                         const float* enc, const int* enc_len, const float* dlogits,
                         float* denc, float keep_lm, unsigned seed);
/* The LM cell chain's part of asr_attn_decoder_bwd (BPTT over all steps, embedding / LM kernel / LM bias gradients) when that
 * call was made with bw->lm_deferred = 1: everything on `stream`.  To be enqueued behind the encoder's backward pass on the same
 * stream -- its persistent BPTT (4 B workgroups) then runs beside the side stream's weight-gradient GEMMs instead of holding
 * up the encoder's BPTT (round 5) -- and before asr_side_join / the optimizer.  New: no counterpart in the reference. */
int asr_attn_decoder_bwd_lm(void* stream, const asr_dec_weights* w, const asr_dec_weights* g, const asr_dec_dims* d,
                            const asr_dec_ws* ws, const asr_dec_bwd_ws* bw, float keep_lm, unsigned seed);
int asr_scatter_add_rows_ld(void* stream, float* table_grad, const int* idx, const float* g, int rows, int width, int ldg);

/* Step-level backward kernels of the launch-based decoder path, for callers that compose their own decoder loop on the host
 * (e2e_asr_amd/multi_decoder.py: MultiRNNCell decoders, decoder.py:66-68).
 * asr_lstm_cell_bwd: pointwise backward of one BasicLSTMCell step whose output went through DropoutWrapper(keep):
 *   dh = dout * mask(seed, step, row, unit) + dh_carry; gates [B][4H]: activated i,j,f,o in, dG (pre-activation gradient) out;
 *   dc_carry [B][H] read and updated.  c_prev / dh_carry may be NULL (first / last step).
 * asr_attn_cell_bwd: attention backward for the query q = c of the cell whose activated gates are in `gates` (dG out), dq
 *   folded into that cell's state gradient.  dqc [B][H+D] = this step's [dq | dctx] from AttnProjection; dctx_carry / dh_carry
 *   (row strides ld_carry / ld_dh) = what step + 1 sent back, or NULL; dhf [B][Te][A], dv_part [B][A] accumulate over steps. */
int asr_lstm_cell_bwd(void* stream, float* gates, const float* c, const float* c_prev, const float* dout, int ld_dout,
                      const float* dh_carry, int ld_dh, float* dc_carry, int B, int H, float keep, unsigned seed, unsigned step);
int asr_attn_cell_bwd(void* stream, const float* q, const float* w_att, const float* b_att, const float* v,
                      const float* hf, const float* enc, const int* enc_len, const float* alpha, const float* y_saved,
                      const float* dqc, const float* dctx_carry, int ld_carry, float* dhf, float* dctx_out, float* dy,
                      float* dv_part, float* gates, const float* c_prev, const float* dh_carry, int ld_dh,
                      float* dc_carry, int B, int Te, int H, int A, int D);

/* The attention backward of asr_attn_cell_bwd alone: dq_out [B][H] = dqc[:, :H] + dy . W_att^T = the gradient w.r.t. the query,
 * for a caller whose query is not an LSTM cell state (e2e_asr_amd/gru_decoder.py: with GRUCell the query is the state itself,
 * /root/reference/decoder.py:79-80).  Everything else as asr_attn_cell_bwd. */
int asr_attn_bwd(void* stream, const float* q, const float* w_att, const float* b_att, const float* v,
                 const float* hf, const float* enc, const int* enc_len, const float* alpha,
                 const float* dqc, const float* dctx_carry, int ld_carry, float* dhf, float* dctx_out, float* dy,
                 float* dv_part, float* dq_out, int B, int Te, int H, int A, int D);

/* One beam-search step for the k live hypotheses of one utterance: BeamSearch.get_top_k (beam_search.py:163-221) up
 * to the two logit vectors (decoder and external LM); the float64 scoring / argpartition stays on the host.
 * d->B = k rows; hf [Te,A] = enc . AttnW (asr_gemm_f32), enc [Te,D], enc_len[0] = Te.  State rows are per hypothesis
 * ([k, .]); `in` and `out` must not alias.  scratch: asr_beam_scratch_floats(k, Te, H, E, lm->P) floats. */
typedef struct {              /* the external LM of shallow fusion (beam_search.py:100-134, map_lm_variables) */
    const float* embedding;   /* [V,E] */
    const float* lstm_kernel; /* [E+H,4H] */
    const float* lstm_bias;
    const float* simple_w;    /* [H,P] or NULL */
    const float* simple_b;
    const float* out_w;       /* [P or H, V] */
    const float* out_b;
    int E, H, P, V;
} asr_lm_weights;
typedef struct { float* dc; float* dh; float* dlc; float* dlh; float* lc; float* lh; float* ctx; } asr_beam_state;
size_t asr_beam_scratch_floats(int k, int Te, int H, int E, int lmP);
/* out row r = in row sel[r] for all seven state fields (parents of the surviving hypotheses, beam_search.py:306-318);
 * widths: dc,dh [H]; dlc,dlh [lmH]; lc,lh [extH]; ctx [D]. */
int asr_beam_gather(void* stream, const int* sel, int k, const asr_beam_state* in, const asr_beam_state* out,
                    int H, int lmH, int extH, int D);
int asr_beam_step(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                  const float* hf, const float* enc, const int* enc_len, const int* tokens,
                  const asr_beam_state* in, const asr_beam_state* out, float* scratch,
                  float* logits, float* logits_lm);
/* asr_beam_step with the parent gather folded in: input row r = row sel[r] of `in` (sel on the device; NULL = identity).
 * Needs both SimpleProjections absent (ASR_EUNSUPPORTED otherwise); `in` and `out` are then simply swapped every step. */
int asr_beam_step_sel(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                      const float* hf, const float* enc, const int* enc_len, const int* tokens, const int* sel,
                      const asr_beam_state* in, const asr_beam_state* out, float* scratch,
                      float* logits, float* logits_lm);
/* asr_beam_step_sel with the three LSTM kernels (decoder's inner LM cell, external LM cell, outer cell) additionally given in the
 * column order the step's tiles read them in (asr_lstm_kernel_tile_order, once per weight set; any of them may be NULL): the
 * cells' weight reads become contiguous, results are bit-identical.  New (the reference has no such call). */
int asr_beam_step_perm(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                       const float* hf, const float* enc, const int* enc_len, const int* tokens, const int* sel,
                       const asr_beam_state* in, const asr_beam_state* out, float* scratch,
                       float* logits, float* logits_lm, const float* lm_kernel_t, const float* ext_kernel_t,
                       const float* dec_kernel_t);
/* out[k][16*tile + 4*unit + gate] = W[k][gate*H + 4*tile + unit] for a TF LSTM kernel W [K, 4H] (H a multiple of 4). */
int asr_lstm_kernel_tile_order(void* stream, const float* W, int K, int H, float* out);
/* Device-resident scoring, selection and bookkeeping of one beam step (beam_search.py:196-214, 290-327): float64
 * log-softmax of both logit vectors, score = log p_dec + lm_weight*log p_lm + carried, top-k per hypothesis then over
 * the continuations, parent = candidate / k, EOS -> finished list and k -= 1.  With asr_beam_gather (sel = ints + kmax)
 * and asr_beam_step (tokens = ints, B = kmax rows; rows >= state[0] are ignored) a whole utterance decodes without a
 * host round trip per step.  Caller initialises state = {1, beam, 0, 0}, cum = 0, ints[0] = GO.  V <= 1024, kmax <= 16. */
typedef struct {
    int*    ints;       /* [2*kmax] tokens | parent rows of the next step's rows */
    double* cum;        /* [kmax] carried scores */
    int*    state;      /* [4] rows fed to the step, beam width left, finished count, step index */
    int*    bp;         /* [max_steps,kmax,2] (parent row, token) of the rows leaving each step */
    int*    fin;        /* [beam,2] (step, parent row) of finished hypotheses, in finishing order */
    double* fin_score;  /* [beam] */
    double* cand;       /* [kmax,16] scratch */
    int*    cand_idx;   /* [kmax,16] scratch */
} asr_beam_book;
int asr_beam_select(void* stream, const float* logits, const float* logits_lm, int V, int kmax, int max_steps,
                    int eos_id, double lm_weight, double word_ins_penalty, const asr_beam_book* book);
/* The whole loop of beam_search.py:255-337 for one utterance in ONE persistent launch (csrc/beam.hip): the tile bodies of
 * asr_beam_step_sel and asr_beam_select run as eight phases per token behind grid barriers, results handed from phase to
 * phase through write-once ring slots (no host round trip, no launch chain).  Initial conditions as for the step calls:
 * d->B = beam, book->state = {1, beam, 0, 0}, book->cum = 0, book->ints[0] = GO.  `ws`: asr_beam_decode_ws_floats(d,
 * lm->H, max_steps) floats, 128-byte aligned (token slots: states, projections, both logit vectors of every step);
 * `barrier`: one zeroed 32-bit word; `err_flag`: the device error word (0 = clean, 61 = a grid barrier timed out).  On
 * return `book` is what the step-by-step loop would have left (bit-identical).  ASR_EUNSUPPORTED -> use the step calls:
 * a SimpleProjection is present, V > 1024, beam > 16, LM vocabulary differs. */
size_t asr_beam_decode_ws_floats(const asr_dec_dims* d, int extH, int max_steps);
int asr_beam_decode(void* stream, const asr_dec_weights* w, const asr_lm_weights* lm, const asr_dec_dims* d,
                    const float* hf, const float* enc, const int* enc_len, float* ws, size_t ws_floats,
                    int max_steps, int eos_id, double lm_weight, double word_ins_penalty,
                    const asr_beam_book* book, unsigned* barrier, int* err_flag);

/* The decoder entry points run the LM cell chain on a library-owned side stream (forked from and
 * ordered against `stream` with events; legal under hipGraph capture).  asr_attn_decoder_bwd leaves
 * LM-chain gradient work in flight on it: call asr_side_join(stream) before reading the gradients. */
int asr_side_join(void* stream);
/* `stream` (a third stream, e.g. the one a gradient all-reduce is launched from) waits for the side-stream work queued so
 * far; unlike asr_side_join the pending join is kept, so a later asr_side_join on the caller's stream still covers
 * everything.  New (the reference is single-device): used by e2e_asr_amd/parallel.py's tail overlap. */
int asr_side_wait(void* stream);
/* Persistent decoder chain (csrc/decoder_chain.hip): used inside asr_attn_decoder_fwd when supported (Te <= 512).
 * asr_decoder_chain_bwd_rows: utterances per 16-workgroup group of the BACKWARD chain for this shape: 2 up to 256 encoder
 * positions; beyond, 2 (two passes over the position slots, hf / dhf in registers) while both utterances' enc rows fit the LDS
 * -- 400 positions at config-2 widths -- else 1 (up to 512).  asr_decoder_chain_rows(Te): the rule of rounds 2-4 (2 up to 256
 * positions, else 1), kept for callers that size by it.  The forward chain keeps 2 wherever both utterances' slices fit. */
int asr_decoder_chain_supported(int B, int Te, int D, int A, int H);
int asr_decoder_chain_rows(int Te);
int asr_decoder_chain_bwd_rows(int Te, int D, int A, int H);
/* Inference graph (mode 1) as one persistent launch (csrc/decoder_greedy.hip): argmax feedback, LM cell, attention,
 * projections of all steps on chip.  Used inside asr_attn_decoder_fwd when supported and ws->greedy_ws is set; only the
 * logits and tok are produced (no saved activations).  The same kernel's training instantiation runs the training graph
 * (modes 0 / 2) in one launch.  Supported: config-2 widths (H 256, D 512, A 128, lmH 256, V <= 1024) and Te <= 419 encoder
 * positions (8 per workgroup up to 256, 16 beyond; ASR_DEC_GREEDY_TEMAX lowers the limit). */
int asr_decoder_greedy_supported(int B, int Te, int D, int A, int H, int lmH, int E, int V);
size_t asr_decoder_greedy_ws_bytes(int B, int D, int A, int H, int lmH, int V);
int asr_decoder_greedy_fwd(void* stream, const float* embedding, const float* lm_kernel, const float* lm_bias,
                           const float* wk, const float* bprime, const float* dec_kh, const float* w_att,
                           const float* b_att, const float* v, const float* ap_w, const float* ap_b,
                           const float* out_w, const float* out_b, const float* hf, const float* enc,
                           const int* enc_len, const int* seq_len, int* tok, float* logits, void* ws, int* err,
                           int B, int Te, int D, int A, int H, int lmH, int E, int V, int T);
size_t asr_decoder_chain_ws_bytes(int B, int D, int A, int H);
size_t asr_decoder_chain_bwd_ws_bytes(int B, int D, int A, int H);
/* LM cell chain of the decoder through the persistent recurrent kernels of csrc/lstm.hip / lstm_bwd.hip
 * (time-major, initial state per scheduled-sampling segment); used with the persistent decoder chain. */
int asr_decoder_lm_chain_supported(int B, int lmH);
int asr_zero_finished_rows(void* stream, float* logits, const int* len, int T, int B, int V);

/* Workgroups of one persistent launch that are guaranteed co-resident on the current device: its compute-unit count
 * (hipDeviceAttributeMultiprocessorCount; 256 on a whole MI355X), optionally lowered by the environment variable
 * ASR_LSTM_MAXWG.  The persistent kernels (recurrent LSTM pair, decoder chains, greedy decoder) size every launch by it;
 * larger batches run as consecutive launches, and a shape whose smallest group does not fit returns ASR_EUNSUPPORTED
 * (-3) -- or selects the per-step launch path -- instead of waiting for workgroups that can never be scheduled. */
int asr_resident_wg_budget(void);
/* bf16 mode (asr_set_gemm_precision(1)) only: 1 = recurrent products of the first-version recurrent kernels on the bf16
 * matrix pipe (round 2's config-3 path); 0 (default since round 3) = the fp32 version-2 recurrences in every mode.
 * New (the reference has one precision); the environment variable ASR_LSTM_MFMA=1 sets the initial value. */
int asr_set_lstm_mfma(int on);
int asr_get_lstm_mfma(void);

/* One GRU encoder layer (tf.nn.rnn_cell.GRUCell under [bidirectional_]dynamic_rnn: /root/reference/encoder.py:42-53 with use_lstm False
 * -- the Encoder.class_params() default, encoder.py:27; the reference CLI always sets use_lstm, encoder.py:187, so this cell is off
 * the measured path and built for completeness, not speed).  Per direction d: wg[d] [in+H][2H] / bg[d] [2H] = gru_cell/gates/{kernel,
 * bias}, wc[d] [in+H][H] / bc[d] [H] = gru_cell/candidate/{kernel,bias} (TF layouts; r | u gate order).  x [B][T][in] (row pitch ldx);
 * out [B][Tout][ndir*H], zeros past each length; bw direction walks t = len-1 .. 0.  gx [B][T][ndir][2H], cx [B][T][ndir][H]:
 * workspaces; hprev and rh [B][T][ndir][H] both non-NULL = save for the backward pass.  Output-only dropout (DropoutWrapper).
 * Backward: gx / cx (what the forward left) are overwritten with dG; wt_ws >= ndir*3*H*H floats; weight / bias gradients are
 * ACCUMULATED into dwg / dbg / dwc / dbc; dx [B][T][in] (or NULL) is overwritten.  H <= 1024.  Everything on `stream`.
 * h0 / h_last [B][ndir][H] (optional): initial state in, final plain state out; dh_last / dh0: their gradients (a caller that
 * composes its own time loop -- the GRU attention decoder of decoder.py:56-59 -- runs T = 1 steps with them). */
int asr_gru_layer_fwd(void* stream, const float* x, int B, int T, int in_dim, int ldx, const int* len, int H, int ndir,
                      const float* const* wg, const float* const* bg, const float* const* wc, const float* const* bc,
                      float* out, int Tout, float* gx, float* cx, float* hprev, float* rh, float keep_prob, unsigned seed,
                      const float* h0, float* h_last);
int asr_gru_layer_bwd(void* stream, const float* x, int B, int T, int in_dim, int ldx, const int* len, int H, int ndir,
                      const float* const* wg, const float* const* wc, const float* dout, int Tout,
                      float* gx, float* cx, const float* hprev, const float* rh, float* wt_ws,
                      float* const* dwg, float* const* dbg, float* const* dwc, float* const* dbc, float* dx,
                      float keep_prob, unsigned seed, const float* dh_last, float* dh0);

/* How the K slices of a split-K weight-gradient product (tf.gradients, seq2seq_model.py:148: X^T . dY with K = B*T) meet in C.
 * 0 (default): float atomics into C (order, hence the last bits, vary from run to run).  1 = DETERMINISTIC mode (environment
 * ASR_WGRAD_SLABS=1): each slice stores its partial tile into a per-stream slab arena owned by the library and a streaming
 * kernel adds the slabs into C in slice order; bias column sums run in two stages and the embedding gradient adds a row's
 * occurrences in token order -- every gradient is bit-reproducible run to run, at +0.1 ... +0.3 ms per config-2 train step.
 * New: the reference computes on one CPU thread and is deterministic by construction. */
int asr_set_wgrad_mode(int slabs);
int asr_get_wgrad_mode(void);
/* table[idx[r]] += g[r] (g rows of pitch ldg), the occurrences of a table row added in ascending r, no atomics; width <= 1024.
 * vocab = rows of `table` (0 if unknown: a slower form with the same sums).  The form asr_scatter_add_rows takes in wgrad mode 1. */
int asr_scatter_add_rows_ordered(void* stream, float* table, int vocab, const int* idx, const float* g, int rows, int width, int ldg);

/* 1 in the race-hunt DEBUG build (libe2e_asr_hip_hunt.so: every publish and poll of the persistent kernels preceded by a
 * random ~4 us delay of one wave in eight, csrc/common.h ASR_RACE_HUNT), 0 in the product library.  New: test tooling. */
int asr_race_hunt_build(void);

/* Optional per-kernel HIP-event timing on the launch stream (bench.py roofline leg).
 * tag: 0 lstm recurrent fwd, 1 lstm recurrent bwd, 2 gemm, 3 decoder fwd, 4 decoder bwd, 5 optimizer.
 * asr_prof_read is a HOST call that synchronises on the recorded events. */
int asr_prof_enable(int on);
/* As asr_prof_enable(1) for the families whose bit (1 << tag) is set in `mask` only: each event pair costs the stream a few
 * microseconds, so a timed run records the one family it reports a roofline for.  New: measurement tooling. */
int asr_prof_enable_mask(unsigned mask);
/* Diagnostic: device buffer (>= 8 u64) for in-kernel phase stamps of the stamped LSTM build (ASR_LSTM_STAMP=1). */
int asr_debug_set_buffer(void* dev_buf);
int asr_prof_read(int tag, double* total_ms, int* launches);
/* One elapsed time per recorded occurrence of `tag`, in recording order (at most cap; *n = how many).  Tag 6
 * (side tail) = from the caller's stream reaching asr_side_join to the end of the side stream's work: may be negative. */
int asr_prof_read_each(int tag, double* out, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif
