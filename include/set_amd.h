/*
 * set_amd.h -- C ABI of libset_amd.so: the MI355X (gfx950) implementation of the
 * Speech-Editing-Toolkit acoustic hot path (FluentSpeech spec_denoiser diffusion
 * loop + HiFi-GAN generator forward).
 *
 * The reference (Zain-Jiang/Speech-Editing-Toolkit) is pure Python/PyTorch and has
 * NO native/FFI boundary of its own; its extension points
 * are Python registries.  This header is therefore the boundary a maintainer of
 * the reference would bind with ctypes.  Every entry point
 * cites the reference code whose arithmetic it replaces (paths relative to the
 * upstream repo root).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (hipMalloc'd / torch CUDA tensors) unless
 *     a parameter is documented as host;
 *   - activations are fp32, channel-major: [B][C][T] with T contiguous (the
 *     layout nn.Conv1d uses in the reference); index tensors are int64;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*);
 *     the caller keeps all buffers alive until it synchronises the stream;
 *   - return value: 0 = ok, negative = SET_E_* error code; nothing throws;
 *   - no global mutable state except a lazily created pool of auxiliary HIP streams (set_diffusion_loop
 *     with n_groups > 1); thread-compatible (one caller thread per stream).
 */
#ifndef SET_AMD_H
#define SET_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SET_AMD_ABI_VERSION 2

/* error codes */
#define SET_OK 0
#define SET_E_INVALID (-1)     /* bad argument (null pointer, shape out of range) */
#define SET_E_UNSUPPORTED (-2) /* shape not supported by the requested implementation */
#define SET_E_LAUNCH (-3)      /* hipLaunch / hip runtime error */

/* activation codes (epilogue `act`) */
#define SET_ACT_NONE 0
#define SET_ACT_RELU 1
#define SET_ACT_GELU 2     /* exact erf GELU (nn.GELU default) */
#define SET_ACT_TANH 3
#define SET_ACT_SOFTPLUS 4 /* beta=1, threshold=20 (nn.Softplus default) */
#define SET_ACT_MISH 5     /* x*tanh(softplus(x)), diffnet.py:14-16 */
#define SET_ACT_LRELU 6    /* leaky_relu(x, act_param) */

/* input prologue codes (`pro`) applied to every in-range input sample */
#define SET_PRO_NONE 0
#define SET_PRO_LRELU 1 /* leaky_relu(x, pro_param)   (hifigan.py:53,55,129,138) */
#define SET_PRO_DIV 2   /* x / pro_param              (diffnet.py:128, / sqrt(L)) */

/* implementation selector for set_conv1d (0 is treated as NAIVE) */
#define SET_IMPL_NAIVE 1 /* one thread per output sample; any shape; device-side cross-check */
#define SET_IMPL_MFMA 2  /* implicit-GEMM on v_mfma_f32_32x32x2_f32; needs packed weights */
#define SET_IMPL_MFMA2 3 /* big-tile variant for wide layers (128*RB rows x 64 frames per block, RB = 4/2/1 for
                            Cout >= 384 / >= 192 / else); needs the image of set_pack_conv_weight_v2;
                            receptive field (K-1)*|dil| <= 64 */

#define SET_IMPL_BF16 4  /* bf16 MFMA operands (v_mfma_f32_32x32x16_bf16), fp32 accumulate / epilogue / HBM tensors; `w` =
                            the image of set_pack_conv_weight_bf16; stride-1 output only.  The training rows' "bf16"
                            arithmetic (BASELINE configs[1]; the reference's autocast hooks: utils/commons/trainer.py:325,343) */
#define SET_IMPL_F16X2 5 /* fp32 operands carried as two fp16 pieces, three fp16 MFMAs per product, fp32 accumulate (fp32-
                          * equivalent results, see SetDiffnetStackArgs.wx3_all); `w` = the image of
                          * set_pack_conv_weight_x2; no in_chan_add; an activation of magnitude >= 32768 raises the sticky
                          * flag of set_conv_x2_range_flag (the caller repeats on an fp32 impl) */

#define SET_IMPL_FEWOUT 6 /* Cout <= 2 over a long sequence (HiFi-GAN conv_post): one thread per four output samples, fp32 VALU,
                          * input streamed once with 16-byte loads; `w` = the RAW weight (w_base / w_s* addressing, as the naive
                          * kernel); needs K <= 9, dil 1, 0 <= pad <= 4, K - pad <= 5, T_in == T_out, T % 4 == 0, no res / mask /
                          * in_chan_add / accumulate, 16-byte aligned in / out */

/* operand type selector of the training GEMMs */
#define SET_DTYPE_F32 0
#define SET_DTYPE_BF16 1         /* fp32 tensors in HBM, rounded to bf16 on the way into LDS */
#define SET_DTYPE_BF16_G16 2     /* weight gradient only: the output gradient g already is bf16 [B][Cout][T] in HBM */
#define SET_DTYPE_BF16_G16_X16 3 /* ... and so is the conv input x (no prologue / per-channel add then) */

int set_abi_version(void);
/* last hip error string of the calling thread's most recent failing call (host pointer, static storage) */
const char *set_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * Generic 1-D convolution, stride 1, with fused prologue/epilogue.  Replaces every
 * F.conv1d / nn.Conv1d / nn.Linear / nn.ConvTranspose1d (as u polyphase convs) call on the path:
 * modules/commons/conv.py:45-49,94 ; nar_tts_modules.py:17,22,83,88 ; mel_encoder.py:7-13 ;
 * diffnet.py:49-52,94,106-107 ; hifigan.py:108,114-115,122 and ResBlock1/2 (:31-48,71-76).
 *
 *   for t in [0, T_iter):  n = t*out_stride + out_off   (skipped unless 0 <= n < T_out)
 *     acc  = sum_{ci,tap} W[co][ci][tap] * P(in[b][ci][t + tap*dil - pad])   (0 outside [0,T_in))
 *            where P(x) = pro(x + in_chan_add[b][ci])      (in_chan_add optional)
 *     v    = act((acc + bias[co]) * alpha)
 *     v    = v + res[b][co][n]                              (res optional)
 *     v    = v * mask[b][n]                                 (mask optional, [B][T_out])
 *     out[b][co][n] = accumulate ? out[b][co][n] + v : v
 *     (accumulate && out_div != 0:  out = (out + v) / out_div -- the last term of a running mean, e.g. the MRF mean
 *      `xs / num_kernels` of hifigan.py:131-137 folded into the last ResBlock's final conv; same summation order)
 *
 * `dil` may be negative (ConvTranspose polyphase taps read in[q - j]).
 * Weights: impl NAIVE reads w[w_base + co*w_sco + ci*w_sci + tap*w_stap];
 *          impl MFMA reads `w` as the packed image written by set_pack_conv_weight.
 * ------------------------------------------------------------------------------------------------ */
typedef struct SetConv1dArgs {
    const float *in;          /* [B][Cin][T_in], batch stride in_bs, channel stride in_cs */
    const float *w;           /* see above */
    const float *bias;        /* [Cout] or NULL */
    const float *res;         /* optional residual, batch stride res_bs, channel stride res_cs */
    const float *mask;        /* optional [B][T_out] */
    const float *in_chan_add; /* optional [B][Cin] (batch stride Cin) */
    float *out;               /* [B][Cout][T_out], batch stride out_bs, channel stride out_cs */
    int64_t in_bs, in_cs, out_bs, out_cs, res_bs, res_cs;
    int64_t w_base, w_sco, w_sci, w_stap; /* NAIVE weight addressing */
    int32_t B, Cin, Cout, K, dil, pad;
    int32_t T_in, T_iter, T_out, out_stride, out_off;
    int32_t pro, act, accumulate, impl;
    float pro_param, act_param, alpha;
    float out_div;            /* see above; 0 = plain accumulate */
} SetConv1dArgs;

int set_conv1d(const SetConv1dArgs *args, void *stream);

/* number of floats of the packed image for (Cout, Cin, K) */
int64_t set_packed_conv_weight_size(int32_t Cout, int32_t Cin, int32_t K);
/* pack w[w_base + co*w_sco + ci*w_sci + tap*w_stap] into MFMA A-fragment order:
 * wp[rb][tap][cp][lane] = W[32*rb + (lane&31)][2*cp + (lane>>5)][tap], zero padded
 * (rb < ceil(Cout/32), cp < CinP/2, CinP = Cin rounded up to 16). */
int set_pack_conv_weight(const float *w, float *wp, int32_t Cout, int32_t Cin, int32_t K,
                         int64_t w_base, int64_t w_sco, int64_t w_sci, int64_t w_stap, void *stream);

/* image for SET_IMPL_F16X2: [32-row block][16-channel group][tap][piece][lane][8 fp16] + 4 floats {2^k, 2^-k, 0, 0}; the
 * weights are multiplied by 2^scale_exp before they are split (pick it so that max |w| 2^k is in [8, 16)); ..._size in fp16
 * ELEMENTS */
int64_t set_packed_conv_weight_x2_size(int32_t Cout, int32_t Cin, int32_t K);
int set_pack_conv_weight_x2(const float *w, void *wp, int32_t Cout, int32_t Cin, int32_t K, int64_t w_base, int64_t w_sco,
                            int64_t w_sci, int64_t w_stap, int32_t scale_exp, void *stream);
/* nn.ConvTranspose1d(Cin, Cout, k, stride u, padding P) on the same kernel, EVERY output phase in one launch (the fp32 path
 * runs one strided-output conv per phase): the rows of the GEMM are (channel, phase) pairs, so with u % 4 == 0 the four rows
 * of an accumulator register group are four consecutive output samples -> 16-byte stores instead of stride-u scatters.
 * w: the module's weight [Cin][Cout][k]; out[b][co][n] = bias[co] + sum_ci sum_j w[ci][co][u j + p] pro(in[b][ci][q - j]),
 * n + P = u q + p (hifigan.py:114-115); in / out contiguous [B][C][T], T_out = (T_in - 1) u - 2 P + k. */
int64_t set_packed_conv_transpose_x2_size(int32_t Cout, int32_t Cin, int32_t k, int32_t u);
int set_pack_conv_transpose_x2(const float *w, void *wp, int32_t Cout, int32_t Cin, int32_t k, int32_t u, int32_t scale_exp,
                               void *stream);
int set_conv_transpose1d_x2(const float *in, const void *wp, const float *bias, float *out, int32_t B, int32_t Cin, int32_t Cout,
                            int32_t k, int32_t u, int32_t P, int32_t T_in, int32_t pro, float pro_param, void *stream);
/* *flag = sticky "an activation left the fp16 range of the F16X2 splitting" word (synchronises); reset != 0 clears it */
int set_conv_x2_range_flag(int32_t *flag, int32_t reset);

/* One iteration of HiFi-GAN's ResBlock1 loop, fused (replaces modules/vocoder/hifigan/hifigan.py:51-58
 *     xt = leaky_relu(x, slope); xt = c1(xt); xt = leaky_relu(xt, slope); xt = c2(xt); x = xt + x
 * with c1 = Conv1d(C, C, K, dilation=dil, padding=dil (K-1)/2), c2 = Conv1d(C, C, K, dilation=1, padding=(K-1)/2)):
 *     out = (c2(lrelu(c1(lrelu(x)) )) + x  [+ previous out if accumulate]) [/ out_div if != 0]
 * accumulate / out_div carry the MRF sum and mean of hifigan.py:131-137 like SetConv1dArgs does.  Two-piece fp16 operands
 * (SET_IMPL_F16X2 arithmetic; w1 / w2 = images of set_pack_conv_weight_x2(C, C, K)); the intermediate xt lives in LDS only
 * (20 C T bytes of HBM traffic per pair become 8 C T).  Results are bit-identical to the two set_conv1d(F16X2) launches.
 * x / out: [B][C][T] with unit frame stride; out must not alias x.  Shapes: see set_resblock_pair_x2_supported.
 * An activation of magnitude >= 32768 raises the sticky flag of set_conv_x2_range_flag. */
typedef struct SetResblockPairArgs {
    const float *x;
    const void *w1;
    const float *b1;
    const void *w2;
    const float *b2;
    float *out;
    int64_t x_bs, x_cs, out_bs, out_cs;  /* batch / channel strides in elements */
    int32_t B, C, K, dil, T;
    int32_t accumulate;
    float slope, out_div;
} SetResblockPairArgs;
int64_t set_sizeof_resblock_pair_args(void);
/* SET_OK if the fused kernel takes the shape (16 <= C <= 256, odd 3 <= K <= 15 -- <= 5 above 128 channels --, dil (K-1) <= 128,
 * T >= 64) */
int set_resblock_pair_x2_supported(int32_t C, int32_t K, int32_t dil, int32_t T);
int set_resblock_pair_x2(const SetResblockPairArgs *args, void *stream);

/* packed image for SET_IMPL_MFMA2 (layout depends on Cout, K and |dil| as well as on the weights) */
int64_t set_packed_conv_weight_v2_size(int32_t Cout, int32_t Cin, int32_t K);
int set_pack_conv_weight_v2(const float *w, float *wp, int32_t Cout, int32_t Cin, int32_t K, int32_t dil,
                            int64_t w_base, int64_t w_sco, int64_t w_sci, int64_t w_stap, void *stream);

/* weight-norm fold  w[i][...] = g[i] * v[i][...] / ||v[i][...]||_2   (torch.nn.utils.weight_norm, dim=0;
 * hifigan.py:32-47,108,114,122).  v,w: [n0][inner]; g: [n0]. */
int set_weight_norm_fold(const float *g, const float *v, float *w, int32_t n0, int64_t inner, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Conditioner glue kernels (FastSpeech-style encoder path; all tiny)
 * ------------------------------------------------------------------------------------------------ */
/* LayerNorm over the channel dim of [B][C][T] (modules/commons/layers.py:5-24 with dim=1),
 * then optional * mask[b][t].  */
int set_layernorm_ch(const float *x, const float *gamma, const float *beta, const float *mask, float *out,
                     int32_t B, int32_t C, int32_t T, float eps, void *stream);
/* out[b][c][t] (+)= scale * table[idx[b][t]][c]     (layers.py:45-50, conv.py:138, fs.py:138,164,188) */
int set_embedding_bct(const int64_t *idx, const float *table, float *out, int32_t B, int32_t T, int32_t C,
                      int32_t n_rows, float scale, int32_t accumulate, void *stream);
/* the same with the scale read from device memory (scale_dev[0]): a learnable scale -- the decoder's pos_embed_alpha,
 * transformer.py:795-796 -- without a device-to-host read in the middle of the step */
int set_embedding_bct_dev_scale(const int64_t *idx, const float *table, float *out, int32_t B, int32_t T, int32_t C,
                                int32_t n_rows, const float *scale_dev, int32_t accumulate, void *stream);
/* mask[b][t] = (sum_c |x[b][c][t]|) > 0 ? 1 : 0     (conv.py:58,108) */
int set_abs_sum_mask(const float *x, float *mask, int32_t B, int32_t C, int32_t T, void *stream);
/* mask[i] = idx[i] > 0 ? 1 : 0                       (fs.py:87,93 ; spec_denoiser.py:163,166) */
int set_index_mask(const int64_t *idx, float *mask, int64_t n, void *stream);
/* out[b][c][t] = mel2ph[b][t] > 0 ? enc[b][c][mel2ph[b][t]-1] : 0   (align_ops.py:21-25) */
int set_expand_states(const float *enc, const int64_t *mel2ph, float *out, int32_t B, int32_t C,
                      int32_t T_txt, int32_t T, void *stream);
/* out[b][c][t] = (x[b][c][t] + (add ? add[b][c] : 0)) * (mask ? mask[b][t] : 1)   (fs.py:91,98,102) */
int set_add_chan_mask(const float *x, const float *add, const float *mask, float *out, int32_t B, int32_t C,
                      int32_t T, void *stream);
/* masked_dur[b][j] = #{t : mel2ph[b][t]*(1-(int)tmask[b][t]) == j+1} * (txt[b][j] != 0)
 * (fs.py:136-137, utils/audio/align.py:71-90).  int64 out [B][T_txt]. */
int set_masked_dur(const int64_t *mel2ph, const float *tmask, const int64_t *txt, int64_t *out, int32_t B,
                   int32_t T, int32_t T_txt, void *stream);
/* pitch bins: f = clamp(2^f0,50,900); f = 0 where uv>0 or pad; mel-scale -> bin 1..255
 * (utils/audio/pitch/utils.py:17-28,71-82 ; fs.py:160-163,182-183).
 * f0_in/uv_in [B][T]; tmask optional: inputs are first multiplied by (1-tmask) (fs.py:160-161);
 * uv_from_logit: treat uv_in as logits (uv = uv_in > 0, fs.py:186); pad optional int64 mel2ph (pad where == 0);
 * outputs optional: f0_denorm fp32 [B][T], coarse int64 [B][T]. */
int set_pitch_coarse(const float *f0_in, const float *uv_in, const float *tmask, const int64_t *mel2ph_pad,
                     int32_t uv_from_logit, float *f0_denorm, int64_t *coarse, int64_t n, void *stream);
/* [B][T][C] <-> [B][C][T] */
int set_transpose_btc_to_bct(const float *in, float *out, int32_t B, int32_t T, int32_t C, void *stream);
int set_transpose_bct_to_btc(const float *in, float *out, int32_t B, int32_t C, int32_t T, void *stream);
/* out = (a + b + c) / div  (b, c optional)           (hifigan.py:131-137 MRF mean `xs / num_kernels`) */
int set_sum_scale(const float *a, const float *b, const float *c, float *out, float div, int64_t n, void *stream);
/* out[i] = a[i]*(1-m[i]) + b[i]*m[i] ; m broadcast over `inner` trailing elements
 * (fs.py:176-177 ; tasks/speech_editing/spec_denoiser.py:53) */
int set_blend_mask(const float *a, const float *b, const float *m, float *out, int64_t n, int64_t inner, void *stream);
/* out[i] = x[i] * (1 - m[i / inner])                 (spec_denoiser.py:164 ref_mels*(1-mask)) */
int set_mul_one_minus_mask(const float *x, const float *m, float *out, int64_t n, int64_t inner, void *stream);
/* LengthRegulator: mel2ph[b][p] = sum_j (j+1) * [cs_prev_j <= p < cs_j], dur = round(dur)*(1-pad)
 * (nar_tts_modules.py:42-72).  Two calls: set_dur_total writes total[b] (int64) = sum_j dur_j;
 * the host reads max(total) to size the output (the reference sizes a tensor the same way, :69). */
int set_dur_total(const float *dur, const int64_t *txt, int64_t *total, int32_t B, int32_t T_txt, void *stream);
int set_length_regulate(const float *dur, const int64_t *txt, int64_t *mel2ph, int32_t B, int32_t T_txt,
                        int32_t T_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * DiffNet + diffusion loop
 * ------------------------------------------------------------------------------------------------ */
/* out[j][n] = sin(t[n]*w_j) (j < dim/2), cos(t[n]*w_{j-dim/2}) (j >= dim/2), w_j = exp(-j ln(1e4)/(dim/2-1))
 * (diffnet.py:34-46).  t: fp32 [n] (host fills it with the integer step ids).  out: [dim][n]. */
int set_sinusoid_embed(const float *t, float *out, int32_t dim, int32_t n, void *stream);

/* z = sigmoid(y[:, :C]) * tanh(y[:, C:])   y [B][2C][T] -> z [B][C][T]    (diffnet.py:76-77) */
int set_gate(const float *y, float *z, int32_t B, int32_t C, int32_t T, void *stream);
/* x_out = (x_in + o[:, :C]) / sqrt(2) ; skip (+)= o[:, C:]               (diffnet.py:80-81,128) */
int set_res_skip(const float *x_in, const float *o, float *x_out, float *skip, int32_t B, int32_t C, int32_t T,
                 int32_t first, void *stream);

/* One fused DiffNet residual layer (diffnet.py:60-81) for residual_channels == 256:
 *   y   = W_dil (*) (x_in + d) + b_dil + condproj      k=3, dilation `dil`, zero padded
 *   z   = sigmoid(y[:256]) * tanh(y[256:])
 *   o   = W_out z + b_out
 *   x_out = (x_in + o[:256]) / sqrt(2);  skip = first ? o[256:] : skip + o[256:]
 * x_in/x_out/skip: [B][256][T] (must not alias); condproj: [B][512][T] slab with batch stride cp_bs
 * (= conditioner_projection(cond) + its bias, hoisted out of the step loop: it does not depend on t);
 * d[b][c] = dstep[b*d_bs + c*d_cs]  (diffusion_projection(mlp(emb(t)))[b][c]);
 * w1p / w2p: images written by set_pack_diffnet_layer. */
typedef struct SetDiffnetLayerArgs {
    const float *x_in;
    const float *condproj;
    const float *dstep;
    const float *w1p;
    const float *b_dil;
    const float *w2p;
    const float *b_out;
    float *x_out;
    float *skip;
    int64_t cp_bs, d_bs, d_cs;
    int32_t B, T, dil, first;
    /* diagnostic, normally NULL: [gridDim.y*gridDim.x][8] uint64 s_memtime stamps written by wave 0 of every
     * block at the phase boundaries (start, tile staged, GEMM1 done, gate done, z staged, GEMM2 done, end) */
    uint64_t *dbg_clock;
} SetDiffnetLayerArgs;
int set_diffnet_layer(const SetDiffnetLayerArgs *args, void *stream);
/* floats needed for w1p (512x768) and w2p (512x256) */
int64_t set_diffnet_w1p_size(void);
int64_t set_diffnet_w2p_size(void);
/* pack dilated_conv.weight [512][256][3] and output_projection.weight [512][256][1] (diffnet.py:63,66) */
int set_pack_diffnet_layer(const float *w_dil, const float *w_out, float *w1p, float *w2p, void *stream);
/* every layer of a stack in one launch per image family: layer q's weights at w_dil + q wd_ls / w_out + q wo_ls (element strides),
 * w1p / w2p = [L][set_diffnet_w1p_size()] / [L][set_diffnet_w2p_size()]; w1w / w2w (both or neither) = the Winograd images
 * [L][set_diffnet_w1w_size()] / [L][512 * 256] of set_pack_diffnet_layer_wino */
int set_pack_diffnet_layers(const float *w_dil, const float *w_out, int64_t wd_ls, int64_t wo_ls, float *w1p, float *w2p,
                            float *w1w, float *w2w, int32_t L, void *stream);

/* All L residual layers of one DiffNet pass in ONE persistent launch (residual_channels == 256).
 * Up to 2 x #CU resident blocks pull (layer, tile) tasks in layer-major order from an atomic queue; a task
 * (l, i) starts when tiles i-1, i, i+1 of layer l-1 are published (per-tile epoch flags; agent-scope
 * release/acquire), so no CU idles at layer boundaries (a 416-tile layer does not divide over 256 CUs).
 * Tiles are 64 frames when a layer has >= 3 x #CU of them, else 32 frames (keeps runnable tasks > workers).
 * Layer l reads (l even ? xa : xb) and writes the other buffer; skip accumulates in place; results are
 * bit-identical to L calls of set_diffnet_layer.  sync_ws: >= 16 + 2*B*ceil(T/32) int32, zeroed by the call;
 * sync_ws[1] != 0 afterwards means a dependency wait timed out (SET_E_LAUNCH is NOT raised asynchronously). */
typedef struct SetDiffnetStackArgs {
    float *xa, *xb, *skip;        /* [B][256][T] each */
    const float *condproj;        /* layer l, batch b slab at condproj + l*cp_ls + b*cp_bs : [512][T] */
    const float *dstep;           /* d[l][b][c] = dstep[l*d_ls + b*d_bs + c*d_cs] */
    const float *w1p_all;         /* [L][512*768]  packed (set_pack_diffnet_layer) */
    const float *w2p_all;         /* [L][512*256] */
    const float *b_dil_all;       /* [L][512] */
    const float *b_out_all;       /* [L][512] */
    int32_t *sync_ws;
    int64_t cp_bs, cp_ls, d_bs, d_cs, d_ls;
    int32_t B, T, L, dilation_cycle_length;
    /* optional Winograd F(2,3) images (set_pack_diffnet_layer_wino), [L][512*256*4] and [L][512*256]: when both are
     * given, dilation_cycle_length <= 4 and the batch has at least 0.68 64-frame tiles per CU, the k=3 conv runs as four
     * 512x256 GEMMs over output pairs (2/3 of its MACs); results then agree with the direct kernels to fp32
     * rounding (a few 1e-6), not bit for bit.  SET_AMD_WINO=0 in the environment disables it. */
    const float *w1w_all;
    const float *w2w_all;
    /* optional, Winograd kernel only (training forward): x_all [L+1][B][256][T] replaces the xa/xb ping-pong (layer l
     * reads slab l, writes slab l+1; slab 0 = input), save_y [L][B][512][T] receives the gate/filter pre-activations
     * and save_z [L][B][256][T] the gated activations, i.e. what the backward pass needs (diffnet.py:73-77). */
    float *x_all;
    float *save_y;
    float *save_z;
    /* optional sticky error word (never cleared by the library): set to 1 if a dependency wait ran into its spin
     * limit and the launch gave up (sync_ws[1] says the same for the last launch only) */
    int32_t *err_flag;
    /* optional, small batches (4*B*ceil(T/32) <= 2 x #CU): row-split images [L][512*768] / [L][512*256] in the layout
     * [32-row block v = 4w + rb][k-step][lane] (the same values as w1p_all / w2p_all, which are [w][k-step][lane][rb])
     * plus a workspace z_ws of B*ceil(T/32)*256*32 floats.  Every 32-frame tile is then computed by four co-operating
     * blocks (one 32-row block per wave), bit-identical to the direct kernels.  SET_AMD_SPLIT=0 disables it. */
    const float *w1s_all;
    const float *w2s_all;
    float *z_ws;
    /* optional split-operand images [L][set_diffnet_layer_x3_image_size(x3_mode)] 16-bit (set_pack_diffnet_layer_x3): every
     * fp32 weight and activation is carried as a sum of 16-bit pieces and every product is formed by several 16-bit MFMAs
     * with fp32 accumulation -- fp32-equivalent accuracy (error against fp64 not larger than the fp32 MFMA chain's) at a
     * fraction of the fp32 matrix-pipe time:
     *   x3_mode 3: three bf16 pieces, six products (fp32 range, error ~2^-23 per product);
     *   x3_mode 2: two fp16 pieces, three products (error ~2^-21 per product; an activation of magnitude >= 32768 sets
     *              *err_flag = 2 instead of overflowing silently; weights are pre-scaled by a power of two at pack time).
     * Used from the Winograd crossover on (batches that fill the chip); results agree with the fp32 kernels to fp32
     * rounding, not bit for bit.  SET_AMD_X3=0 disables it, =2 forces it at any size. */
    const void *wx3_all;
    int32_t x3_mode;
} SetDiffnetStackArgs;
int set_diffnet_stack(const SetDiffnetStackArgs *args, void *stream);
int64_t set_sizeof_diffnet_stack_args(void);

/* which kernel set_diffnet_stack picks for this shape on the current device: 0 direct/64-frame tiles,
 * 1 direct/32-frame tiles, 2 Winograd, 3 row-split (diagnostics: bench.py names the kernel in its roofline object);
 * 4 split-operand 3 x bf16, 5 split-operand 2 x fp16; images: bit 0 = Winograd images given, bit 1 = row-split images +
 * z_ws given, bit 2 = split-operand images of mode 3 given, bit 3 = of mode 2 */
int64_t set_diffnet_layer_x3_image_size(int32_t mode); /* 16-bit elements per layer (mode 2 or 3; -1 otherwise) */
/* k1 / k2: the weights of the dilated conv / the output projection are multiplied by 2^k before they are split (mode 2: pick
 * k so that max |w| 2^k is in [8, 16), which keeps the residual pieces of all but the tiniest weights normal; mode 3: 0) */
int set_pack_diffnet_layer_x3(const float *w_dil /*[512][256][3]*/, const float *w_out /*[512][256]*/, void *img, int32_t mode,
                              int32_t k1, int32_t k2, void *stream);
int set_diffnet_stack_variant(int B, int T, int dilation_cycle_length, int images);
/* Non-zero when the split-operand kernel of variant 5 runs GEMM 1 in its Winograd F(2,3) form for this shape (diffnet_stack_x3v_kernel: the
 * k = 3 dilated conv of diffnet.py:70 over output pairs, 3/4 of the layer's matrix instructions; dilation 1, even T): the number of 32-frame
 * column blocks per tile -- 3 (96-frame tiles: shapes with a tile chain for every CU) or 2 (64-frame tiles); 0 = the direct form
 * (SET_AMD_X3_WINO=0 pins it, =2 / =3 pin the tile width).  `images` as for set_diffnet_stack_variant. */
int set_diffnet_stack_x3_winograd(int B, int T, int dilation_cycle_length, int images);
int64_t set_diffnet_w1w_size(void);
int set_pack_diffnet_layer_wino(const float *w_dil, const float *w_out, float *w1w, float *w2w, void *stream);

/* posterior step (spec_denoiser.py:86-101):  x_prev = c1*x0 + c2*x_t + (t != 0) * exp(0.5*logvar) * eps
 * per-batch scalars coef4[b*coef_bs + {0,1,2,3}] = {c1, c2, logvar, nonzero} (coef_bs = 0: shared by the
 * batch); eps == NULL -> counter-based Philox4x32-10 + Box-Muller noise keyed by (seed, offset + i/4).
 * x_prev may alias x_t. */
int set_posterior_step(const float *x0, const float *x_t, const float *eps, const float *coef4, int64_t coef_bs,
                       float *x_prev, int32_t B, int64_t per_batch, uint64_t seed, uint64_t offset, void *stream);
/* x_t = a[b]*x_start + s[b]*eps   (q_sample, spec_denoiser.py:126-132); ab2[b] = {a, s}; optional * nonpad[b][t] */
int set_q_sample(const float *x_start, const float *eps, const float *ab2, const float *nonpad, float *x_t,
                 int32_t B, int32_t M, int32_t T, void *stream);
/* fill with N(0,1): Philox4x32-10 + Box-Muller (throughput runs; spec_denoiser.py:180) */
int set_randn(float *out, int64_t n, uint64_t seed, uint64_t offset, void *stream);
/* Graph capture of a training step: every Philox kernel launched through set_randn / set_posterior_step / set_dropout adds the device
 * word *dev_word to its seed argument (NULL = off, the default).  A captured step carries the seeds of the step it was captured at; the
 * replay of step k stores (seed_k - seed_captured) in the word before it launches the graph and draws exactly the numbers the eager
 * step k draws.  Process-wide; the word must stay allocated while it is set. */
int set_rng_seed_delta(const uint64_t *dev_word);
/* Host-side stream ordering (the training path's leaf stream, autograd_ops.leaf_work): work enqueued on `after` from now on waits for
 * everything enqueued on `first` so far (hipEventRecord + hipStreamWaitEvent on a cached event; slot < 32 selects the event, one per
 * direction and device: slot s belongs to device s / 2 and its event is created there, whatever the calling thread's current device is).
 * Capturable: inside a stream capture the pair is a cross-stream edge. */
int set_stream_order(void *first, void *after, int32_t slot);
/* Progress markers of a stream (round 6; the leaf stream's operand bookkeeping, autograd_ops.leaf_work -- the reference's autograd engine
 * frees a backward operand as soon as its node has run, utils/commons/trainer.py:340-350 `loss.backward()`): set_stream_mark records marker
 * `slot` (0 .. 63, an event of device `dev`, created on first use) on `stream`; set_stream_mark_done(slot) returns 1 once everything enqueued on
 * that stream before the mark has finished, 0 while it has not, < 0 on error (never blocks). */
int set_stream_mark(void *stream, int32_t slot, int32_t dev);
int set_stream_mark_done(int32_t slot);
/* a non-blocking stream of the lowest priority of the current device (the leaf stream); never destroyed */
int set_stream_create_low_priority(void **out);

/* Whole reverse loop (spec_denoiser.py:178-184 + p_sample :103-108 + DiffNet.forward diffnet.py:110-132)
 * for residual_channels == 256.  All weights pre-packed by the caller; workspaces provided by the caller.
 * Launch sequence: input projection once, then per step {layer stack (set_diffnet_stack, or L set_diffnet_layer
 * launches when persistent == 0), step boundary}.  The step boundary -- skip projection + output projection +
 * posterior update (explicit or Philox noise) + the NEXT step's input projection -- is one fused kernel when
 * T % 4 == 0 and M <= 96 (bit-identical to the separate set_conv1d / set_posterior_step launches it replaces, which
 * remain the path for other shapes; SET_AMD_FUSED_BOUNDARY=0 forces them). */
typedef struct SetDiffLoopArgs {
    /* problem */
    int32_t B, T, M, L, steps, dilation_cycle_length;
    /* state */
    float *x;              /* [B][M][T]  in: x_T, out: x_0 */
    const float *noise;    /* optional explicit eps: [steps][B][M][T] in execution order (i = steps-1..0); NULL -> Philox */
    uint64_t seed;
    const float *condproj; /* [B][L*512][T]  hoisted conditioner projections (+bias) */
    const float *dstep;    /* [L*256][steps] : column s = diffusion_projection_l(mlp(emb(s))) */
    const float *coef4;    /* [steps][4] host-computed {c1, c2, logvar, nonzero} per step id, DEVICE pointer */
    /* packed weights */
    const float *w_in_p;   /* input_projection packed (Cout 256, Cin M, K 1) */
    const float *b_in;
    const float *w1p_all;  /* [L][512*768] packed dilated-conv weights (set_pack_diffnet_layer) */
    const float *w2p_all;  /* [L][512*256] packed output-projection weights */
    const float *b_dil_all; /* [L][512] */
    const float *b_out_all; /* [L][512] */
    const float *w1w_all;   /* optional Winograd images (see SetDiffnetStackArgs), used by the persistent path */
    const float *w2w_all;
    const float *w1s_all;   /* optional row-split images + workspace (see SetDiffnetStackArgs), small batches */
    const float *w2s_all;
    float *z_ws;
    const void *wx3_all;    /* optional split-operand images (see SetDiffnetStackArgs), batches that fill the chip */
    int32_t x3_mode;
    /* optional set_pack_conv_weight_x2 images of skip_projection (256 x 256), output_projection (M x 256) and
     * input_projection (256 x M): with x3_mode == 2 the fused step boundary then runs on the two-piece fp16 operands as well
     * (SET_AMD_BOUNDARY_X2=0 keeps the fp32 MFMA boundary kernel); the opt-in bf16 loop (img16_all) takes it whenever the images are
     * given -- it can raise err_flag = 2 (an activation outside the fp16 split range), which the caller must read and answer by
     * repeating the loop without these images */
    const void *w_skip_x2, *w_outp_x2, *w_in_x2;
    const float *w_skip_p; /* skip_projection packed */
    const float *b_skip;
    const float *w_outp_p; /* output_projection packed (Cout M, Cin 256) */
    const float *b_outp;
    /* workspaces, each [B][256][T] floats */
    float *ws_x0, *ws_x1, *ws_skip, *ws_h;
    float *ws_x0pred; /* [B][M][T] */
    /* optional: HOST array [steps] receiving the wall time in ms of the L-layer span of each step (hipEvent
     * pairs on the launching stream; mean over utterance groups); the call then synchronises before returning */
    float *layer_span_ms;
    /* optional: HOST float receiving the wall time in ms of the whole loop (event pair on `stream`); synchronises */
    float *loop_ms;
    /* utterance groups (<= 8; 0/1 = one): the batch is split into n_groups contiguous slices that run as
     * independent chains on auxiliary HIP streams (forked from / joined to `stream` with events), so the tail of one
     * group's layer launch overlaps the next layer of another group.  Results are bit-identical for any n_groups. */
    int32_t n_groups;
    /* persistent != 0: run the L layers of every step as one set_diffnet_stack launch (needs sync_ws,
     * >= 160 + 2*B*ceil(T/32) int32); 0: one set_diffnet_layer launch per layer */
    int32_t persistent;
    int32_t *sync_ws;
    int32_t *err_flag; /* optional sticky error word, see SetDiffnetStackArgs.err_flag (1: time-out, 2: activation out of the
                        * fp16 split range) */
    /* bf16-operand loop (opt-in, NOT the parity path; img16_all != NULL selects it): every residual layer runs as one
     * set_diffnet_layer_fwd_bf16 launch reading cond [B][192][T] (the conditioner projection is a K-chunk of the layer's
     * GEMM, condproj is not used), images [L][set_diffnet_layer_bf16_image_size()], conditioner biases [L][512] */
    const float *cond;
    const void *img16_all;
    const float *b_cond_all;
    /* bf16 loop: with bf16_ws_floats >= set_diffnet_layers_bf16_scratch_floats(B, T, 0, n, dilation_cycle_length) for the group size
     * n in use and dilation_cycle_length <= 2 the layers run set_diffnet_layers_bf16_plan() per launch
     * (set_diffnet_layers_fwd_bf16); NULL: one launch per layer.  SET_AMD_BF16_FUSE=n overrides the group size (1 = per layer). */
    float *bf16_ws;
    int64_t bf16_ws_floats;
} SetDiffLoopArgs;
int set_diffusion_loop(const SetDiffLoopArgs *args, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Training: backward kernels, losses, optimizer.
 * Input gradients of a convolution are a set_conv1d call on the output gradient with transposed weight
 * addressing (w_sco <-> w_sci) and negated dil / pad; the pieces below are what set_conv1d cannot express.
 * ------------------------------------------------------------------------------------------------ */
/* dW[co][ci][tap] += sum_{b,t} G[b][co][t] * P(X[b][ci][t + tap*dil - pad]),  P as in set_conv1d
 * (chan_add, pro).  impl SET_IMPL_MFMA: split-K GEMM on fp32 MFMA with fp32 atomics; else one thread per weight. */
int set_conv1d_wgrad(const float *g, const float *x, const float *chan_add, float *dw, int32_t B, int32_t Cin,
                     int32_t Cout, int32_t K, int32_t dil, int32_t pad, int32_t T, int32_t T_in, int32_t pro,
                     float pro_param, int32_t impl, void *stream);

/* Deterministic weight gradient (same arithmetic contract as set_conv1d_wgrad, MFMA path): the frames are cut into S
 * slices, slice z writes its partial dW to scratch[z][Cout][Cin][K] with plain stores, then dW += sum_z scratch[z] in
 * slice order -- no atomics, bit-identical from run to run.  dtype SET_DTYPE_F32: fp32 MFMA operands; SET_DTYPE_BF16:
 * g and x rounded to bf16 (RNE) on the way into LDS, fp32 accumulation.  `scratch` must hold
 * set_conv1d_wgrad_scratch_floats(...) floats and may be reused by the next call on the same stream. */
int64_t set_conv1d_wgrad_scratch_floats(int32_t B, int32_t Cin, int32_t Cout, int32_t K, int32_t T, int32_t dtype);
int set_conv1d_wgrad_det(const void *g, const void *x, const float *chan_add, float *dw, int32_t B, int32_t Cin,
                         int32_t Cout, int32_t K, int32_t dil, int32_t pad, int32_t T, int32_t T_in, int32_t pro,
                         float pro_param, int32_t dtype, float *scratch, int64_t scratch_floats, void *stream);
/* The same for `groups` equal GEMMs in ONE launch (+ one ordered reduce) -- the weight gradients of all L residual layers of the
 * DiffNet at once: group q reads g + q g_gs, x + q x_gs, chan_add + q add_gs and ADDS into dw + q dw_gs (strides in elements of the
 * respective type).  With L x the tiles a few frame slices per group fill the chip: 13 - 64 slices per GEMM become 2 - 4 (as much
 * less partial-sum traffic), 120 launches become 6.  bf16 dtypes only, no prologue. */
int64_t set_conv1d_wgrad_grouped_scratch_floats(int32_t groups, int32_t B, int32_t Cin, int32_t Cout, int32_t K, int32_t T);
int set_conv1d_wgrad_det_grouped(const void *g, const void *x, const float *chan_add, float *dw, int32_t groups, int64_t g_gs,
                                 int64_t x_gs, int64_t add_gs, int64_t dw_gs, int32_t B, int32_t Cin, int32_t Cout, int32_t K,
                                 int32_t dil, int32_t pad, int32_t T, int32_t T_in, int32_t dtype, float *scratch,
                                 int64_t scratch_floats, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused DiffNet residual layer, bf16 MFMA operands, for the TRAINING path (residual_channels 256, hidden_size 192,
 * kernel 3; modules/speech_editing/spec_denoiser/diffnet.py:60-81 and its transpose).  One launch per layer and
 * direction; see csrc/diffnet_bf16.hip for the arithmetic and the HBM traffic it removes.
 *   img    packed bf16 weight images of ONE layer (set_pack_diffnet_layer_bf16; set_diffnet_layer_bf16_image_size()
 *          bf16 elements): forward GEMM 1 / GEMM 2 and the three transposed images of the backward
 *   y16/z16/dy16/do16   bf16 [B][512|256][T]: pre-gate, gated activation, and their gradients (MFMA operands of the
 *          weight-gradient GEMMs, inputs of the gate derivative)
 *   dstep  d[b][c] = dstep[b*d_bs + c*d_cs]   (diffusion_projection(step embedding), per utterance)
 * ------------------------------------------------------------------------------------------------ */
typedef struct SetDiffnetLayerBf16Args {
    const float *x_in;   /* [B][256][T] */
    float *x_out;        /* [B][256][T] = (x_in + o_res) / sqrt2 */
    float *skip;         /* [B][256][T] += o_skip (first != 0: = o_skip) */
    const float *cond;   /* [B][192][T] */
    const float *dstep;
    const void *img;
    const float *b_dil, *b_cond, *b_out; /* [512] each */
    uint16_t *y16;       /* [B][512][T] bf16 out (both NULL: inference, nothing is saved) */
    uint16_t *z16;       /* [B][256][T] bf16 out */
    int64_t d_bs, d_cs;
    int32_t B, T, dil, first;
} SetDiffnetLayerBf16Args;
int64_t set_sizeof_diffnet_layer_bf16_args(void);
int64_t set_diffnet_layer_bf16_image_size(void);
int set_pack_diffnet_layer_bf16(const float *wdil /*[512][256][3]*/, const float *wcond /*[512][192]*/,
                                const float *wout /*[512][256]*/, void *img, void *stream);
/* L layers in one launch: layer q reads wdil + q * sdil (etc., element strides between the layers' tensors) and writes image q of
 * img [L][image_size] -- the flat optimizer lays every layer's parameters out at one stride, so the whole stack is re-rounded by ONE
 * launch after an update instead of L. */
int set_pack_diffnet_layers_bf16(const float *wdil, const float *wcond, const float *wout, int64_t sdil, int64_t scond, int64_t sout,
                                 void *img, int32_t L, void *stream);
int set_diffnet_layer_fwd_bf16(const SetDiffnetLayerBf16Args *args, void *stream);
/* Several consecutive residual layers per launch (bf16 operands, inference: nothing is saved): a block keeps its tile on chip for
 * `nl` (<= 16) layers l0 .. l0 + nl - 1 (x and the conditioner are read once; the fp32 residual rows stay in registers; the running
 * skip sum stays in registers (64-frame tiles) or goes through L2 per layer (128-frame tiles)) and stores the tile - 2 H frames whose
 * receptive field stayed inside the tile, H = sum of the dilations 2^((l0 + m) % dilation_cycle_length), m = 1 .. nl - 1.  The tile
 * width is chosen per launch: 128 frames when B * ceil(T / (128 - 2 H)) blocks still fill 3/4 of the CUs, else 64.  Same arithmetic
 * per frame as nl set_diffnet_layer_fwd_bf16 launches (x AND the skip sum bit-identical at both tile widths: the 128-frame shape adds
 * the layers' skip contributions in layer order as well).  Arrays are the per-layer operands of those launches laid out one layer after the other: img
 * [nl][image_size], b_dil / b_cond / b_out [nl][512], dstep of layer l at dstep + l * d_ls (l counted from layer 0 of the network).
 * x_in != x_out.  scratch >= set_diffnet_layers_bf16_scratch_floats(B, T, l0, nl, dilation_cycle_length) floats (128-frame tiles:
 * a block's private copy of its skip rows between the layers of the group).
 * set_diffnet_layers_bf16_plan(B, T, L, dilation_cycle_length): the group size the reverse loop uses for a stack of L layers. */
typedef struct SetDiffnetLayersBf16Args {
    const float *x_in;   /* [B][256][T] input of layer l0 */
    float *x_out;        /* [B][256][T] output of layer l0 + nl - 1 */
    float *skip;         /* [B][256][T] += sum of the nl skip contributions (first != 0: =) */
    const float *cond;   /* [B][192][T] */
    const float *dstep;
    const void *img;
    const float *b_dil, *b_cond, *b_out;
    float *scratch;
    int64_t scratch_floats;
    int64_t d_bs, d_cs, d_ls;
    int32_t B, T, l0, nl, dilation_cycle_length, first;
} SetDiffnetLayersBf16Args;
int64_t set_sizeof_diffnet_layers_bf16_args(void);
int64_t set_diffnet_layers_bf16_scratch_floats(int32_t B, int32_t T, int32_t l0, int32_t nl, int32_t dilation_cycle_length);
int32_t set_diffnet_layers_bf16_plan(int32_t B, int32_t T, int32_t L, int32_t dilation_cycle_length);
int set_diffnet_layers_fwd_bf16(const SetDiffnetLayersBf16Args *args, void *stream);
/* debug: block (1,1) of the bf16 layer kernels stamps s_memtime at its phase boundaries into buf[0..7] (NULL = off) */
int set_debug_bf16_phase_buffer(uint64_t *buf);
/* debug: lane 0 of one block (tile 1, part 1) of the row-split stack kernel ADDS the s_memtime ticks it spends in each of
 * its 8 phases (wait for the previous layer, stage, GEMM 1, gate + z publish, wait for z, z load, GEMM 2, epilogue +
 * publish), summed over the layers, to buf[0..7] (NULL = off) */
int set_debug_split_phase_buffer(uint64_t *buf);
/* debug: lane 0 of block 0 of the split-operand stack kernel adds the s_memtime ticks of its phases (claim + wait, stage,
 * GEMM 1, gate, GEMM 2, epilogue + publish), summed over its tasks, to buf[0..5] and its task count to buf[7] */
int set_debug_x3_phase_buffer(uint64_t *buf);
/* debug: thread 0 of every 16th block of batch row 1 of the fused ResBlock-pair kernel adds the s_memtime ticks of its phases (x
 * chunk wait + split, GEMM 1 issue, GEMM 1 drain + epilogue 1, GEMM 2 issue, GEMM 2 drain + epilogue 2) to buf[0..4], block count
 * to buf[7] */
int set_debug_resblock_phase_buffer(uint64_t *buf);

typedef struct SetDiffnetLayerBf16BwdArgs {
    const float *dx_out; /* [B][256][T] gradient w.r.t. x_out; NULL = zero (the last layer's x_out feeds nothing) */
    const float *dskip;  /* [B][256][T] gradient w.r.t. the skip sum (the same tensor for every layer) */
    const uint16_t *y16; /* saved by the forward */
    const void *img;
    float *dx;           /* [B][256][T] out: gradient w.r.t. x_in (= w.r.t. x_in + d) */
    uint16_t *dy16;      /* [B][512][T] bf16 out: gradient w.r.t. the pre-gate y */
    uint16_t *do16;      /* [B][512][T] bf16 out: [dx_out / sqrt2 ; dskip] */
    float *dcond;        /* [B][192][T] (+)= Wcond^T dy   (dcond_first != 0: written, not accumulated) */
    float *part_dbo;     /* [B * tiles][512] per-tile sums of d_o over frames  (bias gradient of the output projection) */
    float *part_dby;     /* [B * tiles][512] per-tile sums of dy               (bias gradient of dilated conv = conditioner) */
    float *part_dd;      /* [B * tiles][256] per-tile sums of Wdil^T (*) dy    (gradient of the per-utterance step offsets) */
    int32_t B, T, dil, dcond_first;
} SetDiffnetLayerBf16BwdArgs;
int64_t set_sizeof_diffnet_layer_bf16_bwd_args(void);
int32_t set_diffnet_layer_bwd_bf16_tiles(int32_t T, int32_t dil); /* tiles per utterance = rows per b of the part_* arrays */
int set_diffnet_layer_bwd_bf16(const SetDiffnetLayerBf16BwdArgs *args, void *stream);
/* the ordered partial sums of one layer's backward in one launch: db_out[512] += sum part_dbo, db_dil[512] and db_cond[512] +=
 * sum part_dby (over all B*tiles rows), dd[b*dd_bs + c] = sum over the tiles of utterance b of part_dd (c < 256) */
int set_diffnet_layer_bwd_reduce(const float *part_dbo, const float *part_dby, const float *part_dd, int32_t B, int32_t tiles,
                                 float *db_out, float *db_dil, float *db_cond, float *dd, int64_t dd_bs, void *stream);
/* the same for L layers swept with one tile count, in ONE launch: partials [L][B*tiles][512|512|256], layer q's targets at
 * db_* + q * s_* (element strides between the layers' bias gradients) and dd + q * dd_ls */
int set_diffnet_layers_bwd_reduce(const float *part_dbo, const float *part_dby, const float *part_dd, int32_t B, int32_t tiles, int32_t L,
                                  float *db_out, int64_t s_out, float *db_dil, int64_t s_dil, float *db_cond, int64_t s_cond, float *dd,
                                  int64_t dd_bs, int64_t dd_ls, void *stream);
/* out[g][j] (+)= scale * sum_{r < rows} part[(g*rows + r)*cols + j]  in row order (deterministic reduction of per-tile partials) */
int set_partial_rows_sum(const float *part, float *out, int32_t groups, int32_t rows, int32_t cols, int32_t accumulate,
                         float scale, void *stream);

/* Every residual layer's diffusion_projection of the step embedding in one launch (reference: modules/speech_editing/spec_denoiser/
 * diffnet.py:66,72 -- one Linear(C, C) per layer on the [N, C] embedding):  out[n][l*C + co] = b_l[co] + sum_ci W_l[co][ci] h[ci][n]
 * with h [C][N], W_l = w + l*w_ls ([C][C] row-major), b_l = b + l*b_ls, out [N][L*C].  fp32 FMAs.  C % 64 == 0, N <= 64.
 * ..._bwd: dh[ci][n] = sum_{l,co} W_l[co][ci] g[n][l*C+co] (partials in scratch, summed in (l, co-quarter) order),
 * dW_l[co][ci] += sum_n g[n][l*C+co] h[ci][n], db_l[co] += sum_n g[n][l*C+co]; scratch >= set_step_proj_bwd_scratch_floats floats. */
int set_step_proj_fwd(const float *h, const float *w, int64_t w_ls, const float *b, int64_t b_ls, float *out, int32_t L, int32_t C,
                      int32_t N, void *stream);
int64_t set_step_proj_bwd_scratch_floats(int32_t L, int32_t C, int32_t N);
int set_step_proj_bwd(const float *h, const float *g, const float *w, int64_t w_ls, float *dh, float *dw, int64_t dw_ls, float *db,
                      int64_t db_ls, float *scratch, int32_t L, int32_t C, int32_t N, void *stream);
/* its two halves on their own (set_step_proj_bwd = _dh then _dw): dh feeds the backward chain, dW / db are parameter gradients
 * the host side may launch on its second stream */
int set_step_proj_bwd_dh(const float *g, const float *w, int64_t w_ls, float *dh, float *scratch, int32_t L, int32_t C, int32_t N,
                         void *stream);
int set_step_proj_bwd_dw(const float *h, const float *g, float *dw, int64_t dw_ls, float *db, int64_t db_ls, int32_t L, int32_t C,
                         int32_t N, void *stream);

/* bf16 weight image for SET_IMPL_BF16: wp[tap][chunk][row][32] bf16, row < Cout rounded up to 128, chunk < ceil(Cin/32),
 * zero padded; ..._size returns the number of bf16 ELEMENTS (2 bytes each). */
int64_t set_packed_conv_weight_bf16_size(int32_t Cout, int32_t Cin, int32_t K);
int set_pack_conv_weight_bf16(const float *w, void *wp, int32_t Cout, int32_t Cin, int32_t K, int64_t w_base,
                              int64_t w_sco, int64_t w_sci, int64_t w_stap, void *stream);
/* All bf16 images of a model in ONE launch (after an optimizer step).  descs_dev: n descriptors IN DEVICE MEMORY, sorted by
 * `start` = index of the image's first element in the concatenated element space [0, total); per image the layout of
 * set_pack_conv_weight_bf16 (CoutP = Cout rounded up to 128, CinP = Cin rounded up to 32). */
typedef struct SetPackBf16Desc {
    const float *w;
    void *wp;
    int64_t w_base, w_sco, w_sci, w_stap, start;
    int32_t Cout, Cin, K, CoutP, CinP, pad_;
} SetPackBf16Desc;
int64_t set_sizeof_pack_bf16_desc(void);
int set_pack_conv_weights_bf16_batch(const SetPackBf16Desc *descs_dev, int32_t n, int64_t total, void *stream);
/* The same for the fp32 images (training in the reference's default precision, egs/spec_denoiser.yaml:4): every image of
 * set_pack_conv_weight (kind 0) / set_pack_conv_weight_v2 (kind 1) a step used, re-packed in ONE launch after the optimizer step.
 * The caller sets w, wp, w_base .. w_stap, Cout, Cin, K and `start`; set_fill_pack_f32_desc (host) fills kind and the layout
 * fields and returns the image's element count (-1: bad arguments).  The table lives on the device. */
typedef struct SetPackF32Desc {
    const float *w;
    float *wp;
    int64_t w_base, w_sco, w_sci, w_stap, start;
    int32_t Cout, Cin, K, CinP, kind, RB, ch_max, pad_;
} SetPackF32Desc;
int64_t set_sizeof_pack_f32_desc(void);
int64_t set_fill_pack_f32_desc(SetPackF32Desc *d, int32_t kind, int32_t dil);
int set_pack_conv_weights_f32_batch(const SetPackF32Desc *descs_dev, int32_t n, int64_t total, void *stream);
/* out[c] += sum_{b,t} x[b][c][t]   (bias gradients) */
int set_channel_sum(const float *x, float *out, int32_t B, int32_t C, int32_t T, void *stream);

/* Deterministic reductions of the training path.  The entry points above that accumulate with fp32 atomics
 * (set_channel_sum, set_weighted_sum, set_sumsq, the `sums` pass of set_dur_loss / set_pitch_loss, set_embedding_bwd,
 * set_expand_states_bwd) give results whose last bits depend on the arrival order of the blocks.  The *_det variants
 * below write one partial result per block / slice into `scratch` and combine them in a fixed order: bit-identical from
 * run to run (what makes a resumed training run repeat the uninterrupted one exactly).  `scratch` may be reused by the
 * next call on the same stream.  Sizes (floats): channel_sum 2048 + C; weighted_sum 1024; sumsq 2048; dur 4 B;
 * pitch 4 ceil(B T / 256); scatter_rows B * set_scatter_rows_segments(T) * n_rows * C. */
int set_channel_sum_det(const float *x, float *out, int32_t B, int32_t C, int32_t T, float *scratch, void *stream);
int set_weighted_sum_det(const float *x, const float *w, float *out, int64_t n, int64_t inner, float *scratch, void *stream);
int set_sumsq_det(const float *g, float *out, int64_t n, float *scratch, void *stream);
int set_dur_loss_sums_det(const float *dur_pred, const int64_t *mel2ph, const int64_t *txt, const int64_t *word_id,
                          float *sums, int32_t B, int32_t T, int32_t T_txt, int32_t n_words, float *scratch, void *stream);
int set_pitch_loss_sums_det(const float *pp, const float *f0, const float *uv, const int64_t *mel2ph, float *sums,
                            int32_t B, int32_t T, float *scratch, void *stream);
/* Scatter-add of gradient rows, no atomics.  doutT: the gradient TRANSPOSED to [B][T][C].
 * mode 0 (embedding backward, layers.py:45-50): table[idx[b][t]][c] += scale * doutT[b][t][c]  (idx clamped to the table,
 *        padding_idx skipped), table [n_rows][C];
 * mode 1 (expand_states backward, align_ops.py:21-25): table[b][idx[b][t] - 1][c] += doutT[b][t][c] for idx > 0,
 *        table [B][n_rows][C] (the gradient of the encoder output, transposed). */
int32_t set_scatter_rows_segments(int32_t T);
int set_scatter_rows_det(const int64_t *idx, const float *doutT, float *table, int32_t B, int32_t T, int32_t C,
                         int32_t n_rows, float scale, int32_t padding_idx, int32_t mode, float *scratch, void *stream);
/* out[row] = scale * sum_t x[row][t] */
int set_row_sum(const float *x, float *out, int64_t rows, int32_t T, float scale, void *stream);
/* G = dY * mask[b][t] * alpha * [act == RELU: y > 0]   (epilogue backward of set_conv1d; act NONE or RELU) */
int set_conv_epilogue_bwd(const float *dy, const float *y, const float *mask, float *g, int32_t B, int32_t C,
                          int32_t T, int32_t act, float alpha, void *stream);
/* y = act(z) / dz = dy * act'(z)  for the SET_ACT_* codes */
int set_act_fwd(const float *z, float *y, int64_t n, int32_t act, float p, void *stream);
int set_act_bwd(const float *z, const float *dy, float *dz, int64_t n, int32_t act, float p, void *stream);
/* dz = (dy * act'(z)) * scale: act backward followed by the producing conv's output scale alpha (set_conv1d's `alpha`), one launch */
int set_act_bwd_scaled(const float *z, const float *dy, float *dz, int64_t n, int32_t act, float p, float scale, void *stream);
/* backward of set_gate: y [B][2C][T] (saved pre-gate), dz [B][C][T] -> dy [B][2C][T] */
int set_gate_bwd(const float *y, const float *dz, float *dy, int32_t B, int32_t C, int32_t T, void *stream);
/* backward of set_res_skip: dx = dx_out/sqrt2 ; d_o[:, :C] = dx_out/sqrt2 ; d_o[:, C:] = dskip */
int set_res_skip_bwd(const float *dx_out, const float *dskip, float *dx, float *d_o, int32_t B, int32_t C, int32_t T,
                     void *stream);
/* backward of set_layernorm_ch; dgamma/dbeta are accumulated (+=).  `partial`: scratch of
 * set_layernorm_ch_bwd_scratch(B, C, T) floats (contents irrelevant) for a contention-free two-pass reduction of
 * dgamma/dbeta; NULL falls back to one atomic per (block, channel), which serialises across XCDs. */
int64_t set_layernorm_ch_bwd_scratch(int32_t B, int32_t C, int32_t T);
int set_layernorm_ch_bwd(const float *x, const float *gamma, const float *mask, const float *dy, float *dx,
                         float *dgamma, float *dbeta, float *partial, int32_t B, int32_t C, int32_t T, float eps,
                         void *stream);
/* the same with dx = (LayerNorm gradient) + add [B][C][T]: the residual branch of a pre-LN sub-block joins inside the launch */
int set_layernorm_ch_bwd_add(const float *x, const float *gamma, const float *mask, const float *dy, const float *add, float *dx,
                             float *dgamma, float *dbeta, float *partial, int32_t B, int32_t C, int32_t T, float eps, void *stream);
/* dtable[idx[b][t]][c] += scale * dout[b][c][t], except for row `padding_idx` (-1: none), whose gradient stays 0
 * as with nn.Embedding(padding_idx=...) (modules/commons/layers.py:45-50) */
int set_embedding_bwd(const int64_t *idx, const float *dout, float *dtable, int32_t B, int32_t T, int32_t C,
                      int32_t n_rows, float scale, int32_t padding_idx, void *stream);
/* denc[b][c][mel2ph[b][t]-1] += dout[b][c][t] */
int set_expand_states_bwd(const int64_t *mel2ph, const float *dout, float *denc, int32_t B, int32_t C, int32_t T_txt,
                          int32_t T, void *stream);
/* inverted dropout with a Philox keep-mask keyed by (seed, offset + i/4): y = keep ? x/(1-p) : 0.  Calling it on the
 * output gradient with the same (seed, offset) is the backward (nar_tts_modules.py:20,86 predictor dropout). */
int set_dropout(const float *x, float *y, int64_t n, float p, uint64_t seed, uint64_t offset, void *stream);
/* w[f] = (sum_m |target[f][m]|) != 0    (weights_nonzero_speech, utils/nn/seq_utils.py:33-37) */
int set_frame_weight(const float *target, float *w, int64_t frames, int32_t M, void *stream);
/* out[0] += sum_i x[i] * (w ? w[i/inner] : 1) */
int set_weighted_sum(const float *x, const float *w, float *out, int64_t n, int64_t inner, void *stream);
/* absd = |pred - target|, sgn = sign(pred - target)  (either may be NULL)   (l1_loss, speech_base.py:223-229) */
int set_l1_elem(const float *pred, const float *target, float *absd, float *sgn, int64_t n, void *stream);
/* out[i] = a[i] * (w ? w[i/inner] : 1) * (scale_dev ? scale_dev[0] : 1) * scale */
int set_scale_bcast(const float *a, const float *w, float *out, int64_t n, int64_t inner, const float *scale_dev,
                    float scale, void *stream);
/* SSIM (utils/metrics/ssim.py:12-44) on [B][H][W] images: 11x11 gaussian (sigma 1.5), zero padded.
 * `bias` (6.0, speech_base.py:247) is added to the in-range pixels of both images.
 * filter: mu1, mu2, E[x^2], E[y^2], E[xy];  map: 1 - ssim and d ssim / d(mu1, E[x^2], E[xy]);
 * bwd: dimg1 = F(gm) + 2 img1 F(g11) + img2 F(g12). */
int set_ssim_filter(const float *img1, const float *img2, float bias, float *mu1, float *mu2, float *s11, float *s22,
                    float *s12, int32_t B, int32_t H, int32_t W, void *stream);
int set_ssim_map(const float *mu1, const float *mu2, const float *s11, const float *s22, const float *s12,
                 float *one_minus, float *d_mu1, float *d_s11, float *d_s12, int64_t n, void *stream);
int set_ssim_bwd(const float *img1, const float *img2, float bias, const float *gm, const float *g11, const float *g12,
                 float *dimg1, int32_t B, int32_t H, int32_t W, void *stream);
/* duration losses (speech_editing_base.py:58-90).  Pass 1 (ddur NULL): sums[0..3] += {sum nonpad*d^2, sum nonpad,
 * sum wordmask*dw^2, sum wordmask}.  Pass 2 (ddur set): ddur = gscale * d(lam_p*s0/s1 + lam_w*s2/s3)/d dur_pred. */
int set_dur_loss(const float *dur_pred, const int64_t *mel2ph, const int64_t *txt, const int64_t *word_id, float *sums,
                 const float *final_sums, float *ddur, int32_t B, int32_t T, int32_t T_txt, int32_t n_words,
                 float lam_p, float lam_w, float gscale, void *stream);
/* pitch losses (speech_editing_base.py:92-108) on channel-major pitch_pred [B][2][T] (row 0 f0, row 1 uv logit).
 * Pass 1: sums += {sum nonpad*bce, sum nonpad, sum nv*|df0|, sum nv};  pass 2: dpp = gradient. */
int set_pitch_loss(const float *pp, const float *f0, const float *uv, const int64_t *mel2ph, float *sums,
                   const float *final_sums, float *dpp, int32_t B, int32_t T, float lam_uv, float lam_f0, float gscale,
                   void *stream);
/* out[0] += sum g[i]^2 */
int set_sumsq(const float *g, float *out, int64_t n, void *stream);
/* torch.optim.AdamW step (speech_base.py:163-170) over flat buffers on the gradient g*grad_scale (grad_scale =
 * 1/world after a SUM all-reduce), with clip_grad_norm_ folded in (base_task.py:129-133): the gradient is further
 * scaled by min(1, max_norm / (sqrt(sumsq[0])*grad_scale + 1e-6)) when sumsq != NULL (sumsq = sum g^2). */
int set_adamw(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1, float beta2, float eps,
              float weight_decay, int32_t step, const float *sumsq, float max_norm, float grad_scale, void *stream);
/* The same update with lr and the bias corrections 1 - beta^step read from device memory hyper = [lr, bc1, bc2] (a captured training
 * step is replayed for every update: its kernel arguments are frozen).  set_adamw_hyper computes bc1, bc2 on the host exactly as
 * set_adamw does (out2[0], out2[1]). */
int set_adamw_hyper(float beta1, float beta2, int32_t step, float *out2);
int set_adamw_dev(float *p, const float *g, float *m, float *v, int64_t n, const float *hyper, float beta1, float beta2, float eps,
                  float weight_decay, const float *sumsq, float max_norm, float grad_scale, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Attention building blocks (CampNet rows; modules/speech_editing/commons/transformer.py)
 * ------------------------------------------------------------------------------------------------ */
/* Strided batched fp32 GEMM on MFMA:  C[b](m,n) = alpha * sum_k A[b](m,k) * B[b](k,n)  (+ C if accumulate).
 * Batch index b = bo * n_inner + bi; every operand is addressed base + bo*X_bo + bi*X_bi + row*rs + col*cs, so the
 * heads of a [B][heads*d][T] activation are batches and Q, K^T, V, O need no transposes (transformer.py:344-347,365,
 * 403-405 do view/transpose/contiguous instead).  The same entry point serves the four gradient products. */
typedef struct SetBmmArgs {
    const float *A, *B;
    float *C;
    int64_t a_bo, a_bi, a_ms, a_ks;
    int64_t b_bo, b_bi, b_ks, b_ns;
    int64_t c_bo, c_bi, c_ms, c_ns;
    int32_t n_outer, n_inner, M, N, K;
    float alpha;
    int32_t accumulate;
} SetBmmArgs;
int64_t set_sizeof_bmm_args(void);
int set_bmm(const SetBmmArgs *args, void *stream);
/* Fused multi-head attention (replaces the bmm -> softmax -> bmm of modules/speech_editing/commons/transformer.py:361-406 and
 * the self-attention of torch's multi_head_attention_forward that EncSALayer / DecSALayer call, :505-513,:566-575):
 *     o[b][h] = softmax_keys(scale q[b][h] k[b][h]^T  (+ key padding -> fill)) v[b][h]
 * on the channel-major layout: element (b, head h, channel c < head_dim, frame t) of q sits at q + b q_bs + (h head_dim + c) q_cs
 * + t, likewise k / v (Tk frames) and o; q / k / v may be channel slices of one packed projection.  The scores never reach HBM
 * (online softmax over 32-key tiles); lse [B][heads][2][Tq] = the softmax statistics the backward needs: row max m and sum l = sum exp(s - m)
 * (log-sum-exp = m + log l; kept apart because log l vanishes in fp32 next to a -1e8 fill).  p (optional):
 * the probabilities [B][heads][Tq][Tk] for callers that return them (transformer.py:396-410).  kpm (optional) [B][Tk], != 0 =
 * padded key; fill = -inf (torch: a fully padded row is NaN) or -1e8 (transformer.py:381-386: a fully padded row is uniform).
 * bf16 != 0: bf16 MFMA operands (fp32 scores / softmax / accumulation), else fp32 MFMA throughout.  head_dim: 32, 64 or 96. */
typedef struct SetAttnArgs {
    const float *q, *k, *v;
    float *o, *lse, *p;
    const float *kpm;
    int64_t q_bs, k_bs, v_bs, o_bs;
    int32_t q_cs, k_cs, v_cs, o_cs;
    int32_t B, heads, head_dim, Tq, Tk;
    float scale, fill;
    int32_t bf16;
} SetAttnArgs;
int64_t set_sizeof_attn_args(void);
int set_attention(const SetAttnArgs *args, void *stream);
/* Gradients of set_attention: P is recomputed from q, k and fwd.lse (fwd.o = the forward's output, fwd.p unused).  d_o has o's
 * layout; dq / dk / dv are addressed like q / k / v with their own strides and are overwritten; delta [B][heads][Tq] is scratch
 * (sum_c dO O).  Two launches (per query tile: dq; per key tile: dk, dv), no atomics: bit-stable from run to run. */
typedef struct SetAttnBwdArgs {
    SetAttnArgs fwd;
    const float *d_o;
    float *delta, *dq, *dk, *dv;
    int64_t dq_bs, dk_bs, dv_bs;
    int32_t dq_cs, dk_cs, dv_cs;
} SetAttnBwdArgs;
int64_t set_sizeof_attn_bwd_args(void);
int set_attention_bwd(const SetAttnBwdArgs *args, void *stream);
/* y[row] = softmax_fp32(x[row]) over `cols`; logits where key_padding_mask[row / rows_per_batch][col] != 0 are
 * replaced by `fill` first (-inf: F.multi_head_attention_forward; -1e8: transformer.py:381-386).  mask may be NULL. */
int set_softmax_rows(const float *x, const float *key_padding_mask, float *y, int64_t rows, int32_t cols,
                     int64_t rows_per_batch, float fill, void *stream);
/* ds = p * (dp - sum_cols(p * dp)) */
int set_softmax_rows_bwd(const float *p, const float *dp, float *ds, int64_t rows, int32_t cols, void *stream);
/* make_positions with padding_idx 0 (utils/nn/seq_utils.py:6-18): pos[b][t] = rank of entry t among the non-zero
 * entries of row b (1-based), 0 where the entry is 0.  Entries: int64 tokens[B][T], or (tokens == NULL) the fp32
 * values x[b*x_bs + t] (first channel of a [B][C][T] tensor: transformer.py:795 numbers frames by `x[..., 0]`). */
int set_make_positions(const int64_t *tokens, const float *x, int64_t x_bs, int64_t *pos, int32_t B, int32_t T,
                       void *stream);
/* out[b][i] = mean over heads of p[b][h][i]   (transformer.py:414-416, need_head_weights=False) */
int set_head_mean(const float *p, float *out, int32_t B, int32_t heads, int64_t n, void *stream);
/* out[b][c][t] = x[b][c][t]*(1-m[b][t]) + e[c]*m[b][t]   (campnet.py:56 `mels*(1-mask) + mask_emb*mask`) */
int set_mask_fill_chan(const float *x, const float *e, const float *m, float *out, int32_t B, int32_t C, int32_t T,
                       void *stream);
/* out[c] += sum_{b,t} d[b][c][t] * m[b][t]   (gradient of mask_emb) */
int set_masked_channel_sum(const float *d, const float *m, float *out, int32_t B, int32_t C, int32_t T, void *stream);

/* MFMA fragment-layout self test: runs a 32x32xK product through v_mfma_f32_32x32x2_f32 with the layout
 * this library assumes and returns the max abs error vs an in-kernel scalar reference via *max_err (HOST).
 * Synchronises.  Used by tests to pin the hardware layout assumption. */
int set_selftest_mfma(float *max_err_host, void *stream);

/* sizeof() of the argument structs, so a foreign-language binding can verify its mirror of the layout */
int64_t set_sizeof_conv1d_args(void);
int64_t set_sizeof_diffnet_layer_args(void);
int64_t set_sizeof_diff_loop_args(void);

#ifdef __cplusplus
}
#endif
#endif /* SET_AMD_H */
