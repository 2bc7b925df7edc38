/*
 * sepr.h - C ABI of libsepr_hip.so: the SepReformer separator forward path for MI355X (gfx950).
 *
 * The reference (dmlguq456/SepReformer) is pure PyTorch Python: it has no FFI layer, every op below is
 * reached there through torch.nn modules calling aten kernels.  This header is therefore the boundary
 * the reference *would* bind if its modules dispatched to a native library: one entry point per fused
 * block of the hot path, each citing the reference module it replaces (paths relative to
 * /root/reference/models/SepReformer_Base_WSJ0/).  See INTEGRATION.md for the ctypes stubs.
 *
 * Conventions
 *   - plain C: raw device pointers, ints, a stream handle; no torch / C++ types cross the boundary;
 *   - every function returns an int status: SEPR_OK (0) or a negative SEPR_E* code; nothing throws;
 *   - no allocation inside: scratch is a caller-provided workspace (sepr_workspace_bytes tells how much);
 *   - the library never frees or retains caller memory, holds no global mutable state besides the
 *     opt-in profiler, the latched A/B switches and the per-thread deferred-finisher window (both below), and launches on the stream it is given (NULL = default stream): re-entrant
 *     from several host threads, one per device, like torch.nn.parallel.data_parallel's replicas
 *     (reference engine.py:64,98,130,167);
 *   - all activations are fp32, *channel-last* rows: a tensor the reference holds as [b, C, T] is
 *     [b, T, C] here (row = one frame); "rows" of a batch are (b, t) flattened;
 *   - weights are fp32 in the layouts noted per struct ("packed" = re-laid-out once on the host by
 *     sepreformer_amd/pack.py from the reference's state_dict tensors).
 */
#ifndef SEPR_H_
#define SEPR_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SEPR_VERSION 412 /* minor*100 + patch ("ABI 4.12").  Any struct-layout or context-size change bumps the MINOR number
                            (3.01 -> 3.03 grew sepr_ega_w under a patch bump: a caller built against 3.01 would have passed a short
                            struct).  A binding must compare sepr_version() with the SEPR_VERSION it was written against before its
                            first call: sepreformer_amd/lib.py refuses to load a library whose version differs. */

#define SEPR_OK 0
#define SEPR_EINVAL (-1)     /* bad shape / unsupported size / null pointer */
#define SEPR_EWORKSPACE (-2) /* workspace too small */
#define SEPR_EHIP (-3)       /* a HIP launch failed (see sepr_last_hip_error) */

typedef void* sepr_stream_t; /* hipStream_t */

/* ---- weight bundles (device pointers) --------------------------------------------------------- */

/* Optional second-precision ("bf16x3") form of one projection: the weight matrix pre-split into bf16
 * hi/lo planes in MFMA fragment order [N/16][K/32][plane][lane][8] with any LayerNorm gamma folded in,
 * and the bias with LayerNorm beta folded in (sepreformer_amd/pack.py::pack_x3).  wp == NULL selects the
 * exact f32-MFMA core for that projection; both cores share prologues and epilogues. */
typedef struct {
  const void* wp;
  const float* bias;
} sepr_x3_w;

/* GCFN, modules/network.py:46-66.  F -> 6F -> (dwconv3, GLU) 3F -> F */
typedef struct {
  const float* ln_g; /* [F]     net1.0.weight */
  const float* ln_b; /* [F]     net1.0.bias */
  const float* w1;   /* [6F,F]  net1.1.weight */
  const float* b1;   /* [6F]    net1.1.bias */
  const float* dw_w; /* [3,6F]  depthwise.weight, packed tap-major */
  const float* dw_b; /* [6F]    depthwise.bias */
  const float* w2;   /* [F,3F]  net2.2.weight */
  const float* b2;   /* [F]     net2.2.bias */
  const float* ls;   /* [F]     Layer_scale.layer_scale */
  sepr_x3_w x3_up;   /* net1 (LayerNorm folded) */
  sepr_x3_w x3_down; /* net2.2 */
  /* optional fully fused form (bf16x3, F = 64 or 128; pack.py::pack_gcfn_fused); both or none */
  const void* fused_w1p;   /* per 32-channel hidden chunk: [v0 v1 g0 g1][F/32][plane][64][8] bf16 (gamma folded)
                              + 4 KB fp32 constants [2][b1v b1g wv0 wv1 wv2 wg0 wg1 wg2 cbv cbg][16] */
  const void* fused_w2p;   /* [3F/32][F/16][plane][64][8] bf16, permuted k-slot order */
} sepr_gcfn_w;

/* CLA, modules/network.py:159-187 (eval-mode BatchNorm folded into w2/b2 by the packer) */
typedef struct {
  const float* ln_g; /* [F] */
  const float* ln_b; /* [F] */
  const float* w1;   /* [2F,F]  linear1.weight */
  const float* b1;   /* [2F] */
  const float* dw_w; /* [K,F]   dw_conv_1d.weight, packed tap-major (K = 65) */
  const float* dw_b; /* [F] */
  const float* w2;   /* [2F,F]  linear2.weight * bn_scale[row] */
  const float* b2;   /* [2F]    (linear2.bias - running_mean) * bn_scale + BN.bias */
  const float* w3;   /* [F,2F]  linear3.1.weight */
  const float* b3;   /* [F] */
  const float* ls;   /* [F] */
  sepr_x3_w x3_1;    /* linear1 (LayerNorm folded) */
  sepr_x3_w x3_2;    /* linear2 (BatchNorm folded) */
  sepr_x3_w x3_3;    /* linear3.1 */
  /* optional fused head / tail forms (bf16x3, F = 128; pack.py::pack_cla_fused); all three or none */
  const void* fused_w1p;   /* per 32 output channels: [v0 v1 g0 g1][F/32][plane][64][8] bf16 (gamma folded)
                              + 4 KB fp32 constants [4 tiles][16] biases */
  const void* fused_w2p;   /* per 64 hidden channels: [4 tiles][F/32][plane][64][8] bf16 + 4 KB [4][16] biases */
  const void* fused_w3p;   /* [2F/32][F/16][plane][64][8] bf16, permuted k-slot order */
} sepr_cla_w;

/* MultiHeadAttention, modules/network.py:69-124 */
typedef struct {
  const float* ln_g; /* [F] */
  const float* ln_b; /* [F] */
  const float* wqkv; /* [3F,F]  linear_q / linear_k / linear_v weights stacked */
  const float* bqkv; /* [3F] */
  const float* wo;   /* [F,F]   linear_out.weight */
  const float* bo;   /* [F] */
  const float* ls;   /* [F] */
  sepr_x3_w x3_qkv;  /* stacked q/k/v (LayerNorm folded) */
  sepr_x3_w x3_out;  /* linear_out */
  /* optional fully fused speaker-attention form (bf16x3, F = 128, 16-channel heads, S = 2;
     pack.py::pack_spk_fused); both or none; only read by sepr_spkattn_fwd */
  const void* fused_qkv_p; /* per head pair: [q0 q1 k0 k1 v0 v1][F/32][plane][64][8] bf16 (gamma folded)
                              + 4 KB fp32 constants [6 tiles][16] biases */
  const void* fused_out_p; /* [F/32][F/16][plane][64][8] bf16, permuted k-slot order */
} sepr_mha_w;

/* EGA, modules/network.py:126-155 + the shared relative-position table, modules/module.py:42-57 */
typedef struct {
  sepr_mha_w attn;
  const float* gate_ln_g; /* [F]   block.linear.0 */
  const float* gate_ln_b; /* [F] */
  const float* gate_w;    /* [F,F] block.linear.1 */
  const float* gate_b;    /* [F] */
  const float* pe_k;      /* [2*maxlen, F/H]  separator.pos_emb.pe_k.weight */
  int maxlen;
  sepr_x3_w x3_gate;      /* block.linear.1 (LayerNorm folded) */
  const void* fused_gate_p; /* optional (bf16x3, F = 128; pack.py::pack_gate_fused): per 64 output channels
                               [4 tiles][F/32][plane][64][8] bf16 (gamma folded) + 4 KB fp32 constants [4][16] biases */
  const void* pe_k_planes;  /* optional (bf16x3; pack.py, ABI 3.02): pe_k split ONCE into bf16 planes [2: hi, lo][2*maxlen][F/H] -
                               the attention kernel then stages its relative-position band without a per-tile VALU split */
  const void* fused_qkv_p;  /* optional (bf16x3, F = 128; pack.py::pack_gate_fused on self_attn's in-projection, ABI 4.11): the [3F, F]
                               q / k / v projection behind its LayerNorm in the gate's chunk form (6 chunks of 64 output channels) -
                               pooling + LayerNorm + projection then run as ONE launch (sepr_cla_fused.hip launch_ega_qkv) */
  const void* fused_out_p;  /* optional (bf16x3, F = 128, with fused_gate_p; pack.py::pack_outproj_fused, ABI 4.11): self_attn.linear_out as
                               fragments [F/16][F/32][plane][64][8] bf16 - the gate launch then computes its tile's rows of
                               LayerScale(linear_out(.)) itself and the separate output projection is gone */
} sepr_ega_w;

/* DownConvLayer, modules/module.py:63-78 (eval BN folded: y = gelu(conv_nobias * scale + shift)) */
typedef struct {
  const float* w;     /* [K,F] down_conv.weight packed tap-major (K = 5) */
  const float* scale; /* [F] */
  const float* shift; /* [F] */
} sepr_down_w;

/* SpkSplitStage, modules/module.py:110-125 */
typedef struct {
  const float* w1;   /* [4FS,F]   linear.0.weight */
  const float* b1;   /* [4FS] */
  const float* w2;   /* [FS,2FS]  linear.2.weight */
  const float* b2;   /* [FS] */
  const float* gn_g; /* [F]       norm.weight */
  const float* gn_b; /* [F] */
  sepr_x3_w x3_1;
  sepr_x3_w x3_2;
  /* optional one-kernel form of linear.0 + GLU + linear.2 (bf16x3, F = 128): the [rows, 2FS] gated tensor never reaches
   * HBM.  fused_w1p: per 32-channel hidden chunk the up-projection fragments + biases (gate rows pre-scaled by -log2 e),
   * fused_w2p: [FS/128][2FS/32][8][2][64][8] bf16 down-projection slices, one block of 128 output channels after the
   * other (pack.pack_glumlp_fused).  NULL = the two generic projections. */
  const void* fused_w1p;
  const void* fused_w2p;
} sepr_split_w;

/* Decoder-side fusion conv, modules/module.py:187,214 */
typedef struct {
  const float* w;    /* [F,2F]  simple_fusion.i.weight */
  const float* b;    /* [F] */
  sepr_x3_w x3;
} sepr_fuse_w;

/* OutputLayer + AudioDecoder, modules/module.py:237-283 */
typedef struct {
  const float* w1;   /* [4F,F]  end_conv1x1.0.weight */
  const float* b1;   /* [4F] */
  const float* w2;   /* [N,2F]  end_conv1x1.2.weight */
  const float* b2;   /* [N] */
  const float* wdec; /* [K,N]   ConvTranspose1d weight [N,1,K] packed tap-major */
  sepr_x3_w x3_1;
  sepr_x3_w x3_2;
  /* optional one-kernel form of end_conv1x1.0 + GLU + end_conv1x1.2 (bf16x3, F = 128, N % 128 == 0), as in sepr_split_w */
  const void* fused_w1p;
  const void* fused_w2p;
  /* optional (ABI 4.10), heads WITHOUT a mask only (masking = False, model.py:28): end_conv1x1.2 (2F -> N, bias) and the
   * ConvTranspose1d(N -> 1, K = 16, stride 4, no bias) have no nonlinearity between them, so they are ONE linear map:
   * fold_w2p = bf16 hi/lo k-slot fragments [2F/32][1][2][64][8] of W_fold[K, 2F] = wdec^T . W2, fold_b [K] = wdec^T . b2 (both formed in
   * fp64, pack.pack_glumlp_fold).  With fused_w1p set, sepr_outlayer_decoder_fwd(idx = NULL, enc = NULL) is then one launch: up-projection,
   * GLU, 16-row down-projection, overlap-add into wav; no [rows, N] tensor, no workspace.  NULL = the two-step form. */
  const void* fold_w2p;
  const float* fold_b;
} sepr_out_w;

/* ---- library ---------------------------------------------------------------------------------- */

int sepr_version(void);
const char* sepr_build_info(void);
/* A/B switches of the launch path (environment variables of the same names with the SEPR_ prefix).  They are read ONCE per process,
 * on first use - no entry point calls getenv on its launch path - and the host-side mirrors (sepreformer_amd/train_engine.py) ask the
 * library instead of parsing the environment themselves, so both sides always agree on a context layout.  sepr_knobs_reload() re-reads
 * the environment (tests that flip a switch inside one process call it between whole forward + backward runs, never inside one). */
enum { SEPR_KNOB_X3_WIDE = 0 /* 0 / 1 (default) / 2: sepr_gemm_x3.hip */, SEPR_KNOB_TRAIN_GCFN_PLANES /* default 1 */,
       SEPR_KNOB_TRAIN_ATTN_ONE /* default 1 */, SEPR_KNOB_TRAIN_CLA16 /* default 1 */,
       SEPR_KNOB_FOLD_HEAD /* default 1: main OutputLayer + AudioDecoder as one launch when sepr_out_w.fold_* are set */,
       SEPR_KNOB_TN16 /* default 1: weight-gradient contractions of two bf16 operands on the LDS-DMA + transposing-read kernel */, SEPR_KNOB_COUNT };
int sepr_knob(int id);
void sepr_knobs_reload(void);
/* text of the last HIP error seen by the calling thread ("" if none) */
const char* sepr_last_hip_error(void);

/* ops for sepr_workspace_bytes */
enum {
  SEPR_OP_ENCODER = 0, SEPR_OP_GCFN, SEPR_OP_CLA, SEPR_OP_EGA, SEPR_OP_SPKATTN,
  SEPR_OP_SPKSPLIT, SEPR_OP_OUTLAYER, SEPR_OP_PIT /* n = utterances, T = samples, S = speakers */, SEPR_OP_COUNT
};
/* Scratch bytes op needs for n sequences of T frames (Tp = pooled frames for EGA, source frames for
 * OUTLAYER; otherwise ignored), width F, encoder channels N, S speakers.  0 on bad arguments. */
size_t sepr_workspace_bytes(int op, int n, int T, int Tp, int F, int N, int S);

/* ---- hot path, in forward order ----------------------------------------------------------------- */

/* AudioEncoder.forward, modules/module.py:19-23: Conv1d(1->N, K=16, stride, no bias) + exact GELU.
 * wav [B,T] -> enc [B,L,N] (L = (T-K)/stride + 1); also the GroupNorm(1,N,eps) statistics of each
 * sample of enc (mean, rstd) -> gn_stats [B,2] for sepr_projector_fwd.  w_enc is [K,N] (packed). */
int sepr_encoder_fwd(const float* wav, int B, int T, const float* w_enc, int N, int K, int stride,
                     float gn_eps, float* enc, float* gn_stats, void* ws, size_t ws_bytes,
                     sepr_stream_t stream);

/* FeatureProjector.forward + Separator.pad_signal, modules/module.py:32-35,220-234:
 * GroupNorm(1,N) apply + 1x1 conv N->F (no bias); rows l >= L of out [B,Lp,F] are written as zeros. */
int sepr_projector_fwd(const float* enc, int B, int L, int Lp, int N, int F, const float* gn_stats,
                       const float* gn_g, const float* gn_b, const float* w, float* out,
                       sepr_stream_t stream);

/* GCFN.forward, modules/network.py:60-66.  x,y [n,T,F]; y may alias x (the fully fused form needs x != y and
 * is skipped when they alias). */
int sepr_gcfn_fwd(const float* x, float* y, int n, int T, int F, const sepr_gcfn_w* w, void* ws,
                  size_t ws_bytes, sepr_stream_t stream);

/* CLA.forward (eval), modules/network.py:174-187.  x,y [n,T,F]; y may alias x. */
int sepr_cla_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_cla_w* w, void* ws,
                 size_t ws_bytes, sepr_stream_t stream);

/* EGA.forward, modules/network.py:138-155 (with MultiHeadAttention.forward :90-124 and the relative
 * position bias of modules/module.py:52-57,196-198 computed from pe_k without materialising pos_k).
 * x,y [n,T,F], T = Tp * 2^k; y must NOT alias x. */
int sepr_ega_fwd(const float* x, float* y, int n, int T, int Tp, int F, int H, const sepr_ega_w* w,
                 void* ws, size_t ws_bytes, sepr_stream_t stream);

/* SpkAttention.forward up to (excluding) its feed_forward GCFN, modules/network.py:233-247:
 * attention across the S speakers of each frame + residual.  x,y [B*S,T,F] (row index b*S+s);
 * y may alias x.  Follow with sepr_gcfn_fwd for :249. */
int sepr_spkattn_fwd(const float* x, float* y, int nS, int S, int T, int F, int H,
                     const sepr_mha_w* w, void* ws, size_t ws_bytes, sepr_stream_t stream);

/* DownConvLayer.forward (eval), modules/module.py:72-78.  x [n,T,F] -> y [n,(T-1)/2+1,F]. */
int sepr_downconv_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_down_w* w,
                      sepr_stream_t stream);

/* SpkSplitStage.forward, modules/module.py:120-125.  x [B,T,F] -> y [B*S,T,F] incl. GroupNorm(1,F). */
int sepr_spksplit_fwd(const float* x, float* y, int B, int S, int T, int F, float gn_eps,
                      const sepr_split_w* w, void* ws, size_t ws_bytes, sepr_stream_t stream);

/* Statistics-threaded forms of the four residual blocks (ABI 4.10).  Every block of the residual stream starts with a LayerNorm over the F
 * channels of each frame (network.py:50,81,133,162) and ends by writing that stream; x_stats / y_stats [rows][2] = (mean, rstd) of the rows
 * of x / y in the library's LayerNorm arithmetic (eps 1e-5).  x_stats (or NULL): as returned through y_stats by the call that produced x -
 * the block's own statistics pass over x is skipped.  y_stats (or NULL): receives the statistics of y; they come out of the last
 * projection's tile tail where a workgroup tile holds whole rows (F = 256: the generic bf16x3 path of the Large variants), else from one
 * extra pass over y.  Results are identical to the plain entry points.  Everything else as in the function without the suffix. */
int sepr_gcfn_fwd_st(const float* x, const float* x_stats, float* y, float* y_stats, int n, int T, int F, const sepr_gcfn_w* w, void* ws,
                     size_t ws_bytes, sepr_stream_t stream);
int sepr_cla_fwd_st(const float* x, const float* x_stats, float* y, float* y_stats, int n, int T, int F, int K, const sepr_cla_w* w,
                    void* ws, size_t ws_bytes, sepr_stream_t stream);
int sepr_ega_fwd_st(const float* x, const float* x_stats, float* y, float* y_stats, int n, int T, int Tp, int F, int H,
                    const sepr_ega_w* w, void* ws, size_t ws_bytes, sepr_stream_t stream);
int sepr_spkattn_fwd_st(const float* x, const float* x_stats, float* y, float* y_stats, int nS, int S, int T, int F, int H,
                        const sepr_mha_w* w, void* ws, size_t ws_bytes, sepr_stream_t stream);

/* Decoder-side fusion, modules/module.py:212-214: nearest x2 upsample of lo [n,T/2,F], concat with
 * skip [n,T,F] along channels, Conv1d(2F->F,k=1) -> y [n,T,F]. */
int sepr_fuse_fwd(const float* lo, const float* skip, float* y, int n, int T, int F, const sepr_fuse_w* w,
                  sepr_stream_t stream);

/* OutputLayer.forward (+Masking when enc != NULL) and AudioDecoder.forward, modules/module.py:249-283,
 * modules/network.py:34-43, model.py:42-52.  x [nS,Tsrc,F]; frame l < L of the head reads source
 * frame idx[l] (idx == NULL: l itself, i.e. the crop of :250; otherwise the nearest-upsample table of
 * model.py:49).  enc [B,L,N] or NULL.  wav [S,B,Tout], Tout = (L-1)*stride + K. */
int sepr_outlayer_decoder_fwd(const float* x, int nS, int S, int Tsrc, int L, const int* idx,
                              const float* enc, int F, int N, int K, int stride, const sepr_out_w* w,
                              float* wav, void* ws, size_t ws_bytes, sepr_stream_t stream);

/* GroupNorm(1 group) statistics of x [n, count] -> stats [n,2] = (mean, rstd).  modules/module.py:28,117 */
int sepr_groupnorm_stats(const float* x, int n, long long count, float eps, float* stats, void* ws,
                         size_t ws_bytes, sepr_stream_t stream);

/* y[M,N] = x[M,K] . w[N,K]^T + bias : the f32-MFMA projection core on its own (tests, roofline bench) */
int sepr_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                    sepr_stream_t stream);
/* the same product on the bf16x3 core; wp = pack_x3(w) */
int sepr_linear_x3_fwd(const float* x, const void* wp, const float* bias, float* y, int M, int N, int K,
                       sepr_stream_t stream);

/* ---- criteria --------------------------------------------------------- */
/* Permutation-invariant SI-SNR of S <= 3 separated waveforms, utils/implements/criterions.py:
 *   PIT_SISNR_time.__call__ :191-217   loss[b]   = min over permutations of sum_s clamp(-SISNR(est_s, tgt_p(s)), clamp_min)
 *   PIT_SISNRi.__call__     :232-260   sisnri[b] = per-estimate SISNR(est_s, tgt_p(s)) - SISNR(mix, tgt_p(s)) of the
 *                                                  permutation maximising their sum
 * est, tgt [S,B,T] (the reference's lists of [B,T] stacked), mix [B,T] or NULL (then sisnri / sisnri_perm must be
 * NULL).  eps_loss = 1e-8 (:199), eps_i = the caller's eps (1e-15 in engine.py:131), clamp_min = -30 (:212).
 * loss [B]; loss_perm / sisnri_perm [B,S] = target index per estimate (NULL to skip); sisnri [B,S].  The batch
 * mean the reference returns (sum / num_utts) is a host-side mean of `loss`.  Scale-invariant form only
 * (scale_inv: true in every shipped config).  Workspace: sepr_workspace_bytes(SEPR_OP_PIT, B, T, 0, 0, 0, S). */
int sepr_pit_sisnr_fwd(const float* est, const float* tgt, const float* mix, int S, int B, int T,
                       double eps_loss, double eps_i, double clamp_min, float* loss, int* loss_perm,
                       float* sisnri, int* sisnri_perm, void* ws, size_t ws_bytes, sepr_stream_t stream);

/* PIT_SISNR_mag.__call__, utils/implements/criterions.py:148-176 with the conv-STFT of :43-113: per utterance
 *   loss[b] = min over permutations of sum_s -20 log10(eps + |M(c t_p(s))| / (|M(est_s) - M(c t_p(s))| + eps)),
 * M = sqrt(re^2 + im^2 + 1e-10) of the STFT of the zero-mean signal (frame_len-point DFT kernel `dft`, hop
 * frame_shift, zero padding to a multiple of the hop, no centre padding), c = max(<e~,t~> / (|t~|^2 + eps), 1e-2),
 * norms over bins and frames.  est, tgt [S,B,T]; dft [(frame_len + 2 rounded up to 4), frame_len] row-major = the
 * reference's STFTBase.K (real rows, then imaginary rows, then zero rows); loss [B]; perm [B,S] or NULL.
 * mel_opt = false only.  Workspace: sepr_pit_sisnr_mag_workspace(S, B, T, frame_len, frame_shift). */
size_t sepr_pit_sisnr_mag_workspace(int S, int B, int T, int frame_len, int frame_shift);
int sepr_pit_sisnr_mag_fwd(const float* est, const float* tgt, int S, int B, int T, const float* dft,
                           int frame_len, int frame_shift, double eps, float* loss, int* perm, void* ws,
                           size_t ws_bytes, sepr_stream_t stream);

/* ================================================================================================================= */
/* Training path: train-mode forward twins that keep what the backward needs, and the       */
/* backward of every block.  The reference trains through torch.autograd over the same modules (engine.py:50-83:     */
/* data_parallel forward, PIT losses, backward, clip_grad_norm_, AdamW); here each block has an explicit backward.   */
/*                                                                                                                   */
/* Conventions (in addition to the ones at the top of this header)                                                   */
/*   - a block's train_fwd writes its saved-for-backward tensors into a caller-provided context buffer `ctx` of       */
/*     sepr_train_ctx_bytes(op, ...) bytes; the matching _bwd reads the same buffer (the caller keeps it and the      */
/*     block's input x alive in between); `ws` is scratch of sepr_train_ws_bytes(op, ...) bytes;                      */
/*   - parameter gradients are ACCUMULATED (+=) into fp32 buffers laid out exactly like the reference parameters      */
/*     (e.g. depthwise.weight [C,1,K]); the caller zeroes them once per step; activation gradients are overwritten;   */
/*   - a projection travels as sepr_lin: exact-f32 form (w: fp32 [N][K] row-major) or packed-bf16 form (wp: pack_x3    */
/*     fragments, hi and lo planes), never both; b may be NULL.  `planes` selects the arithmetic of the packed form:   */
/*     0 or 3 = bf16x3 (hi.hi + hi.lo + lo.hi, ~2^-16), 1 = plain bf16 operands (one MFMA per product, fp32           */
/*     accumulate: the "bf16" training precision of BASELINE configs[4]; activations are rounded to bf16 on the fly,   */
/*     the lo plane is ignored).  The host packs, per weight version, the forward form and the                         */
/*     transposed form the input gradient needs, with LayerNorm / GroupNorm affines and LayerScale folded in           */
/*     (sepreformer_amd/train_pack.py); the raw parameters ride along for the gradient finishers;                      */
/*   - train-mode BatchNorm uses batch statistics over (sequences x frames) and updates running_mean / running_var     */
/*     in place (momentum 0.1, unbiased variance), like torch.nn.BatchNorm1d (network.py:167, module.py:69);          */
/*   - dropout (network.py:55,57,87,121,124,171): inverted dropout from a counter-based generator keyed by             */
/*     (seed, element index); p_drop = 0 disables it exactly; the backward regenerates the masks from the same seed.  */
/*     (The fused GCFN pair - sepr_gcfn_train_fwd / sepr_gcfn_bwd with fused_w1p set - draws its two sites from a      */
/*     cheaper 16-bit-per-element generator, csrc/sepr_train.h sepr_drop_word: p is quantised to p_eff = round(p *     */
/*     65536) / 65536 and the kept values are scaled by 1 / (1 - p_eff).)                                             */
/* ================================================================================================================= */
typedef struct { const float* w; const void* wp; const float* b; int planes; } sepr_lin;
typedef unsigned long long sepr_u64;

/* GCFN, modules/network.py:46-66 */
typedef struct {
  sepr_lin up;      /* [6F,F]  net1.1.weight * ln_gamma, net1.1.bias + W . ln_beta */
  sepr_lin up_t;    /* [F,6F]  (net1.1.weight * ln_gamma)^T */
  sepr_lin down;    /* [F,3F]  net2.2 */
  sepr_lin down_t;  /* [3F,F]  (layer_scale * net2.2.weight)^T */
  const float* dw_w; /* [3,6F] tap-major */
  const float* dw_b; /* [6F] */
  const float* ls;   /* [F] */
  const float* w1; const float* ln_g; const float* ln_b; const float* w2; const float* b2;   /* raw parameters */
  /* Optional fused form (F in {64, 128}; train_pack.py): the forward is then ONE launch of the fused inference kernel
   * (sepr_gcfn_fwd's, with both dropout sites and a LayerNorm-statistics side output) that keeps only the per-row
   * statistics, and the backward recomputes the hidden tensor inside its middle kernel (csrc/sepr_gcfn_bwd_fused.hip).
   * fused_w1p / fused_w2p exactly as in sepr_gcfn_w.  NULL fused_w1p = the unfused path. */
  const void* fused_w1p; const void* fused_w2p;
  const sepr_u64* seed_salt;   /* optional device word XOR-ed into `seed` by every dropout kernel of the call (hipGraph replay) */
} sepr_gcfn_tw;
typedef struct { float* ln_g; float* ln_b; float* w1; float* b1; float* dw_w; float* dw_b; float* w2; float* b2; float* ls; } sepr_gcfn_grad;

/* CLA, modules/network.py:159-187 (BatchNorm NOT folded: batch statistics) */
typedef struct {
  sepr_lin l1, l1_t;   /* [2F,F] linear1 * ln_gamma (+ beta in the bias); [F,2F] transpose */
  const float* dw_w;   /* [K,F] tap-major */
  const float* dw_wf;  /* [K,F] tap-major, taps reversed (input gradient of the 'same' conv) */
  const float* dw_b;   /* [F] */
  const float* zeros;  /* [>= 2F] zeros */
  sepr_lin l2, l2_t;   /* [2F,F] linear2; [F,2F] linear2^T */
  const float* bn_g; const float* bn_b; float* bn_rm; float* bn_rv;   /* BN affine; running stats (updated in place) */
  sepr_lin l3, l3_t;   /* [F,2F] linear3.1; [2F,F] (layer_scale * linear3.1.weight)^T */
  const float* ls;
  const float* w1; const float* ln_g; const float* ln_b; const float* w3; const float* b3;   /* raw parameters */
  const sepr_u64* seed_salt;   /* as in sepr_gcfn_tw */
} sepr_cla_tw;
typedef struct { float* ln_g; float* ln_b; float* w1; float* b1; float* dw_w; float* dw_b; float* w2; float* b2; float* bn_g; float* bn_b;
                 float* w3; float* b3; float* ls; } sepr_cla_grad;

/* MultiHeadAttention, modules/network.py:69-124 */
typedef struct {
  sepr_lin qkv, qkv_t;  /* [3F,F] stacked q/k/v * ln_gamma; [F,3F] transpose */
  sepr_lin out, out_t;  /* [F,F] linear_out; (layer_scale * linear_out.weight)^T */
  const float* ls;
  const float* wqkv; const float* ln_g; const float* ln_b; const float* wo; const float* bo;   /* raw (wqkv stacked [3F,F]) */
  const sepr_u64* seed_salt;   /* as in the GCFN struct; for EGA the attention struct carries it for the whole block */
} sepr_mha_tw;
typedef struct { float* ln_g; float* ln_b; float* wq; float* bq; float* wk; float* bk; float* wv; float* bv; float* wo; float* bo; float* ls; } sepr_mha_grad;

/* EGA, modules/network.py:126-155 */
typedef struct {
  sepr_mha_tw attn;
  sepr_lin gate, gate_t;  /* [F,F] block.linear.1 * gamma; transpose */
  const float* gate_w; const float* gate_ln_g; const float* gate_ln_b;   /* raw */
  const float* pe_k; int maxlen;
} sepr_ega_tw;
typedef struct { sepr_mha_grad attn; float* gate_ln_g; float* gate_ln_b; float* gate_w; float* gate_b; float* pe_k; } sepr_ega_grad;

/* DownConvLayer, modules/module.py:63-78 */
typedef struct { const float* w /* [K,F] tap-major */; const float* b; const float* bn_g; const float* bn_b; float* bn_rm; float* bn_rv; } sepr_down_tw;
typedef struct { float* w /* [F,1,K] */; float* b; float* bn_g; float* bn_b; } sepr_down_grad;

/* SpkSplitStage, modules/module.py:110-125 */
typedef struct { sepr_lin l1, l1_t, l2, l2_t; const float* gn_g; const float* gn_b; } sepr_split_tw;
typedef struct { float* w1; float* b1; float* w2; float* b2; float* gn_g; float* gn_b; } sepr_split_grad;

/* fusion conv, modules/module.py:187,212-214 */
typedef struct { sepr_lin l, l_t; } sepr_fuse_tw;
typedef struct { float* w; float* b; } sepr_fuse_grad;

/* OutputLayer + AudioDecoder, modules/module.py:237-283 */
typedef struct { sepr_lin l1, l1_t, l2, l2_t; const float* wdec /* [K,N] tap-major */; } sepr_out_tw;
typedef struct { float* w1; float* b1; float* w2; float* b2; float* wdec /* [N,1,K] */; } sepr_out_grad;

/* AudioEncoder + FeatureProjector, modules/module.py:12-35.  The projector runs on the exact-f32 core with the GroupNorm
 * affine in its prologue (as in inference: padded frames must come out exactly zero), only its input gradient uses a
 * folded, transposed form. */
typedef struct {
  const float* w_enc;      /* [K,N] tap-major */
  const float* proj_w;     /* [F,N] feature_projector.conv1d.weight */
  const float* gn_g; const float* gn_b;   /* [N] feature_projector.norm */
  sepr_lin proj_t;         /* [N,F] (conv1d.weight * gn_gamma)^T */
  const float* ones;       /* [>= N] ones */
} sepr_front_tw;
typedef struct { float* w_enc /* [N,1,K] */; float* gn_g; float* gn_b; float* proj_w; } sepr_front_grad;

enum { SEPR_TOP_GCFN = 0, SEPR_TOP_CLA, SEPR_TOP_EGA, SEPR_TOP_SPKATTN, SEPR_TOP_DOWN, SEPR_TOP_SPLIT, SEPR_TOP_FUSE, SEPR_TOP_OUT,
       SEPR_TOP_FRONT, SEPR_TOP_GCFN_FUSED /* sizes of the fused GCFN pair (sepr_gcfn_tw.fused_w1p != NULL) */,
       SEPR_TOP_EGA_X3 /* sizes of EGA when its attention runs on the bf16 MFMA (packed-bf16 precisions, dk 16 / 32) */,
       SEPR_TOP_GCFN_FUSED16 /* the fused GCFN pair in the plain-bf16 precision (sepr_lin.planes == 1): its context also keeps the
                                normalised rows as bf16 [n*T][F] for the plane-staged backward (ABI 3.02) */, SEPR_TOP_COUNT };
/* n sequences of T frames (Tp: pooled frames for EGA, source frames for OUT, padded frames Lp for FRONT), width F, encoder
 * channels N, S speakers, K = depthwise taps where the op has them (CLA 65, DOWN 5; else ignored). */
size_t sepr_train_ctx_bytes(int op, int n, int T, int Tp, int F, int N, int S, int H);
size_t sepr_train_ws_bytes(int op, int n, int T, int Tp, int F, int N, int S, int H, int K);

int sepr_gcfn_train_fwd(const float* x, float* y, int n, int T, int F, const sepr_gcfn_tw* w, void* ctx, size_t ctx_bytes, void* ws,
                        size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream);
int sepr_gcfn_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, const sepr_gcfn_tw* w, const sepr_gcfn_grad* g,
                  const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream);

int sepr_cla_train_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_cla_tw* w, void* ctx, size_t ctx_bytes, void* ws,
                       size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream);
int sepr_cla_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, int K, const sepr_cla_tw* w, const sepr_cla_grad* g,
                 const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream);

int sepr_ega_train_fwd(const float* x, float* y, int n, int T, int Tp, int F, int H, const sepr_ega_tw* w, void* ctx, size_t ctx_bytes,
                       void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream);
int sepr_ega_bwd(const float* x, const float* dy, float* dx, int n, int T, int Tp, int F, int H, const sepr_ega_tw* w,
                 const sepr_ega_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed,
                 sepr_stream_t stream);

int sepr_spkattn_train_fwd(const float* x, float* y, int nS, int S, int T, int F, int H, const sepr_mha_tw* w, void* ctx,
                           size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream);
int sepr_spkattn_bwd(const float* x, const float* dy, float* dx, int nS, int S, int T, int F, int H, const sepr_mha_tw* w,
                     const sepr_mha_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed,
                     sepr_stream_t stream);

int sepr_downconv_train_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_down_tw* w, void* ctx, size_t ctx_bytes,
                            void* ws, size_t ws_bytes, sepr_stream_t stream);
int sepr_downconv_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, int K, const sepr_down_tw* w,
                      const sepr_down_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, sepr_stream_t stream);

int sepr_spksplit_train_fwd(const float* x, float* y, int B, int S, int T, int F, float gn_eps, const sepr_split_tw* w, void* ctx,
                            size_t ctx_bytes, void* ws, size_t ws_bytes, sepr_stream_t stream);
/* dx_accumulate != 0: dx += (the skip tensors are also read by the DownConv of their stage) */
int sepr_spksplit_bwd(const float* x, const float* dy, float* dx, int dx_accumulate, int B, int S, int T, int F, const sepr_split_tw* w,
                      const sepr_split_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, sepr_stream_t stream);

/* forward is sepr_fuse_fwd with w->l; backward: dlo [n,T/2,F], dskip [n,T,F] */
int sepr_fuse_bwd(const float* lo, const float* skip, const float* dy, float* dlo, float* dskip, int n, int T, int F,
                  const sepr_fuse_tw* w, const sepr_fuse_grad* g, void* ws, size_t ws_bytes, sepr_stream_t stream);

int sepr_outlayer_decoder_train_fwd(const float* x, int nS, int S, int Tsrc, int L, const int* idx, const float* enc, int F, int N,
                                    int K, int stride, const sepr_out_tw* w, float* wav, void* ctx, size_t ctx_bytes, void* ws,
                                    size_t ws_bytes, sepr_stream_t stream);
/* dwav [S,B,Tout] -> dx [nS,Tsrc,F] (dx_accumulate: +=; rows of the main head beyond L get zero), denc [B,L,N] += for the
 * masked (auxiliary) heads; idx_start [Tsrc+1]: first output frame mapped to each source frame (inverse of idx) */
int sepr_outlayer_decoder_bwd(const float* x, const float* dwav, float* dx, int dx_accumulate, float* denc, int nS, int S, int Tsrc,
                              int L, const int* idx, const int* idx_start, const float* enc, int F, int N, int K, int stride,
                              const sepr_out_tw* w, const sepr_out_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes,
                              sepr_stream_t stream);

/* encoder + projector: wav [B,T] -> enc [B,L,N] (kept by the caller: the auxiliary heads read it), out [B,Lp,F] */
int sepr_front_train_fwd(const float* wav, int B, int T, int N, int K, int stride, int F, int Lp, float gn_eps, const sepr_front_tw* w,
                         float* enc, float* out, void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, sepr_stream_t stream);
/* dout [B,Lp,F]; denc_aux [B,L,N] or NULL = gradient the auxiliary heads sent into enc (overwritten: used as scratch) */
int sepr_front_bwd(const float* wav, const float* enc, const float* dout, float* denc_aux, int B, int T, int N, int K, int stride, int F,
                   int Lp, const sepr_front_tw* w, const sepr_front_grad* g, const void* ctx, size_t ctx_bytes, void* ws,
                   size_t ws_bytes, sepr_stream_t stream);

/* G[N][K] (+)= sum_m A[m][n] B[m][k]: the weight-gradient contraction on its own (tests, roofline bench).
 * x3: 0 = exact f32 MFMA, 1 = bf16x3, 2 = plain bf16 operands. */
size_t sepr_linear_wgrad_workspace(int M, int N, int K);
int sepr_linear_wgrad(const float* A, const float* B, float* G, float* colsum, int M, int N, int K, int accumulate, int x3, void* ws,
                      size_t ws_bytes, sepr_stream_t stream);
/* the same for two bf16 operands A16 [M][lda], B16 [M][ldb] (leading dimensions in elements), as the plain-bf16 training precision stores the
 * large intermediates of the GCFN / CLA backward (ABI 4.12); one bf16 MFMA per product, fp32 accumulate */
int sepr_linear_wgrad_bf16(const void* A16, int lda, const void* B16, int ldb, float* G, float* colsum, int M, int N, int K, int accumulate,
                           void* ws, size_t ws_bytes, sepr_stream_t stream);
/* the same with the normalisation prologue of a projection behind a LayerNorm: B'[m][k] = (B[m][k] - stats[2m]) * stats[2m+1] */
int sepr_linear_wgrad_norm(const float* A, const float* B, const float* stats, float* G, float* colsum, int M, int N, int K,
                           int accumulate, int x3, void* ws, size_t ws_bytes, sepr_stream_t stream);

/* ---- per-step weight re-pack on the device (ABI 4.00) -----------------------------------------------------------------
 * What the reference's modules do implicitly - read the CURRENT value of every nn.Parameter on each forward (network.py:60-66 etc.) - costs
 * this path a re-layout of every projection after each optimizer step (sepreformer_amd/train_pack.py).  One call handles a STACK of G
 * same-shaped projections and reads the parameters where they live:
 *   src    device table of G * panels pointers; block g's [SN][SK] fp32 source matrix is the row-wise concatenation of its `panels`
 *          tensors (panels = 3: linear_q / linear_k / linear_v stacked, network.py:76-78), each [SN / panels][SK] row-major
 *   scale  device table of G pointers (or NULL with scale_kind 0): scale_kind 1 = per source COLUMN [SK] (LayerNorm / GroupNorm gamma
 *          folded into the weight), 2 = per source ROW [SN] (LayerScale folded in)
 *   transpose 0: out[n][k] = src[n][k] * s;  1: out[n][k] = src[k][n] * s  (the input-gradient form), N x K = SK x SN
 *   planes 1: out = G x pack_x3 fragments (bf16 hi / lo planes, [N/16][K/32][2][64][8], as sepr_x3_w.wp); 0: G x fp32 [N][K]
 * N % 16 == 0 and K % 32 == 0.  Bit-identical to the torch formulation fl32((double)w * (double)s) -> bf16 split. */
int sepr_train_pack_lin(const void* const* src, const void* const* scale, int G, int SN, int SK, int panels, int scale_kind, int transpose,
                        int planes, void* out, sepr_stream_t stream);
/* out[g][n] = (float)((double)bias[g][n] + sum_k (double)W[g][n][k] (double)beta[g][k]): the bias with the LayerNorm beta folded in.
 * w / bias: tables of G * panels pointers ([N / panels][K] and [N / panels] each), beta: G pointers [K].  bias NULL: 0; w and beta NULL: a
 * plain gather of the biases into one [G][N] buffer. */
int sepr_train_fold_bias(const void* const* w, const void* const* bias, const void* const* beta, int G, int N, int K, int panels, float* out,
                         sepr_stream_t stream);

/* The fused GCFN kernel's weight forms (sepr_gcfn_tw.fused_w1p / fused_w2p, i.e. sepr_gcfn_w's) for a stack of G blocks in one launch:
 * seven device tables of G pointers to the blocks' parameters (net1.1.weight [6F,F], net1.1.bias, net1.0.weight / bias [F], net2.2.weight
 * [F,3F], depthwise.weight [6F,1,3], depthwise.bias [6F]).  w1p: G x (3F/32) x (4 * (F/32) * 2 KiB + 4 KiB) bytes, w2p: G x (3F/32) x
 * (F/16) x 2 KiB bytes; layouts as documented on sepr_gcfn_w.  F in {64, 128}. */
int sepr_train_pack_gcfn_fused(const void* const* w1, const void* const* b1, const void* const* ln_g, const void* const* ln_b,
                               const void* const* w2, const void* const* dw_w, const void* const* dw_b, int G, int F, void* w1p, void* w2p,
                               sepr_stream_t stream);

/* ---- deferred gradient finishers (ABI 4.00) ---------------------------------------------------------------------------
 * Every sepr_<block>_bwd ends the gradient of a projection behind a LayerNorm / in front of a LayerScale with a weight-sized "finisher"
 * launch (csrc/sepr_train.h); a step has ~320 of them at 5-7 us each - launch latency, not work - and nothing in the backward reads
 * their outputs.  Between sepr_train_defer_begin and sepr_train_defer_flush ON THE SAME HOST THREAD the finishers are queued and the
 * flush runs them as a few batched launches on `stream` (bit-identical arithmetic).  `arena`: device scratch that receives the reduced
 * contractions until the flush - the sum over the deferred projections of (N K + N) floats, i.e. about one model's worth of parameters
 * (Base: 64 MB is plenty); a full arena simply falls back to immediate launches.  flush(close = 0) keeps the window open (a caller that
 * needs the gradients of the blocks processed so far - an early all-reduce bucket - flushes there); flush(close = 1) ends it.  The window
 * is per host thread (thread-local state); without one the entry points behave exactly as before. */
int sepr_train_defer_begin(void* arena, size_t arena_bytes);
int sepr_train_defer_flush(int close, sepr_stream_t stream);
/* Riding reductions (ABI 4.12).  Every weight-gradient contraction is followed by the fixed-order reduction of its row-slice partial tiles:
 * ~300 launches of 5-7 us per step, launch latency again.  With a double buffer registered on an open window (parts: 256-byte aligned device
 * scratch, 2 x 40 MB covers every contraction of the shipped configurations; NULL unregisters), a contraction whose outputs live in the
 * window's arena writes its partial tiles into one half of it and its reduction runs in blocks appended to the NEXT contraction's launch on
 * the same stream (or on its own at the flush) - same arithmetic, bit-identical gradients, ~290 launches fewer.  A contraction that does not
 * fit a half keeps the immediate form.  The registration ends with the window (flush(close = 1)) or the next sepr_train_defer_begin. */
int sepr_train_defer_parts(void* parts, size_t parts_bytes);

/* ---- weight-gradient side stream (ABI 4.10) ----------------------------------------------------------------------------------
 * Nothing inside a backward walk reads what the weight-gradient contractions write (parameter gradients, or the arena slots of deferred
 * finishers), and they are HBM-bound while the input-gradient chain beside them is VALU- / MFMA-bound.  sepr_train_wgrad_stream(side)
 * registers a second stream on the calling host thread: every contraction (+ its split-M reduction, + finishers that are not deferred) issued
 * by the sepr_*_bwd entry points of that thread is then launched on `side`, ordered behind everything the caller's stream has been given so
 * far (one event per launch), and overlaps the rest of the walk.  The caller owns two obligations:
 *   - workspace re-use: a workspace handed to a *_bwd call is read by side-stream kernels after the call returns.  Alternate two
 *     workspaces and bracket them: sepr_train_wgrad_mark(slot) after the call that used workspace `slot`, sepr_train_wgrad_wait(slot, stream)
 *     before the next call that re-uses it (and keep the call's input tensors alive as long);
 *   - joins: sepr_train_defer_flush and sepr_train_wgrad_join(stream) make `stream` wait for everything issued on the side stream - before
 *     gradients are read (all-reduce, optimizer), before the window closes, before the end of a hipGraph capture.
 * sepr_train_wgrad_stream(NULL) unregisters (after a join).  Gradients are bit-identical with and without a side stream. */
int sepr_train_wgrad_stream(sepr_stream_t side);
int sepr_train_wgrad_join(sepr_stream_t stream);
int sepr_train_wgrad_mark(int slot);
int sepr_train_wgrad_wait(int slot, sepr_stream_t stream);

/* PIT_SISNR_time backward (criterions.py:191-217): d(sum_b loss[b] * gl[b]) / d est.  est, tgt, dest [S,B,T]; perm from the forward. */
int sepr_pit_sisnr_bwd(const float* est, const float* tgt, const int* perm, const float* gl, int S, int B, int T, double eps,
                       double clamp_min, float* dest, void* ws, size_t ws_bytes, sepr_stream_t stream);
/* PIT_SISNR_mag backward (criterions.py:148-176).  dft_t [frame_len][ldd]: the transpose of the first frame_len + 2 rows of
 * `dft`, zero padded to ldd = frame_len + 2 rounded up to a multiple of 32 columns. */
int sepr_pit_sisnr_mag_bwd(const float* est, const float* tgt, const int* perm, const float* gl, int S, int B, int T, const float* dft,
                           const float* dft_t, int frame_len, int frame_shift, double eps, float* dest, void* ws, size_t ws_bytes,
                           sepr_stream_t stream);
size_t sepr_pit_sisnr_mag_bwd_workspace(int S, int B, int T, int frame_len, int frame_shift);

/* ---- optimizer step of the reference loop (ABI 3.03) ------------------------------------------------
 * engine.py:76-77: torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.AdamW.step() as three launches over the whole
 * model.  The gradients live in ONE flat fp32 buffer (the training path's gradient buffer: 64-element aligned slices, zero padding);
 * the moments in two flat fp32 buffers of the caller.  All table pointers are DEVICE pointers.
 *   params[i]     parameter tensor i (fp32, contiguous, 16-byte aligned)
 *   grad_off[i]   element offset of its gradient in `grads`;  state_off[i]: of its moments in exp_avg / exp_avg_sq (multiples of 4)
 *   numel[i]      elements of tensor i
 *   blocks[2 b], blocks[2 b + 1] = (tensor, first element) of update block b; a block covers sepr_adamw_block_elems() elements
 * step: device double, incremented by the call; lr: device float (a scheduler refreshes it without re-capturing a graph);
 * max_norm <= 0: no clipping (no norm pass).  scal (device, 8 floats) receives [0] the total gradient norm BEFORE clipping,
 * [1] the clip coefficient min(1, max_norm / (norm + 1e-6)), [2..4] the step's derived scalars.  The gradients are NOT scaled in
 * place: the coefficient is applied inside the update.  Arithmetic: torch's AdamW (decoupled weight decay), fp32 per element.
 * A non-finite total norm yields a NaN coefficient that reaches every element (clip_grad_norm_'s torch.clamp semantics).  The norm
 * pass reads grads[0 .. grads_numel): the 64-element alignment padding between slices must be zero (the training path's buffer is
 * allocated zeroed and only slices are ever written). */
typedef struct {
  float* const* params;
  const long long* grad_off;
  const long long* state_off;
  const int* numel;
  const int* blocks;
  int ntensors, nblocks;
} sepr_adamw_tables;
int sepr_adamw_block_elems(void);
size_t sepr_adamw_workspace(void);
int sepr_adamw_step(const sepr_adamw_tables* t, const float* grads, long long grads_numel, float* exp_avg, float* exp_avg_sq,
                    double* step, const float* lr, double beta1, double beta2, double eps, double weight_decay, double max_norm,
                    float* scal, void* ws, size_t ws_bytes, sepr_stream_t stream);

/* ---- opt-in kernel timer (bench.py roofline) ------------------------------------------------- */
/* Sites a projection launch can be attributed to. */
enum {
  SEPR_SITE_NONE = 0, SEPR_SITE_GCFN_UP, SEPR_SITE_GCFN_DOWN, SEPR_SITE_CLA, SEPR_SITE_ATTN_PROJ,
  SEPR_SITE_EGA_GATE, SEPR_SITE_SPLIT, SEPR_SITE_FUSE, SEPR_SITE_OUT, SEPR_SITE_PROJECTOR,
  SEPR_SITE_LINEAR, SEPR_SITE_WGRAD /* gemm_tn (training) */, SEPR_SITE_GCFN_BWD /* fused GCFN backward middle */,
  SEPR_SITE_COUNT
};
/* Start bracketing every launch of `site` with hipEvents on its own stream (up to max_launches). */
int sepr_prof_start(int site, int max_launches);
/* Synchronise the recorded events; returns launches, summed kernel ms and summed algorithmic FLOPs. */
int sepr_prof_stop(long long* launches, double* total_ms, double* flops);
/* Algorithmic HBM bytes of the launches of the session sepr_prof_stop last closed (sites that report them: the weight-gradient
 * contraction and the fused GCFN backward; 0 otherwise). */
double sepr_prof_last_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* SEPR_H_ */
