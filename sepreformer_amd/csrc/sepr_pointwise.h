// Host launchers of the HBM-bound kernels (definitions in sepr_pointwise.hip / sepr_attention.hip).
#pragma once
#include "sepr_common.h"

namespace sepr {

// LayerNorm statistics of X [M,F] rows (eps inside rstd) -> stats [M,2] = (mean, rstd)
int launch_rowstats(const float* X, float* stats, long long M, int F, float eps, hipStream_t s);

// (sum, sum of squares) partials -> (mean, rstd) per sequence.  part [n][nchunk][2] doubles.
int launch_gn_partial(const float* X, double* part, int n, long long count, int nchunk, hipStream_t s);
int launch_gn_finalize(const double* part, int n, int nchunk, long long count, float eps, float* stats, hipStream_t s);
int gn_chunks(long long count);
// y = (x - mean[seq]) * rstd[seq] * g[f] + b[f], in place, X [n, T*F]
int launch_gn_apply(float* X, const float* stats, const float* g, const float* b, int n, int T, int F, hipStream_t s);

// mean over fac consecutive frames: X [n, Tp*fac, F] -> Y [n, Tp, F]
int launch_pool(const float* X, float* Y, int n, int Tp, int fac, int F, hipStream_t s);
// ... fused with the LayerNorm statistics [mean, rstd] of the pooled rows (bit-identical to launch_pool + launch_rowstats)
int launch_pool_stats(const float* X, float* Y, float* stats, int n, int Tp, int fac, int F, float eps, hipStream_t s);

// GCFN middle: depthwise k=3 conv along frames + GLU.  H [n,T,6F] -> G [n,T,3F]
int launch_dwglu(const float* H, float* G, int n, int T, int F, const float* w, const float* b, hipStream_t s);

// CLA middle: depthwise 'same' conv (K odd, K <= 65) along frames.  U [n,T,F] -> C [n,T,F]
int launch_dwconv_same(const float* U, float* C, int n, int T, int F, int K, const float* w, const float* b, hipStream_t s);
int launch_dwconv_same_glu_bwd(const float* dc, const float* a, float* da, int n, int T, int F, int K, const float* w, const float* zero_bias,
                               hipStream_t s, int out16 = 0 /* da as bf16 */);
int launch_dwconv_same16(const float* U, float* C, int n, int T, int F, int K, const float* w, const float* b, hipStream_t s);   // C as bf16

// DownConv: depthwise K=5 stride 2 + folded BN + GELU.  X [n,T,F] -> Y [n,To,F]
int launch_downconv(const float* X, float* Y, int n, int T, int To, int F, int K, const float* w,
                    const float* scale, const float* shift, hipStream_t s);

// attention across the S speakers of each frame.  QKV [B*S,T,3F] -> O [B*S,T,F]
int launch_spkmix(const float* QKV, float* O, int B, int S, int T, int F, int H, hipStream_t s);

// encoder: Conv1d(1->N,K,stride)+GELU, wav [B,T] -> E [B,L,N]; GroupNorm partials -> part [B][ntile][2]
int encoder_tiles(int L);
int launch_encoder(const float* wav, int B, int T, int L, const float* w, int N, int K, int stride,
                   float* E, double* part, hipStream_t s);

// decoder: ConvTranspose1d(N->1,K,stride) of O2 [nS,L,N] -> wav [S,B,Tout] (overlap-add in LDS)
int launch_decoder(const float* O2, int nS, int S, int L, int N, int K, int stride, const float* wdec, float* wav,
                   int Tout, const int* idx, int Tsrc, const float* enc, hipStream_t s);

// EGA attention with relative-position bias.  QKV [n,Tp,3F] -> O [n,Tp,F]
// pe_planes (optional, x3 only): pe_k pre-split into bf16 planes [2: hi, lo][2*maxlen][F/H] (sepr_ega_w.pe_k_planes)
int launch_relattn(const float* QKV, float* O, int n, int Tp, int F, int H, const float* pe_k, int maxlen, int x3,
                   hipStream_t s, const void* pe_planes = nullptr);

}  // namespace sepr
