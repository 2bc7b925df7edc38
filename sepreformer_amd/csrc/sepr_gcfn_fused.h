// Argument block and host launchers of the fused GCFN kernel (sepr_gcfn_fused.hip), shared with the entry points that
// drive it (sepr_api.hip: sepr_gcfn_fwd, SpkSplit / OutputLayer GLU-MLP; sepr_train_api.hip: the train forward).
#pragma once
#include "sepr_common.h"

namespace sepr {

struct GcfnFusedArgs {
  const float* x;     // [M, F]
  float* y;           // [M, F]
  int M, T;           // rows, frames per sequence
  const void* w1p;    // per chunk: [4 tiles: v0 v1 g0 g1][KS][plane][64][8] bf16 (LayerNorm gamma folded), then 4 KB of
                      // constants [2 tile pairs][10: b1v b1g wv0 wv1 wv2 wg0 wg1 wg2 cbv cbg][16 channels] fp32; the gate's
                      // conv taps and bias (wg*, cbg) are multiplied by -log2(e): see glu_prescaled
  const void* w2p;    // [NCH][F/16][plane][64][8] bf16, k-slot order (g,e) -> e<4 ? 4g+e : 16+4g+e-4; fragment row 4q+r of
                      // tile ft is output channel 32*(ft/2) + 8q + 4*(ft%2) + r (a lane's accumulators of a tile pair are 8
                      // consecutive channels: the same 32 bytes of a frame row it loaded)
  const float* b2;    // [F]
  const float* ls;    // [F]
  float eps;
  int stagger;        // s_sleep units (64 clocks each) by which every other co-resident workgroup starts late; 0 = off
  // MODE 1 (plain GLU-MLP: y = W2 . GLU(W1 x + b1) + b2, no LayerNorm / conv / residual / LayerScale - SpkSplit, OutputLayer):
  int nch;            // hidden chunks of 32 value + 32 gate channels (GCFN: 3F/32)
  int ldy, col_off;   // output row stride in floats and first output column (one launch writes F columns of a wider tensor)
  int in_rows, in_src;          // in_rows > 0: input row of frame m is (m / in_rows) * in_src + m % in_rows (crop of every sequence)
  int out_T, out_S, out_s;      // out_S > 0: output row of frame m is ((m / out_T) * out_S + out_s) * out_T + m % out_T (speaker split)
  // MODE 2 (launch_glumlp_fold: OutputLayer + folded AudioDecoder): sequences, tiles per sequence and wav samples per sequence
  int fold_nseq, fold_tps, fold_Tout, fold_N;   // fold_N: basis size N (FLOP accounting only)
  // TRAIN instantiation (sepr_gcfn_train_fwd, fused form): the LayerNorm statistics of every output frame go to `stats`
  // ([M][2] = mean, rstd: all the backward keeps of this block), and both dropout sites of network.py:55,57 are live when
  // drop_thr > 0 (sepr_train.h sepr_drop_word: site 0 = gated tensor [M][3F], site 1 = block output [M][F]; the keep scale
  // drop_scale = 1 / (1 - p_eff) of both sites is applied in the epilogue, where it costs nothing)
  int train;
  void* xhat16;       // TRAIN + planes == 1 only (optional): [M][F] bf16 = the normalised rows as the kernel's MFMAs consume them; the
                      // backward's middle kernel stages them by LDS-DMA and the first weight-gradient contraction reads them (round 4)
  int planes;         // TRAIN only: 1 = plain bf16 operands (precision "bf16": the hi planes alone, one MFMA per product); else bf16x3
  float* stats;
  unsigned drop_thr;
  float drop_scale;
  unsigned long long seed;
  const unsigned long long* salt;
};

int launch_gcfn_fused(const GcfnFusedArgs& a, int F, int site, hipStream_t stream);
int launch_glumlp_fused(const GcfnFusedArgs& a, int F, int site, hipStream_t stream);
int launch_glumlp_fold(const GcfnFusedArgs& a, int F, int site, hipStream_t stream);

}  // namespace sepr
