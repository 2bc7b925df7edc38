// Instantiations + host launcher of the bf16x3 projection core.
#include "sepr_gemm_x3w.h"
#include "sepr_pointwise.h"
#include <stdlib.h>

namespace sepr {

// workgroups per launch: SEPR_X3_GRID = "tiles" -> one tile per workgroup (no persistent walk), an integer k ->
// k workgroups per CU; default 2 per CU (what fits: 80 KB of LDS each)
static int x3_grid_cap() {
  static const int v = [] {
    const char* e = getenv("SEPR_X3_GRID");
    if (!e || !e[0]) return persistent_grid();
    if (e[0] == 't') return 0;
    const int k = atoi(e);
    return k > 0 ? persistent_grid() / 2 * k : persistent_grid();
  }();
  return v;
}

// SEPR_X3_WIDE: 0 = the 128 x 128 core only, 1 (default) = the 128 x 256 core (sepr_gemm_x3w.h) for launches with an even number of
// 128-column tiles and at least two wide tiles per CU, 2 = for every launch with an even number of column tiles (A/B)
// (latched once per process - include/sepr.h SEPR_KNOB_X3_WIDE; the parity tests that compare the two cores bit for bit inside one process
//  call sepr_knobs_reload() after changing the variable)
static int x3_wide_mode() { return knob(SEPR_KNOB_X3_WIDE); }

template <int PRO, int EPI, int TAG = 0>
static void launch_x3_inst(const GemmArgs& a, hipStream_t stream) {
  const int cap = x3_grid_cap();
  if constexpr (EPI != EPI_LNBWD) {
    const bool glu = (EPI == EPI_GLU) || (EPI == EPI_GLUSAVE) || (EPI == EPI_DWGLU);
    const int NB = glu ? (a.N / 2 + 63) / 64 : (a.N + GEMM_BN - 1) / GEMM_BN;
    const int wmode = x3_wide_mode();
    const int wtiles = gemm_tiles_wide(a, EPI);
    if (wmode > 0 && NB >= 2 && (NB % 2) == 0 && (wmode == 2 || wtiles >= persistent_grid())) {
      const int grid = (cap <= 0 || wtiles < cap) ? wtiles : cap;
      hipLaunchKernelGGL((gemm_x3w_kernel<PRO, EPI, TAG>), dim3(grid), dim3(GEMM_THREADS), 0, stream, a);
      // output-row statistics: in the kernel's tile tail when a wide tile holds whole rows, else by the row-statistics kernel
      if (a.stats_out && !(NB == 2 && EPI == EPI_RES)) (void)launch_rowstats(a.Y, a.stats_out, a.M, a.N, a.stats_eps, stream);
      return;
    }
  }
  const int tiles = gemm_tiles(a, EPI);
  const int grid = (cap <= 0 || tiles < cap) ? tiles : cap;
  hipLaunchKernelGGL((gemm_x3_kernel<PRO, EPI, TAG>), dim3(grid), dim3(GEMM_THREADS), 0, stream, a);
  if (a.stats_out) (void)launch_rowstats(a.Y, a.stats_out, a.M, a.N, a.stats_eps, stream);
}

int launch_gemm_x3(int pro, int epi, const GemmArgs& a, int site, hipStream_t stream) {
  if (a.M <= 0) return SEPR_OK;
  if (a.N <= 0 || a.K <= 0 || (a.K % X3_BKS) != 0 || (a.N % 16) != 0) return SEPR_EINVAL;
  const bool glu = (epi == EPI_GLU || epi == EPI_GLUSAVE || epi == EPI_DWGLU);
  if (epi == EPI_GLUSAVE && !a.Ysave) return SEPR_EINVAL;
  if (epi == EPI_RESDROP && a.drop_thr == 0u) return SEPR_EINVAL;
  if (glu && (((a.N / 2) % 16) != 0 || !a.bias)) return SEPR_EINVAL;
  if (!a.A || !a.Wp || !a.Y) return SEPR_EINVAL;
  if ((a.lda % 4) != 0 || (a.ldc % 4) != 0) return SEPR_EINVAL;
  if (a.stats_out && (a.ldc != a.N || a.N > 512)) return SEPR_EINVAL;   // row statistics: contiguous output rows of at most 512 columns
  if (pro == PRO_CAT2 && (!a.A2 || (a.ksplit % 32) != 0 || (a.lda2 % 4) != 0)) return SEPR_EINVAL;
  if (pro == PRO_NORM && !a.stats) return SEPR_EINVAL;
  if (epi == EPI_DWGLU && (!a.dw_w || !a.dw_b || a.T <= 0)) return SEPR_EINVAL;
  {
    const long long src_rows = a.rows_out > 0 ? ((long long)(a.M + a.rows_out - 1) / a.rows_out) * a.rows_src : a.M;
    if (src_rows * a.lda >= (1LL << 32)) return SEPR_EINVAL;
    if (pro == PRO_CAT2 && (long long)a.M * a.lda2 >= (1LL << 32)) return SEPR_EINVAL;
  }
  const int key = pro * 16 + epi;
  // every argument check comes BEFORE prof_begin: an early return inside an open profiling slot would leave a start event without
  // its end event (sepr_prof_stop would then read an unrecorded event)
  if (epi == EPI_LNBWD && (a.N > GEMM_BN || !a.aux || !a.stats || (a.aux2 && (a.T <= 0 || a.Tp <= 0 || a.fac <= 0)))) return SEPR_EINVAL;
  if (a.bf1 && a.a16 && ((key != PRO_PLAIN * 16 + EPI_STORE && key != PRO_PLAIN * 16 + EPI_LNBWD && key != PRO_PLAIN * 16 + EPI_RES &&
                          key != PRO_PLAIN * 16 + EPI_RESDROP) || a.rows_out > 0 || (a.lda % 8) != 0)) return SEPR_EINVAL;
  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);
  // (an unknown prologue / epilogue combination below closes the slot before it reports the error)
#define SEPR_X3_BAD_KEY do { if (timed) prof_end(slot, 0.0, stream); return SEPR_EINVAL; } while (0)
  if (a.bf1 && a.a16) {
    if (epi == EPI_LNBWD) launch_x3_inst<PRO_PLAIN, EPI_LNBWD, 16 | 32>(a, stream);
    else if (epi == EPI_RES) launch_x3_inst<PRO_PLAIN, EPI_RES, 16 | 32>(a, stream);
    else if (epi == EPI_RESDROP) launch_x3_inst<PRO_PLAIN, EPI_RESDROP, 16 | 32>(a, stream);
    else launch_x3_inst<PRO_PLAIN, EPI_STORE, 16 | 32>(a, stream);
  } else if (a.bf1) {   // plain bf16 operands: the projections of the training path's "bf16" precision
    switch (key) {
      case PRO_PLAIN * 16 + EPI_STORE: launch_x3_inst<PRO_PLAIN, EPI_STORE, 16>(a, stream); break;
      case PRO_PLAIN * 16 + EPI_RES:   launch_x3_inst<PRO_PLAIN, EPI_RES, 16>(a, stream); break;
      case PRO_PLAIN * 16 + EPI_SPLIT: launch_x3_inst<PRO_PLAIN, EPI_SPLIT, 16>(a, stream); break;
      case PRO_PLAIN * 16 + EPI_LNBWD: launch_x3_inst<PRO_PLAIN, EPI_LNBWD, 16>(a, stream); break;
      case PRO_NORM * 16 + EPI_STORE:  launch_x3_inst<PRO_NORM, EPI_STORE, 16>(a, stream); break;
      case PRO_CAT2 * 16 + EPI_STORE:  launch_x3_inst<PRO_CAT2, EPI_STORE, 16>(a, stream); break;
      case PRO_NORM * 16 + EPI_GLUSAVE: launch_x3_inst<PRO_NORM, EPI_GLUSAVE, 16>(a, stream); break;
      case PRO_PLAIN * 16 + EPI_RESDROP: launch_x3_inst<PRO_PLAIN, EPI_RESDROP, 16>(a, stream); break;
      default: SEPR_X3_BAD_KEY;
    }
  } else if (site == SEPR_SITE_GCFN_UP && key == PRO_NORM * 16 + EPI_DWGLU) {
    launch_x3_inst<PRO_NORM, EPI_DWGLU, 1>(a, stream);
  } else if (site == SEPR_SITE_GCFN_DOWN && key == PRO_PLAIN * 16 + EPI_RES) {
    launch_x3_inst<PRO_PLAIN, EPI_RES, 2>(a, stream);
  } else
  switch (key) {
    case PRO_PLAIN * 16 + EPI_STORE: launch_x3_inst<PRO_PLAIN, EPI_STORE>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_GLU:   launch_x3_inst<PRO_PLAIN, EPI_GLU>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_GELU:  launch_x3_inst<PRO_PLAIN, EPI_GELU>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_RES:   launch_x3_inst<PRO_PLAIN, EPI_RES>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_SPLIT: launch_x3_inst<PRO_PLAIN, EPI_SPLIT>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_MASK:  launch_x3_inst<PRO_PLAIN, EPI_MASK>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_LNBWD: launch_x3_inst<PRO_PLAIN, EPI_LNBWD>(a, stream); break;
    case PRO_NORM * 16 + EPI_STORE:  launch_x3_inst<PRO_NORM, EPI_STORE>(a, stream); break;
    case PRO_NORM * 16 + EPI_GLU:    launch_x3_inst<PRO_NORM, EPI_GLU>(a, stream); break;
    case PRO_NORM * 16 + EPI_GATE:   launch_x3_inst<PRO_NORM, EPI_GATE>(a, stream); break;
    case PRO_CAT2 * 16 + EPI_STORE:  launch_x3_inst<PRO_CAT2, EPI_STORE>(a, stream); break;
    case PRO_NORM * 16 + EPI_GLUSAVE: launch_x3_inst<PRO_NORM, EPI_GLUSAVE>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_RESDROP: launch_x3_inst<PRO_PLAIN, EPI_RESDROP>(a, stream); break;
    default: SEPR_X3_BAD_KEY;
  }
#undef SEPR_X3_BAD_KEY
  if (timed) prof_end(slot, 2.0 * (double)a.M * (double)a.N * (double)a.K, stream);
  SEPR_CHECK_LAUNCH("gemm_x3_kernel");
  return SEPR_OK;
}

}  // namespace sepr
