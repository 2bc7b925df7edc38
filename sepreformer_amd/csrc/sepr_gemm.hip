// Instantiations + host launcher of the f32-MFMA projection core, and the opt-in launch timer.
#include <mutex>
#include <vector>

#include "sepr_gemm.h"

namespace sepr {

// ---- opt-in per-site launch timer (bench.py's roofline leg) ---------------------------------------
namespace {
struct Prof {
  std::mutex mu;
  int site = SEPR_SITE_NONE;
  int cap = 0;
  std::vector<hipEvent_t> ev;  // 2 per launch
  long long launches = 0;
  double flops = 0.0;
  double bytes = 0.0, last_bytes = 0.0;   // algorithmic HBM bytes of the timed launches (sites that report them)
};
Prof g_prof;
}  // namespace

bool prof_begin(int site, hipStream_t stream, long long* slot) {
  if (g_prof.site == SEPR_SITE_NONE || site != g_prof.site) return false;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (site != g_prof.site || g_prof.launches >= g_prof.cap) return false;
  *slot = g_prof.launches++;
  (void)hipEventRecord(g_prof.ev[2 * *slot], stream);
  return true;
}
void prof_end(long long slot, double flops, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  (void)hipEventRecord(g_prof.ev[2 * slot + 1], stream);
  g_prof.flops += flops;
}
void prof_bytes(double bytes) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.bytes += bytes;
}

// co-resident workgroups: 2 per CU (LDS-limited), as a multiple of 8 so a workgroup keeps its XCD
int persistent_grid() {
  static int cached = 0;
  if (cached == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    cached = ((2 * cus + 7) / 8) * 8;
  }
  return cached;
}

template <int PRO, int EPI, int TAG = 0>
static void launch_inst(const GemmArgs& a, hipStream_t stream) {
  const int tiles = gemm_tiles(a, EPI);
#if SEPR_GEMM_PERSIST
  const int cap = persistent_grid();
  const int grid = tiles < cap ? tiles : cap;
#else
  const int grid = tiles;
#endif
  hipLaunchKernelGGL((gemm_kernel<PRO, EPI, TAG>), dim3(grid), dim3(GEMM_THREADS), 0, stream, a);
}

int launch_gemm(int pro, int epi, const GemmArgs& a, int site, hipStream_t stream) {
  if (a.M <= 0) return SEPR_OK;
  if (a.N <= 0 || a.K <= 0 || (a.K % GEMM_BK) != 0 || (a.N % 4) != 0) return SEPR_EINVAL;
  if ((epi == EPI_GLU || epi == EPI_GLUSAVE) && ((a.N / 2) % 4) != 0) return SEPR_EINVAL;
  if (epi == EPI_GLUSAVE && !a.Ysave) return SEPR_EINVAL;
  if (epi == EPI_RESDROP && a.drop_thr == 0u) return SEPR_EINVAL;
  if (!a.A || !a.W || !a.Y) return SEPR_EINVAL;
  if ((a.lda % 4) != 0 || (a.ldc % 4) != 0) return SEPR_EINVAL;
  if (pro == PRO_CAT2 && (!a.A2 || (a.ksplit % GEMM_BK) != 0 || (a.lda2 % 4) != 0)) return SEPR_EINVAL;
  if (pro == PRO_NORM && (!a.stats || (a.gamma != nullptr) != (a.beta != nullptr))) return SEPR_EINVAL;
  // staging addresses are 32-bit element offsets from the bases
  {
    const long long src_rows = a.rows_out > 0 ? ((long long)(a.M + a.rows_out - 1) / a.rows_out) * a.rows_src : a.M;
    if (src_rows * a.lda >= (1LL << 32) || (long long)a.N * a.K >= (1LL << 32)) return SEPR_EINVAL;
    if (pro == PRO_CAT2 && (long long)a.M * a.lda2 >= (1LL << 32)) return SEPR_EINVAL;
  }
  if ((epi == EPI_GLU || epi == EPI_GLUSAVE || epi == EPI_DWGLU) && !a.bias) return SEPR_EINVAL;
  if (epi == EPI_DWGLU && (!a.dw_w || !a.dw_b || a.T <= 0 || ((a.N / 2) % 4) != 0)) return SEPR_EINVAL;
  if (epi == EPI_LNBWD && (a.N > GEMM_BN || !a.aux || !a.stats || (a.aux2 && (a.T <= 0 || a.Tp <= 0 || a.fac <= 0)))) return SEPR_EINVAL;

  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);

  const int key = pro * 16 + epi;
  if (site == SEPR_SITE_GCFN_UP && key == PRO_NORM * 16 + EPI_DWGLU) {
    launch_inst<PRO_NORM, EPI_DWGLU, 1>(a, stream);
  } else if (site == SEPR_SITE_GCFN_DOWN && key == PRO_PLAIN * 16 + EPI_RES) {
    launch_inst<PRO_PLAIN, EPI_RES, 2>(a, stream);
  } else
  switch (key) {
    case PRO_PLAIN * 16 + EPI_STORE: launch_inst<PRO_PLAIN, EPI_STORE>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_GLU:   launch_inst<PRO_PLAIN, EPI_GLU>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_GELU:  launch_inst<PRO_PLAIN, EPI_GELU>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_RES:   launch_inst<PRO_PLAIN, EPI_RES>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_SPLIT: launch_inst<PRO_PLAIN, EPI_SPLIT>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_MASK:  launch_inst<PRO_PLAIN, EPI_MASK>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_LNBWD: launch_inst<PRO_PLAIN, EPI_LNBWD>(a, stream); break;
    case PRO_NORM * 16 + EPI_STORE:  launch_inst<PRO_NORM, EPI_STORE>(a, stream); break;
    case PRO_NORM * 16 + EPI_GLU:    launch_inst<PRO_NORM, EPI_GLU>(a, stream); break;
    case PRO_NORM * 16 + EPI_GATE:   launch_inst<PRO_NORM, EPI_GATE>(a, stream); break;
    case PRO_CAT2 * 16 + EPI_STORE:  launch_inst<PRO_CAT2, EPI_STORE>(a, stream); break;
    case PRO_NORM * 16 + EPI_GLUSAVE: launch_inst<PRO_NORM, EPI_GLUSAVE>(a, stream); break;
    case PRO_PLAIN * 16 + EPI_RESDROP: launch_inst<PRO_PLAIN, EPI_RESDROP>(a, stream); break;
    default: return SEPR_EINVAL;
  }
  if (timed) prof_end(slot, 2.0 * (double)a.M * (double)a.N * (double)a.K, stream);
  SEPR_CHECK_LAUNCH("gemm_kernel");
  return SEPR_OK;
}

}  // namespace sepr

extern "C" int sepr_prof_start(int site, int max_launches) {
  using namespace sepr;
  if (site <= SEPR_SITE_NONE || site >= SEPR_SITE_COUNT || max_launches <= 0) return SEPR_EINVAL;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.assign(2 * (size_t)max_launches, nullptr);
  for (auto& e : g_prof.ev)
    if (hipEventCreate(&e) != hipSuccess) return SEPR_EHIP;
  g_prof.cap = max_launches;
  g_prof.launches = 0;
  g_prof.flops = 0.0;
  g_prof.bytes = 0.0;
  g_prof.site = site;
  return SEPR_OK;
}

extern "C" double sepr_prof_last_bytes(void) {
  std::lock_guard<std::mutex> lk(sepr::g_prof.mu);
  return sepr::g_prof.last_bytes;
}

extern "C" int sepr_prof_stop(long long* launches, double* total_ms, double* flops) {
  using namespace sepr;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  g_prof.site = SEPR_SITE_NONE;
  double ms = 0.0;
  for (long long i = 0; i < g_prof.launches; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess) return SEPR_EHIP;
    if (hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess) return SEPR_EHIP;
    ms += t;
  }
  if (launches) *launches = g_prof.launches;
  if (total_ms) *total_ms = ms;
  if (flops) *flops = g_prof.flops;
  g_prof.last_bytes = g_prof.bytes;
  for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
  g_prof.ev.clear();
  g_prof.cap = 0;
  g_prof.launches = 0;
  return SEPR_OK;
}
