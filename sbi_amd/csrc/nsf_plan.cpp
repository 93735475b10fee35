// nsf_plan.cpp -- host-side plan builder (see nsf_plan.h).
#include "nsf_plan.h"
#include "nsf_plan_layout.h"
#include "debug_env.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int nsf_build_plan(const sbi_amd_nsf_config* cfg, int nw, NsfPlan* pl) {
  const int rc = nsf_build_layout(cfg, nw, pl);     // also on SBI_AMD_E_LDS the plan is complete
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  const int D = cfg->D, H = cfg->H, K = cfg->K;
  pl->B = cfg->tail_bound;
  pl->min_w = cfg->min_bin_width; pl->min_h = cfg->min_bin_height; pl->min_d = cfg->min_derivative;
  pl->lu_eps = cfg->lu_eps;
  pl->sqrt_h = (float)sqrt((double)H);
  pl->inv_sqrt_h = (float)(1.0 / sqrt((double)H));
  pl->one_minus_kw = (float)(1.0 - (double)cfg->min_bin_width * K);
  pl->one_minus_kh = (float)(1.0 - (double)cfg->min_bin_height * K);
  pl->d_const = (float)log(exp(1.0 - (double)cfg->min_derivative) - 1.0);
  pl->log_z = (float)(0.5 * D * log(2.0 * M_PI));
  pl->ablate = sbi_amd_dbg_ablate();   // debug aid, read once per process and announced on stderr
  return rc;
}

int nsf_plan_for_rows(const sbi_amd_nsf_config* cfg, int64_t n, NsfPlan* pl, int* nw_out, bool wide) {
  // 16 rows per wave, 1 / 2 / 4 / 8 waves per workgroup, one workgroup per CU at a time (the weight image fills most of a
  // CU's LDS): the launch runs in ceil(workgroups / 256) rounds of roughly equal length whatever the workgroup size
  // (measured per round: 0.085 / 0.083 / 0.10 ms for 2 / 4 / 8 waves), so the size is chosen to minimise the rounds --
  // 12 288 rows as 192 four-wave workgroups in ONE round (0.083 ms) rather than 384 two-wave workgroups in two (0.166),
  // 24 576 rows as 192 eight-wave workgroups rather than 384 four-wave ones -- and, among equals, to spread over more CUs.
  // `wide` (the sampling direction): 12 waves (3 per SIMD, single staging buffer per wave) when there are enough rows
  // and the image leaves room; SBI_AMD_ABLATE bit 4096 switches it off (the density direction is compiled for <= 8 waves and never takes it).
  const int abl = sbi_amd_dbg_ablate();
  if (wide && !(abl & (1024 | 4096)) && (n + 16 * 12 - 1) / (16 * 12) >= 256) {
    if (nsf_build_plan(cfg, 12, pl) == 0) { *nw_out = 12; return 0; }
  }
  const int nw_max = (abl & 1024) ? 4 : 8;               // debug aid: forward kernel with one wave per SIMD
  int best = 0;
  double best_cost = 0.0;
  NsfPlan cand;
  for (int nw = nw_max; nw >= 1; nw >>= 1) {
    const int rc = nsf_build_plan(cfg, nw, &cand);
    if (rc == SBI_AMD_E_LDS) continue;
    if (rc) return rc;
    const int64_t wgs = (n + 16 * nw - 1) / (16 * nw);
    const double cost = (double)((wgs + 255) / 256) * (nw == 8 ? 1.2 : 1.0);      // two waves per SIMD: ~1.2 x per round
    if (best == 0 || cost <= best_cost) {          // (<=: among equals the smaller workgroup = more CUs busy)
      best = nw;
      best_cost = cost;
      *pl = cand;
    }
  }
  if (best == 0) return SBI_AMD_E_LDS;
  *nw_out = best;
  return 0;
}

// host-side answer (tests, tuning): waves per workgroup the throughput forward / sampling kernel uses for an n-row call
extern "C" int sbi_amd_nsf_plan_waves(const sbi_amd_nsf_config* cfg, int64_t n, int32_t sampling) {
  NsfPlan pl;
  int nw = 0;
  const int rc = nsf_plan_for_rows(cfg, n < 1 ? 1 : n, &pl, &nw, sampling != 0);
  return rc ? rc : nw;
}

extern "C" int64_t sbi_amd_nsf_param_count(const sbi_amd_nsf_config* cfg) {
  NsfPlan pl;
  int rc = nsf_build_plan(cfg, 1, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  return pl.n_params;
}
extern "C" int64_t sbi_amd_nsf_layer_offset(const sbi_amd_nsf_config* cfg, int32_t t) {
  NsfPlan pl;
  int rc = nsf_build_plan(cfg, 1, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  if (t < 0 || t >= pl.T) return SBI_AMD_E_BADARG;
  return pl.g_layer[t];
}
extern "C" int64_t sbi_amd_nsf_lu_offset(const sbi_amd_nsf_config* cfg, int32_t t) {
  NsfPlan pl;
  int rc = nsf_build_plan(cfg, 1, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  if (t < 0 || t >= pl.T) return SBI_AMD_E_BADARG;
  return pl.g_layer[t] + pl.shape[t & 1].g_lu;
}
extern "C" int sbi_amd_nsf_abi_version(void) { return SBI_AMD_NSF_ABI_VERSION; }
extern "C" const char* sbi_amd_nsf_arch(void) { return "gfx950"; }
