// nsf_flow_inv.hip -- inverse (sampling) instantiations of the fused flow kernel.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "nsf_flow_kernel.h"

template <bool INV>
int dispatch_flow(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats, const float* in,
                         const float* x, int64_t n, int64_t x_rows, float* out_main, float* out_aux,
                         float* z_stash, float* astash, float* pstash, void* stream, bool fp32_bin) {
  if (n == 0) return 0;
  if (!cfg || !packed || !zstats || !in || !x || !out_main || n < 0 || x_rows < 1) return SBI_AMD_E_BADARG;
  NsfPlan pl;
  int nw = 0;
  int rc = nsf_plan_for_rows(cfg, n, &pl, &nw, true);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  switch (cfg->K) {
    case 4: return launch_flow_ksh<4, INV>(pl, nw, packed, zstats, in, x, n, x_rows, out_main, out_aux, z_stash, astash, pstash, st, fp32_bin);
    case 5: return launch_flow_ksh<5, INV>(pl, nw, packed, zstats, in, x, n, x_rows, out_main, out_aux, z_stash, astash, pstash, st, fp32_bin);
    case 8: return launch_flow_ksh<8, INV>(pl, nw, packed, zstats, in, x, n, x_rows, out_main, out_aux, z_stash, astash, pstash, st, fp32_bin);
    case 10: return launch_flow_ksh<10, INV>(pl, nw, packed, zstats, in, x, n, x_rows, out_main, out_aux, z_stash, astash, pstash, st, fp32_bin);
    case 16: return launch_flow_ksh<16, INV>(pl, nw, packed, zstats, in, x, n, x_rows, out_main, out_aux, z_stash, astash, pstash, st, fp32_bin);
    default: return SBI_AMD_E_UNSUPPORTED;
  }
}


template int dispatch_flow<true>(const sbi_amd_nsf_config*, const float*, const float*, const float*, const float*,
                                 int64_t, int64_t, float*, float*, float*, float*, float*, void*, bool);
