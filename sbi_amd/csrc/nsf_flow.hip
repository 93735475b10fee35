// nsf_flow.hip -- forward (log_prob) instantiations of the fused flow kernel + pack kernel + C ABI.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "nsf_flow_kernel.h"
#include "nsf_coop_host.h"

// flat parameters -> packed MFMA weight image, grid (T, workgroups per transform).  inverses = 0: everything but the
// explicit LU inverses (what log_prob and the training pass read: re-packed every optimizer step); 1: only U^-1 / L^-1,
// one workgroup per transform (a serial fp64 substitution that only the sampling direction needs)
__global__ void __launch_bounds__(512)
nsf_pack_kernel(const NsfPlan pl, const float* __restrict__ params, float* __restrict__ packed, const int inverses) {
  const int t = blockIdx.x;
  const ShapeDesc& S = pl.shape[pl.ctx_mlp ? 0 : (t & 1)];
  float* img = packed + (long long)t * pl.img_floats;
  if (inverses) {
    if (S.l_Ui >= 0) pack_lu_inverse(img, params + pl.g_layer[t], S, pl.D, pl.lu_eps, threadIdx.x, blockDim.x);
    return;
  }
  pack_layer(img, params + pl.g_layer[t], pl, S, blockIdx.y * blockDim.x + threadIdx.x, gridDim.y * blockDim.x);
}


// ------------------------------------------------------------------ host side
template <bool INV>
int dispatch_flow(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats, const float* in,
                         const float* x, int64_t n, int64_t x_rows, float* out_main, float* out_aux,
                         float* z_stash, float* astash, float* pstash, void* stream, bool fp32_bin) {
  if (n == 0) return 0;
  if (!cfg || !packed || !zstats || !in || !x || !out_main || n < 0 || x_rows < 1) return SBI_AMD_E_BADARG;
  NsfPlan pl;
  int nw = 0;
  int rc = nsf_plan_for_rows(cfg, n, &pl, &nw);
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


template int dispatch_flow<false>(const sbi_amd_nsf_config*, const float*, const float*, const float*, const float*,
                                  int64_t, int64_t, float*, float*, float*, float*, float*, void*, bool);
// inverse instantiations live in nsf_flow_inv.hip (separate TU: parallel build)
extern template int dispatch_flow<true>(const sbi_amd_nsf_config*, const float*, const float*, const float*,
                                        const float*, int64_t, int64_t, float*, float*, float*, float*, float*, void*, bool);

// used by the training path (nsf_train.hip): forward with per-layer state stash
int nsf_log_prob_stash(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats, const float* theta,
                       const float* x, int64_t n, int64_t x_rows, float* logp_out, float* noise_out,
                       float* z_stash, float* astash, float* pstash, void* stream, bool fp32_bin) {
  return dispatch_flow<false>(cfg, packed, zstats, theta, x, n, x_rows, logp_out, noise_out, z_stash, astash, pstash, stream,
                              fp32_bin);
}

// The packed buffer holds TWO images: [throughput image: T LDS images | cooperative image: fragment-ordered, forward
// and transposed matrices (nsf_coop.h); absent for shapes the cooperative kernels do not take].
extern "C" int64_t sbi_amd_nsf_packed_floats(const sbi_amd_nsf_config* cfg) {
  NsfPlan pl;
  int rc = nsf_build_plan(cfg, 1, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  return nsf_packed_floats(pl) + coop_packed_floats(cfg);
}

// which image does an n-row call read?  0: the throughput image, 1: the cooperative image
extern "C" int sbi_amd_nsf_image_kind(const sbi_amd_nsf_config* cfg, int64_t n, int32_t training) {
  if (!cfg) return SBI_AMD_E_BADARG;
  NsfPlan pl;
  CoopPlan cp;
  if (cfg->H > 16 * NSF_HT) return coop_shape_ok(cfg, &pl, &cp) ? 1 : SBI_AMD_E_UNSUPPORTED;   // wide nets: always
  return coop_applies(cfg, n, training != 0, &pl, &cp) ? 1 : 0;
}

// images: bit 0 the throughput image, bit 1 the cooperative image, bit 2 the cooperative image's explicit LU inverses
// (only the sampling direction of nets with hidden > 64 reads them), bit 3 the throughput image's explicit LU inverses
// (only sbi_amd_nsf_sample reads them) (a training loop at a fixed batch size only ever
// needs one of them re-packed per step)
extern "C" int sbi_amd_nsf_pack_images(const sbi_amd_nsf_config* cfg, const float* params, float* packed,
                                       int32_t images, void* stream) {
  if (!cfg || !params || !packed) return SBI_AMD_E_BADARG;
  NsfPlan pl;
  int rc = nsf_build_plan(cfg, 1, &pl);
  if (rc && rc != SBI_AMD_E_LDS) return rc;
  if ((images & 1) && pl.img_floats > 0)
    hipLaunchKernelGGL(nsf_pack_kernel, dim3(pl.T, 24), dim3(256), 0, (hipStream_t)stream, pl, params, packed, 0);
  if ((images & 8) && pl.img_floats > 0)      // the throughput image's explicit LU inverses (sampling direction only)
    hipLaunchKernelGGL(nsf_pack_kernel, dim3(pl.T, 1), dim3(256), 0, (hipStream_t)stream, pl, params, packed, 1);
  if (images & 6) {     // bit 1: the cooperative image; bit 2: its explicit LU inverses (sampling direction, wide nets)
    rc = coop_pack(cfg, params, packed + nsf_packed_floats(pl), (images >> 1) & 3, stream);
    if (rc) return rc;
  }
  return (int)hipGetLastError();
}

extern "C" int sbi_amd_nsf_pack(const sbi_amd_nsf_config* cfg, const float* params, float* packed, void* stream) {
  return sbi_amd_nsf_pack_images(cfg, params, packed, 15, stream);
}

extern "C" int sbi_amd_nsf_log_prob(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats,
                                    const float* theta, const float* x, int64_t n, int64_t x_rows,
                                    float* logp_out, float* noise_out, void* stream) {
  if (n == 0) return 0;
  if (!cfg || !packed || !zstats || !theta || !x || !logp_out || n < 0 || x_rows < 1) return SBI_AMD_E_BADARG;
  NsfPlan pl;
  CoopPlan cp;
  if (coop_applies(cfg, n, false, &pl, &cp))   // small batches: four cooperating waves per 16-row tile (nsf_coop.h)
    return coop_log_prob(cfg, pl, cp, packed + nsf_packed_floats(pl), zstats, theta, x, n, x_rows, logp_out,
                         noise_out, stream);
  return dispatch_flow<false>(cfg, packed, zstats, theta, x, n, x_rows, logp_out, noise_out, nullptr, nullptr, nullptr, stream, false);
}

extern "C" int sbi_amd_nsf_sample(const sbi_amd_nsf_config* cfg, const float* packed, const float* zstats,
                                  const float* noise, const float* x, int64_t n, int64_t x_rows,
                                  float* theta_out, float* logabsdet_out, void* stream) {
  if (n > 0 && cfg) {
    // hidden > 64: the wide cooperative kernel at every batch size; narrower nets: its one-m-tile-per-wave instantiation
    // for the small calls the cooperative family takes (half the latency of the throughput kernel below ~4 000 draws)
    NsfPlan pl;
    CoopPlan cp;
    const bool wide = cfg->H > 16 * NSF_HT;
    if (coop_applies(cfg, n, false, &pl, &cp)) {
      if (!packed || !zstats || !noise || !x || !theta_out || x_rows < 1) return SBI_AMD_E_BADARG;
      return coop_sample(cfg, pl, cp, packed + nsf_packed_floats(pl), zstats, noise, x, n, x_rows, theta_out,
                         logabsdet_out, stream);
    }
    if (wide) return SBI_AMD_E_UNSUPPORTED;
  }
  return dispatch_flow<true>(cfg, packed, zstats, noise, x, n, x_rows, theta_out, logabsdet_out, nullptr, nullptr, nullptr, stream, false);
}
