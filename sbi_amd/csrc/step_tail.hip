// step_tail.hip -- table-driven re-pack of the NSF weight images after an optimizer step (gfx950).
//
// Every training step ends with the weight image of the next step's kernels being rebuilt from the flat parameters
// (what nflows redoes inside every forward call: LULinear._create_lower_upper and the .t() views, nflows
// transforms/lu.py; here: nsf_pack_kernel / nsf_coop_pack_kernel, 11 - 13 us -- kernels that run an integer division
// chain and a 35-entry descriptor search per float to re-derive WHERE each parameter goes, which never changes between
// steps).  sbi_amd_nsf_build_step_map computes the "where" ONCE per network: a gather table
//       image position j  ->  parameter that lives there (| softplus flag), constant, or a transform's logabsdet
// and sbi_amd_nsf_table_pack rewrites the image from it: one coalesced 16-byte store per four positions, the table read
// coalesced, the parameters (392 KB) gathered out of L2.
//
// The table is not derived from a second copy of the image layouts: it is MEASURED by running the real pack kernels on
// three probe parameter vectors (0; 32 + i; 2 (32 + i)) and reading off, per image position, which parameter landed
// there (a copy doubles with its source, a constant does not move, softplus(u) + eps is the only non-linear entry) --
// and then VERIFIED bit for bit against the real pack kernel on a fourth, random vector before it is handed out.
// Whatever layout the pack kernels implement, the table reproduces it or the build call fails.
//
// Non-linear entries (LULinear): U_ii = softplus(u_i) + eps at the flagged positions, and the transform's
// logabsdet = sum_i log U_ii (precomputed into the image so no kernel re-derives it): one wavefront per transform
// (lane k <-> u_k) sums the logs in the pack kernels' order, so the images are bit-identical.
//
// (A first version fused the scatter into the Adam kernel -- parameter i's thread storing to the positions that hold
// it.  Measured: 4-byte stores scattered over the image made that kernel 11 - 14 us and the NEXT forward kernel 3 us
// slower at 8 192 rows (partial lines written from eight L2s); profiles/r4_step_tail.txt.  The gather form writes
// whole lines.)
#include <hip/hip_runtime.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include "nsf_coop_host.h"
#include "nsf_device.h"

#define ST_THREADS 256
#define ST_NWG 256
static inline int st_main_wgs(int n) { const int w = (n / 4 + ST_THREADS - 1) / ST_THREADS; return w < ST_NWG ? (w < 1 ? 1 : w) : ST_NWG; }
#define ST_HDR 96            // header ints in front of the tables
#define ST_MAGIC 0x53544d31  // "STM1"
#define ST_SOFTPLUS 0x40000000
// header: [0] magic [1] images [2] P [3], [7] end / begin of the image(s)' positions, in groups of four [4] T
//         [5] D (0: no LULinear) [6] first table int  [8 + t] first diagonal parameter of transform t
//         [24 + 2 t + {0, 1}] image positions of transform t's logabsdet (-1: none)
//         (build only: [56 + t] logabsdet slots found, [72] unclassifiable positions)
#define ST_CK(e) do { const hipError_t e_ = (e); if (e_ != hipSuccess) return (int)e_; } while (0)
#define ST_H_SLOT 56
#define ST_H_BAD 72

// ---------------------------------------------------------------------------------------------- table construction
__global__ void st_probe_fill(float* __restrict__ pa, float* __restrict__ pb, float* __restrict__ pr, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  pa[i] = 32.f + (float)i;
  pb[i] = 2.f * (32.f + (float)i);
  // verification vector: a cheap hash in (-1, 1) (no two neighbours alike; softplus inputs on both sides of 0)
  unsigned h = (unsigned)i * 2654435761u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  pr[i] = ((float)(h & 0xffffff) / 8388608.f) - 1.f;
}

// code[j]: -1 constant (never rewritten), i plain copy of parameter i, i | ST_SOFTPLUS softplus(parameter i) + eps,
//          -2 - t logabsdet of transform t, -100 unclassifiable (build fails)
__global__ void st_classify(const float* __restrict__ imz, const float* __restrict__ ima, const float* __restrict__ imb,
                            int* __restrict__ code, int* __restrict__ hdr, int n_img, int P,
                            int base1, int stride1, int base2, int stride2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_img) return;
  const float z = imz[j], a = ima[j], b = imb[j];
  int c = -100;
  if (a == b) c = -1;                                                 // same value whatever the parameters
  else if (z == 0.f && b == 2.f * a && a == rintf(a) && a >= 32.f && a < 32.f + (float)P) c = (int)a - 32;
  else if (z != 0.f) {
    const float ra = rintf(a);
    if (fabsf((b - a) - ra) <= 0.5f && ra >= 32.f && ra < 32.f + (float)P) {
      const int i = (int)ra - 32;                                     // only LULinear's diagonal goes through softplus
      bool diag = false;
      for (int t = 0; t < hdr[4]; ++t) diag = diag || (i >= hdr[8 + t] && i < hdr[8 + t] + hdr[5]);
      c = (diag && hdr[5] > 0) ? (i | ST_SOFTPLUS) : -100;
    } else {                                                            // a sum of logs: the transform's logabsdet
      const int t = (stride2 > 0 && j >= base2) ? (j - base2) / stride2 : (stride1 > 0 ? (j - base1) / stride1 : 0);
      c = -100;
      if (t >= 0 && t < hdr[4]) {
        const int slot = atomicAdd(&hdr[ST_H_SLOT + t], 1);
        if (slot < 2) { hdr[24 + 2 * t + slot] = j; c = -2 - t; }
      }
    }
  }
  code[j] = c;
  if (c == -100) atomicAdd(&hdr[ST_H_BAD], 1);
}

// the transform's logabsdet: lane k < D holds log(softplus(u_k) + eps); summed in the pack kernels' order
__device__ __forceinline__ void st_logabsdet(const float* __restrict__ p, float* __restrict__ packed,
                                             const int* __restrict__ map, int t, int lane, float lu_eps) {
  const int D = map[5];
  const float lg = lane < D ? logf(softplus_f(p[map[8 + t] + lane]) + lu_eps) : 0.f;
  float a = 0.f;
  for (int k = 0; k < D; ++k) a += __shfl(lg, k);
  if (lane == 0) {
    if (map[24 + 2 * t] >= 0) packed[map[24 + 2 * t]] = a;
    if (map[25 + 2 * t] >= 0) packed[map[25 + 2 * t]] = a;
  }
}
// grid: main workgroups walk the image four positions per thread; the last (T + 3) / 4 workgroups (LULinear only)
// give one wavefront to each transform's logabsdet
__global__ void __launch_bounds__(ST_THREADS)
nsf_table_pack_kernel(const float* __restrict__ p, float* __restrict__ packed, const int* __restrict__ map,
                      float lu_eps) {
  const int T = map[4], D = map[5], n4 = map[3];      // positions [4 map[7], 4 map[3]) belong to the table's image(s)
  const int4* code = reinterpret_cast<const int4*>(map + map[6]);
  const int nmain = gridDim.x - (D > 0 ? (T + 3) / 4 : 0);
  if ((int)blockIdx.x >= nmain) {
    const int t = 4 * (blockIdx.x - nmain) + (threadIdx.x >> 6);
    if (t < T) st_logabsdet(p, packed, map, t, threadIdx.x & 63, lu_eps);
    return;
  }
  f4* out = reinterpret_cast<f4*>(packed);
  for (int q = map[7] + blockIdx.x * ST_THREADS + threadIdx.x; q < n4; q += nmain * ST_THREADS) {
    const int4 c = code[q];
    if ((c.x & c.y & c.z & c.w) == -1) continue;          // four constants: nothing to rewrite
    const int cs[4] = {c.x, c.y, c.z, c.w};
    f4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float x = p[cs[r] >= 0 ? (cs[r] & ~ST_SOFTPLUS) : 0];
      v[r] = (cs[r] & ST_SOFTPLUS) ? softplus_f(x) + lu_eps : x;
    }
    if (c.x >= 0 && c.y >= 0 && c.z >= 0 && c.w >= 0) out[q] = v;       // the common case: one 16-byte store
    else {                                                             // constants / a logabsdet slot in the group stay
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (cs[r] >= 0) packed[4 * q + r] = v[r];
    }
  }
}
__global__ void st_compare(const float* __restrict__ x, const float* __restrict__ y, int n, int* __restrict__ bad) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && __float_as_uint(x[j]) != __float_as_uint(y[j])) atomicAdd(bad, 1);
}

static bool st_plans(const sbi_amd_nsf_config* cfg, NsfPlan* pl, CoopPlan* cp, bool* has_coop) {
  int rc = nsf_build_plan(cfg, 1, pl);
  if (rc && rc != SBI_AMD_E_LDS) return false;
  NsfPlan p2;
  *has_coop = coop_shape_ok(cfg, &p2, cp);
  return true;
}
// the table covers the whole `packed` buffer (both images), rounded up to a multiple of four positions
static int64_t st_table_ints(const sbi_amd_nsf_config* cfg) {
  const int64_t n = sbi_amd_nsf_packed_floats(cfg);
  return n < 0 ? n : (n + 3) / 4 * 4;
}

extern "C" int64_t sbi_amd_nsf_step_map_ints(const sbi_amd_nsf_config* cfg) {
  if (!cfg) return SBI_AMD_E_BADARG;
  const int64_t n = st_table_ints(cfg);
  return n < 0 ? n : ST_HDR + n;
}
extern "C" int64_t sbi_amd_nsf_step_map_workspace_floats(const sbi_amd_nsf_config* cfg) {
  if (!cfg) return SBI_AMD_E_BADARG;
  const int64_t P = sbi_amd_nsf_param_count(cfg), n = st_table_ints(cfg);
  if (P < 0) return P;
  if (n < 0) return n;
  return 4 * (P + 4) /* probe vectors */ + 3 * n /* their images */ + 16;
}

// Builds the gather table for the image(s) in `images` (bit 0 throughput, bit 1 cooperative: what a training loop at
// a fixed batch size re-packs per step).  Synchronises `stream` (one-time set-up).  `packed` must be the image buffer
// the steps will use: it is fully packed from `params` here (constants included), so that sbi_amd_nsf_table_pack only
// ever has to rewrite parameter-dependent positions.
// Tables this process built: map pointer -> the configuration it was built (and verified) for.  sbi_amd_nsf_table_pack
// launches a kernel that trusts the table's header and gather indices, so it only accepts a table registered here for
// the SAME configuration (a table of another network, a buffer that was never built or whose build failed: E_BADARG).
#include <mutex>
#include <unordered_map>
static std::mutex g_map_mu;
static std::unordered_map<const void*, sbi_amd_nsf_config> g_maps;
// (entries are never evicted behind a live table's back: the owner releases its table with
//  sbi_amd_nsf_release_step_map when it drops the buffer -- 48 bytes per entry otherwise stay until process exit)
static void st_register(const void* map, const sbi_amd_nsf_config* cfg) {
  std::lock_guard<std::mutex> lk(g_map_mu);
  g_maps[map] = *cfg;
}
static void st_unregister(const void* map) {
  std::lock_guard<std::mutex> lk(g_map_mu);
  g_maps.erase(map);
}
// field by field (a C caller's struct padding is not part of the configuration)
static bool st_same_cfg(const sbi_amd_nsf_config& a, const sbi_amd_nsf_config& b) {
  return a.D == b.D && a.C == b.C && a.H == b.H && a.K == b.K && a.T == b.T && a.NB == b.NB &&
         a.tail_bound == b.tail_bound && a.min_bin_width == b.min_bin_width && a.min_bin_height == b.min_bin_height &&
         a.min_derivative == b.min_derivative && a.lu_eps == b.lu_eps && a.ctx_layers == b.ctx_layers;
}
static bool st_registered_for(const void* map, const sbi_amd_nsf_config* cfg) {
  std::lock_guard<std::mutex> lk(g_map_mu);
  auto it = g_maps.find(map);
  return it != g_maps.end() && st_same_cfg(it->second, *cfg);
}
// The owner of a table is about to free (or rebuild) its buffer: sbi_amd_nsf_table_pack refuses the address from now
// on, so a later allocation that happens to reuse it cannot pass for a built table.
extern "C" int sbi_amd_nsf_release_step_map(const int32_t* map) {
  if (!map) return SBI_AMD_E_BADARG;
  st_unregister(map);
  return 0;
}

extern "C" int sbi_amd_nsf_build_step_map(const sbi_amd_nsf_config* cfg, int32_t images, const float* params,
                                          float* packed, int32_t* map, float* workspace, void* stream) {
  if (!cfg || !params || !packed || !map || !workspace || !(images & 3) || (images & ~3)) return SBI_AMD_E_BADARG;
  st_unregister(map);        // (registered again only when the build below completes)
  if (!(cfg->lu_eps >= 0.f && cfg->lu_eps < 0.25f)) return SBI_AMD_E_UNSUPPORTED;   // the probe rounds softplus + eps
  NsfPlan pl;
  CoopPlan cp;
  bool has_coop = false;
  if (!st_plans(cfg, &pl, &cp, &has_coop)) return SBI_AMD_E_UNSUPPORTED;
  const int P = pl.n_params;
  if (P + 32 >= (1 << 23) || pl.T > 16 || pl.D > 64) return SBI_AMD_E_UNSUPPORTED;
  const int64_t n64 = st_table_ints(cfg);
  if (n64 <= 0 || n64 >= ST_SOFTPLUS) return SBI_AMD_E_UNSUPPORTED;
  const int n = (int)n64;                                    // (multiple of 4; the last <= 3 positions are padding)
  const int n_real = (int)sbi_amd_nsf_packed_floats(cfg);
  const int n_thr = (int)nsf_packed_floats(pl);
  if ((images & 1) && pl.img_floats <= 0) return SBI_AMD_E_UNSUPPORTED;
  if ((images & 2) && !has_coop) return SBI_AMD_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int P4 = P + 4;
  float* pz = workspace;        // all-zero parameters
  float* pa = pz + P4;          // 32 + i
  float* pb = pa + P4;          // 2 (32 + i)
  float* pr = pb + P4;          // verification vector
  float* imz = pr + P4;
  float* ima = imz + n;
  float* imb = ima + n;
  int* bad = (int*)(imb + n);
  int* code = (int*)map + ST_HDR;
  // ---- probes through the real pack kernels
  ST_CK(hipMemsetAsync(workspace, 0, sizeof(float) * (size_t)sbi_amd_nsf_step_map_workspace_floats(cfg), st));
  ST_CK(hipMemsetAsync(map, 0, sizeof(int32_t) * (size_t)sbi_amd_nsf_step_map_ints(cfg), st));
  hipLaunchKernelGGL(st_probe_fill, dim3((P + 255) / 256), dim3(256), 0, st, pa, pb, pr, P);
  int rc = sbi_amd_nsf_pack_images(cfg, pz, imz, images, stream);
  if (!rc) rc = sbi_amd_nsf_pack_images(cfg, pa, ima, images, stream);
  if (!rc) rc = sbi_amd_nsf_pack_images(cfg, pb, imb, images, stream);
  if (rc) return rc;
  // ---- header
  int h[ST_HDR];
  for (int i = 0; i < ST_HDR; ++i) h[i] = 0;
  h[0] = ST_MAGIC; h[1] = images; h[2] = P; h[4] = pl.T; h[5] = pl.ctx_mlp ? 0 : pl.D;
  h[6] = ST_HDR;
  h[7] = (images & 1) ? 0 : n_thr / 4;                       // groups of four positions the pack walks
  h[3] = (images & 2) ? n / 4 : (n_thr + 3) / 4;
  for (int t = 0; t < pl.T; ++t) {
    const ShapeDesc& S = pl.shape[pl.ctx_mlp ? 0 : (t & 1)];
    const int ntri = pl.D * (pl.D - 1) / 2;
    h[8 + t] = pl.g_layer[t] + S.g_lu + 2 * ntri;
  }
  for (int i = 24; i < 24 + 32; ++i) h[i] = -1;     // logabsdet positions: filled in by the classification
  ST_CK(hipMemcpyAsync(map, h, sizeof(h), hipMemcpyHostToDevice, st));
  ST_CK(hipStreamSynchronize(st));   // (h is on the stack)
  // ---- classify every position of the buffer (positions of images not asked for: constants)
  hipLaunchKernelGGL(st_classify, dim3((n + 255) / 256), dim3(256), 0, st, imz, ima, imb, code, (int*)map, n, P, 0,
                     pl.img_floats, has_coop ? n_thr : 0x7fffffff, has_coop ? cp.img_floats : 0);
  int hb[ST_HDR];
  ST_CK(hipMemcpyAsync(hb, map, sizeof(hb), hipMemcpyDeviceToHost, st));
  ST_CK(hipStreamSynchronize(st));
  if (hb[ST_H_BAD] != 0) return SBI_AMD_E_UNSUPPORTED;      // a position that is neither constant, copy, softplus nor logabsdet
  for (int i = ST_H_SLOT; i < ST_HDR; ++i) hb[i] = 0;     // (scratch counters of the classification)
  ST_CK(hipMemcpyAsync(map, hb, sizeof(hb), hipMemcpyHostToDevice, st));
  ST_CK(hipStreamSynchronize(st));
  // ---- verification: the table-driven pack of a random vector over an image that holds the constants must equal the
  //      real pack of it, bit for bit
  rc = sbi_amd_nsf_pack_images(cfg, pr, ima, images, stream);          // reference
  if (rc) return rc;
  const int extra = (h[5] > 0) ? (pl.T + 3) / 4 : 0;                  // imb holds the constants (packed from pb)
  hipLaunchKernelGGL(nsf_table_pack_kernel, dim3(st_main_wgs(n) + extra), dim3(ST_THREADS), 0, st, pr, imb,
                     (const int*)map, cfg->lu_eps);
  hipLaunchKernelGGL(st_compare, dim3((n_real + 255) / 256), dim3(256), 0, st, ima, imb, n_real, bad);
  int nbad = -1;
  ST_CK(hipMemcpyAsync(&nbad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
  ST_CK(hipStreamSynchronize(st));
  if (nbad != 0) return SBI_AMD_E_UNSUPPORTED;
  // ---- the caller's image: everything (constants included) from the caller's parameters
  rc = sbi_amd_nsf_pack_images(cfg, params, packed, images, stream);
  if (rc) return rc;
  if (hipStreamSynchronize(st) != hipSuccess) return (int)hipGetLastError();
  st_register(map, cfg);
  return 0;
}

// Re-pack of the image(s) the table was built for, from the current `params`: bit-identical to
// sbi_amd_nsf_pack_images(cfg, params, packed, images) on a buffer that build_step_map (or any full pack) initialised.
extern "C" int sbi_amd_nsf_table_pack(const sbi_amd_nsf_config* cfg, const float* params, float* packed,
                                      const int32_t* map, void* stream) {
  if (!cfg || !params || !packed || !map) return SBI_AMD_E_BADARG;
  const int64_t n = st_table_ints(cfg);
  if (n <= 0) return n < 0 ? (int)n : 0;
  if (!st_registered_for(map, cfg)) return SBI_AMD_E_BADARG;   // not a table sbi_amd_nsf_build_step_map completed for cfg
  // (the kernel takes the number of logabsdet workgroups from the table's header, h[5] = theta-dim or 0: same rule)
  const int extra = (cfg->D > 1) ? (cfg->T + 3) / 4 : 0;      // theta-dim 1 has no LULinear (ContextSplineMap)
  hipLaunchKernelGGL(nsf_table_pack_kernel, dim3(st_main_wgs((int)n) + extra), dim3(ST_THREADS), 0,
                     (hipStream_t)stream, params, packed, (const int*)map, cfg->lu_eps);
  return (int)hipGetLastError();
}
