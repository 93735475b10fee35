// ode.hip -- device-resident Dormand-Prince 5(4) step machinery for the probability-flow ODE of the FMPE posterior
// (include/sbi_amd_fmpe.h, "ODE sampler" block).  The right-hand side is the velocity kernel (fmpe.hip); these
// kernels do everything else of a step on the device: the stage combinations y + h sum_j a_ij k_j (one pass over the
// operands instead of one axpy each), the error norm, the accept / reject decision, the step-size controller and the
// FSAL hand-over.  The host only ever reads the "reached t1" flag, one step late.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sbi_amd_fmpe.h"
#include "../../include/sbi_amd_nsf.h"

// state block (SBI_AMD_DOPRI5_STATE_FLOATS fp32 slots): doubles first, then what other kernels / the host read
//   double d[0] t, [1] h, [2] t_end, [3] direction, [4] atol, [5] rtol, [6] tiny,
//          d[7] signed step of the current attempt in double (t advances by THIS: the fp32 copy the stage kernels
//          use may round past t_end on the last step)                                              = floats 0..15
//   float  f[16] hs (signed step of the CURRENT attempt), f[17..22] stage times t + c_i hs (i = 2..7),
//          f[23] t (fp32 copy), f[24] finished flag (1.0 once |t_end - t| <= tiny),
//          f[25] accepted steps, f[26] rejected steps, f[27] last error ratio,
//          f[28] 1.0 if the CURRENT attempt reaches t_end when accepted (the host then waits for it instead of
//          enqueueing another attempt behind it)
#define ST_HS 16
#define ST_TT 17
#define ST_T 23
#define ST_FIN 24
#define ST_ACC 25
#define ST_REJ 26
#define ST_RATIO 27
#define ST_LAST 28
#define ODE_BLOCKS 256

__constant__ double DP_C[7] = {0.0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
__constant__ float DP_A[7][6] = {
    {0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
    {1.f / 5, 0.f, 0.f, 0.f, 0.f, 0.f},
    {3.f / 40, 9.f / 40, 0.f, 0.f, 0.f, 0.f},
    {44.f / 45, -56.f / 15, 32.f / 9, 0.f, 0.f, 0.f},
    {19372.f / 6561, -25360.f / 2187, 64448.f / 6561, -212.f / 729, 0.f, 0.f},
    {9017.f / 3168, -355.f / 33, 46732.f / 5247, 49.f / 176, -5103.f / 18656, 0.f},
    {35.f / 384, 0.f, 500.f / 1113, 125.f / 192, -2187.f / 6784, 11.f / 84},
};
// b5 - b4 for k1 .. k7
__constant__ float DP_E[7] = {35.f / 384 - 5179.f / 57600, 0.f, 500.f / 1113 - 7571.f / 16695,
                              125.f / 192 - 393.f / 640, -2187.f / 6784 + 92097.f / 339200,
                              11.f / 84 - 187.f / 2100, -1.f / 40};

__device__ __forceinline__ void dp_set_step(float* st) {
  double* d = reinterpret_cast<double*>(st);
  double rem = d[2] - d[0];
  rem = rem < 0 ? -rem : rem;
  const double hh = d[1] < rem ? d[1] : rem;         // 0 once t has reached t_end: the next attempt is a no-op
  const double hs = d[3] * hh;
  d[7] = hs;
  st[ST_HS] = (float)hs;
#pragma unroll
  for (int i = 1; i < 7; ++i) st[ST_TT + i - 1] = (float)(d[0] + hs * DP_C[i]);
  st[ST_T] = (float)d[0];
  st[ST_FIN] = rem <= d[6] ? 1.f : 0.f;
  st[ST_LAST] = (rem > d[6] && d[1] >= rem) ? 1.f : 0.f;
}

__global__ void dopri5_init_kernel(float* st, double t0, double t1, double first_step, double atol, double rtol) {
  if (threadIdx.x || blockIdx.x) return;
  double* d = reinterpret_cast<double*>(st);
  const double span = t1 >= t0 ? t1 - t0 : t0 - t1;
  d[0] = t0; d[1] = first_step < span ? first_step : span; d[2] = t1; d[3] = t1 >= t0 ? 1.0 : -1.0;
  d[4] = atol; d[5] = rtol; d[6] = 1e-12 * (span > 1.0 ? span : 1.0); d[7] = 0.0;
  st[ST_ACC] = 0.f; st[ST_REJ] = 0.f; st[ST_RATIO] = 0.f;
  dp_set_step(st);
}

struct DpK {
  const float* k[7];
};

// yi = y + hs * sum_{j < i} A[i][j] k_j     (i = 1 .. 6: the state handed to stage i + 1)
__global__ void __launch_bounds__(256) dopri5_stage_kernel(const float* __restrict__ y, DpK ks, int i,
                                                           const float* __restrict__ st, float* __restrict__ yi,
                                                           long long n) {
  const float hs = st[ST_HS];
  float a[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) a[j] = DP_A[i][j];
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j < i && a[j] != 0.f) s += a[j] * ks.k[j][e];
    yi[e] = y[e] + hs * s;
  }
}

// partials[b] = sum over this block's elements of (hs * sum_j E_j k_j / (atol + rtol max(|y|, |y5|)))^2
__global__ void __launch_bounds__(256) dopri5_err_kernel(const float* __restrict__ y, const float* __restrict__ y5,
                                                         DpK ks, const float* __restrict__ st,
                                                         double* __restrict__ partials, long long n) {
  const double* d = reinterpret_cast<const double*>(st);
  const float hs = st[ST_HS], atol = (float)d[4], rtol = (float)d[5];
  float e_[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) e_[j] = DP_E[j];
  double acc = 0.0;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j != 1) s += e_[j] * ks.k[j][e];
    const float sc = atol + rtol * fmaxf(fabsf(y[e]), fabsf(y5[e]));
    const float r = hs * s / sc;
    acc += (double)r * (double)r;
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

// every block: error ratio from the partials (same order everywhere) -> accept; accepted: y <- y5, k1 <- k7 (FSAL).
// block 0 then advances (t, h), prepares the next attempt's scalars and the finished flag.  No block reads the
// state here (the decision needs only the partials), so the update races with nothing.
__global__ void __launch_bounds__(256) dopri5_control_kernel(float* __restrict__ y, const float* __restrict__ y5,
                                                             float* __restrict__ k1, const float* __restrict__ k7,
                                                             const double* __restrict__ partials, float* st,
                                                             long long n) {
  __shared__ double red[256];
  red[threadIdx.x] = threadIdx.x < ODE_BLOCKS ? partials[threadIdx.x] : 0.0;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const double ratio = sqrt(red[0] / (double)n);
  const bool accept = ratio <= 1.0;            // NaN -> reject (and shrink below)
  if (accept)
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (long long)gridDim.x * blockDim.x) {
      y[e] = y5[e];
      k1[e] = k7[e];
    }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double* d = reinterpret_cast<double*>(st);
    const double hs = d[7];
    const double hh = hs < 0 ? -hs : hs;
    if (accept) { d[0] += hs; st[ST_ACC] += 1.f; } else { st[ST_REJ] += 1.f; }
    double factor;
    if (!(ratio == ratio)) factor = 0.2;
    else if (ratio <= 0.0) factor = 5.0;
    else {
      factor = 0.9 * pow(ratio, -0.2);
      factor = factor > 5.0 ? 5.0 : (factor < 0.2 ? 0.2 : factor);
    }
    if (hh > 0.0) d[1] = hh * factor;          // (a no-op attempt after the end leaves h alone)
    st[ST_RATIO] = (float)ratio;
    dp_set_step(st);
  }
}

static int ode_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

extern "C" int sbi_amd_dopri5_init(float* state, double t0, double t1, double first_step, double atol, double rtol,
                                   void* stream) {
  if (!state || !(first_step > 0.0) || !(atol >= 0.0) || !(rtol >= 0.0)) return SBI_AMD_E_BADARG;
  hipLaunchKernelGGL(dopri5_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, t0, t1, first_step, atol,
                     rtol);
  return (int)hipGetLastError();
}

extern "C" int sbi_amd_dopri5_stage(const float* y, const float* const* k, int32_t stage, const float* state,
                                    float* y_stage, int64_t n, void* stream) {
  if (!y || !k || !state || !y_stage || stage < 1 || stage > 6 || n < 1) return SBI_AMD_E_BADARG;
  DpK ks;
  for (int j = 0; j < 7; ++j) ks.k[j] = j < stage ? k[j] : nullptr;
  for (int j = 0; j < stage; ++j)
    if (!ks.k[j]) return SBI_AMD_E_BADARG;
  hipLaunchKernelGGL(dopri5_stage_kernel, dim3(ode_grid(n)), dim3(256), 0, (hipStream_t)stream, y, ks, (int)stage,
                     state, y_stage, (long long)n);
  return (int)hipGetLastError();
}

extern "C" int sbi_amd_dopri5_finish(float* y, const float* y5, const float* const* k, float* state,
                                     double* scratch, int64_t n, void* stream) {
  if (!y || !y5 || !k || !state || !scratch || n < 1) return SBI_AMD_E_BADARG;
  DpK ks;
  for (int j = 0; j < 7; ++j) {
    if (!k[j]) return SBI_AMD_E_BADARG;
    ks.k[j] = k[j];
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dopri5_err_kernel, dim3(ODE_BLOCKS), dim3(256), 0, st, (const float*)y, y5, ks,
                     (const float*)state, scratch, (long long)n);
  hipLaunchKernelGGL(dopri5_control_kernel, dim3(ode_grid(n)), dim3(256), 0, st, y, y5, const_cast<float*>(k[0]),
                     k[6], (const double*)scratch, state, (long long)n);
  return (int)hipGetLastError();
}
