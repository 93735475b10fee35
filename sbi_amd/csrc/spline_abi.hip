// spline_abi.hip -- the rational-quadratic spline of the NSF coupling transform as a standalone C-ABI entry point
// (SURVEY.md 8b "spline_coupling_fwd / inv"): the SAME device routine the flow kernels run (nsf_device.h,
// rq_spline_pair: one (row) task per lane pair, widths on lanes 0-31, heights on lanes 32-63), fed from global memory.
// Restates nflows 0.14 transforms/splines/rational_quadratic.py::unconstrained_rational_quadratic_spline
// (tails="linear") as sbi calls it (flow.py:425-432: tail_bound, default minima; coupling.py: widths / heights logits
// divided by sqrt(hidden_features) -- `logit_scale`).  Test hook and binder convenience; the flow kernels never call it.
#include <hip/hip_runtime.h>
#include <math.h>
#include "nsf_device.h"
#include "../../include/sbi_amd_nsf.h"

struct SplineConst {      // the fields rq_spline_pair reads from its PL argument
  float B, min_w, min_h, min_d, inv_sqrt_h, one_minus_kw, one_minus_kh, d_const;
  int ablate;
};

template <int K, bool INV>
__global__ void __launch_bounds__(256)
rq_spline_kernel(const SplineConst c, const float* __restrict__ params, const float* __restrict__ inputs, long long n,
                 float* __restrict__ outputs, float* __restrict__ logabsdet) {
  constexpr int P = 3 * K - 1;
  __shared__ float stage[4][32][P + 1];              // one wave = 32 tasks: both lanes of a pair read the task's P logits
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int task = lane & 31, part = lane >> 5;
  const long long row = ((long long)blockIdx.x * 4 + wave) * 32 + task;
  const long long src = row < n ? row : n - 1;       // idle pairs recompute the last task and drop the result
  for (int k = part; k < P; k += 2) stage[wave][task][k] = params[src * P + k];
  wave_lds_fence();
  float y, ld;
  rq_spline_pair<K, INV>(stage[wave][task], inputs[src], c, part, y, ld);
  if (row < n && part == 0) {
    outputs[row] = y;
    if (logabsdet) logabsdet[row] = ld;     // (inverse direction: nflows' `-logabsdet` of the forward map = log|d output / d input| here too)
  }
}

template <int K>
static void launch_spline(bool inv, const SplineConst& c, const float* params, const float* inputs, int64_t n,
                          float* outputs, float* logabsdet, hipStream_t st) {
  const unsigned grid = (unsigned)((n + 127) / 128);
  if (inv)
    hipLaunchKernelGGL((rq_spline_kernel<K, true>), dim3(grid), dim3(256), 0, st, c, params, inputs, (long long)n, outputs,
                       logabsdet);
  else
    hipLaunchKernelGGL((rq_spline_kernel<K, false>), dim3(grid), dim3(256), 0, st, c, params, inputs, (long long)n, outputs,
                       logabsdet);
}

// params (n, 3K-1): [K width logits | K height logits | K-1 interior derivative pre-activations]; inputs (n);
// outputs (n); logabsdet (n, optional): log|d output / d input| of the direction that was run.
extern "C" int sbi_amd_rq_spline(int32_t num_bins, int32_t inverse, float tail_bound, float min_bin_width,
                                 float min_bin_height, float min_derivative, float logit_scale, const float* params,
                                 const float* inputs, int64_t n, float* outputs, float* logabsdet, void* stream) {
  if (n == 0) return 0;
  if (!params || !inputs || !outputs || n < 0 || !(tail_bound > 0.f) || !(logit_scale > 0.f)) return SBI_AMD_E_BADARG;
  if (min_bin_width * num_bins > 1.0f || min_bin_height * num_bins > 1.0f) return SBI_AMD_E_BADARG;
  SplineConst c;
  c.B = tail_bound;
  c.min_w = min_bin_width;
  c.min_h = min_bin_height;
  c.min_d = min_derivative;
  c.inv_sqrt_h = logit_scale;
  c.one_minus_kw = (float)(1.0 - (double)min_bin_width * num_bins);
  c.one_minus_kh = (float)(1.0 - (double)min_bin_height * num_bins);
  c.d_const = (float)log(exp(1.0 - (double)min_derivative) - 1.0);
  c.ablate = 0;
  hipStream_t st = (hipStream_t)stream;
  switch (num_bins) {
    case 4: launch_spline<4>(inverse != 0, c, params, inputs, n, outputs, logabsdet, st); break;
    case 5: launch_spline<5>(inverse != 0, c, params, inputs, n, outputs, logabsdet, st); break;
    case 8: launch_spline<8>(inverse != 0, c, params, inputs, n, outputs, logabsdet, st); break;
    case 10: launch_spline<10>(inverse != 0, c, params, inputs, n, outputs, logabsdet, st); break;
    case 16: launch_spline<16>(inverse != 0, c, params, inputs, n, outputs, logabsdet, st); break;
    default: return SBI_AMD_E_UNSUPPORTED;
  }
  return (int)hipGetLastError();
}
