// Timing harness for maf_dw_kernel (csrc/maf_kernel.h): the weight-gradient GEMMs of one maf_rqs transform at the
// benchmark shape (65 536 rows, theta-dim 10, hidden 50 -> final 290 x 50, three 50 x 50, one 50 x 10, one 50 x 10)
// with the phases switched off one at a time.  build: hipcc --offload-arch=gfx950 -O3 -I../../sbi_amd/csrc -o maf_dw_bench maf_dw_bench.hip
#define MAF_MAIN_TU
#include "maf_kernel.h"
#include <cstdio>
#include <cstring>
#include <vector>
int main() {
  const long long n = 65536;
  const int D = 10, P = 29, PTW = 32, DP = D * PTW, H = 50, NB = 2;
  const int GW = (MAF_MAX_NB + 2) * MAF_AW, AWS = (MAF_MAX_NB + 1) * MAF_AW;
  float *GP, *ACT, *G, *CTX, *part;
  hipMalloc(&GP, n * DP * 4); hipMalloc(&ACT, n * AWS * 4); hipMalloc(&G, n * GW * 4); hipMalloc(&CTX, n * MAF_CW * 4);
  hipMemset(GP, 0, n * DP * 4); hipMemset(ACT, 0, n * AWS * 4); hipMemset(G, 0, n * GW * 4); hipMemset(CTX, 0, n * MAF_CW * 4);
  const int n_layer = H * D + H + H * 10 + H + NB * (H * H + H) + D * P * H + D * P;
  const int nchunks = (int)(n / MAF_DW_CHUNK);
  hipMalloc(&part, (size_t)nchunks * n_layer * 4);
  MafDwArgs d;
  __builtin_memset((void*)&d, 0, sizeof(d));
  int g = 0;
  const long long gts = n * 16;
  auto set = [&](int i, const float* Gp, int, const float* A, int lda, int out, int in, int group, int gpad, int kind) {
    d.lin[i].G = Gp; d.lin[i].gts = gts; d.lin[i].A = A; d.lin[i].lda = lda; d.lin[i].out = out; d.lin[i].in = in; d.lin[i].in_total = in; d.lin[i].col0 = 0;
    d.lin[i].group = group; d.lin[i].group_pad = gpad; d.lin[i].g_w = g; g += out * in; d.lin[i].g_b = g; g += out;
    d.lin[i].kind = kind;
  };
  set(0, GP, DP, ACT + 64 * NB, AWS, D * P, H, P, PTW, 3);
  set(1, G + 4 * 2 * gts, GW, ACT, AWS, H, H, H, 64, 2);
  set(2, G + 4 * 3 * gts, GW, ACT + 64, AWS, H, H, H, 64, 2);
  set(3, G, GW, CTX, MAF_CW, H, D, H, 64, 0);
  set(4, G + 4 * gts, GW, CTX + D, MAF_CW, H, 10, H, 64, 1);
  d.n = n; d.rows_per_chunk = MAF_DW_CHUNK; d.nchunks = nchunks; d.n_layer = n_layer; d.D = D; d.P = P; d.partial = part;
  const int lds = MAF_DW_LDS_BYTES;
  hipFuncSetAttribute((const void*)maf_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int abls[] = {0, 1, 2, 4, 8, 1 | 2, 1 | 2 | 4, 15};
  for (int abl : abls) {
    d.abl = abl;
    for (int only = -1; only < (abl == 0 ? 5 : 0); ++only) {      // abl 0: also each linear alone
      MafDwArgs dd = d;
      int np = 0;
      for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < 5; ++i) {
          if (only >= 0 && i != only) continue;
          const MafLin& L = d.lin[i];
          const int mtiles = ((L.out + L.group - 1) / L.group * L.group_pad + 15) / 16, cap = 4 * MAF_DW_MTW;
          for (int mt0 = 0; mt0 < mtiles; mt0 += cap) {
            const int mtn = mtiles - mt0 < cap ? mtiles - mt0 : cap;
            if ((mtn == cap) != (pass == 0)) continue;
            dd.lin[np] = L; dd.lin[np].mt0 = mt0; dd.lin[np].mtn = mtn; ++np;
          }
        }
      dim3 grid(nchunks, np);
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(maf_dw_kernel, grid, dim3(256), lds, 0, dd);
      hipEventRecord(e0);
      for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(maf_dw_kernel, grid, dim3(256), lds, 0, dd);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("abl %2d linear %2d: %7.1f us per launch\n", abl, only, ms * 100.f);
    }
  }
  return 0;
}
