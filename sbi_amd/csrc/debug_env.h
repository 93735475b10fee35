// debug_env.h -- the timing / ablation hooks of the kernels are DEBUG aids: each environment variable is read
// ONCE per process (not per launch) and, when set, announced loudly on stderr -- a stray variable can no longer
// change results silently.
#pragma once
#include <stdio.h>
#include <stdlib.h>

inline int sbi_amd_debug_env_int(const char* name, int* cache, int* done) {
  if (!*done) {
    const char* a = getenv(name);
    *cache = a ? atoi(a) : 0;
    *done = 1;
    if (a && *cache)
      fprintf(stderr, "sbi_amd: WARNING: %s=%d is set: DEBUG / timing mode, kernel results may be INVALID\n", name,
              *cache);
  }
  return *cache;
}
#define SBI_AMD_DEBUG_ENV(fn, name)                       \
  inline int fn() {                                       \
    static int cache = 0, done = 0;                       \
    return sbi_amd_debug_env_int(name, &cache, &done);    \
  }
SBI_AMD_DEBUG_ENV(sbi_amd_dbg_ablate, "SBI_AMD_ABLATE")
SBI_AMD_DEBUG_ENV(sbi_amd_dbg_timeline, "SBI_AMD_TIMELINE")
SBI_AMD_DEBUG_ENV(sbi_amd_dbg_fm_ablate, "SBI_AMD_FM_ABLATE")
SBI_AMD_DEBUG_ENV(sbi_amd_dbg_fm_timeline, "SBI_AMD_FM_TIMELINE")
