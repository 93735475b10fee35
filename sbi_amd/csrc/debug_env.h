// debug_env.h -- the timing / ablation hooks of the kernels are DEBUG aids and exist in -DNSF_DEBUG builds only
// (`python -m sbi_amd._build --debug` -> libsbi_amd_nsf_debug.so, selected with SBI_AMD_LIB; tools/timeline.py,
// tools/ablate.sh).  The shipped library reads no environment variable and carries neither cycle-counter stores nor
// run-time ablation switches: the accessors below are constant 0 and every use folds away.
// In a debug build each variable is read ONCE per process (not per launch) and, when set, announced loudly on stderr.
#pragma once
// Debug switches of the kernels: a field test in -DNSF_DEBUG builds, the constant 0 in the shipped library.
#ifdef NSF_DEBUG
#define NSF_DBG_ABL(field, bits) ((field) & (bits))
#else
#define NSF_DBG_ABL(field, bits) (0)
#endif
#ifdef NSF_DEBUG
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
#else
#define SBI_AMD_DEBUG_ENV(fn, name) \
  constexpr int fn() { return 0; }
#endif
SBI_AMD_DEBUG_ENV(sbi_amd_dbg_ablate, "SBI_AMD_ABLATE")
SBI_AMD_DEBUG_ENV(sbi_amd_dbg_timeline, "SBI_AMD_TIMELINE")
SBI_AMD_DEBUG_ENV(sbi_amd_dbg_fm_ablate, "SBI_AMD_FM_ABLATE")
SBI_AMD_DEBUG_ENV(sbi_amd_dbg_fm_timeline, "SBI_AMD_FM_TIMELINE")
