// Shared host-side helpers for libsparsefusion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/sparsefusion_hip.h"
#include "sf_operand.h"

extern thread_local char sf_err_buf[512];

#define SF_FAIL(code, ...)                                   \
  do {                                                       \
    snprintf(sf_err_buf, sizeof(sf_err_buf), __VA_ARGS__);   \
    return (code);                                           \
  } while (0)

#define SF_CHECK_LAUNCH(name)                                             \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess)                                                \
      SF_FAIL(SF_ERR_LAUNCH, "%s: %s", (name), hipGetErrorString(e__));   \
  } while (0)

static inline uint32_t sf_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Number of workgroups that fills the chip for a grid-stride kernel:
// 256 CUs x 8 resident 256-thread blocks (cdna guide, Guideline 11).
static inline uint32_t sf_grid_cap(uint64_t want) {
  const uint64_t cap = 256ull * 8ull;
  return (uint32_t)(want < cap ? (want ? want : 1) : cap);
}
