// CPU harness for k_gemm_rows (sparsefusion_amd/csrc/gemm_rows.h): the kernel source runs on CPU threads (hip_emu.h).
#ifndef SF_HOST_EMU
#define SF_HOST_EMU
#endif
#define HIPEMU_IMPLEMENTATION
#include "hip_emu.h"
#include <algorithm>
using std::min;
using std::max;
#include "../../sparsefusion_amd/csrc/gemm_rows.h"

extern "C" void emu_gemm_rows(const float* x, const uint16_t* W, const float* bias, float* y, int M, int N, int K, int Kp, int ldx,
                              int ldy, int in_silu, int out_act) {
  GemmRowsArgs a{x, reinterpret_cast<const __bf16*>(W), bias, y, M, N, K, Kp, ldx, ldy, in_silu, out_act};
  hipemu::launch((unsigned)((N + 63) / 64), 256, 0, [&] { k_gemm_rows(a); });
}

// r06: the K-sliced form (N / 16 workgroups, the 4 waves split K by chunk)
extern "C" void emu_gemm_rows_ks(const float* x, const uint16_t* W, const float* bias, float* y, int M, int N, int K, int Kp, int ldx,
                                 int ldy, int in_silu, int out_act) {
  GemmRowsArgs a{x, reinterpret_cast<const __bf16*>(W), bias, y, M, N, K, Kp, ldx, ldy, in_silu, out_act};
  if (M <= 32 && K > 4 * GR_KC) hipemu::launch((unsigned)((N + 15) / 16), 512, GemmRowsKs<8>::LDS_BYTES, [&] { k_gemm_rows_ks<8>(a); });      // (run_gemv's rule)
  else hipemu::launch((unsigned)((N + 15) / 16), 256, GemmRowsKs<4>::LDS_BYTES, [&] { k_gemm_rows_ks<4>(a); });
}
