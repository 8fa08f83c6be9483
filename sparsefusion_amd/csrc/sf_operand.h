// MFMA operand type of every conv / linear / attention GEMM of this library: bf16 (the default build, BASELINE configs[1]) or,
// with -DSF_OPERAND_F16=1, IEEE half (libsparsefusion_hip_f16.so, selected with SF_OPERAND=f16: BASELINE configs[4] "fp16
// UNet").  fp32 accumulation, fp32 residual stream, norms, softmax and PLMS arithmetic either way.  Same kernels, same tiling:
// v_mfma_f32_16x16x32_f16 has the shape and rate of v_mfma_f32_16x16x32_bf16.  The vector typedefs keep their names
// (bf16x8 = 8 operand values).
#pragma once
#ifndef SF_OPERAND_F16
#define SF_OPERAND_F16 0
#endif
#if SF_OPERAND_F16
typedef _Float16 sf_opnd;
#else
typedef __bf16 sf_opnd;
#endif
