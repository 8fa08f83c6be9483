// fp16 types exist only so that the half branches of the reference's templates parse; they are never executed
// (AT_DISPATCH below instantiates float only).
#pragma once
#include "cuda_runtime.h"
struct __half {
  float v;
  __half() = default;
  __half(float f) : v(f) {}
  operator float() const { return v; }
};
struct __half2 { __half x, y; };
inline __half2 operator+(__half2 a, __half2 b) { return {__half((float)a.x + (float)b.x), __half((float)a.y + (float)b.y)}; }
