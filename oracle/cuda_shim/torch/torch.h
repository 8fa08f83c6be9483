// The slice of ATen the reference's host wrappers touch: a tensor is (pointer, dtype, contiguity, "is_cuda").
#pragma once
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include "../cuda_fp16.h"

namespace at {
enum class ScalarType { Byte, Int, Float, Half, Double };
struct Half {
  float v;
  Half() = default;
  Half(float f) : v(f) {}
  operator float() const { return v; }
};
struct Device { bool cuda; bool is_cuda() const { return cuda; } };
struct Tensor {
  void* ptr = nullptr;
  ScalarType dtype = ScalarType::Float;
  bool contiguous = true, cuda = true;
  Device device() const { return {cuda}; }
  bool is_contiguous() const { return contiguous; }
  ScalarType scalar_type() const { return dtype; }
  template <class T> T* data_ptr() const { return reinterpret_cast<T*>(ptr); }
};
template <class T> using optional = std::optional<T>;
}  // namespace at

template <class... A>
inline std::string shim_cat(A&&... a) { std::ostringstream s; (void)std::initializer_list<int>{((s << a), 0)...}; return s.str(); }
#define TORCH_CHECK(cond, ...) do { if (!(cond)) throw std::runtime_error(shim_cat(__VA_ARGS__)); } while (0)

// float is the only dtype the reference path (and this build) uses: "scalar_t should always be float in use"
// (raymarching.cu:90).  Other dtypes raise instead of silently running an fp16 stand-in.
#define AT_DISPATCH_FLOATING_TYPES_AND_HALF(TYPE, NAME, ...)                                   \
  do {                                                                                           \
    if ((TYPE) != at::ScalarType::Float) throw std::runtime_error(std::string(NAME) + ": host reference build is float only"); \
    using scalar_t = float;                                                                      \
    __VA_ARGS__();                                                                               \
  } while (0)
