// Host emulation of the HIP execution model for kernel-LOGIC tests on a GPU-less machine (test infrastructure only).
//
// A kernel written against csrc/sf_dev.h compiles unchanged with the host clang (-DSF_HOST_EMU): every thread of a
// workgroup is one OS thread, __syncthreads() and the wave-wide primitives (shuffles, MFMA) are rendezvous points,
// workgroups run one after the other.  `__shared__` becomes function-local static storage (valid because only one
// workgroup is alive at a time).  The MFMA emulation implements the v_mfma_f32_16x16x32_bf16 operand layout
//   A: lane l holds A[m = l & 15][k = 8 * (l >> 4) + j], j < 8;   B: lane l holds B[k = 8 * (l >> 4) + j][n = l & 15]
//   D: lane l holds D[m = 4 * (l >> 4) + r][n = l & 15], r < 4
// which is the layout the shipped kernels (k_conv_igemm) were validated with on hardware.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace hipemu {

struct Dim3 { unsigned x, y, z; };

// Barrier whose participant count shrinks when a thread retires (kernel `return` before a later barrier).
class Barrier {
 public:
  void reset(int n) { live_ = n; waiting_ = 0; gen_ = 0; }
  void wait() {
    std::unique_lock<std::mutex> lk(mu_);
    const unsigned g = gen_;
    if (++waiting_ == live_) { waiting_ = 0; ++gen_; cv_.notify_all(); return; }
    cv_.wait(lk, [&] { return gen_ != g; });
  }
  void retire() {
    std::unique_lock<std::mutex> lk(mu_);
    --live_;
    if (live_ > 0 && waiting_ == live_) { waiting_ = 0; ++gen_; cv_.notify_all(); }
  }
 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int live_ = 0, waiting_ = 0;
  unsigned gen_ = 0;
};

struct WaveState {
  Barrier bar;
  uint32_t xchg[64][8];        // per-lane exchange slots (up to 32 bytes)
};

struct BlockState {
  Barrier bar;
  std::vector<WaveState> waves;
  std::vector<char> dyn_smem;
};

extern thread_local Dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockState* t_block;
extern thread_local WaveState* t_wave;
extern thread_local int t_lane;

inline void syncthreads() { t_block->bar.wait(); }
inline char* dyn_smem() { return t_block->dyn_smem.data(); }

template <class T>
inline T shfl_xor(T v, int mask) {
  static_assert(sizeof(T) <= 32, "exchange slot too small");
  WaveState* w = t_wave;
  memcpy(w->xchg[t_lane], &v, sizeof(T));
  w->bar.wait();
  T r;
  memcpy(&r, w->xchg[(t_lane ^ mask) & 63], sizeof(T));
  w->bar.wait();
  return r;
}

// Runs `body` for every thread of every workgroup.
void launch(unsigned grid, unsigned block, size_t dyn_smem_bytes, const std::function<void()>& body);

}  // namespace hipemu

#define threadIdx hipemu::t_threadIdx
#define blockIdx hipemu::t_blockIdx
#define blockDim hipemu::t_blockDim
#define gridDim hipemu::t_gridDim

#ifdef HIPEMU_IMPLEMENTATION
namespace hipemu {
thread_local Dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockState* t_block;
thread_local WaveState* t_wave;
thread_local int t_lane;

void launch(unsigned grid, unsigned block, size_t dyn_smem_bytes, const std::function<void()>& body) {
  const unsigned n_waves = (block + 63) / 64;
  for (unsigned b = 0; b < grid; ++b) {
    BlockState bs;
    bs.waves = std::vector<WaveState>(n_waves);
    bs.bar.reset((int)block);
    bs.dyn_smem.assign(dyn_smem_bytes + 64, 0);
    for (unsigned w = 0; w < n_waves; ++w) {
      const unsigned lanes = (w + 1) * 64 <= block ? 64 : block - w * 64;
      bs.waves[w].bar.reset((int)lanes);
    }
    std::vector<std::thread> th;
    th.reserve(block);
    for (unsigned t = 0; t < block; ++t) {
      th.emplace_back([&, t, b] {
        t_threadIdx = Dim3{t, 0, 0};
        t_blockIdx = Dim3{b, 0, 0};
        t_blockDim = Dim3{block, 1, 1};
        t_gridDim = Dim3{grid, 1, 1};
        t_block = &bs;
        t_wave = &bs.waves[t / 64];
        t_lane = (int)(t % 64);
        body();
        t_wave->bar.retire();
        bs.bar.retire();
      });
    }
    for (auto& x : th) x.join();
  }
}
}  // namespace hipemu
#endif
