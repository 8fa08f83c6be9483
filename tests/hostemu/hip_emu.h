// Host emulation of the HIP execution model for kernel-LOGIC tests on a GPU-less machine (test infrastructure only).
//
// A kernel written against csrc/sf_dev.h compiles unchanged with the host clang (-DSF_HOST_EMU).  Execution model (r04): every lane
// of a workgroup is a FIBER (hand-rolled x86-64 context switch) on the launching thread.  A lane runs until its next rendezvous -- a
// wave-wide primitive (shuffle, MFMA, sf_wave_sync) or __syncthreads() --; a wave runs AHEAD, through all its wave-wide rendezvous,
// until every live lane of it stands at __syncthreads() (or has returned), then the next wave runs; lanes and waves are visited in
// alternating order (ascending, then descending).  The schedule is deterministic and adversarial: a missing barrier shows as a
// wave reading what another has not written yet or has already overwritten -- on every run, not when thread timing happens to expose
// it.  (r01-r03: one OS thread per lane: a 512-thread workgroup spent its time in futex wake-ups -- the CPU suite ran 9 minutes, 17 of
// its 25 CPU-minutes in the kernel -- and cross-wave races were a matter of luck.)  Workgroups run one after the other.  `__shared__` becomes function-local static storage (valid because only one
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
#include <functional>
#include <vector>

namespace hipemu {

struct Dim3 { unsigned x, y, z; };

enum LaneState { LANE_READY = 0, LANE_WAIT_WAVE = 1, LANE_WAIT_BLOCK = 2, LANE_DONE = 3 };
void lane_yield(int state);                // the running lane parks in `state`; returns when the scheduler resumes it

// wave-wide rendezvous of the lanes (fibers) of one wave: kept under the name the kernels' emulation layer uses (t_wave->bar.wait())
struct WaveBarrier { void wait() { lane_yield(LANE_WAIT_WAVE); } };

struct WaveState {
  WaveBarrier bar;
  uint32_t xchg[64][8];        // per-lane exchange slots (up to 32 bytes)
};

struct BlockState {
  std::vector<WaveState> waves;
  std::vector<char> dyn_smem;
};

extern thread_local Dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockState* t_block;
extern thread_local WaveState* t_wave;
extern thread_local int t_lane;

inline void syncthreads() { lane_yield(LANE_WAIT_BLOCK); }
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
// void hipemu_switch(void** save_sp, void* load_sp): callee-saved registers + MXCSR / x87 control word on the stack, swap stacks
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n"
    ".globl hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  subq $8, %rsp\n  stmxcsr (%rsp)\n  fnstcw 4(%rsp)\n"
    "  movq %rsp, (%rdi)\n"
    "  movq %rsi, %rsp\n"
    "  ldmxcsr (%rsp)\n  fldcw 4(%rsp)\n  addq $8, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");

namespace hipemu {
thread_local Dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockState* t_block;
thread_local WaveState* t_wave;
thread_local int t_lane;

namespace {
constexpr size_t kStackBytes = 512 * 1024;
struct Lane { void* sp = nullptr; char* stack = nullptr; int state = LANE_READY; };
struct BlockRun {
  std::vector<Lane> lanes;                 // [wave * 64 + lane]
  void* sched_sp = nullptr;
  int cur = -1;
  const std::function<void()>* body = nullptr;
};
thread_local BlockRun* t_run = nullptr;

void lane_entry() {                        // first activation of a lane: runs the kernel body, then parks as DONE for good
  BlockRun* r = t_run;
  (*r->body)();
  r = t_run;
  r->lanes[r->cur].state = LANE_DONE;
  hipemu_switch(&r->lanes[r->cur].sp, r->sched_sp);
  fprintf(stderr, "hipemu: a finished lane was resumed\n");
  abort();
}

void run_block(BlockState& bs, unsigned block, unsigned grid, unsigned b, const std::function<void()>& body) {
  const unsigned n_waves = (block + 63) / 64;
  BlockRun run;
  run.lanes.resize((size_t)n_waves * 64);
  run.body = &body;
  t_run = &run;
  t_block = &bs;
  t_blockIdx = Dim3{b, 0, 0};
  t_blockDim = Dim3{block, 1, 1};
  t_gridDim = Dim3{grid, 1, 1};
  auto n_lanes = [&](unsigned w) { return (w + 1) * 64 <= block ? 64u : block - w * 64; };
  for (unsigned w = 0; w < n_waves; ++w)
    for (unsigned i = 0; i < 64; ++i) {
      Lane& L = run.lanes[w * 64 + i];
      if (i >= n_lanes(w)) { L.state = LANE_DONE; continue; }
      L.stack = static_cast<char*>(malloc(kStackBytes));
      if (!L.stack) { fprintf(stderr, "hipemu: out of memory for lane stacks\n"); abort(); }
      // initial frame hipemu_switch pops: [mxcsr | fcw][r15 r14 r13 r12 rbx rbp][return address = lane_entry]; at lane_entry's first
      // instruction rsp must be 8 mod 16 (as after a call)
      uintptr_t top = (reinterpret_cast<uintptr_t>(L.stack) + kStackBytes) & ~uintptr_t(15);
      uint64_t* f = reinterpret_cast<uint64_t*>(top - 72);
      const uint32_t mx = 0x1f80;
      const uint16_t cw = 0x037f;
      f[0] = 0;
      memcpy(reinterpret_cast<char*>(f), &mx, 4);
      memcpy(reinterpret_cast<char*>(f) + 4, &cw, 2);
      for (int k = 1; k <= 6; ++k) f[k] = 0;
      f[7] = reinterpret_cast<uint64_t>(&lane_entry);
      f[8] = 0;
      L.sp = f;
      L.state = LANE_READY;
    }
  // a wave runs until every live lane of it stands at __syncthreads() or has returned; false = nothing left to run in it
  auto run_wave = [&](unsigned w, bool forward) {
    t_wave = &bs.waves[w];
    for (;;) {
      bool ran = false;
      for (unsigned k = 0; k < 64; ++k) {
        const unsigned i = forward ? k : 63 - k;
        Lane& L = run.lanes[w * 64 + i];
        if (L.state != LANE_READY) continue;
        run.cur = (int)(w * 64 + i);
        t_lane = (int)i;
        t_threadIdx = Dim3{w * 64 + i, 0, 0};
        hipemu_switch(&run.sched_sp, L.sp);            // returns when the lane parks (lane_yield) or finishes
        ran = true;
      }
      forward = !forward;
      int n_wave = 0, n_block = 0;
      for (unsigned i = 0; i < 64; ++i) {
        const int st = run.lanes[w * 64 + i].state;
        n_wave += st == LANE_WAIT_WAVE;
        n_block += st == LANE_WAIT_BLOCK;
      }
      if (n_wave && n_block) {
        fprintf(stderr, "hipemu: wave %u of workgroup %u diverged across a rendezvous (%d lanes at a wave primitive, %d at __syncthreads)\n",
                w, b, n_wave, n_block);
        abort();
      }
      if (n_wave) {                                    // every live lane reached the wave-wide rendezvous: release, keep running this wave
        for (unsigned i = 0; i < 64; ++i)
          if (run.lanes[w * 64 + i].state == LANE_WAIT_WAVE) run.lanes[w * 64 + i].state = LANE_READY;
        continue;
      }
      (void)ran;
      return;                                          // all live lanes at __syncthreads(), or the wave has finished
    }
  };
  bool forward = true;
  for (;;) {
    for (unsigned k = 0; k < n_waves; ++k) run_wave(forward ? k : n_waves - 1 - k, forward);
    forward = !forward;
    size_t n_block = 0, n_done = 0;
    for (const Lane& L : run.lanes) { n_block += L.state == LANE_WAIT_BLOCK; n_done += L.state == LANE_DONE; }
    if (n_done == run.lanes.size()) break;
    if (n_block + n_done != run.lanes.size()) { fprintf(stderr, "hipemu: scheduler stuck in workgroup %u\n", b); abort(); }
    for (Lane& L : run.lanes)                          // every live lane of the workgroup reached __syncthreads(): release
      if (L.state == LANE_WAIT_BLOCK) L.state = LANE_READY;
  }
  for (Lane& L : run.lanes) free(L.stack);
  t_run = nullptr;
}
}  // namespace

void lane_yield(int state) {
  BlockRun* r = t_run;
  Lane& L = r->lanes[r->cur];
  L.state = state;
  hipemu_switch(&L.sp, r->sched_sp);
}

void launch(unsigned grid, unsigned block, size_t dyn_smem_bytes, const std::function<void()>& body) {
  const unsigned n_waves = (block + 63) / 64;
  for (unsigned b = 0; b < grid; ++b) {
    BlockState bs;
    bs.waves = std::vector<WaveState>(n_waves);
    bs.dyn_smem.assign(dyn_smem_bytes + 64, 0);
    run_block(bs, block, grid, b, body);
  }
}
}  // namespace hipemu
#endif
