// Table-gradient scatter of the wrapped (hashed / tiled) levels WITHOUT per-sample device atomics (r04): bin, then reduce in LDS.
//
// Why.  Device-memory fp32 atomics execute at the memory side on MI355X (~20 G 16-byte granules per second, measured r02), and
// a 128 x 128 x 128 render puts 16.8 M corner contributions into every level (the reference field, network_grid.py:63: 16 levels,
// 2^16 rows each from level 3 on, `tiled`): on the levels whose cell is smaller than the footprint of an 8 x 8 ray patch the LDS
// cache of k_ngp_scatter finds nothing to merge, and the backward was bound by the number of atomic granules (k_ngp_scatter_fine
// 1.67 ms + the upper levels of k_ngp_scatter ~1.3 ms, alone).  Globally, though, every table row is hit ~256 times per render.  So:
//   k_ngp_bin         one workgroup = a tile of 1024 samples, four levels at a time: the contributions of an x-corner PAIR (two table
//                     rows, 2 x 2 values; both rows lie in one BUCKET of 1024 consecutive rows but for ~1 pair in 1024) are ranked
//                     inside the bucket by a returning LDS atomic, the workgroup reserves a contiguous run per touched bucket with
//                     ONE returning device atomic on the bucket's cursor, and the 16-byte entries are stored there (plain stores:
//                     the runs of a tile are assembled in its XCD's L2; the kernel is bound by the number of store instructions,
//                     all of them divergent -- pairs halve it);
//   k_ngp_bin_reduce  one workgroup per bucket: its entries stream in coalesced, accumulate in a 16 KB LDS slice of DOUBLES (ds_add_f64:
//                     fp32 LDS atomics retire 0.33 lane-operations per clock and CU on gfx950, fp64 ones 2.3 -- sf_dev.h),
//                     and the slice is added to the gradient table with plain vector read-modify-writes -- the workgroup is the
//                     only writer of those rows while it runs (stream order; the cached-level kernel owns other levels).
// Device atomics per level: one per (tile, touched bucket), <= 64 per 4096 pairs, instead of one granule per pair.  A bucket that
// overflows its capacity (a skewed `tiled` level, adversarial input) falls back to direct atomics for the overflow: always correct.
// 16 KB of LDS each: both kernels run beside k_ngp_field_bwd_mfma's 113 KB on the same CU.
// Reference semantics: external/gridencoder/src/gridencoder.cu:203-262 (kernel_grid_backward: atomicAdd per corner and channel);
// values equal ngp_scatter (ngp_device.h) up to fp32 summation order, which atomics never fixed either.
#pragma once
#include "sf_dev.h"
#include "ngp_device.h"

#ifndef SB_ROWS_LOG
#define SB_ROWS_LOG 10          // measured on the reference field (2^16-row `tiled` levels), render fwd + bwd: 2048 / 1024 / 512 rows = 4.27 / 3.90 / 4.13 ms
#endif
#define SB_ROWS (1u << SB_ROWS_LOG)   // table rows per bucket
static_assert(SB_ROWS_LOG <= 10, "the entry word keeps row1 - row0 + 1024 in 11 bits");
#define SB_MAX_BUCKETS (1u << (19 - SB_ROWS_LOG))            // per level: hsize <= 2^19, the reference's log2_hashmap_size (larger tables keep k_ngp_scatter_fine)
#define SB_THREADS 1024

#ifdef SF_HOST_EMU
static inline uint32_t sb_lds_inc(uint32_t* p) { return __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED); }
static inline uint32_t sb_reserve(uint32_t* p, uint32_t n) { return __atomic_fetch_add(p, n, __ATOMIC_RELAXED); }
#else
SF_DEV uint32_t sb_lds_inc(uint32_t* p) { return atomicAdd(p, 1u); }
SF_DEV uint32_t sb_reserve(uint32_t* p, uint32_t n) { return atomicAdd(p, n); }
#endif

struct SBArgs {
  NgpLevels lv;
  float bound;
  const float* rays_o; const float* rays_d; const float* aabb; const float* z_s;   // pre-offset to this launch's first ray
  const float* dfeat;          // level-major [L][P_stride][2], this launch's points at p_off
  float* gtable;
  uint32_t* cursor;            // [buckets]: entries reserved so far (zero before the first k_ngp_bin of a reduce round)
  f32x4* ent;                  // [buckets][cap] 16-byte entries {word, a, b, ratio} (sb_pack below)
  uint32_t P, T2, first_level, P_stride, p_off, cap;
  uint32_t bucket0[NGP_MAX_LEVELS + 1];      // first bucket of level l (levels below first_level hold none)
};

// One 16-byte entry per x-corner pair.  The four contributions of a pair are the outer product (w0, w1) x (dF0, dF1): the entry keeps
// the products of the corner with the LARGER weight (a, b) and the weight ratio <= 1; the reducer rebuilds the other corner's as a * ratio,
// b * ratio (two roundings instead of one: <= 2 ulp on the smaller contribution).  word = row0 | (row1 - row0 + 1024) << 20 | swap << 31
// (swap: (a, b) belong to row1); |row1 - row0| < SB_ROWS because both rows lie in the entry's bucket.
SF_DEV f32x4 sb_pack(uint32_t r0, uint32_t r1, float w0, float w1, float d0, float d1) {
  const bool swap = w1 > w0;
  const float wh = swap ? w1 : w0, wl = swap ? w0 : w1;
  const uint32_t word = r0 | ((r1 - r0 + 1024u) << 20) | (swap ? 0x80000000u : 0u);
  f32x4 e;
  e[0] = __builtin_bit_cast(float, word);
  e[1] = SF_MUL(wh, d0);
  e[2] = SF_MUL(wh, d1);
  e[3] = wh > 0.0f ? SF_DIV(wl, wh) : 0.0f;
  return e;
}

// Levels are taken SB_G at a time: the device atomics of a group go out together (one per thread), and a group's feature-gradient
// rows are one batch of loads.  The ranks of a group wait in registers (16 bits each); the cells are computed again for the stores.
#define SB_G 4
SF_DEV bool sb_z_dropped(const NgpLevels& lv, uint32_t l) {
  const uint32_t step = lv.resolution[l] + 1;
  return lv.gridtype == 1 && (uint64_t)step * step > lv.hsize[l] && step <= lv.hsize[l];
}
SF_KERNEL(SB_THREADS) void k_ngp_bin(SBArgs a) {
  SF_SHARED uint32_t cnt[SB_G][SB_MAX_BUCKETS];
  SF_SHARED uint32_t base[SB_G][SB_MAX_BUCKETS];
  const uint32_t tid = threadIdx.x;
  for (uint32_t b = tid; b < SB_G * SB_MAX_BUCKETS; b += SB_THREADS) (&cnt[0][0])[b] = 0;
  sf_sync();
  float box[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) box[i] = a.aabb[i];
  const f32x2* df2 = reinterpret_cast<const f32x2*>(a.dfeat);
  for (uint32_t t0 = blockIdx.x * SB_THREADS; t0 < a.P; t0 += gridDim.x * SB_THREADS) {      // uniform trip count per workgroup
    const uint32_t p = t0 + tid;
    bool inside = false;
    float x01[3] = {0.f, 0.f, 0.f};
    if (p < a.P) {
      const uint32_t n = p / a.T2;
      const float o[3] = {a.rays_o[n * 3], a.rays_o[n * 3 + 1], a.rays_o[n * 3 + 2]};
      const float d[3] = {a.rays_d[n * 3], a.rays_d[n * 3 + 1], a.rays_d[n * 3 + 2]};
      float x[3];
      ngp_point(o, d, a.z_s[p], box, x);
      inside = ngp_unit(x, a.bound, x01);
    }
    for (uint32_t l0 = a.first_level; l0 < a.lv.L; l0 += SB_G) {
      f32x2 d[SB_G];
      uint32_t rk[SB_G][2];                                              // ranks of the 4 corner pairs, two per word
#pragma unroll
      for (int g = 0; g < SB_G; ++g) {                                   // the group's gradient rows in one batch of loads
        d[g] = f32x2{0.f, 0.f};
        if (inside && l0 + g < a.lv.L) d[g] = df2[(size_t)(l0 + g) * a.P_stride + a.p_off + p];
      }
#pragma unroll
      for (int g = 0; g < SB_G; ++g) {
        const uint32_t l = l0 + g;
        if (d[g][0] != 0.0f || d[g][1] != 0.0f) {                        // outside points / dead samples contribute nothing
          const int np = sb_z_dropped(a.lv, l) ? 2 : 4;                  // corners i and i + 4 share their row on z-dropped levels
          NgpCell c;
          ngp_cell(a.lv, l, x01, c);
          uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < np) r[j] = sb_lds_inc(&cnt[g][c.row[2 * j] >> SB_ROWS_LOG]);
          rk[g][0] = r[0] | (r[1] << 16);                                // <= 4 * SB_THREADS pairs per tile and level
          rk[g][1] = r[2] | (r[3] << 16);
        }
      }
      sf_sync();
      for (uint32_t q = tid; q < SB_G * SB_MAX_BUCKETS; q += SB_THREADS) {
        const uint32_t g = q / SB_MAX_BUCKETS, b = q - g * SB_MAX_BUCKETS, l = l0 + g;
        if (l < a.lv.L && b < a.bucket0[l + 1] - a.bucket0[l]) {
          const uint32_t k = cnt[g][b];
          if (k) { base[g][b] = sb_reserve(&a.cursor[a.bucket0[l] + b], k); cnt[g][b] = 0; }
        }
      }
      sf_sync();
#pragma unroll
      for (int g = 0; g < SB_G; ++g) {
        const uint32_t l = l0 + g;
        if (d[g][0] != 0.0f || d[g][1] != 0.0f) {
          const bool zd = sb_z_dropped(a.lv, l);
          const int np = zd ? 2 : 4;
          NgpCell c;
          ngp_cell(a.lv, l, x01, c);
          if (zd) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c.w[i] = SF_ADD(c.w[i], c.w[i + 4]);
          }
          float* tab = a.gtable + (size_t)a.lv.offset[l] * 2;
          const uint32_t b0 = a.bucket0[l];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j < np) {
              const uint32_t r0 = c.row[2 * j], r1 = c.row[2 * j + 1], b = r0 >> SB_ROWS_LOG;
              const uint32_t pos = base[g][b] + ((rk[g][j >> 1] >> (16 * (j & 1))) & 0xffffu);
              const float w0 = c.w[2 * j], w1 = c.w[2 * j + 1];
              const bool full = pos >= a.cap;                            // bucket full: the pair goes straight to the table
              const bool split = (r1 >> SB_ROWS_LOG) != b;               // second row in another bucket (~1 pair in 1024): direct adds
              if (full || split) {
                sf_global_add(tab + (size_t)r1 * 2, SF_MUL(w1, d[g][0]));
                sf_global_add(tab + (size_t)r1 * 2 + 1, SF_MUL(w1, d[g][1]));
              }
              if (full) {
                sf_global_add(tab + (size_t)r0 * 2, SF_MUL(w0, d[g][0]));
                sf_global_add(tab + (size_t)r0 * 2 + 1, SF_MUL(w0, d[g][1]));
              } else {
                a.ent[(size_t)(b0 + b) * a.cap + pos] = split ? sb_pack(r0, r0, w0, 0.0f, d[g][0], d[g][1]) : sb_pack(r0, r1, w0, w1, d[g][0], d[g][1]);
              }
            }
          }
        }
      }
      // base[] of this group is read above and rewritten only behind the next group's first barrier; cnt[] is zero again
    }
  }
}

struct SBRArgs {
  NgpLevels lv;
  float* gtable;
  uint32_t* cursor; const f32x4* ent;
  uint32_t first_level, cap;
  uint32_t bucket0[NGP_MAX_LEVELS + 1];
};

// one workgroup per bucket; leaves the bucket's cursor at zero for the next round.  Every global access is issued in batches of
// SBR_U per thread before its first use: with one load in flight per thread (the first version) the entries of a bucket were
// 115 dependent round trips per thread -- 1.37 ms per launch for 400 MB.
#define SBR_THREADS (SB_ROWS < 1024u ? (int)SB_ROWS : 1024)
#define SBR_U 4
SF_KERNEL(SBR_THREADS) void k_ngp_bin_reduce(SBRArgs a) {
  SF_SHARED double acc[SB_ROWS * 2];
  const uint32_t gb = blockIdx.x, tid = threadIdx.x;
  uint32_t n = a.cursor[gb];
  if (n == 0) return;                                                    // (uniform: every thread read the same word)
  if (n > a.cap) n = a.cap;
  uint32_t l = a.first_level;
  while (l + 1 < a.lv.L && gb >= a.bucket0[l + 1]) ++l;
  const f32x4* ent = a.ent + (size_t)gb * a.cap;
  f32x4 v[SBR_U];
  auto fetch = [&](uint32_t i0) {
#pragma unroll
    for (int u = 0; u < SBR_U; ++u) {
      uint32_t i = i0 + u * SBR_THREADS + tid;
      if (i > n - 1) i = n - 1;                                          // unconditional loads (clamped): one straight-line block
      v[u] = ent[i];
    }
  };
  fetch(0);                                                              // the first batch is in flight while the slice is zeroed
  for (uint32_t k = tid; k < SB_ROWS * 2; k += SBR_THREADS) acc[k] = 0.0;
  sf_sync();
  if (tid == 0) a.cursor[gb] = 0;
  for (uint32_t i0 = 0; i0 < n; i0 += SBR_U * SBR_THREADS) {
    f32x4 vc[SBR_U];
#pragma unroll
    for (int u = 0; u < SBR_U; ++u) vc[u] = v[u];
    if (i0 + SBR_U * SBR_THREADS < n) fetch(i0 + SBR_U * SBR_THREADS);   // next batch under this batch's LDS atomics
#pragma unroll
    for (int u = 0; u < SBR_U; ++u) {
      if (i0 + u * SBR_THREADS + tid < n) {
        const uint32_t word = __builtin_bit_cast(uint32_t, vc[u][0]);
        const uint32_t q0 = word & (SB_ROWS - 1), q1 = q0 + ((word >> 20) & 2047u) - 1024u;      // both rows of the pair lie in this bucket
        const uint32_t qh = (word >> 31) ? q1 : q0, ql = (word >> 31) ? q0 : q1;
        sf_lds_add_f64(&acc[2 * qh], (double)vc[u][1]);
        sf_lds_add_f64(&acc[2 * qh + 1], (double)vc[u][2]);
        if (vc[u][3] != 0.0f) {
          sf_lds_add_f64(&acc[2 * ql], (double)(vc[u][1] * vc[u][3]));
          sf_lds_add_f64(&acc[2 * ql + 1], (double)(vc[u][2] * vc[u][3]));
        }
      }
    }
  }
  sf_sync();
  const uint32_t row0 = (gb - a.bucket0[l]) << SB_ROWS_LOG;
  const uint32_t hs = a.lv.hsize[l];
  const uint32_t nrows = hs - row0 < SB_ROWS ? hs - row0 : SB_ROWS;
  f32x2* tab = reinterpret_cast<f32x2*>(a.gtable + (size_t)a.lv.offset[l] * 2) + row0;
  constexpr int RPT = SB_ROWS / SBR_THREADS;                             // rows per thread: all their loads first, then the stores
  f32x2 t[RPT];
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    uint32_t q = u * SBR_THREADS + tid;
    if (q > nrows - 1) q = nrows - 1;
    t[u] = tab[q];
  }
#pragma unroll
  for (int u = 0; u < RPT; ++u) {
    const uint32_t q = u * SBR_THREADS + tid;
    const f32x2 s = {(float)acc[2 * q], (float)acc[2 * q + 1]};
    if (q < nrows && (s[0] != 0.0f || s[1] != 0.0f)) tab[q] = f32x2{t[u][0] + s[0], t[u][1] + s[1]};
  }
}
