// HBM-bound kernels of the CalciumGAN hot path: LayerNorm(+LeakyReLU),
// discriminator head, phase-unshuffle, WGAN-GP interpolation / penalty norm,
// bias gradients, Keras Adam, signal metrics.  All loads/stores are 16-byte
// (8 x bf16 or 4 x f32) per lane and row-contiguous; reductions use wavefront
// shuffles (64 lanes) then one atomic per wave/block.
#include "cg_common.h"

namespace {

constexpr int kThreads = 256;

inline unsigned grid1d(long long work, int per_block, long long cap = 1 << 20) {
  long long b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------
// Ordered two-stage reductions (the `ws` argument of the C ABI).  A reducing
// kernel whose blocks used to meet through f32 atomics -- which land in a
// different order every run -- instead leaves ONE partial row per block in the
// caller's workspace, and finish_cols_kernel adds the rows in a fixed order and
// STORES the result: bit-identical from run to run, no zeroed output needed.
// Grids are capped at kMaxParts blocks so that the workspace has a fixed size
// (cg_reduce_ws_elems()).
// ---------------------------------------------------------------------------
constexpr int kMaxParts = 2048;
constexpr long long kReduceWsElems = 4ll << 20;  // 16 MiB of f32

struct FinishArgs {
  const float* ws;
  int nparts, ncol;      // partial rows; columns per output
  long long pstride;     // floats between consecutive partial rows
  int cstride;           // floats between the outputs' sections inside a row
  int nout;
  float* out[3];
  int cvalid[3];         // columns actually stored per output
  float scale;
};

// block = 64 columns x 16 row classes: wave j adds rows j, j + 16, ... (fixed
// order), the sixteen sums meet through LDS in order 0..15
__global__ __launch_bounds__(1024) void finish_cols_kernel(FinishArgs f) {
  __shared__ float sm[16][64];
  const int lane = threadIdx.x & 63;
  const int j = threadIdx.x >> 6;
  const int o = blockIdx.y;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < f.ncol) {
    const float* p = f.ws + (long long)o * f.cstride + c;
    int r = j;
    // (eight loads in flight, added in row order: the same sum as one at a time)
    for (; r + 112 < f.nparts; r += 128) {
      float a[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = p[(long long)(r + 16 * k) * f.pstride];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += a[k];
    }
    for (; r < f.nparts; r += 16) s += p[(long long)r * f.pstride];
  }
  sm[j][lane] = s;
  __syncthreads();
  if (j == 0 && c < f.cvalid[o]) {
    float t = sm[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sm[k][lane];
    f.out[o][c] = t * f.scale;
  }
}

// Deferred finishing launches (round 5).  The finishing launch of a reduction
// whose result only the optimizer reads -- the generator backward's seven: bias,
// gamma / beta gradients -- does not have to sit between its producer and the
// next kernel of the chain: between cg_finish_defer(1) and cg_finish_flush() on a
// thread every launch_finish is queued, and the flush adds all of them in ONE
// launch (same order of sums per output: same bits).  The caller gives every
// queued reduction its OWN workspace region (the partial rows must survive until
// the flush).
constexpr int kMaxDeferred = 12;
struct FinishBatch {
  int n;
  FinishArgs f[kMaxDeferred];
};
thread_local bool g_finish_defer = false;
thread_local FinishBatch g_finish_batch;

__global__ __launch_bounds__(1024) void finish_cols_batched_kernel(FinishBatch b) {
  const FinishArgs& f = b.f[blockIdx.z];
  if ((int)blockIdx.y >= f.nout || (int)blockIdx.x * 64 >= f.ncol) return;
  __shared__ float sm[16][64];
  const int lane = threadIdx.x & 63;
  const int j = threadIdx.x >> 6;
  const int o = blockIdx.y;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < f.ncol) {
    const float* p = f.ws + (long long)o * f.cstride + c;
    int r = j;
    for (; r + 112 < f.nparts; r += 128) {
      float a[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = p[(long long)(r + 16 * k) * f.pstride];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += a[k];
    }
    for (; r < f.nparts; r += 16) s += p[(long long)r * f.pstride];
  }
  sm[j][lane] = s;
  __syncthreads();
  if (j == 0 && c < f.cvalid[o]) {
    float t = sm[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sm[k][lane];
    f.out[o][c] = t * f.scale;
  }
}

inline void flush_finishes(hipStream_t s) {
  FinishBatch& b = g_finish_batch;
  if (b.n == 0) return;
  int gx = 1, gy = 1;
  for (int i = 0; i < b.n; ++i) {
    if ((b.f[i].ncol + 63) / 64 > gx) gx = (b.f[i].ncol + 63) / 64;
    if (b.f[i].nout > gy) gy = b.f[i].nout;
  }
  hipLaunchKernelGGL(finish_cols_batched_kernel, dim3(gx, gy, b.n), dim3(1024), 0, s, b);
  b.n = 0;
}

inline void launch_finish(const FinishArgs& f, hipStream_t s) {
  if (g_finish_defer) {
    if (g_finish_batch.n == kMaxDeferred) flush_finishes(s);
    g_finish_batch.f[g_finish_batch.n++] = f;
    return;
  }
  hipLaunchKernelGGL(finish_cols_kernel, dim3((f.ncol + 63) / 64, f.nout),
                     dim3(1024), 0, s, f);
}

__device__ __forceinline__ void unpack8(const uint4 u, float* v) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = act_lo(w[i]);
    v[2 * i + 1] = act_hi(w[i]);
  }
}
__device__ __forceinline__ uint4 ldg16(const uint16_t* p) {
  return *reinterpret_cast<const uint4*>(p);
}
__device__ __forceinline__ void load8(const uint16_t* p, float* v) {
  unpack8(ldg16(p), v);
}
__device__ __forceinline__ void store8(uint16_t* p, const float* v) {
  *reinterpret_cast<uint4*>(p) =
      make_uint4(pack2act(v[0], v[1]), pack2act(v[2], v[3]), pack2act(v[4], v[5]),
                 pack2act(v[6], v[7]));
}

// 8 consecutive f32 channels of a row whose pitch is `pitch` floats, with the
// widest aligned loads the pitch allows (16 / 8 / 4 bytes); channels at or past
// `nvalid` read as zero.
__device__ __forceinline__ void load8f(const float* p, int nvalid, int pitch,
                                       float* v) {
  if (nvalid >= 8 && (pitch & 3) == 0) {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(p);
    const f32x4 a1 = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a0[0]; v[1] = a0[1]; v[2] = a0[2]; v[3] = a0[3];
    v[4] = a1[0]; v[5] = a1[1]; v[6] = a1[2]; v[7] = a1[3];
  } else if (nvalid >= 8 && (pitch & 1) == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float2 t = *reinterpret_cast<const float2*>(p + 2 * e);
      v[2 * e] = t.x;
      v[2 * e + 1] = t.y;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = e < nvalid ? p[e] : 0.f;
  }
}

// ---------------------------------------------------------------------------
// LayerNorm + LeakyReLU.  A row (Cp <= 512 channels) is covered by LPR lanes
// of 8 channels each (LPR = power of two >= Cp/8), so one wave processes
// 64/LPR rows at a time; row statistics reduce with xor-shuffles inside the
// LPR-lane group.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float group_sum(float v, int lpr) {
  return group_sum_n(v, lpr);  // DPP / permlane swaps, no LDS round trips
}

__global__ __launch_bounds__(kThreads) void ln_fwd_kernel(
    const uint16_t* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, uint16_t* __restrict__ h,
    float* __restrict__ mean_o, float* __restrict__ rstd_o, long long rows,
    int C, int Cp, float eps, float alpha, int lpr, int log2lpr,
    int rows_per_slot) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & (lpr - 1);        // channel group inside the row
  const int slot = lane >> log2lpr;        // row slot inside the wave
  const int rpw = 64 >> log2lpr;           // rows per wave per iteration
  const long long wave_id =
      (long long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const int c0 = sub * 8;
  const bool active = c0 < Cp;
  float gam[8], bet[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool ok = active && c0 + e < C;
    gam[e] = ok ? gamma[c0 + e] : 0.f;
    bet[e] = ok ? beta[c0 + e] : 0.f;
  }
  const float invC = 1.f / C;
  const long long row0 = wave_id * (long long)rpw * rows_per_slot + slot;
  // kRB rows' loads are issued before any of them is reduced: one 16-byte load
  // per lane in flight cannot cover the HBM latency at this occupancy
#ifndef CG_LN_RB
#define CG_LN_RB 4
#endif
  constexpr int kRB = CG_LN_RB;
  for (int it0 = 0; it0 < rows_per_slot; it0 += kRB) {
   uint4 raw[kRB];
#pragma unroll
   for (int k = 0; k < kRB; ++k) {
     const long long row = row0 + (long long)(it0 + k) * rpw;
     raw[k] = (row < rows && active) ? ldg16(y + row * Cp + c0)
                                     : make_uint4(0u, 0u, 0u, 0u);
   }
#pragma unroll
   for (int k = 0; k < kRB; ++k) {
    const long long row = row0 + (long long)(it0 + k) * rpw;
    const bool rv = row < rows;  // uniform inside the LPR group
    float v[8];
    unpack8(raw[k], v);
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (c0 + e < C) s += v[e];
    const float mean = group_sum(s, lpr) * invC;
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (active && c0 + e < C) {
        const float d = v[e] - mean;
        s2 += d * d;
      }
    const float rstd = rsqrtf(group_sum(s2, lpr) * invC + eps);
    if (rv && sub == 0) {
      if (mean_o) mean_o[row] = mean;
      if (rstd_o) rstd_o[row] = rstd;
    }
    if (rv && active) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = 0.f;
        if (c0 + e < C) {
          t = (v[e] - mean) * rstd * gam[e] + bet[e];
          t = t > 0.f ? t : alpha * t;
        }
        o[e] = t;
      }
      store8(h + row * Cp + c0, o);
    }
   }
  }
}

// Rows whose 8-channel groups are not a power of two (320 -> 40 groups, 192 ->
// 24) left 37 / 25 % of the lanes of ln_fwd_kernel idle (64 / 32 lanes per row:
// 3.97 TB/s on the generator's 320- and 192-channel blocks).  Here a row is
// always 8 lanes, each holding GPL groups (lane j takes groups j, j + 8, ...: a
// load instruction covers 128 contiguous bytes per row), so every lane works
// for any pitch that is a multiple of 64 channels and a wave carries 8 rows.
// Gamma / beta live in LDS (GPL x 16 registers otherwise).
template <int GPL>
__global__ __launch_bounds__(kThreads) void ln_fwd8_kernel(
    const uint16_t* __restrict__ y, const float* __restrict__ gamma,
    const float* __restrict__ beta, uint16_t* __restrict__ h,
    float* __restrict__ mean_o, float* __restrict__ rstd_o, long long rows,
    int C, int Cp, float eps, float alpha, int rows_per_slot) {
  __shared__ float sgam[GPL * 64], sbet[GPL * 64];
  for (int i = threadIdx.x; i < GPL * 64; i += kThreads) {
    sgam[i] = i < C ? gamma[i] : 0.f;
    sbet[i] = i < C ? beta[i] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int sub = lane & 7;
  const int slot = lane >> 3;
  constexpr int rpw = 8;
  const long long wave_id =
      (long long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const int ng = Cp >> 3;
  const float invC = 1.f / C;
  const long long row0 = wave_id * (long long)rpw * rows_per_slot + slot;
  constexpr int kRB = 2;
  for (int it0 = 0; it0 < rows_per_slot; it0 += kRB) {
    uint4 raw[kRB][GPL];
#pragma unroll
    for (int k = 0; k < kRB; ++k) {
      const long long row = row0 + (long long)(it0 + k) * rpw;
#pragma unroll
      for (int j = 0; j < GPL; ++j) {
        const int g = j * 8 + sub;
        raw[k][j] = (row < rows && g < ng) ? ldg16(y + row * Cp + g * 8)
                                           : make_uint4(0u, 0u, 0u, 0u);
      }
    }
#pragma unroll
    for (int k = 0; k < kRB; ++k) {
      const long long row = row0 + (long long)(it0 + k) * rpw;
      const bool rv = row < rows;  // uniform inside the 8-lane group
      float v[GPL][8];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < GPL; ++j) {
        unpack8(raw[k][j], v[j]);
        const int c0 = (j * 8 + sub) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c0 + e < C) s += v[j][e];
      }
      const float mean = group_sum_n(s, 8) * invC;
      float s2 = 0.f;
#pragma unroll
      for (int j = 0; j < GPL; ++j) {
        const int c0 = (j * 8 + sub) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c0 + e < C) {
            const float d = v[j][e] - mean;
            s2 += d * d;
          }
      }
      const float rstd = rsqrtf(group_sum_n(s2, 8) * invC + eps);
      if (rv && sub == 0) {
        if (mean_o) mean_o[row] = mean;
        if (rstd_o) rstd_o[row] = rstd;
      }
      if (!rv) continue;
#pragma unroll
      for (int j = 0; j < GPL; ++j) {
        const int g = j * 8 + sub;
        if (g >= ng) continue;
        const int c0 = g * 8;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = 0.f;
          if (c0 + e < C) {
            t = (v[j][e] - mean) * rstd * sgam[c0 + e] + sbet[c0 + e];
            t = t > 0.f ? t : alpha * t;
          }
          o[e] = t;
        }
        store8(h + row * Cp + c0, o);
      }
    }
  }
}

// dy = rstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat)), dyh = do * gamma,
// do = dh * lrelu'(h); dgamma += do * xhat, dbeta += do: per-lane partials ->
// a fixed-order sum inside the block -> one partial row per block (ws; summed by
// finish_cols_kernel) or, without a workspace, one global atomic per channel.
__global__ __launch_bounds__(kThreads) void ln_bwd_kernel(
    const uint16_t* __restrict__ dh, const uint16_t* __restrict__ h,
    const uint16_t* __restrict__ y, const float* __restrict__ mean_i,
    const float* __restrict__ rstd_i, const float* __restrict__ gamma,
    uint16_t* __restrict__ dy, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ dbias, long long rows, int C,
    int Cp, float alpha, int lpr, int log2lpr, int rows_per_slot,
    float* __restrict__ ws) {
  __shared__ float sg[4 * 512];
  __shared__ float sb[4 * 512];
  __shared__ float sd[4 * 512];
  const int lane = threadIdx.x & 63;
  const int sub = lane & (lpr - 1);
  const int slot = lane >> log2lpr;
  const int rpw = 64 >> log2lpr;
  const long long wave_id =
      (long long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  const int c0 = sub * 8;
  const bool active = c0 < Cp;
  float gam[8], accg[8], accb[8], accd[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    gam[e] = (active && c0 + e < C) ? gamma[c0 + e] : 0.f;
    accg[e] = 0.f;
    accb[e] = 0.f;
    accd[e] = 0.f;
  }
  const float invC = 1.f / C;
  const long long row0 = wave_id * (long long)rpw * rows_per_slot + slot;
  constexpr int kRB = 2;  // rows whose loads are in flight together
  for (int it0 = 0; it0 < rows_per_slot; it0 += kRB) {
   uint4 rd[kRB], rh[kRB], ry[kRB];
   float rmean[kRB], rrstd[kRB];
#pragma unroll
   for (int k = 0; k < kRB; ++k) {
     const long long row = row0 + (long long)(it0 + k) * rpw;
     const bool ok = row < rows;
     const uint4 z = make_uint4(0u, 0u, 0u, 0u);
     rmean[k] = ok ? mean_i[row] : 0.f;
     rrstd[k] = ok ? rstd_i[row] : 0.f;
     rd[k] = (ok && active) ? ldg16(dh + row * Cp + c0) : z;
     rh[k] = (ok && active) ? ldg16(h + row * Cp + c0) : z;
     ry[k] = (ok && active) ? ldg16(y + row * Cp + c0) : z;
   }
#pragma unroll
   for (int k = 0; k < kRB; ++k) {
    const long long row = row0 + (long long)(it0 + k) * rpw;
    const bool rv = row < rows;
    float vd[8], vh[8], vy[8];
    unpack8(rd[k], vd);
    unpack8(rh[k], vh);
    unpack8(ry[k], vy);
    const float mean = rmean[k], rstd = rrstd[k];
    float xh[8], dyh[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xh[e] = 0.f;
      dyh[e] = 0.f;
      if (rv && active && c0 + e < C) {
        const float d_o = vd[e] * (vh[e] > 0.f ? 1.f : alpha);
        xh[e] = (vy[e] - mean) * rstd;
        dyh[e] = d_o * gam[e];
        accg[e] += d_o * xh[e];
        accb[e] += d_o;
        s1 += dyh[e];
        s2 += dyh[e] * xh[e];
      }
    }
    s1 = group_sum(s1, lpr) * invC;
    s2 = group_sum(s2, lpr) * invC;
    if (rv && active) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        o[e] = (c0 + e < C) ? rstd * (dyh[e] - s1 - xh[e] * s2) : 0.f;
        accd[e] += act2f(f2act(o[e]));  // bias gradient of the producing conv
      }
      store8(dy + row * Cp + c0, o);
    }
   }
  }
  // rows of the lane slots that share a channel group: xor tree inside the wave
  // (a fixed order), slot 0 holds the wave's sums
  for (int o = lpr; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      accg[e] += __shfl_xor(accg[e], o, 64);
      accb[e] += __shfl_xor(accb[e], o, 64);
      accd[e] += __shfl_xor(accd[e], o, 64);
    }
  }
  // the four waves through LDS, added in wave order
  const int wv = threadIdx.x >> 6;
  if (slot == 0 && active) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sg[wv * 512 + c0 + e] = accg[e];
      sb[wv * 512 + c0 + e] = accb[e];
      sd[wv * 512 + c0 + e] = accd[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Cp; c += kThreads) {
    const float g4 = ((sg[c] + sg[512 + c]) + sg[1024 + c]) + sg[1536 + c];
    const float b4 = ((sb[c] + sb[512 + c]) + sb[1024 + c]) + sb[1536 + c];
    const float d4 = ((sd[c] + sd[512 + c]) + sd[1024 + c]) + sd[1536 + c];
    if (ws) {
      // one partial row [3][Cp] per block; finish_cols_kernel adds the rows
      float* row = ws + (long long)blockIdx.x * 3 * Cp;
      row[c] = g4;
      row[Cp + c] = b4;
      row[2 * Cp + c] = d4;
    } else if (c < C) {
      atomicAdd(dgamma + c, g4);
      atomicAdd(dbeta + c, b4);
      if (dbias) atomicAdd(dbias + c, d4);
    }
  }
}

// ---------------------------------------------------------------------------
// BatchNormalization(axis=-1) of the generator blocks (calciumgan.py:42-43;
// Keras defaults: momentum 0.99, epsilon 1e-3, biased batch variance).  Batch
// statistics are column sums over ALL rows: two-stage ordered reductions through
// the shared workspace (no atomics form).
// ---------------------------------------------------------------------------
// per block: sum of v and of v * w over its rows for every channel, v / w chosen by
// MODE: 0 (statistics) v = y, w = y; 1 (backward) v = do, w = xhat with
// do = dout * (act ? lrelu'(h) : 1), xhat = (y - mean) * rstd.
// Partial row [2][Cp] per block; thread -> (8-channel group, row lane) as colsum.
template <int MODE>
__global__ __launch_bounds__(kThreads) void bn_sums_kernel(
    const uint16_t* __restrict__ y, const uint16_t* __restrict__ dout,
    const uint16_t* __restrict__ h, const float* __restrict__ mean,
    const float* __restrict__ var, long long rows, int C, int Cp,
    int rows_per_block, float eps, float alpha, int act, float* __restrict__ ws) {
  __shared__ float s1[kThreads * 8];
  __shared__ float s2[kThreads * 8];
  const int groups = Cp / 8;
  const int rlanes = kThreads / groups;
  const int grp = threadIdx.x % groups;
  const int rl = threadIdx.x / groups;
  float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < rlanes) {
    float mu[8], rs[8];
    if (MODE == 0) {
      // statistics: sums of (y - K) and (y - K)^2 around the block's own first
      // row K (the one-pass E[y^2] - E[y]^2 over raw values cancels for
      // channels whose mean is large against their spread); bn_stats_finish_kernel
      // combines the blocks' (count, mean, M2) -- tf.nn.moments' mean((y - mean)^2)
      load8(y + (long long)blockIdx.x * rows_per_block * Cp + grp * 8, mu);
    }
    if (MODE == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = grp * 8 + e;
        mu[e] = c < C ? mean[c] : 0.f;
        rs[e] = c < C ? rsqrtf(var[c] + eps) : 0.f;
      }
    }
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    for (int rr = rl; rr < rows_per_block; rr += rlanes) {
      const long long row = r0 + rr;
      if (row >= rows) break;
      float vy[8];
      load8(y + row * Cp + grp * 8, vy);
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = vy[e] - mu[e];
          a1[e] += d;
          a2[e] += d * d;
        }
      } else {
        float vd[8], vh[8];
        load8(dout + row * Cp + grp * 8, vd);
        if (act) load8(h + row * Cp + grp * 8, vh);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d_o = act ? vd[e] * (vh[e] > 0.f ? 1.f : alpha) : vd[e];
          a1[e] += d_o;
          a2[e] += d_o * ((vy[e] - mu[e]) * rs[e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[rl * Cp + grp * 8 + e] = a1[e];
      s2[rl * Cp + grp * 8 + e] = a2[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Cp; c += kThreads) {
    float t1 = s1[c], t2 = s2[c];
    for (int r = 1; r < rlanes; ++r) {
      t1 += s1[r * Cp + c];
      t2 += s2[r * Cp + c];
    }
    float* row = ws + (long long)blockIdx.x * (MODE == 0 ? 3 : 2) * Cp;
    row[c] = t1;
    row[Cp + c] = t2;
    if (MODE == 0)  // the block's shift (its first row)
      row[2 * Cp + c] = act2f(y[(long long)blockIdx.x * rows_per_block * Cp + c]);
  }
}

// mean / biased variance from the partial rows (64 channels x 16 row classes per
// block, as finish_cols_kernel) and the moving averages of the layer
// Partial rows [3][Cp] per block: S1 = sum(y - K), S2 = sum((y - K)^2), K.  The
// blocks' (count, mean, M2) are combined in a fixed order (16 row classes through
// LDS, classes added in order): mean first, then M2 = sum(M2_b + n_b (mean_b -
// mean)^2) -- the biased batch variance mean((y - mean)^2) of tf.nn.moments.
__global__ __launch_bounds__(1024) void bn_stats_finish_kernel(
    const float* __restrict__ ws, int nparts, int C, int Cp, long long rows,
    int rows_per_block, float* __restrict__ mean, float* __restrict__ var,
    float* __restrict__ moving_mean, float* __restrict__ moving_var,
    float momentum) {
  __shared__ float sm[16][64];
  __shared__ float smean[64];
  const int lane = threadIdx.x & 63;
  const int j = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const float inv_rows = 1.f / (float)rows;
  auto count = [&](int r) {
    const long long left = rows - (long long)r * rows_per_block;
    return (float)(left < rows_per_block ? left : rows_per_block);
  };
  float a = 0.f;
  if (c < Cp)
    for (int r = j; r < nparts; r += 16) {
      const float* row = ws + (long long)r * 3 * Cp;
      a += row[c] + count(r) * row[2 * Cp + c];  // the block's sum of y
    }
  sm[j][lane] = a;
  __syncthreads();
  if (j == 0) {
    float t = sm[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += sm[k][lane];
    smean[lane] = t * inv_rows;
  }
  __syncthreads();
  const float m = smean[lane];
  float b = 0.f;
  if (c < Cp)
    for (int r = j; r < nparts; r += 16) {
      const float* row = ws + (long long)r * 3 * Cp;
      const float nb = count(r);
      const float s1 = row[c];
      const float d = row[2 * Cp + c] + s1 / nb - m;
      b += (row[Cp + c] - s1 * s1 / nb) + nb * d * d;
    }
  sm[j][lane] = b;
  __syncthreads();
  if (j == 0 && c < C) {
    float t2 = sm[0][lane];
#pragma unroll
    for (int k = 1; k < 16; ++k) t2 += sm[k][lane];
    const float v = fmaxf(t2 * inv_rows, 0.f);
    mean[c] = m;
    var[c] = v;
    if (moving_mean) {
      moving_mean[c] = moving_mean[c] * momentum + m * (1.f - momentum);
      moving_var[c] = moving_var[c] * momentum + v * (1.f - momentum);
    }
  }
}

// out = f((y - mean) * rsqrt(var + eps) * gamma + beta), f = max(t, alpha t)
// (alpha = 1: no activation)
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(
    const uint16_t* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ gamma,
    const float* __restrict__ beta, uint16_t* __restrict__ out, int C, int Cp,
    float eps, float alpha, long long total8) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const int per_row = Cp / 8;
  const long long row = idx / per_row;
  const int c0 = (int)(idx - row * per_row) * 8;
  float v[8], o[8];
  load8(y + row * Cp + c0, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    float t = 0.f;
    if (c < C) {
      t = (v[e] - mean[c]) * rsqrtf(var[c] + eps) * gamma[c] + beta[c];
      t = fmaxf(t, alpha * t);
    }
    o[e] = t;
  }
  store8(out + row * Cp + c0, o);
}

// dy = gamma * rstd * (do - dbeta / R - xhat * dgamma / R)
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(
    const uint16_t* __restrict__ dout, const uint16_t* __restrict__ h,
    const uint16_t* __restrict__ y, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ gamma,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta,
    uint16_t* __restrict__ dy, int C, int Cp, float eps, float alpha, int act,
    float inv_rows, long long total8) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const int per_row = Cp / 8;
  const long long row = idx / per_row;
  const int c0 = (int)(idx - row * per_row) * 8;
  float vd[8], vh[8], vy[8], o[8];
  load8(dout + row * Cp + c0, vd);
  load8(y + row * Cp + c0, vy);
  if (act) load8(h + row * Cp + c0, vh);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    float t = 0.f;
    if (c < C) {
      const float rs = rsqrtf(var[c] + eps);
      const float d_o = act ? vd[e] * (vh[e] > 0.f ? 1.f : alpha) : vd[e];
      const float xh = (vy[e] - mean[c]) * rs;
      t = gamma[c] * rs * (d_o - dbeta[c] * inv_rows - xh * dgamma[c] * inv_rows);
    }
    o[e] = t;
  }
  store8(dy + row * Cp + c0, o);
}

// ---------------------------------------------------------------------------
// discriminator head
// ---------------------------------------------------------------------------
// (delta != null: the same pass also seeds the backward chain,
// delta[b][t][c] = coef[b / seg_size] * bf16(w[t*C+c]) * lrelu'(h[b][t][c]) --
// cg_dense1_bwd's work without a second read of h: the seed does not depend on
// the head's output)
// (NT threads per sample: one block per sample leaves a CU with one or two blocks
// at cfg2 -- 384 samples, 256 CUs -- so the block is as wide as the row allows:
// 1024 threads when F >= 8192 groups' worth, 13.7 -> ~8 us)
template <int NT>
__global__ __launch_bounds__(NT) void dense1_fwd_kernel(
    const uint16_t* __restrict__ h, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ out, int F, int C,
    int Cp, const float* __restrict__ coef, uint16_t* __restrict__ delta,
    int seg_size, float alpha) {
  constexpr int kThreads = NT;  // (shadows the file's 256 inside this kernel)
  __shared__ float part[NT / 64];
  const int b = blockIdx.x;
  const float cf = delta ? coef[b / seg_size] : 0.f;
  float s = 0.f;
  constexpr int kNB = NT >= 1024 ? 2 : 4;  // load pairs in flight per lane
  for (int i0 = threadIdx.x * 8; i0 < F; i0 += kNB * kThreads * 8) {
    uint4 raw[kNB];
    float wv[kNB][8];
#pragma unroll
    for (int k = 0; k < kNB; ++k) {
      const int i = i0 + k * kThreads * 8;
      const bool ok = i < F;
      raw[k] = ok ? ldg16(h + (long long)b * F + i) : make_uint4(0u, 0u, 0u, 0u);
      const int t = ok ? i / Cp : 0;
      const int c = ok ? i - t * Cp : 0;
      // w is (Lt, C) row-major: 8 consecutive channels of one timestep
      load8f(w + (long long)t * C + c, ok ? C - c : 0, C, wv[k]);
    }
#pragma unroll
    for (int k = 0; k < kNB; ++k) {
      float v[8], o[8];
      unpack8(raw[k], v);
      const int i = i0 + k * kThreads * 8;
      const int ch = i - (i / Cp) * Cp;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float wq = act2f(f2act(wv[k][e]));
        s += v[e] * wq;
        o[e] = (ch + e < C) ? cf * wq * (v[e] > 0.f ? 1.f : alpha) : 0.f;
      }
      if (delta && i < F) store8(delta + (long long)b * F + i, o);
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = part[0];
#pragma unroll
    for (int k = 1; k < NT / 64; ++k) t += part[k];
    out[b] = t + bias[0];
  }
}

__global__ __launch_bounds__(kThreads) void dense1_bwd_kernel(
    const float* __restrict__ w, const float* __restrict__ coef,
    const uint16_t* __restrict__ h, uint16_t* __restrict__ delta, int F, int C,
    int Cp, int seg_size, float alpha, long long total8) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const int per_row = F / 8;
  const int b = (int)(idx / per_row);
  const int i = (int)(idx - (long long)b * per_row) * 8;
  const float c = coef[b / seg_size];
  float vh[8], o[8], wv[8];
  const uint4 raw = ldg16(h + (long long)b * F + i);
  const int t = i / Cp;
  const int ch = i - t * Cp;
  load8f(w + (long long)t * C + ch, C - ch, C, wv);
  unpack8(raw, vh);
#pragma unroll
  for (int e = 0; e < 8; ++e)
    o[e] = (ch + e < C) ? c * act2f(f2act(wv[e])) * (vh[e] > 0.f ? 1.f : alpha)
                        : 0.f;
  store8(delta + (long long)b * F + i, o);
}

// dw[i] += sum_b coef[seg(b)] * x[b][i]; grid (F/8/256, bsplit).  With a
// workspace the sample splits leave partial rows ws[split][Fpad] (+ their share of
// the bias sum behind them) and dense1_wgrad_finish adds them in split order.
__global__ __launch_bounds__(kThreads) void dense1_wgrad_kernel(
    const uint16_t* __restrict__ x, const float* __restrict__ coef,
    const float* __restrict__ bias_coef, float* __restrict__ dw,
    float* __restrict__ db, int nB, int F, int C, int Cp, int seg_size,
    float* __restrict__ ws, int Fpad) {
  const int i = (blockIdx.x * kThreads + threadIdx.x) * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float bsum = 0.f;
  constexpr int kNB = 4;  // samples whose loads are in flight together
  for (int b0 = blockIdx.y; b0 < nB; b0 += kNB * gridDim.y) {
    uint4 raw[kNB];
    float cf[kNB];
#pragma unroll
    for (int k = 0; k < kNB; ++k) {
      const int b = b0 + k * gridDim.y;
      const bool ok = b < nB;
      cf[k] = ok ? coef[b / seg_size] : 0.f;
      if (ok && i == 0 && bias_coef) bsum += bias_coef[b / seg_size];
      raw[k] = (ok && i < F) ? ldg16(x + (long long)b * F + i)
                             : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < kNB; ++k) {
      float v[8];
      unpack8(raw[k], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += cf[k] * v[e];
    }
  }
  if (ws) {
    float* row = ws + (long long)blockIdx.y * Fpad;
    if (i < F) {
      *reinterpret_cast<f32x4*>(row + i) = f32x4{acc[0], acc[1], acc[2], acc[3]};
      *reinterpret_cast<f32x4*>(row + i + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    }
    if (i == 0) ws[(long long)gridDim.y * Fpad + blockIdx.y] = bsum;
    return;
  }
  // transpose through LDS so that one atomic instruction covers 64 consecutive
  // features (256 contiguous bytes) instead of 64 features 32 bytes apart
  __shared__ float tr[kThreads * 8];
#pragma unroll
  for (int e = 0; e < 8; ++e) tr[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  const int i0 = blockIdx.x * kThreads * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int j = e * kThreads + threadIdx.x;  // feature offset inside the block
    const int f = i0 + j;
    if (f < F) {
      const int t = f / Cp;
      const int ch = f - t * Cp;
      if (ch < C) atomicAdd(dw + t * C + ch, tr[j]);
    }
  }
  if (i == 0 && db && bias_coef) atomicAdd(db, bsum);
}

// dw[t * C + ch] = sum over the splits, in split order (one thread per feature)
__global__ __launch_bounds__(kThreads) void dense1_wgrad_finish(
    const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ db,
    int nsplit, int F, int Fpad, int C, int Cp, int has_bias) {
  const int f = blockIdx.x * kThreads + threadIdx.x;
  if (f < F) {
    // (independent loads, eight in flight: one at a time the walk over ~50
    // splits was 15 us of dependent latency)
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= nsplit; z += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = ws[(long long)(z + k) * Fpad + f];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; z < nsplit; ++z) s += ws[(long long)z * Fpad + f];
    const int t = f / Cp;
    const int ch = f - t * Cp;
    if (ch < C) dw[t * C + ch] = s;
  }
  if (f == 0 && db && has_bias) {
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += ws[(long long)nsplit * Fpad + z];
    db[0] = s;
  }
}

// ---------------------------------------------------------------------------
// phase unshuffle + LeakyReLU mask
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void unshuffle_mask_kernel(
    const uint16_t* __restrict__ e, const uint16_t* __restrict__ h,
    uint16_t* __restrict__ delta, const int* __restrict__ shifts, int w, int Cp,
    int seg_size, float alpha, long long total8) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const int per_row = Cp / 8;
  const long long rowg = idx / per_row;
  const int c = (int)(idx - rowg * per_row) * 8;
  const int b = (int)(rowg / w);
  const int r = (int)(rowg - (long long)b * w);
  const int s = shifts ? shifts[b / seg_size] : 0;
  // all t with shuffle_src(t, s, w) == r
  int t0, t1 = -1;
  if (s > 0) {
    t0 = r - s;                    // direct branch, valid if t0 >= 0
    const int tr = 2 * (w - 1) - s - r;  // reflected branch, t in [w-s, w-1]
    if (tr >= w - s && tr <= w - 1) t1 = tr;
  } else {
    const int a = -s;
    t0 = r + a;                    // direct branch t >= a, valid if t0 < w
    if (t0 >= w) t0 = -1;
    const int tr = a - r;          // reflected branch t in [0, a)
    if (tr >= 0 && tr < a) t1 = tr;
  }
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float v[8];
  const long long base = (long long)b * w;
  if (t0 >= 0) {
    load8(e + (base + t0) * Cp + c, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
  if (t1 >= 0) {
    load8(e + (base + t1) * Cp + c, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += v[k];
  }
  load8(h + rowg * Cp + c, v);
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] *= (v[k] > 0.f ? 1.f : alpha);
  store8(delta + rowg * Cp + c, acc);
}

// Second half of the unshuffle fused into the producing convolution
// (cg_conv_desc.out_shifts): thread (b, j, 8 channels), j < |s|, zeroes the
// row nothing maps to and folds reflected row j into its source row.
// (One launch per layer: the next input-gradient launch of the chain reads the
// rows this one completes, so the layers of a pass cannot share a launch.)
__global__ __launch_bounds__(kThreads) void unshuffle_fixup_kernel(
    const uint16_t* __restrict__ side, const uint16_t* __restrict__ h,
    uint16_t* __restrict__ delta, const int* __restrict__ shifts, int w, int Cp,
    int seg_size, int side_rows, float alpha, long long total8) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const int per_row = Cp / 8;
  const long long rowg = idx / per_row;
  const int c = (int)(idx - rowg * per_row) * 8;
  const int b = (int)(rowg / side_rows);
  const int j = (int)(rowg - (long long)b * side_rows);
  const int s = shifts[b / seg_size];
  const int n = s > 0 ? s : -s;
  if (j >= n) return;
  // s > 0: reflected output row t = w - s + j came from r = w - 2 - j, rows
  // [0, s) receive nothing; s < 0 (a = -s): t = j came from r = a - j, rows
  // [w - a, w) receive nothing
  const int r_add = s > 0 ? w - 2 - j : n - j;
  const int r_zero = s > 0 ? j : w - n + j;
  const long long base = (long long)b * w;
  float sv[8], hv[8], dv[8];
  load8(side + rowg * Cp + c, sv);
  load8(h + (base + r_add) * Cp + c, hv);
  load8(delta + (base + r_add) * Cp + c, dv);
#pragma unroll
  for (int k = 0; k < 8; ++k) dv[k] += sv[k] * (hv[k] > 0.f ? 1.f : alpha);
  store8(delta + (base + r_add) * Cp + c, dv);
  const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  store8(delta + (base + r_zero) * Cp + c, z);
}

// ---------------------------------------------------------------------------
// WGAN-GP pieces
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void interp_pack_kernel(
    const float* __restrict__ real, const float* __restrict__ fake,
    const float* __restrict__ alpha, uint16_t* __restrict__ x0, int B, int L,
    int C, int Cr, int Cf, int Cp, long long total8, int write_real) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const int per_row = Cp / 8;
  const long long row = idx / per_row;  // b*L + t
  const int c = (int)(idx - row * per_row) * 8;
  const int b = (int)(row / L);
  // (alpha == nullptr: [real | fake] only -- the caller takes layer 1 of x^ from
  // the layer's outputs on the other two segments, cg_lrelu_mix)
  const float al = alpha ? alpha[b] : 0.f;
  float r[8], f[8], x[8];
  // (the 8-channel group base c is a multiple of 8, so the loads are aligned
  // whenever the pitch is)
  load8f(real + row * Cr + c, C - c, Cr, r);
  load8f(fake + row * Cf + c, C - c, Cf, f);
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = al * r[e] + (1.f - al) * f[e];
  const long long seg = (long long)B * L * Cp;
  if (write_real) store8(x0 + row * Cp + c, r);
  store8(x0 + seg + row * Cp + c, f);
  if (alpha) store8(x0 + 2 * seg + row * Cp + c, x);
}

__global__ __launch_bounds__(kThreads) void cast_pad_kernel(
    const float* __restrict__ src, uint16_t* __restrict__ dst, int C, int Cs,
    int Cp, long long total8) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const int per_row = Cp / 8;
  const long long row = idx / per_row;
  const int c = (int)(idx - row * per_row) * 8;
  float v[8];
  load8f(src + row * Cs + c, C - c, Cs, v);
  store8(dst + row * Cp + c, v);
}

// sumsq[b] += partial of g bf16 [B][n]; grid (chunks, B).  With a workspace the
// chunks leave ws[b][chunk] and sqrt_sum_kernel adds them in chunk order.
__global__ __launch_bounds__(kThreads) void sumsq_kernel(
    const uint16_t* __restrict__ g, float* __restrict__ sumsq, long long n,
    float* __restrict__ ws) {
  __shared__ float part[4];
  const int b = blockIdx.y;
  const uint16_t* p = g + (long long)b * n;
  float s = 0.f;
  for (long long i = ((long long)blockIdx.x * kThreads + threadIdx.x) * 8; i < n;
       i += (long long)gridDim.x * kThreads * 8) {
    float v[8];
    load8(p + i, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e] * v[e];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = ((part[0] + part[1]) + part[2]) + part[3];
    if (ws) ws[(long long)b * gridDim.x + blockIdx.x] = t;
    else atomicAdd(sumsq + b, t);
  }
}

__global__ void sqrt_sum_kernel(const float* __restrict__ ws, float* v, int n,
                                int chunks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < chunks; ++c) s += ws[(long long)i * chunks + c];
  v[i] = sqrtf(s);
}

// (a kernel, not hipMemsetAsync: inside the captured step that call is a memset
// NODE, and replayed after other work had run it no longer left zeros in front
// of the atomics below -- the round-2 "stale graph" penalties of 1e25, DESIGN.md
// section 8; tools/stale_graph_hunt.py memsetprobe)
__global__ void zero_f32_kernel(float* v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = 0.f;
}

__global__ void sqrt_kernel(float* v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = sqrtf(v[i]);
}

__global__ void gp_finalize_kernel(float* __restrict__ norm,
                                   float* __restrict__ gp,
                                   float* __restrict__ coef, int B, float scale,
                                   int squared) {
  __shared__ float part[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += kThreads) {
    float nv = norm[b];
    if (squared) {
      nv = sqrtf(nv);
      norm[b] = nv;
    }
    const float d = nv - 1.f;
    s += d * d;
    coef[b] = scale * 2.f * d / (B * nv);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) gp[0] = (part[0] + part[1] + part[2] + part[3]) / B;
}

__global__ __launch_bounds__(kThreads) void scale_rows_kernel(
    const uint16_t* __restrict__ g, const float* __restrict__ coef,
    uint16_t* __restrict__ a0, long long n, long long total8) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const long long i = idx * 8;
  const int b = (int)(i / n);
  const float c = coef[b];
  float v[8], o[8];
  load8(g + i, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = c * v[e];
  store8(a0 + i, o);
}

__global__ void critic_loss_kernel(const float* __restrict__ d_out,
                                   const float* __restrict__ gp, float penalty,
                                   float* __restrict__ out, int B) {
  __shared__ float pr[4], pf[4];
  float sr = 0.f, sf = 0.f;
  for (int b = threadIdx.x; b < B; b += kThreads) {
    sr += d_out[b];
    sf += d_out[B + b];
  }
  sr = wave_sum(sr);
  sf = wave_sum(sf);
  if ((threadIdx.x & 63) == 0) {
    pr[threadIdx.x >> 6] = sr;
    pf[threadIdx.x >> 6] = sf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float mr = (pr[0] + pr[1] + pr[2] + pr[3]) / B;
    const float mf = (pf[0] + pf[1] + pf[2] + pf[3]) / B;
    out[0] = -mr + mf + penalty * gp[0];
    out[1] = -mf;
  }
}

__global__ void neg_mean_kernel(const float* __restrict__ d_out,
                                float* __restrict__ out, int B) {
  __shared__ float pr[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += kThreads) s += d_out[b];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) pr[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = -(pr[0] + pr[1] + pr[2] + pr[3]) / B;
}

// column sums: block handles `rows_per_block` rows of one slab of up to 2048
// channels (blockIdx.y); thread -> (8-channel group, row lane); the row lanes
// meet through LDS in a fixed order, then one partial row per block (ws, summed
// by finish_cols_kernel) or one atomic per channel per block.
__global__ __launch_bounds__(kThreads) void colsum_kernel(
    const uint16_t* __restrict__ x, float* __restrict__ out, long long rows,
    int C, int Cp, int rows_per_block, float* __restrict__ ws) {
  __shared__ float sacc[kThreads * 8];  // [row lane][slab channel]
  const int c_base = blockIdx.y * (kThreads * 8);
  const int slab = min(Cp - c_base, kThreads * 8);
  const int groups = slab / 8;
  const int rlanes = kThreads / groups;  // rows processed concurrently
  const int grp = threadIdx.x % groups;
  const int rl = threadIdx.x / groups;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < rlanes) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const uint16_t* xg = x + c_base + grp * 8;
    // four rows' loads in flight per thread before the first add
    for (int rr = rl; rr < rows_per_block; rr += 4 * rlanes) {
      uint4 raw[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const long long row = r0 + rr + k * rlanes;
        raw[k] = (rr + k * rlanes < rows_per_block && row < rows)
                     ? ldg16(xg + row * Cp)
                     : make_uint4(0u, 0u, 0u, 0u);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float v[8];
        unpack8(raw[k], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sacc[rl * slab + grp * 8 + e] = acc[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < slab; c += kThreads) {
    float t = sacc[c];
    for (int r = 1; r < rlanes; ++r) t += sacc[r * slab + c];
    if (ws) ws[(long long)blockIdx.x * Cp + c_base + c] = t;
    else if (c_base + c < C) atomicAdd(out + c_base + c, t);
  }
}

__global__ __launch_bounds__(kThreads) void sigmoid_bwd_kernel(
    const uint16_t* __restrict__ dfake, const float* __restrict__ fake,
    uint16_t* __restrict__ dz, int C, int Cf, int Cp, long long total8) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const int per_row = Cp / 8;
  const long long row = idx / per_row;
  const int c = (int)(idx - row * per_row) * 8;
  float o[8], sv[8], dv[8];
  load8f(fake + row * Cf + c, C - c, Cf, sv);
  load8(dfake + row * Cp + c, dv);  // padding channels are zero in dfake
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = dv[e] * sv[e] * (1.f - sv[e]);
  store8(dz + row * Cp + c, o);
}

__global__ __launch_bounds__(kThreads) void lrelu_bwd_kernel(
    const uint16_t* __restrict__ dh, const uint16_t* __restrict__ h,
    uint16_t* __restrict__ dpre, float alpha, long long total8) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  float a[8], b[8];
  load8(dh + idx * 8, a);
  load8(h + idx * 8, b);
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] *= (b[e] > 0.f ? 1.f : alpha);
  store8(dpre + idx * 8, a);
}

// Layer 1 of the critic on x^ = a real + (1 - a) fake without the convolution
// (wgan_gp.py:41-47, 68-72; round 5): a convolution is linear, so the layer's
// pre-activation on x^ is a y_real + (1 - a) y_fake (the bias rides along, a +
// (1 - a) = 1).  h = max(y, alpha y) with 0 < alpha <= 1 is inverted exactly up to
// the rounding of the stored activation: y = h > 0 ? h : h / alpha.
__global__ __launch_bounds__(kThreads) void lrelu_mix_kernel(
    const uint16_t* __restrict__ ha, const uint16_t* __restrict__ hb,
    const float* __restrict__ mix, uint16_t* __restrict__ out,
    long long per_sample8, long long total8, float alpha, float inv_alpha) {
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const float al = mix[idx / per_sample8];
  float a[8], b[8];
  load8(ha + idx * 8, a);
  load8(hb + idx * 8, b);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float ya = a[e] > 0.f ? a[e] : a[e] * inv_alpha;
    const float yb = b[e] > 0.f ? b[e] : b[e] * inv_alpha;
    const float y = al * ya + (1.f - al) * yb;
    a[e] = fmaxf(y, alpha * y);
  }
  store8(out + idx * 8, a);
}

__global__ __launch_bounds__(kThreads) void adam_kernel(
    float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ m,
    float* __restrict__ v, long long n, float lr_t, float b1, float b2,
    float eps, float gscale, const float* __restrict__ lr_t_dev) {
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  if (lr_t_dev) lr_t = lr_t_dev[0];
  const float g = grad[i] * gscale;
  const float mi = b1 * m[i] + (1.f - b1) * g;
  const float vi = b2 * v[i] + (1.f - b2) * g * g;
  m[i] = mi;
  v[i] = vi;
  p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

// ---- dynamic loss scaling (mixed_float16: optimizer.py:10-12,23-29) ---------
// state ls[4] = {scale S, consecutive finite steps, applied Adam steps t,
// "gradients finite" flag of the update in progress (1 between updates)}
__global__ __launch_bounds__(kThreads) void grad_finite_kernel(
    const float* __restrict__ grad, long long n4, float* __restrict__ ls) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n4;
       i += (long long)gridDim.x * kThreads) {
    const f32x4 g = reinterpret_cast<const f32x4*>(grad)[i];
    bad |= !(__builtin_isfinite(g[0]) && __builtin_isfinite(g[1]) &&
             __builtin_isfinite(g[2]) && __builtin_isfinite(g[3]));
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) ls[3] = 0.f;  // (only ever 0)
}

__global__ __launch_bounds__(kThreads) void adam_ls_kernel(
    float* __restrict__ p, const float* __restrict__ grad, float* __restrict__ m,
    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
    float gscale, const float* __restrict__ ls) {
  // (the step size once per block, not two powf and a sqrtf per element)
  __shared__ float s_lr_t;
  if (threadIdx.x == 0) {
    const float t = ls[2] + 1.f;
    s_lr_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));
  }
  __syncthreads();
  const long long i = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n) return;
  if (ls[3] == 0.f) return;  // non-finite gradients: the update is skipped
  const float lr_t = s_lr_t;
  const float g = grad[i] * (gscale / ls[0]);
  const float mi = b1 * m[i] + (1.f - b1) * g;
  const float vi = b2 * v[i] + (1.f - b2) * g * g;
  m[i] = mi;
  v[i] = vi;
  p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}

__global__ void loss_scale_update_kernel(float* __restrict__ ls, float interval) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (ls[3] != 0.f) {
    ls[2] += 1.f;
    if (ls[1] + 1.f >= interval) {
      const float s2 = ls[0] * 2.f;
      if (__builtin_isfinite(s2)) ls[0] = s2;  // (stays put once doubling overflows)
      ls[1] = 0.f;
    } else {
      ls[1] += 1.f;
    }
  } else {
    ls[0] = fmaxf(ls[0] * 0.5f, 1.f);
    ls[1] = 0.f;
  }
  ls[3] = 1.f;
}

// min/max/mean/std over the channels of each (b,t) row of real and of fake,
// squared differences summed into out[4].  A row is covered by LPR lanes of 8
// channels (LPR = power of two >= C/8, <= 64), 64/LPR rows per wave at a time;
// rows wider than 512 channels loop over 512-channel spans.
__global__ __launch_bounds__(kThreads) void signal_metrics_kernel(
    const float* __restrict__ real, const float* __restrict__ fake,
    float* __restrict__ out, long long rows, int C, int Cr, int Cf, float smin,
    float scale, int lpr, int log2lpr, int rows_per_slot,
    float* __restrict__ ws) {
  __shared__ float part[4][4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int sub = lane & (lpr - 1);
  const int slot = lane >> log2lpr;
  const int rpw = 64 >> log2lpr;
  float acc[4] = {0, 0, 0, 0};
  const long long wave_id = (long long)blockIdx.x * 4 + wave;
  const long long row0 = wave_id * (long long)rpw * rows_per_slot + slot;
  const float invC = 1.f / C;
  if (C <= lpr * 8) {
    // one 8-channel group per lane covers the row: two rows of both tensors
    // are loaded before the first reduction and stay in registers for the
    // second (variance) pass
    const int c = sub * 8;
    const int nv = C - c;
    for (int it0 = 0; it0 < rows_per_slot; it0 += 2) {
      float v[2][2][8];
      bool rvk[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const long long row = row0 + (long long)(it0 + k) * rpw;
        rvk[k] = it0 + k < rows_per_slot && row < rows;
        if (rvk[k] && nv > 0) {
          load8f(real + row * Cr + c, nv, Cr, v[k][0]);
          load8f(fake + row * Cf + c, nv, Cf, v[k][1]);
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float st[2][4];
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          float mn = INFINITY, mx = -INFINITY, sum = 0.f;
          float t[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const bool ok = rvk[k] && e < nv;
            t[e] = ok ? v[k][which][e] * scale + smin : 0.f;
            mn = ok ? fminf(mn, t[e]) : mn;
            mx = ok ? fmaxf(mx, t[e]) : mx;
            sum += t[e];
          }
          mn = group_min_n(mn, lpr);
          mx = group_max_n(mx, lpr);
          const float mean = group_sum(sum, lpr) * invC;
          float s2 = 0.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = t[e] - mean;
            s2 += (rvk[k] && e < nv) ? d * d : 0.f;
          }
          st[which][0] = mn;
          st[which][1] = mx;
          st[which][2] = mean;
          st[which][3] = sqrtf(group_sum(s2, lpr) * invC);
        }
        if (rvk[k] && sub == 0) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float d = st[0][q] - st[1][q];
            acc[q] += d * d;
          }
        }
      }
    }
  } else
  for (int it = 0; it < rows_per_slot; ++it) {
    const long long row = row0 + (long long)it * rpw;
    const bool rv = row < rows;
    float st[2][4];
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const float* p = which ? fake + row * Cf : real + row * Cr;
      const int pitch = which ? Cf : Cr;
      float mn = INFINITY, mx = -INFINITY, s = 0.f;
      for (int c = sub * 8; c < C; c += lpr * 8) {
        float v[8];
        if (rv) load8f(p + c, C - c, pitch, v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (rv && c + e < C) {
            const float t = v[e] * scale + smin;
            mn = fminf(mn, t);
            mx = fmaxf(mx, t);
            s += t;
          }
      }
      for (int o = lpr >> 1; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
      }
      const float mean = group_sum(s, lpr) * invC;
      float s2 = 0.f;
      for (int c = sub * 8; c < C; c += lpr * 8) {
        float v[8];
        if (rv) load8f(p + c, C - c, pitch, v);  // L1/L2 hit (or registers)
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (rv && c + e < C) {
            const float d = v[e] * scale + smin - mean;
            s2 += d * d;
          }
      }
      st[which][0] = mn;
      st[which][1] = mx;
      st[which][2] = mean;
      st[which][3] = sqrtf(group_sum(s2, lpr) * invC);
    }
    if (rv && sub == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = st[0][k] - st[1][k];
        acc[k] += d * d;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = wave_sum(acc[k]);
  if (lane == 0)
    for (int k = 0; k < 4; ++k) part[wave][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 4) {
    const float t = ((part[0][threadIdx.x] + part[1][threadIdx.x]) +
                     part[2][threadIdx.x]) + part[3][threadIdx.x];
    if (ws) ws[(long long)blockIdx.x * 4 + threadIdx.x] = t;
    else atomicAdd(out + threadIdx.x, t);
  }
}

// the seven scalars train() returns, in one launch: [gen_loss, mean critic loss,
// mean penalty, metrics x 4] (wgan_gp.py:82-95 averages the critic's values over
// its n_critic updates)
__global__ void step_outputs_kernel(const float* __restrict__ gen_loss,
                                    const float* __restrict__ loss,
                                    const float* __restrict__ gp,
                                    const float* __restrict__ metrics, int n,
                                    float* __restrict__ out) {
  const int i = threadIdx.x;
  if (i == 0) out[0] = gen_loss[0];
  if (i == 1 || i == 2) {
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += (i == 1) ? loss[2 * k] : gp[k];
    out[i] = n > 0 ? s / n : 0.f;
  }
  if (i >= 3 && i < 7) out[i] = metrics[i - 3];
}

inline int rows_per_slot_for(long long rows, int rpw, int lo, int hi) {
  long long want = rows / ((long long)rpw * 8192);
  int rps = lo;
  while (rps * 2 <= want && rps * 2 <= hi) rps *= 2;
  return rps;
}

}  // namespace

#define S_(x) ((hipStream_t)(x))
#define U16(x) (reinterpret_cast<const uint16_t*>(x))
#define U16W(x) (reinterpret_cast<uint16_t*>(x))

extern "C" int cg_ln_lrelu_fwd(const void* y_pre, const float* gamma,
                               const float* beta, void* h, float* mean,
                               float* rstd, long long rows, int C, int Cp,
                               float eps, float alpha, void* stream) {
  if (Cp % 8 || Cp > 512 || C > Cp || rows < 1) return CG_EINVAL;
  int lpr = 1, l2 = 0;
  while (lpr * 8 < Cp) { lpr <<= 1; ++l2; }
  const int rpw = 64 / lpr;
  // rows per lane slot: enough waves to fill the chip (256 CUs x 32) before
  // each wave gets a longer sequential run; a multiple of the load batch
#ifndef CG_LN_RPS_LO
#define CG_LN_RPS_LO 4
#define CG_LN_RPS_HI 8
#endif
  static const bool pow2_only = getenv("CALCIUMGAN_LN_POW2") != nullptr;  // (A/B)
  if (lpr * 8 != Cp && Cp >= 64 && !pow2_only) {
    // a pitch whose 8-channel groups are not a power of two: 8 lanes per row
    const int gpl = (Cp / 8 + 7) / 8;
    const int rps = rows_per_slot_for(rows, 8, 2, 8);
    const dim3 grid(grid1d(rows, 4 * 8 * rps, 1LL << 31));
#define CG_LN8(G)                                                                 \
  case G:                                                                         \
    hipLaunchKernelGGL(ln_fwd8_kernel<G>, grid, dim3(kThreads), 0, S_(stream),    \
                       U16(y_pre), gamma, beta, U16W(h), mean, rstd, rows, C, Cp, \
                       eps, alpha, rps);                                          \
    break;
    switch (gpl) {
      CG_LN8(2) CG_LN8(3) CG_LN8(4) CG_LN8(5) CG_LN8(6) CG_LN8(7) CG_LN8(8)
      default: return CG_EINVAL;
    }
#undef CG_LN8
    CG_LAUNCH_CHECK();
  }
  const int rows_per_slot = rows_per_slot_for(rows, rpw, CG_LN_RPS_LO, CG_LN_RPS_HI);
  hipLaunchKernelGGL(ln_fwd_kernel,
                     dim3(grid1d(rows, 4 * rpw * rows_per_slot, 1LL << 31)),
                     dim3(kThreads), 0, S_(stream), U16(y_pre), gamma, beta,
                     U16W(h), mean, rstd, rows, C, Cp, eps, alpha, lpr, l2,
                     rows_per_slot);
  CG_LAUNCH_CHECK();
}

extern "C" long long cg_reduce_ws_elems(void) { return kReduceWsElems; }

extern "C" int cg_finish_defer(int on) {
  const int was = g_finish_defer ? 1 : 0;
  if (on && !g_finish_defer) g_finish_batch.n = 0;
  g_finish_defer = on != 0;
  return was;
}

extern "C" int cg_finish_flush(void* stream) {
  flush_finishes(S_(stream));
  g_finish_defer = false;
  CG_LAUNCH_CHECK();
}

extern "C" int cg_ln_lrelu_bwd(const void* dh, const void* h, const void* y_pre,
                               const float* mean, const float* rstd,
                               const float* gamma, void* dy, float* dgamma,
                               float* dbeta, float* dbias, long long rows,
                               int C, int Cp, float alpha, float* ws,
                               void* stream) {
  if (Cp % 8 || Cp > 512 || C > Cp || rows < 1) return CG_EINVAL;
  int lpr = 1, l2 = 0;
  while (lpr * 8 < Cp) { lpr <<= 1; ++l2; }
  const int rpw = 64 / lpr;
  // atomics: long runs (every block ends with 3*C global atomics onto the same
  // addresses, which serialise -- fewer, longer blocks); partial rows: a block
  // ends with one coalesced store, so shorter runs and more blocks in flight,
  // capped at kMaxParts blocks (the workspace's size)
  int rows_per_slot = ws ? rows_per_slot_for(rows, rpw, 4, 16)
                         : rows_per_slot_for(rows, rpw, 16, 64);
  if (ws) {
    while ((rows + 4ll * rpw * rows_per_slot - 1) / (4ll * rpw * rows_per_slot) >
           kMaxParts)
      rows_per_slot *= 2;
  }
  const unsigned blocks = grid1d(rows, 4 * rpw * rows_per_slot, 1LL << 31);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(kThreads), 0, S_(stream),
                     U16(dh), U16(h), U16(y_pre), mean, rstd, gamma, U16W(dy),
                     dgamma, dbeta, dbias, rows, C, Cp, alpha, lpr, l2,
                     rows_per_slot, ws);
  if (ws) {
    FinishArgs f;
    f.ws = ws; f.nparts = (int)blocks; f.ncol = Cp; f.pstride = 3ll * Cp;
    f.cstride = Cp; f.nout = dbias ? 3 : 2; f.scale = 1.f;
    f.out[0] = dgamma; f.out[1] = dbeta; f.out[2] = dbias;
    f.cvalid[0] = f.cvalid[1] = f.cvalid[2] = C;
    launch_finish(f, S_(stream));
  }
  CG_LAUNCH_CHECK();
}

// rows per block of the BatchNorm column sums: partial rows [2][Cp] within the
// workspace, at most kMaxParts blocks
static int bn_rows_per_block(long long rows, int Cp) {
  int rpb = 256;
  while (rpb < 4096 && rows / (rpb * 2) >= 512) rpb *= 2;
  while ((rows + rpb - 1) / rpb > kMaxParts ||
         ((rows + rpb - 1) / rpb) * 3ll * Cp > kReduceWsElems)
    rpb *= 2;
  return rpb;
}

extern "C" int cg_bn_stats(const void* y, long long rows, int C, int Cp,
                           float* mean, float* var, float* moving_mean,
                           float* moving_var, float momentum, float* ws,
                           void* stream) {
  if (!y || !mean || !var || !ws || Cp % 8 || Cp > 2048 || C > Cp || rows < 1 ||
      (!moving_mean != !moving_var))
    return CG_EINVAL;
  const int rpb = bn_rows_per_block(rows, Cp);
  const unsigned blocks = grid1d(rows, rpb, 1LL << 31);
  hipLaunchKernelGGL(bn_sums_kernel<0>, dim3(blocks), dim3(kThreads), 0,
                     S_(stream), U16(y), (const uint16_t*)nullptr,
                     (const uint16_t*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, rows, C, Cp, rpb, 0.f, 1.f, 0, ws);
  hipLaunchKernelGGL(bn_stats_finish_kernel, dim3((Cp + 63) / 64), dim3(1024), 0,
                     S_(stream), ws, (int)blocks, C, Cp, rows, rpb, mean, var,
                     moving_mean, moving_var, momentum);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_bn_apply(const void* y, const float* mean, const float* var,
                           const float* gamma, const float* beta, void* out,
                           long long rows, int C, int Cp, float eps, float alpha,
                           void* stream) {
  if (!y || !mean || !var || !gamma || !beta || !out || Cp % 8 || C > Cp ||
      rows < 1)
    return CG_EINVAL;
  const long long total8 = rows * Cp / 8;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid1d(total8, kThreads, 1LL << 31)),
                     dim3(kThreads), 0, S_(stream), U16(y), mean, var, gamma, beta,
                     U16W(out), C, Cp, eps, alpha, total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_bn_bwd(const void* dout, const void* h, const void* y,
                         const float* mean, const float* var, const float* gamma,
                         void* dy, float* dgamma, float* dbeta, long long rows,
                         int C, int Cp, float eps, float alpha, int act, float* ws,
                         void* stream) {
  if (!dout || !y || !mean || !var || !gamma || !dy || !dgamma || !dbeta || !ws ||
      (act && !h) || Cp % 8 || Cp > 2048 || C > Cp || rows < 1)
    return CG_EINVAL;
  const int rpb = bn_rows_per_block(rows, Cp);
  const unsigned blocks = grid1d(rows, rpb, 1LL << 31);
  hipLaunchKernelGGL(bn_sums_kernel<1>, dim3(blocks), dim3(kThreads), 0,
                     S_(stream), U16(y), U16(dout), U16(h), mean, var, rows, C, Cp,
                     rpb, eps, alpha, act, ws);
  FinishArgs f;
  f.ws = ws; f.nparts = (int)blocks; f.ncol = Cp; f.pstride = 2ll * Cp;
  f.cstride = Cp; f.nout = 2; f.scale = 1.f;
  f.out[0] = dbeta; f.out[1] = dgamma; f.out[2] = nullptr;
  f.cvalid[0] = f.cvalid[1] = C; f.cvalid[2] = 0;
  launch_finish(f, S_(stream));
  const long long total8 = rows * Cp / 8;
  hipLaunchKernelGGL(bn_bwd_apply_kernel,
                     dim3(grid1d(total8, kThreads, 1LL << 31)), dim3(kThreads), 0,
                     S_(stream), U16(dout), U16(h), U16(y), mean, var, gamma,
                     dgamma, dbeta, U16W(dy), C, Cp, eps, alpha, act,
                     1.f / (float)rows, total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_dense1_fwd(const void* h, const float* w, const float* bias,
                             float* out, int nB, int Lt, int C, int Cp,
                             void* stream) {
  const int F = Lt * Cp;
  if (Cp % 8 || C > Cp || nB < 1) return CG_EINVAL;
  if (F >= 8192)
    hipLaunchKernelGGL(dense1_fwd_kernel<1024>, dim3(nB), dim3(1024), 0, S_(stream),
                       U16(h), w, bias, out, F, C, Cp, (const float*)nullptr,
                       (uint16_t*)nullptr, 1, 1.f);
  else
    hipLaunchKernelGGL(dense1_fwd_kernel<256>, dim3(nB), dim3(256), 0, S_(stream),
                       U16(h), w, bias, out, F, C, Cp, (const float*)nullptr,
                       (uint16_t*)nullptr, 1, 1.f);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_dense1_fwd_bwd(const void* h, const float* w, const float* bias,
                                 float* out, const float* coef, void* delta,
                                 int nB, int Lt, int C, int Cp, int seg_size,
                                 float alpha, void* stream) {
  const int F = Lt * Cp;
  if (Cp % 8 || C > Cp || nB < 1 || seg_size < 1 || !coef || !delta)
    return CG_EINVAL;
  if (F >= 8192)
    hipLaunchKernelGGL(dense1_fwd_kernel<1024>, dim3(nB), dim3(1024), 0, S_(stream),
                       U16(h), w, bias, out, F, C, Cp, coef, U16W(delta), seg_size,
                       alpha);
  else
    hipLaunchKernelGGL(dense1_fwd_kernel<256>, dim3(nB), dim3(256), 0, S_(stream),
                       U16(h), w, bias, out, F, C, Cp, coef, U16W(delta), seg_size,
                       alpha);
  CG_LAUNCH_CHECK();
}

// gp_finalize + critic_loss in one launch (both are single-block reductions over
// the batch; the loss only needs the penalty this launch has just formed)
__global__ void gp_critic_loss_kernel(float* __restrict__ norm, float* __restrict__ gp,
                                      float* __restrict__ coef,
                                      const float* __restrict__ d_out,
                                      float* __restrict__ loss, int B, float scale,
                                      int squared, float coef_mul) {
  __shared__ float part[3][4];
  float s = 0.f, sr = 0.f, sf = 0.f;
  for (int b = threadIdx.x; b < B; b += kThreads) {
    float nv = norm[b];
    if (squared) {
      nv = sqrtf(nv);
      norm[b] = nv;
    }
    const float d = nv - 1.f;
    s += d * d;
    coef[b] = scale * 2.f * d / (B * nv) * coef_mul;
    sr += d_out[b];
    sf += d_out[B + b];
  }
  s = wave_sum(s);
  sr = wave_sum(sr);
  sf = wave_sum(sf);
  if ((threadIdx.x & 63) == 0) {
    part[0][threadIdx.x >> 6] = s;
    part[1][threadIdx.x >> 6] = sr;
    part[2][threadIdx.x >> 6] = sf;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float g = (part[0][0] + part[0][1] + part[0][2] + part[0][3]) / B;
    const float mr = (part[1][0] + part[1][1] + part[1][2] + part[1][3]) / B;
    const float mf = (part[2][0] + part[2][1] + part[2][2] + part[2][3]) / B;
    gp[0] = g;
    loss[0] = -mr + mf + scale * g;
    loss[1] = -mf;
  }
}

extern "C" int cg_gp_critic_loss(float* norm, float* gp, float* coef,
                                 const float* d_out, float* loss, int B,
                                 float penalty, int squared, float coef_mul,
                                 void* stream) {
  if (!norm || !gp || !coef || !d_out || !loss || B < 1) return CG_EINVAL;
  hipLaunchKernelGGL(gp_critic_loss_kernel, dim3(1), dim3(kThreads), 0, S_(stream),
                     norm, gp, coef, d_out, loss, B, penalty, squared, coef_mul);
  CG_LAUNCH_CHECK();
}

// dgp/dnorm per sample (x coef_mul), the one expression both halves of
// gp_loss_scale_kernel use
__device__ __forceinline__ float gp_coef(float nv, int B, float scale, float coef_mul) {
  const float d = nv - 1.f;
  return scale * 2.f * d / (B * nv) * coef_mul;
}
__device__ __forceinline__ float slot_sum(const float* __restrict__ ws, int P, int b) {
  float s = 0.f;
  for (int j = 0; j < P; ++j) s += ws[(long long)b * P + j];  // slot order
  return s;
}

// rowsumsq_finish + gp_critic_loss + scale_rows in one launch: the LAST block is
// gp_critic_loss_kernel on norms it forms from the slots; every other block
// scales 2048 elements of g by its sample's coefficient, formed from the same
// slots with the same expression (so the rows carry exactly coef[b]).
__global__ __launch_bounds__(kThreads) void gp_loss_scale_kernel(
    const float* __restrict__ ssq_ws, int P, float* __restrict__ norm,
    float* __restrict__ gp, float* __restrict__ coef,
    const float* __restrict__ d_out, float* __restrict__ loss, int B, float scale,
    float coef_mul, const uint16_t* __restrict__ g, uint16_t* __restrict__ dst,
    long long n, long long total8) {
  if (blockIdx.x + 1 == gridDim.x) {
    __shared__ float part[3][4];
    float s = 0.f, sr = 0.f, sf = 0.f;
    for (int b = threadIdx.x; b < B; b += kThreads) {
      const float nv = sqrtf(slot_sum(ssq_ws, P, b));
      norm[b] = nv;
      const float d = nv - 1.f;
      s += d * d;
      coef[b] = gp_coef(nv, B, scale, coef_mul);
      sr += d_out[b];
      sf += d_out[B + b];
    }
    s = wave_sum(s);
    sr = wave_sum(sr);
    sf = wave_sum(sf);
    if ((threadIdx.x & 63) == 0) {
      part[0][threadIdx.x >> 6] = s;
      part[1][threadIdx.x >> 6] = sr;
      part[2][threadIdx.x >> 6] = sf;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float gm = (part[0][0] + part[0][1] + part[0][2] + part[0][3]) / B;
      const float mr = (part[1][0] + part[1][1] + part[1][2] + part[1][3]) / B;
      const float mf = (part[2][0] + part[2][1] + part[2][2] + part[2][3]) / B;
      gp[0] = gm;
      loss[0] = -mr + mf + scale * gm;
      loss[1] = -mf;
    }
    return;
  }
  const long long idx = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total8) return;
  const long long i = idx * 8;
  const int b = (int)(i / n);
  const float c = gp_coef(sqrtf(slot_sum(ssq_ws, P, b)), B, scale, coef_mul);
  float v[8], o[8];
  load8(g + i, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = c * v[e];
  store8(dst + i, o);
}

extern "C" int cg_gp_loss_scale(const float* ssq_ws, int P, float* norm, float* gp,
                                float* coef, const float* d_out, float* loss,
                                int B, float penalty, float coef_mul,
                                const void* g, void* dst, long long n,
                                void* stream) {
  if (!ssq_ws || P < 1 || !norm || !gp || !coef || !d_out || !loss || B < 1)
    return CG_EINVAL;
  if ((g != nullptr) != (dst != nullptr) || (g && (n < 8 || n % 8))) return CG_EINVAL;
  const long long total8 = g ? (long long)B * n / 8 : 0;
  const unsigned blocks = (unsigned)((total8 + kThreads - 1) / kThreads) + 1;
  hipLaunchKernelGGL(gp_loss_scale_kernel, dim3(blocks), dim3(kThreads), 0,
                     S_(stream), ssq_ws, P, norm, gp, coef, d_out, loss, B, penalty,
                     coef_mul, U16(g), U16W(dst), n, total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_dense1_bwd(const float* w, const float* coef, const void* h,
                             void* delta, int nB, int Lt, int C, int Cp,
                             int seg_size, float alpha, void* stream) {
  const int F = Lt * Cp;
  if (Cp % 8 || C > Cp || nB < 1 || seg_size < 1) return CG_EINVAL;
  const long long total8 = (long long)nB * F / 8;
  hipLaunchKernelGGL(dense1_bwd_kernel, dim3(grid1d(total8, kThreads, 1LL << 31)),
                     dim3(kThreads), 0, S_(stream), w, coef, U16(h),
                     U16W(delta), F, C, Cp, seg_size, alpha, total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_dense1_wgrad(const void* x, const float* coef,
                               const float* bias_coef, float* dw, float* db,
                               int nB, int Lt, int C, int Cp, int seg_size,
                               float* ws, void* stream) {
  const int F = Lt * Cp;
  if (Cp % 8 || C > Cp || nB < 1 || seg_size < 1) return CG_EINVAL;
  const int gx = (F / 8 + kThreads - 1) / kThreads;
  int gy = 512 / gx;
  if (gy < 1) gy = 1;
  if (gy > nB) gy = nB;
  const int Fpad = gx * kThreads * 8;
  if (ws && (long long)gy * Fpad + gy > kReduceWsElems) return CG_EINVAL;
  hipLaunchKernelGGL(dense1_wgrad_kernel, dim3(gx, gy), dim3(kThreads), 0,
                     S_(stream), U16(x), coef, bias_coef, dw, db, nB, F, C,
                     Cp, seg_size, ws, Fpad);
  if (ws)
    hipLaunchKernelGGL(dense1_wgrad_finish, dim3((F + kThreads - 1) / kThreads),
                       dim3(kThreads), 0, S_(stream), ws, dw, db, gy, F, Fpad, C,
                       Cp, bias_coef ? 1 : 0);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_unshuffle_mask(const void* e, const void* h, void* delta,
                                 const int* shifts, int nB, int w, int Cp,
                                 int seg_size, float alpha, void* stream) {
  if (Cp % 8 || nB < 1 || w < 1 || seg_size < 1) return CG_EINVAL;
  const long long total8 = (long long)nB * w * Cp / 8;
  hipLaunchKernelGGL(unshuffle_mask_kernel,
                     dim3(grid1d(total8, kThreads, 1LL << 31)), dim3(kThreads),
                     0, S_(stream), U16(e), U16(h), U16W(delta), shifts, w, Cp,
                     seg_size, alpha, total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_unshuffle_fixup(const void* side, const void* h, void* delta,
                                  const int* shifts, int nB, int w, int Cp,
                                  int seg_size, int side_rows, float alpha,
                                  void* stream) {
  if (Cp % 8 || nB < 1 || w < 1 || seg_size < 1 || side_rows < 1 || !shifts ||
      2 * side_rows + 1 > w)
    return CG_EINVAL;
  const long long total8 = (long long)nB * side_rows * Cp / 8;
  hipLaunchKernelGGL(unshuffle_fixup_kernel,
                     dim3(grid1d(total8, kThreads, 1LL << 31)), dim3(kThreads),
                     0, S_(stream), U16(side), U16(h), U16W(delta), shifts, w,
                     Cp, seg_size, side_rows, alpha, total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_interp_pack(const float* real, const float* fake,
                              const float* alpha, void* x0, int B, int L, int C,
                              int Cr, int Cf, int Cp, int write_real,
                              void* stream) {
  if (Cp % 8 || C > Cp || C > Cr || C > Cf) return CG_EINVAL;
  const long long total8 = (long long)B * L * Cp / 8;
  hipLaunchKernelGGL(interp_pack_kernel,
                     dim3(grid1d(total8, kThreads, 1LL << 31)), dim3(kThreads),
                     0, S_(stream), real, fake, alpha, U16W(x0), B, L, C, Cr,
                     Cf, Cp, total8, write_real);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_cast_pad(const float* src, void* dst, long long rows, int C,
                           int Cs, int Cp, void* stream) {
  if (Cp % 8 || C > Cp || C > Cs) return CG_EINVAL;
  const long long total8 = rows * Cp / 8;
  hipLaunchKernelGGL(cast_pad_kernel, dim3(grid1d(total8, kThreads, 1LL << 31)),
                     dim3(kThreads), 0, S_(stream), src, U16W(dst), C, Cs, Cp,
                     total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_rownorm(const void* g, float* norm, int B, long long n,
                          float* ws, void* stream) {
  if (n % 8 || B < 1) return CG_EINVAL;
  int chunks = (int)((n / 8 + kThreads - 1) / kThreads);
  if (chunks > 64) chunks = 64;
  if (ws) {
    if ((long long)B * chunks > kReduceWsElems) return CG_EINVAL;
    hipLaunchKernelGGL(sumsq_kernel, dim3(chunks, B), dim3(kThreads), 0,
                       S_(stream), U16(g), norm, n, ws);
    hipLaunchKernelGGL(sqrt_sum_kernel, dim3((B + 255) / 256), dim3(256), 0,
                       S_(stream), ws, norm, B, chunks);
    CG_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(zero_f32_kernel, dim3((B + 255) / 256), dim3(256), 0,
                     S_(stream), norm, B);
  hipLaunchKernelGGL(sumsq_kernel, dim3(chunks, B), dim3(kThreads), 0,
                     S_(stream), U16(g), norm, n, (float*)nullptr);
  hipLaunchKernelGGL(sqrt_kernel, dim3((B + 255) / 256), dim3(256), 0,
                     S_(stream), norm, B);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_gp_finalize(float* norm, float* gp, float* coef, int B,
                              float scale, int squared, void* stream) {
  hipLaunchKernelGGL(gp_finalize_kernel, dim3(1), dim3(kThreads), 0, S_(stream),
                     norm, gp, coef, B, scale, squared);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_scale_rows(const void* g, const float* coef, void* a0, int B,
                             long long n, void* stream) {
  if (n % 8) return CG_EINVAL;
  const long long total8 = (long long)B * n / 8;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(grid1d(total8, kThreads, 1LL << 31)),
                     dim3(kThreads), 0, S_(stream), U16(g), coef, U16W(a0), n,
                     total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_critic_loss(const float* d_out, const float* gp, float penalty,
                              float* out, int B, void* stream) {
  hipLaunchKernelGGL(critic_loss_kernel, dim3(1), dim3(kThreads), 0, S_(stream),
                     d_out, gp, penalty, out, B);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_neg_mean(const float* d_out, float* out, int B, void* stream) {
  hipLaunchKernelGGL(neg_mean_kernel, dim3(1), dim3(kThreads), 0, S_(stream),
                     d_out, out, B);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_colsum(const void* x, float* out, long long rows, int C,
                         int Cp, float* ws, void* stream) {
  if (Cp % 8 || C > Cp || rows < 1) return CG_EINVAL;
  // enough rows per block to amortise the block's closing reduction, still
  // >= ~512 blocks on the large activations
  int rows_per_block = 256;
  while (rows_per_block < 4096 && rows / (rows_per_block * 2) >= 512)
    rows_per_block *= 2;
  if (ws) {
    // partial rows [block][Cp] must fit the workspace
    while ((rows + rows_per_block - 1) / rows_per_block > kMaxParts ||
           ((rows + rows_per_block - 1) / rows_per_block) * Cp > kReduceWsElems)
      rows_per_block *= 2;
  }
  const int slabs = (Cp + kThreads * 8 - 1) / (kThreads * 8);
  const unsigned blocks = grid1d(rows, rows_per_block, 1LL << 31);
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks, slabs), dim3(kThreads), 0,
                     S_(stream), U16(x), out, rows, C, Cp, rows_per_block, ws);
  if (ws) {
    FinishArgs f;
    f.ws = ws; f.nparts = (int)blocks; f.ncol = Cp; f.pstride = Cp;
    f.cstride = 0; f.nout = 1; f.scale = 1.f;
    f.out[0] = out; f.out[1] = f.out[2] = nullptr;
    f.cvalid[0] = C; f.cvalid[1] = f.cvalid[2] = 0;
    launch_finish(f, S_(stream));
  }
  CG_LAUNCH_CHECK();
}

extern "C" int cg_sigmoid_bwd(const void* dfake, const float* fake, void* dz,
                              long long rows, int C, int Cf, int Cp,
                              void* stream) {
  if (Cp % 8 || C > Cp || C > Cf) return CG_EINVAL;
  const long long total8 = rows * Cp / 8;
  hipLaunchKernelGGL(sigmoid_bwd_kernel,
                     dim3(grid1d(total8, kThreads, 1LL << 31)), dim3(kThreads),
                     0, S_(stream), U16(dfake), fake, U16W(dz), C, Cf, Cp,
                     total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_lrelu_bwd(const void* dh, const void* h, void* dpre,
                            long long n, float alpha, void* stream) {
  if (n % 8) return CG_EINVAL;
  const long long total8 = n / 8;
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid1d(total8, kThreads, 1LL << 31)),
                     dim3(kThreads), 0, S_(stream), U16(dh), U16(h), U16W(dpre),
                     alpha, total8);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_lrelu_mix(const void* h_a, const void* h_b, const float* mix,
                            void* out, int B, long long n_per_sample, float alpha,
                            void* stream) {
  if (n_per_sample % 8 || B < 1 || !(alpha > 0.f && alpha <= 1.f)) return CG_EINVAL;
  const long long per8 = n_per_sample / 8, total8 = per8 * B;
  hipLaunchKernelGGL(lrelu_mix_kernel, dim3(grid1d(total8, kThreads, 1LL << 31)),
                     dim3(kThreads), 0, S_(stream), U16(h_a), U16(h_b), mix,
                     U16W(out), per8, total8, alpha, 1.f / alpha);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_adam(float* p, const float* grad, float* m, float* v,
                       long long n, float lr_t, float beta1, float beta2,
                       float eps, float grad_scale, const float* lr_t_dev,
                       void* stream) {
  if (n < 1) return CG_EINVAL;
  hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n, kThreads, 1LL << 31)),
                     dim3(kThreads), 0, S_(stream), p, grad, m, v, n, lr_t,
                     beta1, beta2, eps, grad_scale, lr_t_dev);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_grad_finite(const float* grad, long long n, float* ls,
                              void* stream) {
  if (n < 4 || (n & 3)) return CG_EINVAL;
  hipLaunchKernelGGL(grad_finite_kernel, dim3(grid1d(n / 4, kThreads, 2048)),
                     dim3(kThreads), 0, S_(stream), grad, n / 4, ls);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_adam_scaled(float* p, const float* grad, float* m, float* v,
                              long long n, float lr, float beta1, float beta2,
                              float eps, float grad_scale, const float* ls,
                              void* stream) {
  if (n < 1 || !ls) return CG_EINVAL;
  hipLaunchKernelGGL(adam_ls_kernel, dim3(grid1d(n, kThreads, 1LL << 31)),
                     dim3(kThreads), 0, S_(stream), p, grad, m, v, n, lr, beta1,
                     beta2, eps, grad_scale, ls);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_loss_scale_update(float* ls, int growth_interval,
                                    void* stream) {
  if (!ls || growth_interval < 1) return CG_EINVAL;
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, S_(stream),
                     ls, (float)growth_interval);
  CG_LAUNCH_CHECK();
}

extern "C" int cg_signal_metrics(const float* real, const float* fake,
                                 float* out, long long rows, int C, int Cr,
                                 int Cf, float smin, float smax, float* ws,
                                 void* stream) {
  if (rows < 1 || C < 1 || Cr < C || Cf < C) return CG_EINVAL;
  int lpr = 1, l2 = 0;
  while (lpr * 8 < C && lpr < 64) { lpr <<= 1; ++l2; }
  const int rpw = 64 / lpr;
  // every block ends with 4 atomics on the same 4 addresses, which serialise
  // (or one partial row of 4 floats): ~1024 blocks on the large inputs
  int rows_per_slot = 4;
  while (rows_per_slot < 64 && rows / ((long long)4 * rpw * rows_per_slot) > 1024)
    rows_per_slot *= 2;
  if (ws) {
    while ((rows + 4ll * rpw * rows_per_slot - 1) / (4ll * rpw * rows_per_slot) >
           kMaxParts)
      rows_per_slot *= 2;
  }
  const unsigned blocks = grid1d(rows, 4 * rpw * rows_per_slot, 1LL << 31);
  hipLaunchKernelGGL(signal_metrics_kernel, dim3(blocks), dim3(kThreads), 0,
                     S_(stream), real, fake, out, rows, C, Cr, Cf, smin,
                     smax - smin, lpr, l2, rows_per_slot, ws);
  if (ws) {
    FinishArgs f;
    f.ws = ws; f.nparts = (int)blocks; f.ncol = 4; f.pstride = 4;
    f.cstride = 0; f.nout = 1; f.scale = 1.f / (float)rows;  // the means
    f.out[0] = out; f.out[1] = f.out[2] = nullptr;
    f.cvalid[0] = 4; f.cvalid[1] = f.cvalid[2] = 0;
    launch_finish(f, S_(stream));
  }
  CG_LAUNCH_CHECK();
}

extern "C" int cg_step_outputs(const float* gen_loss, const float* loss,
                               const float* gp, const float* metrics, int n,
                               float* out, void* stream) {
  if (!gen_loss || !loss || !gp || !metrics || !out || n < 0) return CG_EINVAL;
  hipLaunchKernelGGL(step_outputs_kernel, dim3(1), dim3(64), 0, S_(stream),
                     gen_loss, loss, gp, metrics, n, out);
  CG_LAUNCH_CHECK();
}
