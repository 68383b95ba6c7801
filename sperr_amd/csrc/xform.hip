// xform.hip -- floating-point stages of the chunk pipeline as HIP kernels for gfx950:
// conditioner (mean / constant test), CDF 9/7 lifting DWT and IDWT, max-reduce, mid-tread
// quantiser and its inverse, chunk gather/scatter.  Everything is fp64 with the reference's exact
// operation order; compile with -ffp-contract=off (every fused multiply-add of the canonical
// reference build is an explicit fma()).
//
// Reference behaviour restated (file:line under /root/reference):
//   src/Conditioner.cpp:10-64,119-163   strided sequential mean, constant test
//   src/CDF97.cpp:132-148,284-302,345-474,598-666   dyadic / wavelet-packet 3D transform
//   src/SPECK_FLT.cpp:282-301,311-399   q, quantise, inverse quantise
//   src/SPERR3D_OMP_C.cpp:236-261, src/SPERR3D_OMP_D.cpp:167-184   gather / scatter
#include <type_traits>

#include "xform.h"
#include "speck_dec.h"

namespace sperrhip {

// ------------------------------------------------------------------------------------------
// conditioner
// ------------------------------------------------------------------------------------------

// Strided mean (src/Conditioner.cpp:119-135): every stride is summed strictly sequentially in fp64
// (the order is part of the result), so one thread owns one stride -- but reading a stride per
// thread straight from HBM touches a different cache line per lane.  A workgroup therefore takes
// kSumStrides strides and stages them segment by segment through LDS: all threads load
// kSumStrides x kSumSeg samples with coalesced row reads through the chunk's gather map, then
// kSumStrides threads add their row of the tile in order.  Also tests "all samples equal".
constexpr int kSumStrides = 64, kSumSeg = 64;


template <typename T>
__global__ void __launch_bounds__(kThreads)
k_stride_sums(const T* __restrict__ vol, VolDesc vd, const ChunkGeom* geom, uint32_t cx,
              uint32_t cy, uint32_t cz, uint32_t nstrides, uint32_t ssz, double* strideMean,
              size_t strideMeanStride, CoderState* st, int want_range)
{
  __shared__ double tile[kSumStrides][kSumSeg + 1];
  __shared__ uint32_t sh_differs;
  double vmax = -INFINITY, vnegmax = -INFINITY;   // (PSNR mode) range of the chunk's samples
  const uint32_t c = blockIdx.y;
  const uint32_t s0 = blockIdx.x * kSumStrides;
  const ChunkGeom g = geom[c];
  const size_t vx = vd.dims[0], vy = vd.dims[1];
  const T first = vol[((size_t)g.org[2] * vy + g.org[1]) * vx + g.org[0]];
  if (threadIdx.x == 0)
    sh_differs = 0;
  double acc = 0.0;
  bool differs = false;
  (void)cz;
  for (uint32_t seg = 0; seg < ssz; seg += kSumSeg) {
    const uint32_t len = min((uint32_t)kSumSeg, ssz - seg);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < (uint32_t)(kSumStrides * kSumSeg); k += kThreads) {
      const uint32_t r = k / kSumSeg, j = k % kSumSeg;
      const uint32_t sidx = s0 + r;
      if (sidx < nstrides && j < len) {
        const uint32_t e = sidx * ssz + seg + j;      // sample index inside the chunk
        const uint32_t x = e % cx, q = e / cx;
        const uint32_t y = q % cy, z = q / cy;
        const T v = vol[((size_t)(g.org[2] + z) * vy + (g.org[1] + y)) * vx + g.org[0] + x];
        differs |= (v != first);
        tile[r][j] = (double)v;
        vmax = fmax(vmax, (double)v);
        vnegmax = fmax(vnegmax, -(double)v);
      }
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)kSumStrides && s0 + threadIdx.x < nstrides)
      for (uint32_t j = 0; j < len; j++)
        acc += tile[threadIdx.x][j];
  }
  if (differs)
    sh_differs = 1;
  __syncthreads();
  if (threadIdx.x < (uint32_t)kSumStrides && s0 + threadIdx.x < nstrides)
    strideMean[c * strideMeanStride + s0 + threadIdx.x] = acc / (double)ssz;
  if (threadIdx.x == 0 && sh_differs)
    st[c].not_const_flag = 1;
  if (want_range) {
    for (int d = 32; d > 0; d >>= 1) {
      vmax = fmax(vmax, __shfl_xor(vmax, d, 64));
      vnegmax = fmax(vnegmax, __shfl_xor(vnegmax, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMax(reinterpret_cast<unsigned long long*>(&st[c].vmaxKey), order_key(vmax));
      atomicMax(reinterpret_cast<unsigned long long*>(&st[c].vnegmaxKey), order_key(vnegmax));
    }
  }
}

// The same sums when a stride is a whole number of chunk rows (every power-of-two chunk): one
// thread per stride streams its rows with 16-byte loads, several in flight, and adds them in
// order.  Eight wavefronts per CU keep enough bytes in flight to follow HBM.
template <typename T>
__global__ void __launch_bounds__(64)
k_stride_sums_rows(const T* __restrict__ vol, VolDesc vd, const ChunkGeom* geom, uint32_t cx,
                   uint32_t cy, uint32_t nstrides, uint32_t ssz, double* strideMean,
                   size_t strideMeanStride, CoderState* st, int want_range)
{
  constexpr int V = 16 / sizeof(T);                 // samples per 16-byte load
  struct alignas(16) Pack { T v[V]; };
  const uint32_t c = blockIdx.y;
  const uint32_t sidx = blockIdx.x * 64 + threadIdx.x;
  const ChunkGeom g = geom[c];
  const size_t vx = vd.dims[0], vy = vd.dims[1];
  const T first = vol[((size_t)g.org[2] * vy + g.org[1]) * vx + g.org[0]];
  double acc = 0.0, vmax = -INFINITY, vnegmax = -INFINITY;
  bool differs = false;
  if (sidx < nstrides) {
    const uint32_t rows = ssz / cx, row0 = sidx * rows;   // chunk rows of this stride
    for (uint32_t r = 0; r < rows; r++) {
      const uint32_t y = (row0 + r) % cy, z = (row0 + r) / cy;
      const T* src = vol + ((size_t)(g.org[2] + z) * vy + (g.org[1] + y)) * vx + g.org[0];
      for (uint32_t x = 0; x < cx; x += 8 * V) {
        Pack pk[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
          pk[k] = *reinterpret_cast<const Pack*>(src + x + k * V);
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
          for (int e = 0; e < V; e++) {
            const T v = pk[k].v[e];
            differs |= (v != first);
            acc += (double)v;
            vmax = fmax(vmax, (double)v);
            vnegmax = fmax(vnegmax, -(double)v);
          }
      }
    }
    strideMean[c * strideMeanStride + sidx] = acc / (double)ssz;
  }
  if (__any(differs) && threadIdx.x == 0)
    st[c].not_const_flag = 1;
  if (want_range) {
    for (int d = 32; d > 0; d >>= 1) {
      vmax = fmax(vmax, __shfl_xor(vmax, d, 64));
      vnegmax = fmax(vnegmax, __shfl_xor(vnegmax, d, 64));
    }
    if (threadIdx.x == 0) {
      atomicMax(reinterpret_cast<unsigned long long*>(&st[c].vmaxKey), order_key(vmax));
      atomicMax(reinterpret_cast<unsigned long long*>(&st[c].vnegmaxKey), order_key(vnegmax));
    }
  }
}

// The stride means added up in the reference's order (src/Conditioner.cpp:137-163): one thread does the
// additions, but out of LDS, where the workgroup has put the means with coalesced loads (round 3: one
// thread reading them from global memory one dependent load at a time took 0.44 ms for 2048 strides,
// on the critical path of every compression call)
constexpr int kMeanStage = 2048;
template <typename T>
__global__ void __launch_bounds__(kThreads)
k_mean_finalize(const T* __restrict__ vol, VolDesc vd, const ChunkGeom* geom, uint32_t nstrides,
                const double* strideMean, size_t strideMeanStride, CoderState* st)
{
  __shared__ double stage[kMeanStage];
  const uint32_t c = blockIdx.x;
  const double* sm = strideMean + c * strideMeanStride;
  double total = 0.0;
  for (uint32_t base = 0; base < nstrides; base += kMeanStage) {
    const uint32_t n = min((uint32_t)kMeanStage, nstrides - base);
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < n; k += kThreads)
      stage[k] = sm[base + k];
    __syncthreads();
    if (threadIdx.x == 0)
      for (uint32_t k = 0; k < n; k++)
        total += stage[k];
  }
  if (threadIdx.x != 0)
    return;
  const ChunkGeom g = geom[c];
  const bool is_const = st[c].not_const_flag == 0;
  st[c].is_const = is_const ? 1u : 0u;
  if (is_const)  // the header stores the value itself (Conditioner.cpp:28-44)
    st[c].mean = (double)vol[((size_t)g.org[2] * vd.dims[1] + g.org[1]) * vd.dims[0] + g.org[0]];
  else
    st[c].mean = total / (double)nstrides;
}

// vals[i] = (double)vol[gather(i)] - mean
template <typename T>
__global__ void k_gather_condition(const T* __restrict__ vol, VolDesc vd, const ChunkGeom* geom,
                                   uint32_t cx, uint32_t cy, uint32_t n, double* vals,
                                   size_t valsStride, const CoderState* st)
{
  const uint32_t c = blockIdx.y;
  const ChunkGeom g = geom[c];
  const double mean = st[c].mean;
  const size_t vx = vd.dims[0], vy = vd.dims[1];
  double* out = vals + c * valsStride;
  for (uint32_t i = blockIdx.x * blockDim.x * 4 + threadIdx.x, k = 0; k < 4 && i < n;
       k++, i += blockDim.x) {
    const uint32_t x = i % cx, r = i / cx;
    const uint32_t y = r % cy, z = r / cy;
    const T v = vol[((size_t)(g.org[2] + z) * vy + (g.org[1] + y)) * vx + g.org[0] + x];
    out[i] = (double)v - mean;
  }
}

// out[scatter(i)] = (T)(vals[i] + mean), or the constant value (Conditioner.cpp:66-96)
template <typename T>
__global__ void k_scatter_uncondition(T* __restrict__ vol, VolDesc vd, const ChunkGeom* geom,
                                      uint32_t cx, uint32_t cy, uint32_t n, const double* vals,
                                      size_t valsStride, const CoderState* st)
{
  const uint32_t c = blockIdx.y;
  const ChunkGeom g = geom[c];
  const double mean = st[c].mean;
  const bool is_const = st[c].is_const != 0;
  const size_t vx = vd.dims[0], vy = vd.dims[1];
  const double* in = vals + c * valsStride;
  for (uint32_t i = blockIdx.x * blockDim.x * 4 + threadIdx.x, k = 0; k < 4 && i < n;
       k++, i += blockDim.x) {
    const uint32_t x = i % cx, r = i / cx;
    const uint32_t y = r % cy, z = r / cy;
    const double v = is_const ? mean : in[i] + mean;
    vol[((size_t)(g.org[2] + z) * vy + (g.org[1] + y)) * vx + g.org[0] + x] = (T)v;
  }
}

// ------------------------------------------------------------------------------------------
// CDF 9/7 lifting along one axis, LDS-staged line tiles
// ------------------------------------------------------------------------------------------
//
// A workgroup stages NL adjacent lines of length `len` in LDS as sm[pos * (NL+1) + line] with the
// samples already de-interleaved ([even | odd], src/CDF97.cpp:476-519), runs the four lifting
// steps + scaling with a barrier between dependent steps, and writes the lines back.  For the Y
// and Z axes the NL lines are NL consecutive x positions, so every global access is a coalesced
// NL*8-byte segment; for the X axis the lines are NL consecutive rows and lanes run along x.

extern __shared__ __attribute__((aligned(16))) char dyn_smem[];

// IO: 0 = in place on the fp64 chunk buffer; 1 / 2 = the pass also moves the chunk between the
// float / double VOLUME and the chunk buffer: the first forward pass reads the volume through the
// gather map and subtracts the mean (src/SPERR3D_OMP_C.cpp:236-261, Conditioner.cpp:48-50), the
// last inverse pass adds the mean, narrows and scatters (Conditioner.cpp:66-96,
// SPERR3D_OMP_D.cpp:167-184, SPERR_C_API.cpp:246-250).  Both passes cover the whole chunk.
constexpr int kLoadBatch = 8;

template <bool FORWARD, int IO>
__global__ void __launch_bounds__(kThreads)
k_lift_axis(double* vals, size_t valsStride, uint32_t cx, uint32_t cy, int axis, uint32_t rx,
            uint32_t ry, uint32_t rz, int NL, LiftConsts K, CoderState* st, void* volume,
            VolDesc vd, const ChunkGeom* geom, LiftFuse F)
{
  const uint32_t c = blockIdx.y;
  const bool is_const = st[c].is_const != 0;
  if (is_const && !(IO != 0 && !FORWARD))
    return;
  double* sm = reinterpret_cast<double*>(dyn_smem);
  double* buf = vals + c * valsStride;
  const uint32_t region[3] = {rx, ry, rz};
  const uint32_t bx = (!FORWARD && F.bufx) ? F.bufx : cx, by = (!FORWARD && F.bufy) ? F.bufy : cy;   // (compact buffer)
  const size_t stride[3] = {1, bx, (size_t)bx * by};
  const int ua = (axis == 0) ? 1 : 0;            // axis along which the NL lines are adjacent
  const int wa = (axis == 2) ? 1 : 2;            // remaining axis
  const uint32_t len = region[axis];
  const uint32_t ntu = (region[ua] + NL - 1) / NL;
  const uint32_t tu = blockIdx.x % ntu, tw = blockIdx.x / ntu;
  const uint32_t u0 = tu * NL;
  const uint32_t nl = min((uint32_t)NL, region[ua] - u0);  // valid lines in this tile
  double* tile = buf + (size_t)u0 * stride[ua] + (size_t)tw * stride[wa];
  const size_t sl = stride[axis], su = stride[ua];
  const int NLP = NL + 1;
  const uint32_t even_len = len - len / 2, odd_len = len / 2;
  const int tid = threadIdx.x;

  // volume address of chunk sample (line l, position p) of this tile
  using VT = typename std::conditional<IO == 1, float, double>::type;
  VT* vol = reinterpret_cast<VT*>(volume);
  size_t vbase = 0, vsl = 0, vsu = 0;
  double mean = 0.0;
  if (IO != 0) {
    const ChunkGeom g = geom[c];
    const size_t vstride[3] = {1, (size_t)vd.dims[0], (size_t)vd.dims[0] * vd.dims[1]};
    vbase = (size_t)g.org[0] * vstride[0] + (size_t)g.org[1] * vstride[1] +
            (size_t)g.org[2] * vstride[2] + (size_t)u0 * vstride[ua] + (size_t)tw * vstride[wa];
    vsl = vstride[axis];
    vsu = vstride[ua];
    mean = st[c].mean;
  }

  // ---- load: a thread issues kLoadBatch loads before it uses the first value (one at a time
  // leaves the pass waiting on HBM latency) ----
  using LT = typename std::conditional<(IO != 0 && FORWARD), VT, double>::type;
  // (l, p) of this tile as chunk coordinates, and whether a later pass of the forward order
  // touches the sample (LiftFuse)
  auto outside_inner = [&](uint32_t l, uint32_t p, size_t& idx) -> bool {
    uint32_t xyz[3];
    xyz[axis] = p;
    xyz[ua] = u0 + l;
    xyz[wa] = tw;
    idx = ((size_t)xyz[2] * cy + xyz[1]) * cx + xyz[0];
    return !(xyz[0] < F.inner[0] && xyz[1] < F.inner[1] && xyz[2] < F.inner[2]);
  };
  const bool dequant = !FORWARD && F.mode == 2 && !st[c].wide;
  const double fq = dequant ? st[c].q : 0.0;
  // value of a coefficient that became significant on the last decoded plane / the one before and
  // was never refined (k_inv_quantize)
  const uint32_t lastPl = (dequant && F.dst) ? (uint32_t)F.dst[c].lastPlane : 0u;
  const int scheme = (dequant && F.coefSigned != 0 && F.dst) ? coef_scheme(F.dst[c]) : 0;
  const uint32_t initNew = (1u << lastPl) + (1u << lastPl) - (1u << lastPl) / 2 - 1;
  const uint32_t initOld = lastPl < 31 ? (2u << lastPl) + (2u << lastPl) - (2u << lastPl) / 2 - 1 : 0u;
  auto fetch = [&](uint32_t l, uint32_t p) -> LT {
    if (IO != 0 && FORWARD)
      return (LT)vol[vbase + l * vsu + p * vsl];
    if (!FORWARD && dequant) {
      size_t idx;
      if (outside_inner(l, p, idx)) {   // k_inv_quantize for this one sample; the loads are independent
        if (scheme) {   // (the sign in bit 31, the value complete: LiftFuse::coefSigned)
          const uint32_t sv = F.coef[c * F.coefStride + idx];
          return (LT)(fq * (double)coef_scheme_mag(sv, scheme == 2) * ((sv >> 31) ? -1.0 : 1.0));
        }
        const uint32_t w = (uint32_t)(idx >> 6), sh = (uint32_t)(idx & 63);
        uint32_t v = F.coef[c * F.coefStride + idx];
        const uint64_t sgw = F.sign[c * F.signStride + w];
        const uint64_t mnw = F.sigNew ? F.sigNew[c * F.maskStride + w] : 0ull;
        const uint64_t mow = F.sigNew ? F.sigOld[c * F.maskStride + w] : 0ull;
        const uint32_t mn = (uint32_t)(mnw >> sh) & 1u, mo = (uint32_t)(mow >> sh) & 1u;
        const uint32_t fill = mn ? initNew : (mo ? initOld : 0u);
        v = v ? v : fill;
        return (LT)(fq * (double)v * (((sgw >> sh) & 1ull) ? 1.0 : -1.0));
      }
    }
    return (LT)tile[l * su + p * sl];
  };
  auto deposit = [&](uint32_t l, uint32_t p, LT v) {
    const uint32_t dst = FORWARD ? ((p & 1) ? even_len + (p >> 1) : (p >> 1)) : p;
    sm[dst * NLP + l] = (IO != 0 && FORWARD) ? (double)v - mean : (double)v;
  };
  if (!(is_const && !(IO != 0 && FORWARD))) {
    if (axis == 0) {  // lanes along the line: a wavefront moves 64 consecutive samples of a line
      const uint32_t nseg = (len + 63) / 64, nsg = nl * nseg, step = kThreads / 64;
      for (uint32_t sg0 = tid / 64; sg0 < nsg; sg0 += step * kLoadBatch) {
        LT v[kLoadBatch];
#pragma unroll
        for (int u = 0; u < kLoadBatch; u++) {
          const uint32_t sg = sg0 + u * step, l = sg / nseg, p = (sg % nseg) * 64 + tid % 64;
          v[u] = (sg < nsg && p < len) ? fetch(l, p) : (LT)0;
        }
#pragma unroll
        for (int u = 0; u < kLoadBatch; u++) {
          const uint32_t sg = sg0 + u * step, l = sg / nseg, p = (sg % nseg) * 64 + tid % 64;
          if (sg < nsg && p < len)
            deposit(l, p, v[u]);
        }
      }
    }
    else {            // lanes across the lines
      const uint32_t l = tid % NL, k0 = tid / NL, kg = kThreads / NL;
      if (l < nl)
        for (uint32_t p0 = k0; p0 < len; p0 += kg * kLoadBatch) {
          LT v[kLoadBatch];
#pragma unroll
          for (int u = 0; u < kLoadBatch; u++)
            v[u] = p0 + u * kg < len ? fetch(l, p0 + u * kg) : (LT)0;
#pragma unroll
          for (int u = 0; u < kLoadBatch; u++)
            if (p0 + u * kg < len)
              deposit(l, p0 + u * kg, v[u]);
        }
    }
  }
  __syncthreads();

  // ---- lift ----
  if (!is_const) {
    const uint32_t l = tid % NL, k0 = tid / NL, kg = kThreads / NL;
    double* E = sm + l;
    double* O = sm + (size_t)even_len * NLP + l;
#define EV(i) E[(size_t)(i) * NLP]
#define OD(i) O[(size_t)(i) * NLP]
    auto odd_step = [&](double k) {
      if (l < nl)
        for (uint32_t i = k0; i < odd_len; i += kg) {
          const uint32_t r = min(i + 1, even_len - 1);
          OD(i) = fma(k, EV(i) + EV(r), OD(i));
        }
      __syncthreads();
    };
    auto even_step = [&](double k) {
      if (l < nl)
        for (uint32_t i = k0; i < even_len; i += kg) {
          const uint32_t a = max(i, 1u) - 1, b = min(i, odd_len - 1);
          EV(i) = fma(k, OD(a) + OD(b), EV(i));
        }
      __syncthreads();
    };
    if (FORWARD) {  // src/CDF97.cpp:598-631
      odd_step(K.alpha);
      even_step(K.beta);
      odd_step(K.gamma);
      if (l < nl) {
        for (uint32_t i = k0; i < even_len; i += kg) {
          const uint32_t a = max(i, 1u) - 1, b = min(i, odd_len - 1);
          EV(i) = K.eps * fma(K.delta, OD(a) + OD(b), EV(i));
        }
      }
      __syncthreads();
      if (l < nl)
        for (uint32_t i = k0; i < odd_len; i += kg)
          OD(i) = (-K.inv_eps) * OD(i);
      __syncthreads();
    }
    else {          // src/CDF97.cpp:633-666
      if (l < nl)
        for (uint32_t i = k0; i < odd_len; i += kg)
          OD(i) = (-K.eps) * OD(i);
      __syncthreads();
      if (l < nl)
        for (uint32_t i = k0; i < even_len; i += kg) {
          const uint32_t a = max(i, 1u) - 1, b = min(i, odd_len - 1);
          const double t = K.delta * (OD(a) + OD(b));
          EV(i) = fma(EV(i), K.inv_eps, -t);
        }
      __syncthreads();
      odd_step(-K.gamma);
      even_step(-K.beta);
      odd_step(-K.alpha);
    }
#undef EV
#undef OD
  }

  // ---- store ----
  const bool wantMax = FORWARD && F.mode == 1 && !is_const;
  double vmax = 0.0;
  if (axis == 0) {
    const uint32_t nseg = (len + 63) / 64;
    for (uint32_t sg = tid / 64; sg < nl * nseg; sg += kThreads / 64) {
      const uint32_t l = sg / nseg, p = (sg % nseg) * 64 + tid % 64;
      if (p >= len)
        continue;
      const uint32_t src = FORWARD ? p : ((p & 1) ? even_len + (p >> 1) : (p >> 1));
      if (IO != 0 && !FORWARD)
        vol[vbase + l * vsu + p * vsl] = (VT)(is_const ? mean : sm[src * NLP + l] + mean);
      else {
        const double v = sm[src * NLP + l];
        tile[l * su + p] = v;
        size_t idx;
        if (wantMax && outside_inner(l, p, idx))
          vmax = fmax(vmax, fabs(v));
      }
    }
  }
  else {
    const uint32_t l = tid % NL, k0 = tid / NL, kg = kThreads / NL;
    if (l < nl)
      for (uint32_t p = k0; p < len; p += kg) {
        const uint32_t src = FORWARD ? p : ((p & 1) ? even_len + (p >> 1) : (p >> 1));
        if (IO != 0 && !FORWARD)
          vol[vbase + l * vsu + p * vsl] = (VT)(is_const ? mean : sm[src * NLP + l] + mean);
        else {
          const double v = sm[src * NLP + l];
          tile[l * su + p * sl] = v;
          size_t idx;
          if (wantMax && outside_inner(l, p, idx))
            vmax = fmax(vmax, fabs(v));
        }
      }
  }
  if (FORWARD && F.mode == 1) {   // (uniform: every wavefront reduces, one atomic each)
    for (int d = 32; d > 0; d >>= 1)
      vmax = fmax(vmax, __shfl_xor(vmax, d, 64));
    if ((tid & 63) == 0 && vmax > 0.0)  // non-negative doubles order like their bit patterns
      atomicMax(reinterpret_cast<unsigned long long*>(&st[c].maxabs),
                (unsigned long long)__double_as_longlong(vmax));
  }
}

// ------------------------------------------------------------------------------------------
// The two passes of the finest level that touch the volume, fused: forward x-lift then y-lift
// (volume -> fp64 buffer), inverse y-lift then x-lift (fp64 buffer -> volume).  A workgroup owns
// R rows of one z-slice and stages them in LDS together with a halo of 4 rows on each side: the
// four lifting steps reach one row further each, so rows y0-4 .. y0+R+3 determine the R rows
// exactly (the symmetric boundary rule only ever refers to rows of the slice itself).  Every
// sample goes through the very same operations as in k_lift_axis; halo rows are recomputed by the
// neighbouring tiles.  Saves one full read + write of the fp64 buffer per direction.
// The lifting itself runs in registers (lift16): the first version kept every intermediate in LDS
// and was bound by LDS bandwidth (about 40 eight-byte LDS accesses per sample against 12 bytes of
// HBM traffic); now a sample costs about 6.
// ------------------------------------------------------------------------------------------
constexpr int kXYHalo = 4;
constexpr int kXYThreads = 512;    // two workgroups per CU (LDS) = 4 waves per SIMD: 128 VGPRs each
constexpr int kStage = 16;         // global loads a lane has in flight while staging
constexpr int kSeg = 8;            // samples a thread produces per pass (plus 4 + 4 halo = 16 registers)

// position q of the whole-sample symmetric extension of a signal of n >= 2 samples
__device__ __forceinline__ uint32_t reflect_index(int q, int n)
{
  const int period = 2 * (n - 1);
  q = q < 0 ? -q : q;
  if (q >= period)
    q %= period;
  return (uint32_t)(q < n ? q : period - q);
}

// The lifting steps of QccWAVCDF97AnalysisSymmetric / SynthesisSymmetric (src/CDF97.cpp:598-666) on
// 16 consecutive samples r[0..16) of the symmetrically extended signal, r[0] at an even position:
// r[4..12) come out exactly as the whole-signal loops compute them -- a step reaches one sample to
// each side, the extension is symmetric about the first and the last sample and a + b == b + a, so
// the mirrored copies stay equal to the samples the reference's clamped indices refer to.
template <bool FORWARD>
__device__ __forceinline__ void lift16(double (&r)[16], const LiftConsts& K)
{
  if (FORWARD) {
#pragma unroll
    for (int k = 1; k <= 13; k += 2)
      r[k] = fma(K.alpha, r[k - 1] + r[k + 1], r[k]);
#pragma unroll
    for (int k = 2; k <= 12; k += 2)
      r[k] = fma(K.beta, r[k - 1] + r[k + 1], r[k]);
#pragma unroll
    for (int k = 3; k <= 11; k += 2)
      r[k] = fma(K.gamma, r[k - 1] + r[k + 1], r[k]);
#pragma unroll
    for (int k = 4; k <= 10; k += 2)
      r[k] = K.eps * fma(K.delta, r[k - 1] + r[k + 1], r[k]);
#pragma unroll
    for (int k = 5; k <= 11; k += 2)
      r[k] = (-K.inv_eps) * r[k];
  }
  else {
#pragma unroll
    for (int k = 1; k <= 15; k += 2)
      r[k] = (-K.eps) * r[k];
#pragma unroll
    for (int k = 2; k <= 14; k += 2) {
      const double t = K.delta * (r[k - 1] + r[k + 1]);
      r[k] = fma(r[k], K.inv_eps, -t);
    }
#pragma unroll
    for (int k = 3; k <= 13; k += 2)
      r[k] = fma(-K.gamma, r[k - 1] + r[k + 1], r[k]);
#pragma unroll
    for (int k = 4; k <= 12; k += 2)
      r[k] = fma(-K.beta, r[k - 1] + r[k + 1], r[k]);
#pragma unroll
    for (int k = 5; k <= 11; k += 2)
      r[k] = fma(-K.alpha, r[k - 1] + r[k + 1], r[k]);
  }
}

template <bool FORWARD, int IO>
__global__ void __launch_bounds__(kXYThreads)
k_lift_xy(double* vals, size_t valsStride, uint32_t cx, uint32_t cy, uint32_t cz, int R,
          LiftConsts K, const CoderState* st, void* volume, VolDesc vd, const ChunkGeom* geom)
{
  static_assert(IO == 1 || IO == 2, "float or double volume");
  using VT = typename std::conditional<IO == 1, float, double>::type;
  const uint32_t c = blockIdx.y;
  const bool is_const = st[c].is_const != 0;
  if (is_const && FORWARD)
    return;
  double* sm = reinterpret_cast<double*>(dyn_smem);
  const uint32_t ntile = (cy + R - 1) / R;
  const uint32_t z = blockIdx.x / ntile, y0 = (blockIdx.x % ntile) * R;
  const uint32_t ylo = y0 >= (uint32_t)kXYHalo ? y0 - kXYHalo : 0u;
  const uint32_t yhi = min(cy, y0 + R + kXYHalo);       // rows [ylo, yhi) are staged
  const uint32_t yend = min(cy, y0 + R);                // rows [y0, yend) are this tile's output
  const uint32_t nrow = yhi - ylo;
  const uint32_t RS = cx + 1;                           // row stride in LDS
  const uint32_t xe = cx - cx / 2, ye = cy - cy / 2;    // even samples of a row / of a column
  const ChunkGeom g = geom[c];
  VT* vol = reinterpret_cast<VT*>(volume);
  const size_t vsy = vd.dims[0], vsz = (size_t)vd.dims[0] * vd.dims[1];
  const size_t vbase = (size_t)(g.org[2] + z) * vsz + (size_t)g.org[1] * vsy + g.org[0];
  double* buf = vals + c * valsStride + (size_t)z * cx * cy;
  const double mean = st[c].mean;
  const uint32_t tid = threadIdx.x;

  if (is_const) {   // inverse only: the chunk is its constant
    for (uint32_t k = tid; k < (yend - y0) * cx; k += kXYThreads)
      vol[vbase + (size_t)(y0 + k / cx) * vsy + k % cx] = (VT)mean;
    return;
  }

  // ---- stage the rows.  In LDS everything is interleaved (sample x of row y at [y - ylo][x]).
  // One wavefront per row, 64 samples per load; a lane issues kStage loads before it touches the
  // first value, so a workgroup keeps its whole tile in flight (one load at a time per wavefront
  // left the kernel waiting on HBM latency).
  const uint32_t lane = tid & 63u, wave = tid >> 6, nwaves = kXYThreads / 64;
  {
    using ST = typename std::conditional<FORWARD, VT, double>::type;
    const uint32_t nxb = (cx + 63) / 64;
    const uint32_t rowsW = nrow > wave ? (nrow - wave + nwaves - 1) / nwaves : 0;
    const uint32_t items = rowsW * nxb;        // (row, 64-sample block) pairs of this wavefront
    uint32_t ri = 0, bi = 0;                   // pair m = (row wave + nwaves * ri, block bi)
    for (uint32_t m0 = 0; m0 < items; m0 += kStage) {
      ST v[kStage];
      uint32_t ri2 = ri, bi2 = bi;
#pragma unroll
      for (int u = 0; u < kStage; u++) {
        const uint32_t j = wave + nwaves * ri2, x = lane + 64 * bi2;
        v[u] = 0;
        if (m0 + u < items && x < cx) {
          const uint32_t y = ylo + j;
          if (FORWARD)
            v[u] = (ST)vol[vbase + (size_t)y * vsy + x];
          else   // the buffer holds low | high halves along x and along y
            v[u] = (ST)buf[(size_t)((y & 1) ? ye + (y >> 1) : (y >> 1)) * cx + ((x & 1) ? xe + (x >> 1) : (x >> 1))];
        }
        if (++bi2 == nxb) {
          bi2 = 0;
          ri2++;
        }
      }
#pragma unroll
      for (int u = 0; u < kStage; u++) {
        const uint32_t j = wave + nwaves * ri, x = lane + 64 * bi;
        if (m0 + u < items && x < cx)
          sm[j * RS + x] = FORWARD ? (double)v[u] - mean : (double)v[u];
        if (++bi == nxb) {
          bi = 0;
          ri++;
        }
      }
    }
  }
  __syncthreads();

  // ---- the passes.  A thread takes 8 consecutive samples of a row (or of a column) together
  // with a halo of 4 on each side into registers (lift16) and writes only its 8 results back.
  // Rows are independent in the x pass and columns in the y pass, so one round handles whole rows
  // (columns): all loads of a round precede its stores.
  const uint32_t nseg = (cx + kSeg - 1) / kSeg;
  auto lift_x = [&](uint32_t jlo, uint32_t jhi) {          // staged rows [jlo, jhi), in place
    // lanes of a wavefront take different rows: addresses differ by the (odd) row stride
    const uint32_t rpr = min(32u, (uint32_t)kXYThreads / nseg);
    const uint32_t l = tid % rpr, sg = tid / rpr;
    for (uint32_t rb = jlo; rb < jhi; rb += rpr) {
      const bool on = sg < nseg && rb + l < jhi;
      double* row = sm + (size_t)(rb + l) * RS;
      const int s = (int)(sg * kSeg);
      double r[16];
      if (on) {
        if (s >= 4 && s + 12 <= (int)cx) {
#pragma unroll
          for (int k = 0; k < 16; k++)
            r[k] = row[s - 4 + k];
        }
        else {
#pragma unroll
          for (int k = 0; k < 16; k++)
            r[k] = row[reflect_index(s - 4 + k, (int)cx)];
        }
        lift16<FORWARD>(r, K);
      }
      __syncthreads();
      if (on) {
#pragma unroll
        for (int k = 0; k < kSeg; k++)
          if ((uint32_t)(s + k) < cx)
            row[s + k] = r[4 + k];
      }
      __syncthreads();
    }
  };

  // y direction: thread = (column, 8 rows of the tile); lanes take consecutive columns
  const uint32_t nq = (yend - y0 + kSeg - 1) / kSeg;
  const uint32_t cpr = (uint32_t)kXYThreads / nq;          // columns per round
  auto load_column = [&](uint32_t x, uint32_t q, double (&r)[16]) {
    const int s = (int)(y0 + q * kSeg);
    const double* col = sm + x;
    if (s - 4 >= (int)ylo && s + 12 <= (int)yhi) {   // all 16 rows are staged rows of the slice
      const double* top = col + (size_t)(s - 4 - (int)ylo) * RS;
#pragma unroll
      for (int k = 0; k < 16; k++)
        r[k] = top[(size_t)k * RS];
    }
    else {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        // rows outside the slice mirror into it; what lies outside the staged rows is clamped
        // into them: it cannot reach the tile's own rows (see above)
        const uint32_t yy = min(max(reflect_index(s - 4 + k, (int)cy), ylo), yhi - 1);
        r[k] = col[(size_t)(yy - ylo) * RS];
      }
    }
  };

  if (FORWARD) {
    lift_x(0, nrow);
    // y pass straight from LDS into the buffer: low | high halves along both axes
    for (uint32_t xb = 0; xb < cx; xb += cpr) {
      const uint32_t x = xb + tid % cpr, q = tid / cpr;
      if (x < cx && q < nq) {
        double r[16];
        load_column(x, q, r);
        lift16<true>(r, K);
        const uint32_t dcol = (x & 1) ? xe + (x >> 1) : (x >> 1);
#pragma unroll
        for (int k = 0; k < kSeg; k++) {
          const uint32_t y = y0 + q * kSeg + k;
          if (y < yend)
            buf[(size_t)((y & 1) ? ye + (y >> 1) : (y >> 1)) * cx + dcol] = r[4 + k];
        }
      }
    }
  }
  else {
    for (uint32_t xb = 0; xb < cx; xb += cpr) {
      const uint32_t x = xb + tid % cpr, q = tid / cpr;
      const bool on = x < cx && q < nq;
      double r[16];
      if (on) {
        load_column(x, q, r);
        lift16<false>(r, K);
      }
      __syncthreads();
      if (on) {
#pragma unroll
        for (int k = 0; k < kSeg; k++) {
          const uint32_t y = y0 + q * kSeg + k;
          if (y < yend)
            sm[(size_t)(y - ylo) * RS + x] = r[4 + k];
        }
      }
    }
    __syncthreads();
    lift_x(y0 - ylo, yend - ylo);   // (only the tile's own rows go on)
    for (uint32_t y = y0 + wave; y < yend; y += nwaves) {
      VT* dstrow = vol + vbase + (size_t)y * vsy;
      const double* srow = sm + (size_t)(y - ylo) * RS;
      for (uint32_t x = lane; x < cx; x += 64)
        dstrow[x] = (VT)(srow[x] + mean);
    }
  }
}

// ------------------------------------------------------------------------------------------
// The THREE passes of the finest level fused with the volume access (round 3): the z pass joins
// k_lift_xy's x and y passes, so the fp64 chunk buffer is written once (forward) / the integer
// coefficients are read once (inverse) instead of a 16-byte-per-sample round trip through HBM for
// the z pass alone.  A brick that is whole along all three axes is the chunk, so the z direction
// is a SLIDING WINDOW: a workgroup owns kXYZRows rows (all of x) and marches through the slices;
// every slice is staged with a halo of four rows, x- and y-lifted in LDS exactly as in k_lift_xy,
// and its samples then enter per-position lifting PIPELINES along z held in registers.
//
// The four lifting steps along z as a pipeline (forward; x = input, d = odd, e = even samples):
//     d1[2m-1] = x[2m-1] + a (x[2m-2] + x[2m])         known when slice 2m arrives
//     e1[2m-2] = x[2m-2] + b (d1[2m-3] + d1[2m-1])
//     d2[2m-3] = d1[2m-3] + g (e1[2m-4] + e1[2m-2])    -> high[m-2] = -1/eps d2[2m-3]
//     e2[2m-4] = eps (e1[2m-4] + d (d2[2m-5] + d2[2m-3]))   -> low[m-2]
// so five values per position (x[2m], x[2m+1], d1[2m-1], e1[2m-2], d2[2m-3]) carry everything;
// the ends follow the reference's clamped indices (src/CDF97.cpp:598-666: a = max(i,1)-1,
// b = min(i, odd_len-1), r = min(i+1, even_len-1)), written out below.  Every sample goes through
// the very same operations, in the same order, as in k_lift_axis: bit-identical.
// The inverse runs the mirror image: pairs (low[m], high[m]) enter, four values per position stay
// (o1[m], e1[m], o2[m-1], e2[m-1]), slices 2m-3 and 2m-2 leave and go through the y and x passes.
// ------------------------------------------------------------------------------------------
// (the pipelines live in registers: the forward kernel fits 16 wavefronts of 128 VGPRs, the inverse one,
//  which inverts the halo rows along z too, as well -- with a few spills outside its main loop)
#ifndef XYZ_INV_THREADS
#define XYZ_INV_THREADS 1024
#endif
#ifndef XYZ_INV_PREFETCH
#define XYZ_INV_PREFETCH 2   // 0: the loads of rounds 1-3 (kept for A/B builds, tools/build_variant.sh)
#endif
#ifndef XYZ_FWD_THREADS
#define XYZ_FWD_THREADS 1024
#endif
constexpr int kXYZThreadsF = XYZ_FWD_THREADS, kXYZThreadsI = XYZ_INV_THREADS;
constexpr int kXYZRows = 16;
constexpr int kXYZStaged = kXYZRows + 2 * kXYHalo;   // LDS rows of a slice
constexpr int kXYZPosF = (4096 + kXYZThreadsF - 1) / kXYZThreadsF;    // z pipelines per thread, forward: kXYZRows * cx <= 4096
constexpr int kXYZStageF = (6144 + kXYZThreadsF - 1) / kXYZThreadsF;  // staged samples per thread, forward: kXYZStaged * cx <= 6144
constexpr int kXYZPosI = 6144 / kXYZThreadsI;   // inverse: kXYZStaged * cx <= 6144
#ifndef XYZ_INV_GROUP
#define XYZ_INV_GROUP 3
#endif
constexpr int kXYZGroupI = XYZ_INV_GROUP;       // positions whose loads are in flight together
constexpr int kXYZBoxSlots = 48;                // inverse, sign-in-word kernel: 256-byte rows of box samples on their way in (12 rows x 4)

// LDS layout of a slice: kXYZStaged rows; row r holds row reflect_index(y0 - 4 + r, cy) of the slice
// (the four rows above and below the tile -- mirrored at the ends of the slice, so the first and the
// last tile need nothing special), each row as [4 mirrored samples | cx samples | 4 mirrored samples]:
// lift16 then never has to reflect an index along x or y.
__host__ __device__ inline uint32_t xyz_row_stride(uint32_t cx)
{
  return ((cx + 7u) & ~7u) + 9u;   // (odd: rows start in different banks)
}

// A workgroup barrier that waits for the LDS traffic only.  __syncthreads() also waits for every global
// load and store in flight (s_waitcnt vmcnt(0)): here those are the NEXT slice's samples on their way
// in and the last slice's results on their way out, which no other thread of the workgroup looks at.
#define XYZ_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// sample x of LDS row `row`, with its mirror images in the row's apron
__device__ __forceinline__ void xyz_put(double* row, uint32_t x, uint32_t cx, double v)
{
  row[4 + x] = v;
  if (x - 1u < 4u)
    row[4 - x] = v;
  if (cx - 2u - x < 4u)
    row[2 * cx + 2 - x] = v;   // 4 + (cx - 1) + ((cx - 1) - x)
}

// x pass: rows [jlo, jhi) of `src` (aprons filled) -> the same rows of `dst` (another buffer: no barrier
// inside); aprons of dst are filled when `apron`
template <bool FORWARD, int NT>
__device__ __forceinline__ void xyz_lift_x(const double* src, double* dst, uint32_t RS, uint32_t cx, uint32_t jlo,
                                           uint32_t jhi, uint32_t tid, const LiftConsts& K)
{
  const uint32_t nseg = (cx + kSeg - 1) / kSeg;
  const uint32_t ntask = (jhi - jlo) * nseg;
#pragma unroll 1
  for (uint32_t t = tid; t < ntask; t += NT) {
    // lanes of a wavefront take different rows: their addresses differ by the (odd) row stride
    const uint32_t rows = jhi - jlo, sg = t / rows, rb = jlo + (t - sg * rows);
    const double* row = src + (size_t)rb * RS + sg * kSeg;
    double r[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
      r[k] = row[k];
    lift16<FORWARD>(r, K);
    double* out = dst + (size_t)rb * RS + 4 + sg * kSeg;
#pragma unroll
    for (int k = 0; k < kSeg; k++)
      if (sg * kSeg + k < cx)
        out[k] = r[4 + k];
  }
}

// y pass: tile rows (LDS rows 4 .. 4 + nt) of `src` -> the same rows of `dst`
template <bool FORWARD, bool APRON, int NT>
__device__ __forceinline__ void xyz_lift_y(const double* src, double* dst, uint32_t RS, uint32_t cx, uint32_t nt,
                                           uint32_t tid, const LiftConsts& K)
{
  const uint32_t nq = (nt + kSeg - 1) / kSeg;
  const uint32_t ntask = nq * cx;
#pragma unroll 1
  for (uint32_t t = tid; t < ntask; t += NT) {
    const uint32_t q = t / cx, x = t - q * cx;   // lanes take consecutive columns
    const double* top = src + (size_t)(q * kSeg) * RS + 4 + x;
    double r[16];
#pragma unroll
    for (int k = 0; k < 16; k++)
      r[k] = top[(size_t)k * RS];
    lift16<FORWARD>(r, K);
#pragma unroll
    for (int k = 0; k < kSeg; k++) {
      if (q * kSeg + k < nt) {
        double* row = dst + (size_t)(4 + q * kSeg + k) * RS;
        if (APRON)
          xyz_put(row, x, cx, r[4 + k]);
        else
          row[4 + x] = r[4 + k];
      }
    }
  }
}

template <int IO>
__global__ void __launch_bounds__(kXYZThreadsF) __attribute__((amdgpu_waves_per_eu(kXYZThreadsF / 256, kXYZThreadsF / 256)))
k_lift_xyz_fwd(double* vals, size_t valsStride, uint32_t cx, uint32_t cy, uint32_t cz, LiftConsts K,
               CoderState* st, const void* volume, VolDesc vd, const ChunkGeom* geom, int wantMax,
               uint32_t in0, uint32_t in1, uint32_t in2, uint32_t nseg)
{
  static_assert(IO == 1 || IO == 2, "float or double volume");
  using VT = typename std::conditional<IO == 1, float, double>::type;
  const uint32_t c = blockIdx.y;
  if (st[c].is_const != 0)
    return;
  const uint32_t tid = threadIdx.x;
  // A small batch has too few tiles for the device: the slices are then dealt to `nseg` workgroups
  // per tile.  Segment g emits what the even slices 2m, m in [mA, mB), complete (the last one also
  // the end of the lines); a pipeline's output depends on the ten slices before it, so the segment
  // starts that much earlier and throws away what comes out before its own part.
  const uint32_t seg = blockIdx.x % nseg, npairsAll = (cz + 1) / 2;   // even slices 0, 2, ...: m < npairsAll
  const uint32_t mA = seg == 0 ? 0u : (uint32_t)((uint64_t)npairsAll * seg / nseg);
  const uint32_t mB = seg + 1 == nseg ? npairsAll : (uint32_t)((uint64_t)npairsAll * (seg + 1) / nseg);
  const uint32_t zFirst = mA >= 6 ? 2 * (mA - 6) : 0u;          // first slice this workgroup lifts
  const uint32_t zEnd = seg + 1 == nseg ? cz : 2 * mB - 1;       // one past its last slice
  const uint32_t y0 = (blockIdx.x / nseg) * kXYZRows;
  const uint32_t nt = min((uint32_t)kXYZRows, cy - y0);   // rows of this tile
  const uint32_t RS = xyz_row_stride(cx);
  // two staging buffers that swap roles from pass to pass (three barriers per slice instead of six);
  // (offsets, not an array of pointers: the compiler must see that these are LDS addresses)
  double* sm = reinterpret_cast<double*>(dyn_smem);
  const uint32_t bufN = (uint32_t)kXYZStaged * RS;
  const uint32_t xe = cx - cx / 2, ye = cy - cy / 2, ze = cz - cz / 2;
  const VT* vol = reinterpret_cast<const VT*>(volume);
  const size_t vsy = vd.dims[0], vsz = (size_t)vd.dims[0] * vd.dims[1];
  const VT* volc = vol + ((size_t)geom[c].org[2] * vsz + (size_t)geom[c].org[1] * vsy + geom[c].org[0]);
  double* buf = vals + c * valsStride;
  const size_t sliceN = (size_t)cx * cy;
  const double mean = st[c].mean;

  // staging map: value k of this thread is sample x of LDS row j = row ysrc of the slice:
  // j << 27 | ysrc << 12 | x
  const uint32_t nstage = (uint32_t)kXYZStaged * cx;
  uint32_t pk[kXYZStageF];
#pragma unroll
  for (int k = 0; k < kXYZStageF; k++) {
    const uint32_t q = tid + (uint32_t)k * kXYZThreadsF;
    const uint32_t j = q / cx, x = q - j * cx;
    pk[k] = (j << 27) | (reflect_index((int)y0 - kXYHalo + (int)j, (int)cy) << 12) | x;
  }
  // z pipelines: position k of this thread is (row y0 + row, column col) of the tile
  const uint32_t npos = nt * cx;
  uint32_t srow[kXYZPosF], ooff[kXYZPosF];   // (LDS row << 16 | column), offset in a slice of the chunk buffer
  uint32_t outerMask = 0;   // bit k: position k lies outside the next level's box along x or y
#pragma unroll
  for (int k = 0; k < kXYZPosF; k++) {
    const uint32_t q = tid + (uint32_t)k * kXYZThreadsF;
    const uint32_t row = q / cx, col = q - row * cx, y = y0 + row;
    srow[k] = ((row + kXYHalo) << 16) | col;
    const uint32_t drow = (y & 1) ? ye + (y >> 1) : (y >> 1), dcol = (col & 1) ? xe + (col >> 1) : (col >> 1);
    ooff[k] = drow * cx + dcol;
    if (!(dcol < in0 && drow < in1))
      outerMask |= 1u << k;
  }
  double sxe[kXYZPosF], sxo[kXYZPosF], d1p[kXYZPosF], e1p[kXYZPosF], d2p[kXYZPosF];
#pragma unroll
  for (int k = 0; k < kXYZPosF; k++)
    sxe[k] = sxo[k] = d1p[k] = e1p[k] = d2p[k] = 0.0;
  double vmax = 0.0;
  auto emit = [&](int k, uint32_t zp, double v) {   // sample (ooff, zp) of the transformed chunk
    buf[(size_t)zp * sliceN + ooff[k]] = v;
    if (wantMax && (((outerMask >> k) & 1u) || zp >= in2))
      vmax = fmax(vmax, fabs(v));
  };

  VT pre[kXYZStageF];
  auto issue = [&](uint32_t z) {
    const VT* src = volc + (size_t)z * vsz;
#pragma unroll
    for (int k = 0; k < kXYZStageF; k++)
      pre[k] = (tid + (uint32_t)k * kXYZThreadsF) < nstage
                   ? src[(size_t)((pk[k] >> 12) & 0x7fffu) * vsy + (pk[k] & 0xfffu)] : (VT)0;
  };
  issue(zFirst);
  for (uint32_t z = zFirst; z < zEnd; z++) {
    // (what the addresses below are made of goes through an empty asm once per slice: otherwise the
    //  compiler computes every one of them before the loop and keeps -- spills -- some 200 values)
    uint32_t RSv = RS, tidv = tid;
    asm volatile("" : "+s"(RSv), "+v"(tidv));
#pragma unroll
    for (int k = 0; k < kXYZStageF; k++)
      asm volatile("" : "+v"(pk[k]));
#pragma unroll
    for (int k = 0; k < kXYZPosF; k++)
      asm volatile("" : "+v"(srow[k]), "+v"(ooff[k]));
    double* A = sm + ((z & 1) ? bufN : 0u);
    double* B = sm + ((z & 1) ? 0u : bufN);
#pragma unroll
    for (int k = 0; k < kXYZStageF; k++)
      if ((tid + (uint32_t)k * kXYZThreadsF) < nstage)
        xyz_put(A + (pk[k] >> 27) * RSv, pk[k] & 0xfffu, cx, (double)pre[k] - mean);
    if (z + 1 < zEnd)
      issue(z + 1);   // (in flight while this slice is lifted)
    XYZ_LDS_BARRIER();
    xyz_lift_x<true, kXYZThreadsF>(A, B, RSv, cx, 0, kXYZStaged, tidv, K);
    XYZ_LDS_BARRIER();
    xyz_lift_y<true, false, kXYZThreadsF>(B, A, RSv, cx, nt, tidv, K);
    XYZ_LDS_BARRIER();   // (A's tile rows: this slice after x and y; the next slice is staged into B)
    const uint32_t m = z >> 1;
    if (z & 1) {
#pragma unroll
      for (int k = 0; k < kXYZPosF; k++)
        if ((tid + (uint32_t)k * kXYZThreadsF) < npos)
          sxo[k] = A[(srow[k] >> 16) * RSv + 4 + (srow[k] & 0xffffu)];
    }
    else {
#pragma unroll
      for (int k = 0; k < kXYZPosF; k++) {
        if ((tid + (uint32_t)k * kXYZThreadsF) >= npos)
          continue;
        const double v = A[(srow[k] >> 16) * RSv + 4 + (srow[k] & 0xffffu)];
        if (m >= 1) {
          const double d1n = fma(K.alpha, sxe[k] + v, sxo[k]);                          // d1[2m-1]
          const double e1n = fma(K.beta, (m == 1 ? d1n : d1p[k]) + d1n, sxe[k]);        // e1[2m-2]
          if (m >= 2) {
            const double d2n = fma(K.gamma, e1p[k] + e1n, d1p[k]);                      // d2[2m-3]
            const double e2 = K.eps * fma(K.delta, (m == 2 ? d2n : d2p[k]) + d2n, e1p[k]);   // e2[2m-4]
            if (m >= mA) {   // (before that: the segment's run-up)
              emit(k, m - 2, e2);
              emit(k, ze + m - 2, (-K.inv_eps) * d2n);
            }
            d2p[k] = d2n;
          }
          d1p[k] = d1n;
          e1p[k] = e1n;
        }
        sxe[k] = v;
      }
    }
  }
  // ---- the end of the lines: the samples still in the pipelines (the last segment's to emit)
#pragma unroll
  for (int k = 0; k < kXYZPosF; k++) {
    if ((tid + (uint32_t)k * kXYZThreadsF) >= npos || seg + 1 != nseg)
      continue;
    if ((cz & 1) == 0) {   // the last sample is odd: x[2M+1], M = cz / 2 - 1
      const uint32_t M = cz / 2 - 1;
      const double d1L = fma(K.alpha, sxe[k] + sxe[k], sxo[k]);          // d1[2M+1]
      const double e1M = fma(K.beta, d1p[k] + d1L, sxe[k]);              // e1[2M]
      const double d2a = fma(K.gamma, e1p[k] + e1M, d1p[k]);             // d2[2M-1]
      const double e2a = K.eps * fma(K.delta, d2p[k] + d2a, e1p[k]);     // e2[2M-2]
      const double d2L = fma(K.gamma, e1M + e1M, d1L);                   // d2[2M+1]
      const double e2L = K.eps * fma(K.delta, d2a + d2L, e1M);           // e2[2M]
      emit(k, M - 1, e2a);
      emit(k, ze + M - 1, (-K.inv_eps) * d2a);
      emit(k, M, e2L);
      emit(k, ze + M, (-K.inv_eps) * d2L);
    }
    else {                 // the last sample is even: x[2M], M = cz / 2 (its slice went through the loop)
      const uint32_t M = cz / 2;
      const double e1M = fma(K.beta, d1p[k] + d1p[k], sxe[k]);           // e1[2M]
      const double d2a = fma(K.gamma, e1p[k] + e1M, d1p[k]);             // d2[2M-1]
      const double e2a = K.eps * fma(K.delta, d2p[k] + d2a, e1p[k]);     // e2[2M-2]
      const double e2L = K.eps * fma(K.delta, d2a + d2a, e1M);           // e2[2M]
      emit(k, M - 1, e2a);
      emit(k, ze + M - 1, (-K.inv_eps) * d2a);
      emit(k, M, e2L);
    }
  }
  if (wantMax) {
    for (int d = 32; d > 0; d >>= 1)
      vmax = fmax(vmax, __shfl_xor(vmax, d, 64));
    if ((tid & 63) == 0 && vmax > 0.0)
      atomicMax(reinterpret_cast<unsigned long long*>(&st[c].maxabs),
                (unsigned long long)__double_as_longlong(vmax));
  }
}

// SG: the coefficients carry their sign where the chunk allows it (LiftFuse::coefSigned, coef_scheme), a kernel of its own -- with both fast paths in
// one function the register allocator spilled inside the slice loop (88 registers against 38) and the kernel took 5.4 ms
// instead of 3.7
template <int IO, bool SG>
__global__ void __launch_bounds__(kXYZThreadsI) __attribute__((amdgpu_waves_per_eu(kXYZThreadsI / 256, kXYZThreadsI / 256)))
k_lift_xyz_inv(const double* vals, size_t valsStride, uint32_t cx, uint32_t cy, uint32_t cz, LiftConsts K,
               const CoderState* st, void* volume, VolDesc vd, const ChunkGeom* geom, LiftFuse F, uint32_t nseg)
{
  static_assert(IO == 1 || IO == 2, "float or double volume");
  using VT = typename std::conditional<IO == 1, float, double>::type;
  const uint32_t c = blockIdx.y;
  const uint32_t tid = threadIdx.x;
  // (small batches: `nseg` workgroups per tile share the slices, see k_lift_xyz_fwd; segment g finishes
  //  the slices that the pairs [mA, mB) complete, the last one also the end of the lines, and runs
  //  its pipelines six pairs ahead of that)
  const uint32_t seg = blockIdx.x % nseg, npairsAll = cz / 2;
  const uint32_t mA = seg == 0 ? 0u : (uint32_t)((uint64_t)npairsAll * seg / nseg);
  const uint32_t mB = seg + 1 == nseg ? npairsAll : (uint32_t)((uint64_t)npairsAll * (seg + 1) / nseg);
  const uint32_t mFirst = mA >= 6 ? mA - 6 : 0u;
  const uint32_t y0 = (blockIdx.x / nseg) * kXYZRows;
  const uint32_t nt = min((uint32_t)kXYZRows, cy - y0);
  const uint32_t RS = xyz_row_stride(cx);
  double* sm = reinterpret_cast<double*>(dyn_smem);
  const uint32_t bufN = (uint32_t)kXYZStaged * RS;
  const uint32_t xe = cx - cx / 2, ye = cy - cy / 2, ze = cz - cz / 2;
  VT* vol = reinterpret_cast<VT*>(volume);
  const size_t vsy = vd.dims[0], vsz = (size_t)vd.dims[0] * vd.dims[1];
  VT* volc = vol + ((size_t)geom[c].org[2] * vsz + (size_t)geom[c].org[1] * vsy + geom[c].org[0]);
  const double* buf = vals + c * valsStride;
  const size_t sliceN = (size_t)cx * cy;
  const uint32_t bufx = F.bufx ? F.bufx : cx;                       // the chunk buffer may be compact:
  const size_t bufSlice = (size_t)bufx * (F.bufy ? F.bufy : cy);   // only the next level's box
  const double mean = st[c].mean;
  const uint32_t lane = tid & 63u, wave = tid >> 6, nwaves = kXYZThreadsI / 64;

  if (st[c].is_const != 0) {   // the chunk is its constant
    for (uint32_t z = seg; z < cz; z += nseg)
      for (uint32_t k = tid; k < nt * cx; k += kXYZThreadsI)
        volc[(size_t)z * vsz + (size_t)(y0 + k / cx) * vsy + k % cx] = (VT)mean;
    return;
  }

  // position k of this thread: sample x of LDS row j = row ysrc of a slice (j << 27 | ysrc << 12 | x);
  // in the transformed chunk that is sample (dcol, drow): low | high halves along x and y
  const uint32_t npos = (uint32_t)kXYZStaged * cx;
  uint32_t pk[kXYZPosI];
  uint32_t innerMask = 0;   // bit k: inside the next level's box along x and y
#pragma unroll
  for (int k = 0; k < kXYZPosI; k++) {
    const uint32_t q = tid + (uint32_t)k * kXYZThreadsI;
    const uint32_t j = q / cx, x = q - j * cx, y = reflect_index((int)y0 - kXYHalo + (int)j, (int)cy);
    const uint32_t drow = (y & 1) ? ye + (y >> 1) : (y >> 1), dcol = (x & 1) ? xe + (x >> 1) : (x >> 1);
    pk[k] = (j << 27) | (y << 12) | x;
    if (dcol < F.inner[0] && drow < F.inner[1])
      innerMask |= 1u << k;
  }
  const bool dequant = F.mode == 2 && !st[c].wide;
  const double fq = dequant ? st[c].q : 0.0;
  const uint32_t* coef = F.coef + c * F.coefStride;
  const uint64_t* sign = F.sign + c * F.signStride;
  // a coefficient that became significant on the last decoded plane / the one before and was never
  // refined still holds 0: its value comes from the decoder's masks (k_inv_quantize, k_lift_axis)
  const bool haveMasks = dequant && F.sigNew != nullptr && F.dst != nullptr;
  const uint64_t* mNew = haveMasks ? F.sigNew + c * F.maskStride : sign;
  const uint64_t* mOld = haveMasks ? F.sigOld + c * F.maskStride : sign;
  const uint32_t lastPl = haveMasks ? (uint32_t)F.dst[c].lastPlane : 0u;
  const uint32_t initNew = haveMasks ? (1u << lastPl) + (1u << lastPl) - (1u << lastPl) / 2 - 1 : 0u;
  const uint32_t initOld = (haveMasks && lastPl < 31) ? (2u << lastPl) + (2u << lastPl) - (2u << lastPl) / 2 - 1 : 0u;
  // the sign in bit 31, the value complete (k_ref_assemble, LiftFuse::coefSigned): neither the sign nor the mask words are read
  const int scheme = (SG && dequant && F.dst) ? coef_scheme(F.dst[c]) : 0;   // (0: this chunk keeps magnitudes and masks)
  const bool two = scheme == 2;
  auto sg_value = [&](uint32_t sv) -> double {
    double d = fq * (double)coef_scheme_mag(sv, two);
    return __hiloint2double(__double2hiint(d) ^ (int)(sv & 0x80000000u), __double2loint(d));   // * -1.0, exactly
  };
  // sample (dcol, drow, zp): straight from the decoder (q * double(c) * (+-1.0), src/SPECK_FLT.cpp:373-399)
  // unless a coarser level's passes have produced it
  auto fetch = [&](int k, uint32_t zp) -> double {
    const uint32_t y = (pk[k] >> 12) & 0x7fffu, x = pk[k] & 0xfffu;
    const uint32_t drow = (y & 1) ? ye + (y >> 1) : (y >> 1), dcol = (x & 1) ? xe + (x >> 1) : (x >> 1);
    const size_t idx = (size_t)zp * sliceN + drow * cx + dcol;
    const bool inBox = ((innerMask >> k) & 1u) && zp < F.inner[2];
    if (!dequant && !inBox && F.bufx)
      return 0.0;   // (a compact buffer holds the box only; the host asks for one only when every chunk dequantises here)
    if (dequant && !inBox) {
      if (SG && scheme)
        return sg_value(coef[idx]);
      uint32_t v = coef[idx];
      const uint32_t sh = (uint32_t)(idx & 63);
      const uint64_t sgw = sign[idx >> 6], mnw = mNew[idx >> 6], mow = mOld[idx >> 6];   // (independent loads)
      const uint32_t fill = ((mnw >> sh) & 1ull) ? initNew : (((mow >> sh) & 1ull) ? initOld : 0u);
      v = v ? v : fill;
      return fq * (double)v * (((sgw >> sh) & 1ull) ? 1.0 : -1.0);
    }
    return buf[(size_t)zp * bufSlice + drow * bufx + dcol];
  };
  // The slice of z-inverted samples staged in X (all staged rows) -> y pass into Y, x pass back into
  // X, volume.  X and Y swap from slice to slice: three barriers per slice.
  uint32_t flip = 0;
  // sample k of the slice that is finished next (its staging buffer is free: see finish_slice)
  auto stage = [&](int k, double v) {
    uint32_t p = pk[k];
    asm volatile("" : "+v"(p));   // (keeps the LDS address from being computed ahead and spilled)
    (sm + (flip ? bufN : 0u))[(p >> 27) * RS + 4 + (p & 0xfffu)] = v;
  };
  auto stage_all = [&](const double (&v)[kXYZPosI]) {
#pragma unroll
    for (int k = 0; k < kXYZPosI; k++)
      if ((tid + (uint32_t)k * kXYZThreadsI) < npos)
        stage(k, v[k]);
  };
  // rows of a finished slice (in staging buffer `which`) -> volume, mean added, narrowed
  auto store_rows = [&](uint32_t z, uint32_t which) {
    const double* X = sm + (which ? bufN : 0u);
    for (uint32_t r = wave; r < nt; r += nwaves) {
      VT* dstrow = volc + (size_t)z * vsz + (size_t)(y0 + r) * vsy;
      const double* srow = X + (size_t)(kXYHalo + r) * RS + 4;
      if (F.noMean) {   // (uniform)
        for (uint32_t x = lane; x < cx; x += 64)
          dstrow[x] = (VT)srow[x];
      }
      else
        for (uint32_t x = lane; x < cx; x += 64)
          dstrow[x] = (VT)(srow[x] + mean);
    }
  };
  auto finish_slice = [&](uint32_t z) {
    uint32_t RSv = RS, tidv = tid;   // (see k_lift_xyz_fwd)
    asm volatile("" : "+s"(RSv), "+v"(tidv));
    const uint32_t which = flip;
    double* X = sm + (flip ? bufN : 0u);
    double* Y = sm + (flip ? 0u : bufN);
    flip ^= 1u;
    XYZ_LDS_BARRIER();
    xyz_lift_y<false, true, kXYZThreadsI>(X, Y, RSv, cx, nt, tidv, K);
    XYZ_LDS_BARRIER();
    xyz_lift_x<false, kXYZThreadsI>(Y, X, RSv, cx, kXYHalo, kXYHalo + nt, tidv, K);
    XYZ_LDS_BARRIER();
    store_rows(z, which);
    // (the next slice is staged into Y, which nobody reads any more; its y pass writes X only after
    //  the barrier behind that staging, when every row above has been stored)
  };

  double o1p[kXYZPosI], e1p[kXYZPosI], o2p[kXYZPosI], e2p[kXYZPosI];
#pragma unroll
  for (int k = 0; k < kXYZPosI; k++)
    o1p[k] = e1p[k] = o2p[k] = e2p[k] = 0.0;
  const uint32_t npairs = cz / 2;
  // one pair (low[m], high[m]) of position k enters its pipeline; slice 2m-3's sample is staged
  auto zstep = [&](int k, uint32_t m, bool mine, bool active, double E, double O) {
    const double o1 = (-K.eps) * O;                                        // o1[m]
    const double t = K.delta * ((m == 0 ? o1 : o1p[k]) + o1);
    const double e1 = fma(E, K.inv_eps, -t);                               // e1[m]
    if (m >= 1) {
      const double o2 = fma(-K.gamma, e1p[k] + e1, o1p[k]);                // o2[m-1]
      const double e2 = fma(-K.beta, (m == 1 ? o2 : o2p[k]) + o2, e1p[k]); // e2[m-1]
      if (m >= 2 && mine && active)
        stage(k, fma(-K.alpha, e2p[k] + e2, o2p[k]));                      // o3[m-2]: slice 2m-3
      o2p[k] = o2;
      e2p[k] = e2;
    }
    o1p[k] = o1;
    e1p[k] = e1;
  };
#if XYZ_INV_PREFETCH == 2
  // Round 4.  The kernel spent half its time waiting for loads, one after the other: every sample's sign and
  // mask words sat behind a branch of their own (box or not), so a pair was some thirty-six dependent round
  // trips.  Now (chunks that dequantise here):
  //  * pair m + 1's coefficients travel from HBM straight into LDS (global_load_lds: no register holds them)
  //    while pair m's two slices go through the y and x passes; every wavefront has its own 64-dword row
  //    per (position, low / high half) behind the two staging buffers.  They are issued AFTER the pair's
  //    own loads: the memory counter retires in order, a wait for those would wait for these too;
  //  * the sign / mask dwords and the fp64 samples of the coarser levels' box (an eighth of all) are loaded
  //    without any branch -- a lane that does not need one reads a harmless address --, three positions'
  //    worth at a time: two round trips per pair.
  // The arithmetic per sample is what it was.
  uint32_t* myPre = reinterpret_cast<uint32_t*>(sm + 2 * (size_t)bufN) + (size_t)wave * (kXYZPosI * 2 * 64);
  const uint32_t* sign32 = reinterpret_cast<const uint32_t*>(sign);
  const uint32_t* mNew32 = reinterpret_cast<const uint32_t*>(mNew);
  const uint32_t* mOld32 = reinterpret_cast<const uint32_t*>(mOld);
  uint32_t activeMask = 0;
#pragma unroll
  for (int k = 0; k < kXYZPosI; k++)
    if ((tid + (uint32_t)k * kXYZThreadsI) < npos)
      activeMask |= 1u << k;
    else
      pk[k] = pk[0];   // (a valid position: its loads are harmless, nothing of it is staged)
  uint32_t boxAny = 0;   // bit k: some lane of this wavefront has position k inside the coarser levels' box (along x and y)
#pragma unroll
  for (int k = 0; k < kXYZPosI; k++)
    boxAny |= __ballot((innerMask >> k) & 1u) != 0ull ? 1u << k : 0u;
  boxAny = (uint32_t)__builtin_amdgcn_readfirstlane((int)boxAny);
  auto pos_off = [&](int k, uint32_t& drow, uint32_t& dcol) {
    const uint32_t y = (pk[k] >> 12) & 0x7fffu, x = pk[k] & 0xfffu;
    drow = (y & 1) ? ye + (y >> 1) : (y >> 1);
    dcol = (x & 1) ? xe + (x >> 1) : (x >> 1);
  };
  // (a high-half sample never lies in the next level's box when that box ends at or before the low half)
  const bool fastLoads = dequant && F.inner[2] <= ze && (!SG || scheme != 0);
  // The fp64 samples of the coarser levels' box THROUGH LDS as well (sign-in-word kernel, round 5): where the lanes of
  // a wavefront that are in the box are exactly its even ones -- rows of a multiple of 64 samples: 32 doubles in a row
  // of the chunk buffer --, the 256 bytes travel like a row of coefficients, a pair ahead (lane i fetches dword i),
  // and lane 2 i reads double i.  Two load round trips per pair in half the wavefronts, with the other half waiting
  // for them at the barrier, before.  A slot per (wavefront, position) that has a box row: 12 rows x cx / 64 <= 48.
  uint32_t boxLds = 0, boxSlot0 = 0;   // (wave-uniform) bit k: position k's box samples come through LDS; the wavefront's first slot
  uint32_t* boxRows = reinterpret_cast<uint32_t*>(sm + 2 * (size_t)bufN) + (size_t)nwaves * (kXYZPosI * 2 * 64);
  if (SG && fastLoads) {
    uint32_t reg = 0;
#pragma unroll
    for (int k = 0; k < kXYZPosI; k++) {
      const uint64_t inb = __ballot((innerMask >> k) & 1u), act = __ballot((activeMask >> k) & 1u);
      if (inb == 0x5555555555555555ull && act == ~0ull && (cx & 63u) == 0)
        reg |= 1u << k;
    }
    reg = (uint32_t)__builtin_amdgcn_readfirstlane((int)reg);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(sm);   // (the staging buffers are not in use yet)
    if (lane == 0)
      cnt[wave] = (uint32_t)__popc(reg);
    XYZ_LDS_BARRIER();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < nwaves; w++) {
      const uint32_t v = cnt[w];
      before += w < wave ? v : 0u;
      total += v;
    }
    XYZ_LDS_BARRIER();
    if (total <= (uint32_t)kXYZBoxSlots && reg == (1u << kXYZPosI) - 1u) {   // (all of the wavefront's positions or none)
      boxLds = reg;
      boxSlot0 = before;
    }
  }
  uint32_t* myBox = boxRows + (size_t)boxSlot0 * 64;   // position k's row: myBox + k * 64
  auto pre_issue = [&](uint32_t m) {
#pragma unroll
    for (int k = 0; k < kXYZPosI; k++) {
      uint32_t drow, dcol;
      pos_off(k, drow, dcol);
      const uint32_t off = drow * cx + dcol;
      __builtin_amdgcn_global_load_lds(coef + ((size_t)m * sliceN + off), myPre + (k * 2) * 64, 4, 0, 0);
      __builtin_amdgcn_global_load_lds(coef + ((size_t)(ze + m) * sliceN + off), myPre + (k * 2 + 1) * 64, 4, 0, 0);
    }
    if (SG && boxLds != 0 && m < F.inner[2]) {   // (uniform)
#pragma unroll
      for (int k = 0; k < kXYZPosI; k++) {
        uint32_t drow, dcol;
        pos_off(k, drow, dcol);
        const uint32_t x0h = ((pk[k] & 0xfffu) - lane) >> 1;   // the box column of the wavefront's first lane
        const uint32_t* src = reinterpret_cast<const uint32_t*>(buf + ((size_t)m * bufSlice + (size_t)drow * bufx + x0h)) + lane;
        __builtin_amdgcn_global_load_lds(src, myBox + k * 64, 4, 0, 0);
      }
    }
  };
  if (fastLoads && mFirst < mB)
    pre_issue(mFirst);
#endif
  for (uint32_t m = mFirst; m < mB; m++) {
    const bool mine = m >= mA;   // (else: the segment's run-up)
#pragma unroll
    for (int k = 0; k < kXYZPosI; k++)
      asm volatile("" : "+v"(pk[k]));
#if XYZ_INV_PREFETCH == 2
    if (SG && fastLoads) {
      // (round 5) the coefficients carry their sign and are complete: a pair's global loads are the fp64 samples of
      // the coarser levels' box alone -- and only in the wavefronts that have a lane in the box (with 256 samples a
      // row: those whose rows are the even ones)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this pair's coefficients have landed in LDS (and its box rows)
      const bool boxWave = boxAny != 0 && m < F.inner[2];   // (uniform)
      if (boxWave && boxLds != 0) {   // the box samples were prefetched with the coefficients
#pragma unroll
        for (int k = 0; k < kXYZPosI; k++) {
          const double lo = sg_value(myPre[(k * 2) * 64 + lane]);
          const double hi = sg_value(myPre[(k * 2 + 1) * 64 + lane]);
          const double bxv = reinterpret_cast<const double*>(myBox + k * 64)[lane >> 1];
          zstep(k, m, mine, true, (lane & 1u) ? lo : bxv, hi);   // (the even lanes are the ones in the box)
        }
      }
      else {
#pragma unroll
      for (int g = 0; g < kXYZPosI; g += kXYZGroupI) {
        __builtin_amdgcn_sched_barrier(0);   // (kXYZGroupI positions' loads in flight at a time)
        double bx[kXYZGroupI];
        uint32_t boxm = 0;
        if (boxWave) {
#pragma unroll
          for (int kk = 0; kk < kXYZGroupI; kk++) {
            const int k = g + kk;
            if (k >= kXYZPosI)
              continue;
            uint32_t drow, dcol;
            pos_off(k, drow, dcol);
            const bool inBox = ((innerMask >> k) & 1u) != 0;
            boxm |= inBox ? 1u << kk : 0u;
            bx[kk] = buf[inBox ? (size_t)m * bufSlice + drow * bufx + dcol : (size_t)0];
          }
        }
#pragma unroll
        for (int kk = 0; kk < kXYZGroupI; kk++) {
          const int k = g + kk;
          if (k >= kXYZPosI)
            continue;
          const double lo = sg_value(myPre[(k * 2) * 64 + lane]);
          const double hi = sg_value(myPre[(k * 2 + 1) * 64 + lane]);
          zstep(k, m, mine, ((activeMask >> k) & 1u) != 0, ((boxm >> kk) & 1u) ? bx[kk] : lo, hi);
        }
      }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (m + 1 < mB)
        pre_issue(m + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    else if (!SG && fastLoads) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this pair's coefficients have landed in LDS
#pragma unroll
      for (int g = 0; g < kXYZPosI; g += kXYZGroupI) {
        __builtin_amdgcn_sched_barrier(0);   // (kXYZGroupI positions' loads in flight at a time)
        constexpr int NV = 2 * kXYZGroupI;
        uint32_t sgw[NV], mnw[NV], mow[NV], cv[NV], shv[NV], boxm = 0;
        double bx[kXYZGroupI];   // (only a low-half sample can lie in the box: fastLoads)
#pragma unroll
        for (int j = 0; j < NV; j++) {
          const int k = g + j / 2;
          if (k >= kXYZPosI)
            continue;
          const uint32_t zp = (j & 1) ? ze + m : m;
          uint32_t drow, dcol;
          pos_off(k, drow, dcol);
          const size_t idx = (size_t)zp * sliceN + drow * cx + dcol;
          sgw[j] = sign32[idx >> 5];
          mnw[j] = mNew32[idx >> 5];
          mow[j] = mOld32[idx >> 5];
          if ((j & 1) == 0) {
            const bool inBox = ((innerMask >> k) & 1u) && zp < F.inner[2];
            boxm |= inBox ? 1u << j : 0u;
            bx[j / 2] = buf[inBox ? (size_t)zp * bufSlice + drow * bufx + dcol : (size_t)0];
          }
          cv[j] = myPre[(k * 2 + (j & 1)) * 64 + lane];
          shv[j] = (uint32_t)idx & 31u;
        }
        double val[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) {
          if (g + j / 2 >= kXYZPosI)
            continue;
          const uint32_t fill = ((mnw[j] >> shv[j]) & 1u) ? initNew : (((mow[j] >> shv[j]) & 1u) ? initOld : 0u);
          const uint32_t v = cv[j] ? cv[j] : fill;
          const double dq = fq * (double)v * (((sgw[j] >> shv[j]) & 1u) ? 1.0 : -1.0);
          val[j] = ((j & 1) == 0 && ((boxm >> j) & 1u)) ? bx[j / 2] : dq;
        }
#pragma unroll
        for (int kk = 0; kk < kXYZGroupI; kk++)
          if (g + kk < kXYZPosI)
            zstep(g + kk, m, mine, ((activeMask >> (g + kk)) & 1u) != 0, val[2 * kk], val[2 * kk + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (m + 1 < mB)
        pre_issue(m + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    else
#endif
    {
#pragma unroll
      for (int k = 0; k < kXYZPosI; k++) {
        if ((k % 3) == 0)
          __builtin_amdgcn_sched_barrier(0);   // (four positions' loads in flight at a time, not all twelve)
        if ((tid + (uint32_t)k * kXYZThreadsI) >= npos)
          continue;
        const double E = fetch(k, m), O = fetch(k, ze + m);
        zstep(k, m, mine, true, E, O);
      }
    }
    if (m >= 2 && mine)
      finish_slice(2 * m - 3);
    if (m >= 1 && mine) {
      stage_all(e2p);
      finish_slice(2 * m - 2);
    }
  }
  // ---- the end of the lines (the last segment's)
  if (seg + 1 != nseg)
    return;
  if ((cz & 1) == 0) {   // pairs 0 .. M, M = cz / 2 - 1
    const uint32_t M = npairs - 1;
    double outC[kXYZPosI];
#pragma unroll
    for (int k = 0; k < kXYZPosI; k++) {
      outC[k] = 0.0;
      if ((tid + (uint32_t)k * kXYZThreadsI) >= npos)
        continue;
      const double o2 = fma(-K.gamma, e1p[k] + e1p[k], o1p[k]);              // o2[M]
      const double e2 = fma(-K.beta, o2p[k] + o2, e1p[k]);                   // e2[M]
      stage(k, fma(-K.alpha, e2p[k] + e2, o2p[k]));                          // o3[M-1]: slice 2M-1
      e2p[k] = e2;                                                           // slice 2M
      outC[k] = fma(-K.alpha, e2 + e2, o2);                                  // o3[M]: slice 2M+1
    }
    finish_slice(2 * M - 1);
    stage_all(e2p);
    finish_slice(2 * M);
    stage_all(outC);
    finish_slice(2 * M + 1);
  }
  else {                 // pairs 0 .. M-1 and the lone even sample low[M], M = cz / 2
    const uint32_t M = npairs;
    double outC[kXYZPosI], outD[kXYZPosI];
#pragma unroll
    for (int k = 0; k < kXYZPosI; k++) {
      outC[k] = outD[k] = 0.0;
      if ((tid + (uint32_t)k * kXYZThreadsI) >= npos)
        continue;
      const double E = fetch(k, M);
      const double t = K.delta * (o1p[k] + o1p[k]);
      const double e1 = fma(E, K.inv_eps, -t);                               // e1[M]
      const double o2 = fma(-K.gamma, e1p[k] + e1, o1p[k]);                  // o2[M-1]
      const double e2 = fma(-K.beta, o2p[k] + o2, e1p[k]);                   // e2[M-1]
      stage(k, fma(-K.alpha, e2p[k] + e2, o2p[k]));                          // o3[M-2]: slice 2M-3
      e2p[k] = e2;                                                           // slice 2M-2
      const double e2L = fma(-K.beta, o2 + o2, e1);                          // e2[M]: slice 2M
      outC[k] = fma(-K.alpha, e2 + e2L, o2);                                 // o3[M-1]: slice 2M-1
      outD[k] = e2L;
    }
    finish_slice(2 * M - 3);
    stage_all(e2p);
    finish_slice(2 * M - 2);
    stage_all(outC);
    finish_slice(2 * M - 1);
    stage_all(outD);
    finish_slice(2 * M);
  }
}

// ------------------------------------------------------------------------------------------
// quantiser
// ------------------------------------------------------------------------------------------

constexpr int kMaxPer = 16;   // samples per thread, as 16-byte loads
__global__ void __launch_bounds__(kThreads)
k_maxabs(const double* vals, size_t valsStride, uint32_t n, CoderState* st)
{
  const uint32_t c = blockIdx.y;
  if (st[c].is_const)
    return;
  __shared__ double wmax[kThreads / 64];
  const double* in = vals + c * valsStride;
  const uint32_t base = blockIdx.x * (kThreads * kMaxPer);
  double m = 0.0;
#pragma unroll
  for (int k = 0; k < kMaxPer / 2; k++) {
    const uint32_t i = base + (k * kThreads + threadIdx.x) * 2;
    if (i + 1 < n) {
      const double2 v = *reinterpret_cast<const double2*>(in + i);
      m = fmax(m, fmax(fabs(v.x), fabs(v.y)));
    }
    else if (i < n)
      m = fmax(m, fabs(in[i]));
  }
  for (int d = 32; d > 0; d >>= 1)
    m = fmax(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0)
    wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kThreads / 64; w++)
      m = fmax(m, wmax[w]);
    if (m > 0.0)  // non-negative doubles order like their bit patterns
      atomicMax(reinterpret_cast<unsigned long long*>(&st[c].maxabs),
                (unsigned long long)__double_as_longlong(m));
  }
}

// ------------------------------------------------------------------------------------------
// PSNR mode: error of the mid-tread quantiser with step q (src/SPECK_FLT.cpp:237-266).  Strides
// of 4096 coefficients are summed one after the other, each sequentially, with the fused
// multiply-adds of the canonical build: diff = fma(-q, rint(v / q'), v), acc = fma(diff, diff, acc)
// where q' = 1/q is formed first.  Same LDS staging as k_stride_sums.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMseStride = 4096;

__global__ void __launch_bounds__(kThreads)
k_mse_strides(const double* vals, size_t valsStride, uint32_t n, double* partial,
              size_t partialStride, const CoderState* st)
{
  __shared__ double tile[kSumStrides][kSumSeg + 1];
  const uint32_t c = blockIdx.y;
  if (st[c].is_const || !st[c].mse_active)
    return;
  const double q = st[c].q, rcp_q = 1.0 / q;
  const double* in = vals + c * valsStride;
  const uint32_t npart = n / kMseStride + 1;            // the last one may be short or empty
  const uint32_t s0 = blockIdx.x * kSumStrides;
  double acc = 0.0;
  for (uint32_t seg = 0; seg < kMseStride; seg += kSumSeg) {
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < (uint32_t)(kSumStrides * kSumSeg); k += kThreads) {
      const uint32_t r = k / kSumSeg, j = k % kSumSeg;
      const uint64_t e = (uint64_t)(s0 + r) * kMseStride + seg + j;
      tile[r][j] = (s0 + r < npart && e < n) ? in[e] : 0.0;   // (a zero adds exactly nothing)
    }
    __syncthreads();
    if (threadIdx.x < (uint32_t)kSumStrides && s0 + threadIdx.x < npart) {
      const uint64_t e0 = (uint64_t)(s0 + threadIdx.x) * kMseStride + seg;
      const uint32_t len = e0 >= n ? 0u : (uint32_t)min((uint64_t)kSumSeg, n - e0);
      for (uint32_t j = 0; j < len; j++) {
        const double v = tile[threadIdx.x][j];
        const double diff = fma(-q, rint(v * rcp_q), v);
        acc = fma(diff, diff, acc);
      }
    }
  }
  if (threadIdx.x < (uint32_t)kSumStrides && s0 + threadIdx.x < npart)
    partial[c * partialStride + s0 + threadIdx.x] = acc;
}

__global__ void k_mse_final(uint32_t n, const double* partial, size_t partialStride,
                            CoderState* st, uint32_t nchunks)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks || st[c].is_const || !st[c].mse_active)
    return;
  const uint32_t npart = n / kMseStride + 1;
  const double* pp = partial + c * partialStride;
  double total = 0.0;
  for (uint32_t s = 0; s < npart; s++)
    total += pp[s];
  st[c].mse = total / (double)n;
}

// SPECK_FLT.cpp:282-301 (fixed-rate q) ; `wide` selects the high-precision retry
__global__ void k_make_q_rate(CoderState* st, uint32_t nchunks, int wide_pass)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nchunks)
    return;
  if (wide_pass) {
    if (!st[c].need_retry)
      return;
    st[c].q = st[c].maxabs / 0x1.fffffffffffffp52;
    st[c].wide = 1;
  }
  else {
    st[c].q = st[c].maxabs / 4294967295.0;
    st[c].wide = 0;
  }
}

// SPECK_FLT.cpp:345-368 : ll = llrint(v * (1/q)) ; sign bit = (ll >= 0) ; magnitude ; plus the
// msb position that the coder's significance pyramid starts from.  One wave = one sign word.
template <typename CT>
__global__ void k_quantize(const double* vals, size_t valsStride, uint32_t n, CT* coef,
                           size_t coefStride, uint64_t* sign, size_t signStride, int8_t* msb,
                           size_t msbStride, const CoderState* st, int wide_pass)
{
  const uint32_t c = blockIdx.y;
  const CoderState& s = st[c];
  if (s.is_const || (wide_pass && !s.need_retry))
    return;
  const double inv = 1.0 / s.q;
  const double* in = vals + c * valsStride;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  long long ll = 0;
  bool nonneg = false;
  if (i < n) {
    ll = __double2ll_rn(in[i] * inv);
    nonneg = ll >= 0;
    const unsigned long long mag = (unsigned long long)(ll < 0 ? -ll : ll);
    coef[c * coefStride + i] = (CT)mag;
    msb[c * msbStride + i] = mag ? (int8_t)(63 - __clzll((long long)mag)) : (int8_t)-1;
  }
  const unsigned long long word = __ballot(nonneg);
  if ((threadIdx.x & 63) == 0 && i < n)
    sign[c * signStride + (i >> 6)] = word;
}

// The same for 32-bit magnitudes, four consecutive samples per thread: 32-byte loads, 16-byte stores of
// the magnitudes, one 4-byte store of the four msb positions; the sign word of 64 samples is put
// together by the 16 lanes that hold them (all strides and n are multiples of 4 -- launch_quantize).
__global__ void __launch_bounds__(kThreads)
k_quantize4(const double* vals, size_t valsStride, uint32_t n, uint32_t* coef, size_t coefStride,
            uint64_t* sign, size_t signStride, int8_t* msb, size_t msbStride, const CoderState* st)
{
  const uint32_t c = blockIdx.y;
  const CoderState& s = st[c];
  if (s.is_const)
    return;
  const double inv = 1.0 / s.q;
  const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  uint32_t nn = 0;   // bit e: sample i + e is non-negative
  if (i < n) {
    const double2* in = reinterpret_cast<const double2*>(vals + c * valsStride + i);
    const double2 a = in[0], b2 = in[1];
    const double v[4] = {a.x, a.y, b2.x, b2.y};
    uint32_t mag[4], mb = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const long long ll = __double2ll_rn(v[e] * inv);
      nn |= (ll >= 0 ? 1u : 0u) << e;
      const unsigned long long m = (unsigned long long)(ll < 0 ? -ll : ll);
      mag[e] = (uint32_t)m;
      mb |= (uint32_t)(uint8_t)(m ? (int8_t)(63 - __clzll((long long)m)) : (int8_t)-1) << (8 * e);
    }
    *reinterpret_cast<uint4*>(coef + c * coefStride + i) = make_uint4(mag[0], mag[1], mag[2], mag[3]);
    *reinterpret_cast<uint32_t*>(msb + c * msbStride + i) = mb;
  }
  // 16 lanes x 4 bits = one sign word
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t word = (uint64_t)nn << (4 * (lane & 15u));
#pragma unroll
  for (int d = 1; d < 16; d <<= 1)
    word |= __shfl_xor(word, d, 64);
  if ((lane & 15u) == 0 && i < n)
    sign[c * signStride + (i >> 6)] = word;
}

// SPECK_FLT.cpp:373-399 : (q * c) * (+-1.0), left to right.  When the decoder's masks are given,
// the coefficients that became significant but were never refined are completed here instead of
// in a pass of their own (k_dec_finish, speck_dec.hip: 1.5 * 2^plane - 1, SPECK_INT.cpp:462-468).
template <typename CT>
__global__ void __launch_bounds__(kThreads)
k_inv_quantize(const CT* coef, size_t coefStride, const uint64_t* sign, size_t signStride,
               uint32_t n, double* vals, size_t valsStride, const CoderState* st, int wide_pass,
               const uint64_t* sigNew, const uint64_t* sigOld, size_t maskStride,
               const DecState* dst)
{
  const uint32_t c = blockIdx.y;
  const CoderState& s = st[c];
  if (s.is_const || (int)s.wide != wide_pass)
    return;
  const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;   // four samples of one mask word
  if (i0 >= n)
    return;
  const CT* in = coef + c * coefStride + i0;
  double* out = vals + c * valsStride + i0;
  CT v[4];
  const uint32_t cnt = min(4u, n - i0);
  if (cnt == 4) {
    if (sizeof(CT) == 4) {
      const uint4 q = *reinterpret_cast<const uint4*>(in);
      v[0] = (CT)q.x, v[1] = (CT)q.y, v[2] = (CT)q.z, v[3] = (CT)q.w;
    }
    else {
      const ulonglong2 q0 = *reinterpret_cast<const ulonglong2*>(in);
      const ulonglong2 q1 = *reinterpret_cast<const ulonglong2*>(in + 2);
      v[0] = (CT)q0.x, v[1] = (CT)q0.y, v[2] = (CT)q1.x, v[3] = (CT)q1.y;
    }
  }
  else
    for (uint32_t k = 0; k < 4; k++)
      v[k] = k < cnt ? in[k] : (CT)1;
  const uint32_t w = i0 >> 6, sh = i0 & 63;
  if (sigNew != nullptr && (v[0] == 0 || v[1] == 0 || v[2] == 0 || v[3] == 0)) {
    const uint32_t mn = (uint32_t)(sigNew[c * maskStride + w] >> sh) & 15u;
    const uint32_t mo = (uint32_t)(sigOld[c * maskStride + w] >> sh) & 15u;
    if (mn | mo) {
      const int pl = dst[c].lastPlane;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (v[k] == 0 && ((mn | mo) >> k) & 1u) {
          const CT thr = (CT)1 << (pl + (((mn >> k) & 1u) ? 0 : 1));
          v[k] = thr + thr - thr / 2 - 1;
        }
    }
  }
  const uint32_t sg = (uint32_t)(sign[c * signStride + w] >> sh);
  double r[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
    r[k] = s.q * (double)v[k] * (((sg >> k) & 1u) ? 1.0 : -1.0);
  if (cnt == 4) {
    *reinterpret_cast<double2*>(out) = make_double2(r[0], r[1]);
    *reinterpret_cast<double2*>(out + 2) = make_double2(r[2], r[3]);
  }
  else
    for (uint32_t k = 0; k < cnt; k++)
      out[k] = r[k];
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

LiftConsts lift_consts()
{
  // include/CDF97.h:136-147 evaluated in IEEE fp64 (bit patterns: SURVEY.md Appendix B)
  auto bits = [](uint64_t b) {
    double d;
    memcpy(&d, &b, 8);
    return d;
  };
  LiftConsts k;
  k.alpha = bits(0xbff960ce676352c0ull);
  k.beta = bits(0xbfab2035c938c1f9ull);
  k.gamma = bits(0x3fec40ceba5579b7ull);
  k.delta = bits(0x3fdc626a9045727dull);
  k.eps = bits(0x3ff264c795071559ull);
  k.inv_eps = bits(0x3febd5edf975cca4ull);
  return k;
}

// Lines per tile.  The kernel is HBM-bound and synchronises between lifting steps, so it wants
// several workgroups per CU: measured on MI355X (256-sample lines) 8 lines per tile are best along
// x, where a wavefront reads along the line, and 16 across (128-byte rows); larger tiles halve the
// throughput.
static int pick_nl(uint32_t len, int axis, size_t* smem)
{
  // (measured again with the dequantising inverse passes, SPERR_HIP_LIFT_LDS_KB: 20 / 36 / 72 KB
  //  along y and z give 9.1 / 8.3 / 11.5 ms for the 13 inverse passes of the bench volume)
  static const int capYZ = tune_getenv("SPERR_HIP_LIFT_LDS_KB") ? atoi(tune_getenv("SPERR_HIP_LIFT_LDS_KB")) : 36;
  const size_t cap = (size_t)(axis == 0 ? 20 : capYZ) * 1024;
  for (int nl = 32; nl >= 1; nl >>= 1) {
    const size_t bytes = (size_t)len * (nl + 1) * sizeof(double);
    if (bytes <= cap || nl == 1) {
      *smem = bytes;
      return nl;
    }
  }
  return 1;
}

int launch_lift(hipStream_t stream, bool forward, double* vals, size_t valsStride,
                uint32_t nchunks, const uint32_t cdims[3], int axis, const uint32_t region[3],
                CoderState* st, int io, void* volume, VolDesc vd, const ChunkGeom* geom,
                const LiftFuse* fuse)
{
  const LiftFuse F = fuse ? *fuse : LiftFuse{};
  {
    const void* fns[6] = {reinterpret_cast<const void*>(&k_lift_axis<true, 0>),
                          reinterpret_cast<const void*>(&k_lift_axis<true, 1>),
                          reinterpret_cast<const void*>(&k_lift_axis<true, 2>),
                          reinterpret_cast<const void*>(&k_lift_axis<false, 0>),
                          reinterpret_cast<const void*>(&k_lift_axis<false, 1>),
                          reinterpret_cast<const void*>(&k_lift_axis<false, 2>)};
    for (const void* f : fns)
      if (set_max_dyn_lds(f, 160 * 1024))
        return -1;
  }
  const uint32_t len = region[axis];
  if (len < 2)
    return io ? -1 : 0;
  size_t smem = 0;
  const int NL = pick_nl(len, axis, &smem);
  if (smem > 160 * 1024) {
    fprintf(stderr, "[sperr_hip] line of %u samples does not fit in LDS\n", len);
    return -1;
  }
  const int ua = (axis == 0) ? 1 : 0, wa = (axis == 2) ? 1 : 2;
  const uint32_t ntu = (region[ua] + NL - 1) / NL;
  dim3 grid(ntu * region[wa], nchunks);
  const LiftConsts K = lift_consts();
#define LIFT_ARGS vals, valsStride, cdims[0], cdims[1], axis, region[0], region[1], region[2], NL, K, st, volume, vd, geom, F
  if (forward) {
    if (io == 1)
      LAUNCH_K((k_lift_axis<true, 1>), grid, dim3(kThreads), smem, stream, LIFT_ARGS);
    else if (io == 2)
      LAUNCH_K((k_lift_axis<true, 2>), grid, dim3(kThreads), smem, stream, LIFT_ARGS);
    else
      LAUNCH_K((k_lift_axis<true, 0>), grid, dim3(kThreads), smem, stream, LIFT_ARGS);
  }
  else {
    if (io == 1)
      LAUNCH_K((k_lift_axis<false, 1>), grid, dim3(kThreads), smem, stream, LIFT_ARGS);
    else if (io == 2)
      LAUNCH_K((k_lift_axis<false, 2>), grid, dim3(kThreads), smem, stream, LIFT_ARGS);
    else
      LAUNCH_K((k_lift_axis<false, 0>), grid, dim3(kThreads), smem, stream, LIFT_ARGS);
  }
#undef LIFT_ARGS
  HIP_CHECK(hipGetLastError());
  return 0;
}

// rows per tile of k_lift_xy for rows of cx samples, 0 when the fused kernel does not apply
static int xy_rows(uint32_t cx, uint32_t cy)
{
  if (cx < 2 || cy < 2)
    return 0;
  const size_t rowBytes = (size_t)(cx + 1) * sizeof(double);
  static const int ldsKB = tune_getenv("SPERR_HIP_XY_LDS_KB") ? atoi(tune_getenv("SPERR_HIP_XY_LDS_KB")) : 68;
  static const int maxRows = tune_getenv("SPERR_HIP_XY_ROWS") ? atoi(tune_getenv("SPERR_HIP_XY_ROWS")) : 32;
  int rows = (int)(((size_t)ldsKB * 1024) / rowBytes);
  if (rows > maxRows)
    rows = maxRows;
  int R = (rows - 2 * kXYHalo) & ~1;
  if ((uint32_t)R > cy)
    R = (int)((cy + 1) & ~1u);
  return R >= 8 ? R : 0;
}

bool lift_xy_applicable(const uint32_t cdims[3])
{
  return xy_rows(cdims[0], cdims[1]) > 0;
}

int launch_lift_xy(hipStream_t stream, bool forward, double* vals, size_t valsStride,
                   uint32_t nchunks, const uint32_t cdims[3], const CoderState* st, int io,
                   void* volume, VolDesc vd, const ChunkGeom* geom)
{
  {
    const void* fns[4] = {reinterpret_cast<const void*>(&k_lift_xy<true, 1>),
                          reinterpret_cast<const void*>(&k_lift_xy<true, 2>),
                          reinterpret_cast<const void*>(&k_lift_xy<false, 1>),
                          reinterpret_cast<const void*>(&k_lift_xy<false, 2>)};
    for (const void* f : fns)
      if (set_max_dyn_lds(f, 160 * 1024))
        return -1;
  }
  const int R = xy_rows(cdims[0], cdims[1]);
  if (R <= 0 || (io != 1 && io != 2))
    return -1;
  const size_t smem = (size_t)(R + 2 * kXYHalo) * (cdims[0] + 1) * sizeof(double);
  const uint32_t ntile = (cdims[1] + R - 1) / R;
  const dim3 grid(ntile * cdims[2], nchunks);
  const LiftConsts K = lift_consts();
#define XY_ARGS vals, valsStride, cdims[0], cdims[1], cdims[2], R, K, st, volume, vd, geom
  if (forward) {
    if (io == 1)
      LAUNCH_K((k_lift_xy<true, 1>), grid, dim3(kXYThreads), smem, stream, XY_ARGS);
    else
      LAUNCH_K((k_lift_xy<true, 2>), grid, dim3(kXYThreads), smem, stream, XY_ARGS);
  }
  else {
    if (io == 1)
      LAUNCH_K((k_lift_xy<false, 1>), grid, dim3(kXYThreads), smem, stream, XY_ARGS);
    else
      LAUNCH_K((k_lift_xy<false, 2>), grid, dim3(kXYThreads), smem, stream, XY_ARGS);
  }
#undef XY_ARGS
  HIP_CHECK(hipGetLastError());
  return 0;
}

bool lift_xyz_applicable(const uint32_t cdims[3])
{
  static const bool on = !(tune_getenv("SPERR_HIP_LIFT_XYZ") && atoi(tune_getenv("SPERR_HIP_LIFT_XYZ")) == 0);
  if (!on || cdims[0] < 9 || cdims[1] < 9 || cdims[2] < 9)
    return false;
  // a thread's z pipelines are registers: (rows + halo) * cx positions over the workgroup's threads
  // (and the packed staging map holds 12 bits of x, 15 bits of y)
  return (size_t)kXYZStaged * cdims[0] <= (size_t)kXYZPosI * kXYZThreadsI &&
         (size_t)kXYZStaged * cdims[0] <= (size_t)kXYZStageF * kXYZThreadsF &&
         (size_t)kXYZRows * cdims[0] <= (size_t)kXYZPosF * kXYZThreadsF && cdims[0] < 4096 && cdims[1] < 32768;
}

int launch_lift_xyz(hipStream_t stream, bool forward, double* vals, size_t valsStride, uint32_t nchunks,
                    const uint32_t cdims[3], CoderState* st, int io, void* volume, VolDesc vd,
                    const ChunkGeom* geom, const LiftFuse* fuse)
{
  if (!lift_xyz_applicable(cdims) || (io != 1 && io != 2))
    return -1;
  const LiftFuse F = fuse ? *fuse : LiftFuse{};
  size_t smem = 2 * (size_t)kXYZStaged * xyz_row_stride(cdims[0]) * sizeof(double);   // two staging buffers
#if XYZ_INV_PREFETCH == 2
  if (!forward)
    smem += (size_t)kXYZThreadsI * kXYZPosI * 2 * sizeof(uint32_t) +   // + the coefficients on their way in
            (size_t)kXYZBoxSlots * 64 * sizeof(uint32_t);              //   and the box samples (k_lift_xyz_inv<.., true>)
#endif
  {
    const void* fns[6] = {reinterpret_cast<const void*>(&k_lift_xyz_fwd<1>), reinterpret_cast<const void*>(&k_lift_xyz_fwd<2>),
                          reinterpret_cast<const void*>(&k_lift_xyz_inv<1, false>), reinterpret_cast<const void*>(&k_lift_xyz_inv<2, false>),
                          reinterpret_cast<const void*>(&k_lift_xyz_inv<1, true>), reinterpret_cast<const void*>(&k_lift_xyz_inv<2, true>)};
    for (const void* f : fns)
      if (set_max_dyn_lds(f, 160 * 1024))
        return -1;
  }
  // a batch with fewer tiles than two per CU: the slices of a tile are dealt to several workgroups
  const uint32_t ntile = (cdims[1] + kXYZRows - 1) / kXYZRows;
  uint32_t nseg = 1;
  while (nseg < 4 && (size_t)ntile * nchunks * nseg < 512 && cdims[2] / (2 * nseg) >= 24)
    nseg *= 2;
  const dim3 grid(ntile * nseg, nchunks);
  const LiftConsts K = lift_consts();
  if (forward) {
    const int wantMax = F.mode == 1 ? 1 : 0;
    if (io == 1)
      LAUNCH_K((k_lift_xyz_fwd<1>), grid, dim3(kXYZThreadsF), smem, stream, vals, valsStride, cdims[0], cdims[1],
               cdims[2], K, st, volume, vd, geom, wantMax, F.inner[0], F.inner[1], F.inner[2], nseg);
    else
      LAUNCH_K((k_lift_xyz_fwd<2>), grid, dim3(kXYZThreadsF), smem, stream, vals, valsStride, cdims[0], cdims[1],
               cdims[2], K, st, volume, vd, geom, wantMax, F.inner[0], F.inner[1], F.inner[2], nseg);
  }
  else {
    const bool sg = F.mode == 2 && F.coefSigned != 0;
#define XYZ_INV_LAUNCH(io_, sg_)                                                                                    \
  LAUNCH_K((k_lift_xyz_inv<io_, sg_>), grid, dim3(kXYZThreadsI), smem, stream, vals, valsStride, cdims[0], cdims[1], \
           cdims[2], K, st, volume, vd, geom, F, nseg)
    if (io == 1) {
      if (sg)
        XYZ_INV_LAUNCH(1, true);
      else
        XYZ_INV_LAUNCH(1, false);
    }
    else {
      if (sg)
        XYZ_INV_LAUNCH(2, true);
      else
        XYZ_INV_LAUNCH(2, false);
    }
#undef XYZ_INV_LAUNCH
  }
  HIP_CHECK(hipGetLastError());
  return 0;
}

template <typename T>
int launch_condition(hipStream_t stream, const T* vol, VolDesc vd, const ChunkGeom* geom,
                     uint32_t nchunks, const uint32_t cdims[3], uint32_t nstrides,
                     double* strideMean, size_t strideMeanStride, double* vals,
                     size_t valsStride, CoderState* st, bool gather, bool want_range,
                     bool org_x_aligned)
{
  const uint32_t n = cdims[0] * cdims[1] * cdims[2];
  const uint32_t ssz = n / nstrides;
  // rows of whole 128-byte pieces, 16-byte aligned in the volume, a stride = whole rows
  const uint32_t perLoad = 8 * (16 / (uint32_t)sizeof(T));
  const bool rowsPath = cdims[0] % perLoad == 0 && ssz % cdims[0] == 0 &&
                        vd.dims[0] % (16 / sizeof(T)) == 0 &&
                        reinterpret_cast<uintptr_t>(vol) % 16 == 0 && org_x_aligned;
  if (rowsPath)
    LAUNCH_K(k_stride_sums_rows<T>, dim3((nstrides + 63) / 64, nchunks), dim3(64), 0, stream, vol,
             vd, geom, cdims[0], cdims[1], nstrides, ssz, strideMean, strideMeanStride, st,
             want_range ? 1 : 0);
  else
    LAUNCH_K(k_stride_sums<T>, dim3((nstrides + kSumStrides - 1) / kSumStrides, nchunks),
             dim3(kThreads), 0, stream, vol, vd, geom, cdims[0], cdims[1], cdims[2], nstrides, ssz,
             strideMean, strideMeanStride, st, want_range ? 1 : 0);
  LAUNCH_K(k_mean_finalize<T>, dim3(nchunks), dim3(kThreads), 0, stream, vol, vd, geom,
                     nstrides, strideMean, strideMeanStride, st);
  if (gather)   // otherwise the first lifting pass reads the volume itself
    LAUNCH_K(k_gather_condition<T>, dim3((n + kThreads * 4 - 1) / (kThreads * 4), nchunks),
             dim3(kThreads), 0, stream, vol, vd, geom, cdims[0], cdims[1], n, vals, valsStride, st);
  HIP_CHECK(hipGetLastError());
  return 0;
}
template int launch_condition<float>(hipStream_t, const float*, VolDesc, const ChunkGeom*,
                                     uint32_t, const uint32_t[3], uint32_t, double*, size_t,
                                     double*, size_t, CoderState*, bool, bool, bool);
template int launch_condition<double>(hipStream_t, const double*, VolDesc, const ChunkGeom*,
                                      uint32_t, const uint32_t[3], uint32_t, double*, size_t,
                                      double*, size_t, CoderState*, bool, bool, bool);

template <typename T>
int launch_scatter(hipStream_t stream, T* vol, VolDesc vd, const ChunkGeom* geom,
                   uint32_t nchunks, const uint32_t cdims[3], const double* vals,
                   size_t valsStride, const CoderState* st)
{
  const uint32_t n = cdims[0] * cdims[1] * cdims[2];
  LAUNCH_K(k_scatter_uncondition<T>,
                     dim3((n + kThreads * 4 - 1) / (kThreads * 4), nchunks), dim3(kThreads), 0,
                     stream, vol, vd, geom, cdims[0], cdims[1], n, vals, valsStride, st);
  HIP_CHECK(hipGetLastError());
  return 0;
}
template int launch_scatter<float>(hipStream_t, float*, VolDesc, const ChunkGeom*, uint32_t,
                                   const uint32_t[3], const double*, size_t, const CoderState*);
template int launch_scatter<double>(hipStream_t, double*, VolDesc, const ChunkGeom*, uint32_t,
                                    const uint32_t[3], const double*, size_t, const CoderState*);

int launch_maxabs_q(hipStream_t stream, const double* vals, size_t valsStride, uint32_t nchunks,
                    uint32_t n, CoderState* st, bool have_max)
{
  if (!have_max)
    LAUNCH_K(k_maxabs, dim3((n + kThreads * kMaxPer - 1) / (kThreads * kMaxPer), nchunks),
                       dim3(kThreads), 0, stream, vals, valsStride, n, st);
  LAUNCH_K(k_make_q_rate, dim3((nchunks + 63) / 64), dim3(64), 0, stream, st, nchunks,
                     0);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_mse(hipStream_t stream, const double* vals, size_t valsStride, uint32_t nchunks,
               uint32_t n, double* partial, size_t partialStride, CoderState* st)
{
  const uint32_t npart = n / kMseStride + 1;
  LAUNCH_K(k_mse_strides, dim3((npart + kSumStrides - 1) / kSumStrides, nchunks), dim3(kThreads), 0,
           stream, vals, valsStride, n, partial, partialStride, st);
  LAUNCH_K(k_mse_final, dim3((nchunks + 63) / 64), dim3(64), 0, stream, n, partial, partialStride,
           st, nchunks);
  HIP_CHECK(hipGetLastError());
  return 0;
}

// PSNR mode: the chunks flagged for 64-bit coefficients keep their q
__global__ void k_mark_wide(CoderState* st, uint32_t nchunks)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nchunks && st[c].need_retry)
    st[c].wide = 1;
}

int launch_mark_wide(hipStream_t stream, uint32_t nchunks, CoderState* st)
{
  LAUNCH_K(k_mark_wide, dim3((nchunks + 63) / 64), dim3(64), 0, stream, st, nchunks);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_make_q_wide(hipStream_t stream, uint32_t nchunks, CoderState* st)
{
  LAUNCH_K(k_make_q_rate, dim3((nchunks + 63) / 64), dim3(64), 0, stream, st, nchunks,
                     1);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_quantize(hipStream_t stream, bool wide, const double* vals, size_t valsStride,
                    uint32_t nchunks, uint32_t n, void* coef, size_t coefStride, uint64_t* sign,
                    size_t signStride, int8_t* msb, size_t msbStride, const CoderState* st)
{
  dim3 grid((n + kThreads - 1) / kThreads, nchunks);
  if (wide)
    LAUNCH_K(k_quantize<uint64_t>, grid, dim3(kThreads), 0, stream, vals, valsStride, n,
                       (uint64_t*)coef, coefStride, sign, signStride, msb, msbStride, st, 1);
  else if (n % 64 == 0 && valsStride % 4 == 0 && coefStride % 4 == 0 && msbStride % 4 == 0)
    LAUNCH_K(k_quantize4, dim3((n / 4 + kThreads - 1) / kThreads, nchunks), dim3(kThreads), 0, stream, vals,
             valsStride, n, (uint32_t*)coef, coefStride, sign, signStride, msb, msbStride, st);
  else
    LAUNCH_K(k_quantize<uint32_t>, grid, dim3(kThreads), 0, stream, vals, valsStride, n,
                       (uint32_t*)coef, coefStride, sign, signStride, msb, msbStride, st, 0);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_inv_quantize(hipStream_t stream, bool wide, const void* coef, size_t coefStride,
                        const uint64_t* sign, size_t signStride, uint32_t nchunks, uint32_t n,
                        double* vals, size_t valsStride, const CoderState* st,
                        const uint64_t* sigNew, const uint64_t* sigOld, size_t maskStride,
                        const DecState* dst)
{
  dim3 grid((n + kThreads * 4 - 1) / (kThreads * 4), nchunks);
  if (wide)
    LAUNCH_K(k_inv_quantize<uint64_t>, grid, dim3(kThreads), 0, stream,
                       (const uint64_t*)coef, coefStride, sign, signStride, n, vals, valsStride,
                       st, 1, sigNew, sigOld, maskStride, dst);
  else
    LAUNCH_K(k_inv_quantize<uint32_t>, grid, dim3(kThreads), 0, stream,
                       (const uint32_t*)coef, coefStride, sign, signStride, n, vals, valsStride,
                       st, 0, sigNew, sigOld, maskStride, dst);
  HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace sperrhip

