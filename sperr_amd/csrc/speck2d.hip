// speck2d.hip -- SPECK2D_INT_ENC / _DEC for one slice (sperr_comp_2d / sperr_decomp_2d).
//
// The 2D coder differs from the 3D one in its set structure -- quadrants are visited bottom-right
// first, and a type-I set (everything outside the already partitioned top-left box) is tested at
// the end of every sorting pass (src/SPECK2D_INT.cpp:10-98,149-186) -- so it cannot ride on the
// forest machinery of speck_enc.hip / speck_dec.hip.  One wavefront codes the slice: the LIP and
// refinement passes work on whole mask words with all 64 lanes (64 coefficients of a word sit in
// registers), runs of insignificant list entries are skipped 64 at a time, and the set recursion
// -- sequential by nature -- is scalar code with its stack in lane-indexed vector registers
// (see outlier.hip, which uses the same skeleton for the 1D coder).
#include "speck2d.h"

namespace sperrhip {
namespace {

__device__ __forceinline__ uint64_t low_mask(uint32_t n)   // n in [0, 64]
{
  return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
}
__device__ __forceinline__ uint32_t rfl(uint32_t v)
{
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t rfl64(uint64_t v)
{
  return (uint64_t)rfl((uint32_t)v) | ((uint64_t)rfl((uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l)
{
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}
__device__ __forceinline__ void wrlane(uint32_t& v, uint32_t l, uint32_t val)
{
  v = threadIdx.x == l ? val : v;   // (val and l are wave-uniform)
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
  for (int d = 32; d > 0; d >>= 1)
    v = max(v, __shfl_xor(v, d, 64));
  return v;
}
__host__ __device__ inline uint32_t approx_len(uint32_t len, uint32_t lev)
{
  for (uint32_t i = 0; i < lev; i++)
    len -= len / 2;
  return len;
}

// ------------------------------------------------------------------------------------------
// encoder preparation: largest msb of the slice and of every subband (the type-I set and the
// three sets it releases per level are tested against these, SPECK2D_INT_ENC.cpp:64-99)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_s2_prep(Speck2dBufs b)
{
  __shared__ int sm[1 + 3 * kS2MaxLevels];
  for (uint32_t i = threadIdx.x; i < 1 + 3 * kS2MaxLevels; i += blockDim.x)
    sm[i] = -1;
  __syncthreads();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < b.N; i += gridDim.x * blockDim.x) {
    const int m = b.msb[i];
    if (m < 0)
      continue;
    const uint32_t x = i % b.dx, y = i / b.dx;
    atomicMax(&sm[0], m);
    // level lev holds (x, y) when it lies inside the approximation box of level lev - 1 but not
    // inside that of level lev
    uint32_t bx = b.dx, by = b.dy;
    for (uint32_t lev = 1; lev <= b.nxforms; lev++) {
      const uint32_t ax = bx - bx / 2, ay = by - by / 2;
      if (x >= ax || y >= ay) {
        const int k = (x >= ax && y >= ay) ? 0 : (x >= ax ? 1 : 2);   // BR, TR, BL
        atomicMax(&sm[1 + 3 * lev + k], m);
        break;
      }
      bx = ax;
      by = ay;
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < 1 + 3 * kS2MaxLevels; i += blockDim.x)
    if (sm[i] >= 0)
      atomicMax(&b.prep[i], sm[i]);
}

// ------------------------------------------------------------------------------------------
// the coder
// ------------------------------------------------------------------------------------------
template <bool ENC, typename CT>
__global__ void __launch_bounds__(64)
k_speck2d(Speck2dBufs b, uint64_t budget, uint64_t raw_budget, int rate_mode, int wide_pass)
{
  const uint32_t lane = threadIdx.x;
  CoderState& cs = b.cst[0];
  uint64_t avail = ~0ull;
  int nbp;
  if (ENC) {
    if (cs.is_const) {
      if (lane == 0) {
        cs.stream_len = 17;
        cs.need_retry = 0;
      }
      return;
    }
    if (wide_pass ? (cs.need_retry == 0) : (cs.need_retry != 0))
      return;
    nbp = (int)rfl((uint32_t)(b.prep[0] + 1));
  }
  else {
    const DecState& ds = b.dst[0];
    if (!ds.active)
      return;
    nbp = (int)rfl((uint32_t)ds.nbp);
    avail = rfl64(ds.avail);
  }
  const uint32_t dx = b.dx, dy = b.dy, N = b.N, nw = b.nw;
  uint64_t* runs = b.runs;
  int8_t* sval = b.sval;
  unsigned long long* lip = reinterpret_cast<unsigned long long*>(b.lip);
  unsigned long long* lsp = reinterpret_cast<unsigned long long*>(b.lsp);
  unsigned long long* words = reinterpret_cast<unsigned long long*>(b.stream);
  CT* coef = static_cast<CT*>(b.coef);
  unsigned long long* sign = reinterpret_cast<unsigned long long*>(b.sign);
  const int8_t* msb = b.msb;
  uint32_t* fresh = b.fresh;

  // lane-indexed registers: list length / first slot / end slot of level `lane`; recursion stack
  uint32_t vCnt = 0;
  const uint32_t vOff = b.levelOff[min(lane, (uint32_t)kS2MaxLevels)];
  const uint32_t vEnd = b.levelOff[min(lane + 1u, (uint32_t)kS2MaxLevels)];
  uint32_t vRlo = 0, vRhi = 0, vT = 0, vKS = 0, vKG = 0;   // rect, state, children's msb, pixel signs
  // encoder: largest msb of everything the type-I set of level `lane` covers
  uint32_t vImax = 0;
  if (ENC) {
    int m = -1;
    for (uint32_t lev = 1; lev <= min(lane, b.nxforms); lev++)
      for (int k = 0; k < 3; k++)
        m = max(m, b.prep[1 + 3 * lev + k]);
    vImax = (uint32_t)m;
  }
  uint32_t err = 0;

  uint64_t wpos = 0, acc = 0;
  auto flush_acc = [&]() {
    if (acc && lane == 0)
      atomicOr(words + (wpos >> 6), (unsigned long long)acc);
    acc = 0;
  };
  auto put = [&](uint32_t bit) {
    acc |= (uint64_t)bit << (wpos & 63);
    if ((wpos & 63) == 63)
      flush_acc();
    wpos++;
  };
  auto skip_zeros = [&](uint32_t n) {
    if (((wpos + n) >> 6) != (wpos >> 6))
      flush_acc();
    wpos += n;
  };
  // (a truncated stream is zero beyond its last byte, up to total_bits: SPECK_INT.cpp:95-105)
  const uint64_t nwAvail = ENC ? 0ull : (avail + 63) / 64;
  auto word_at = [&](uint64_t wi) -> uint64_t { return wi < nwAvail ? (uint64_t)words[wi] : 0ull; };
  uint64_t rpos = 0, w0 = 0, w1 = 0, cw = 1ull << 62;
  auto window = [&]() -> uint64_t {
    const uint64_t wi = rpos >> 6;
    if (wi != cw) {
      w0 = (wi == cw + 1) ? w1 : rfl64(word_at(wi));
      w1 = rfl64(word_at(wi + 1));
      cw = wi;
    }
    const uint32_t sh = (uint32_t)(rpos & 63);
    return sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
  };
  auto get = [&]() -> uint32_t {
    const uint32_t bit = (uint32_t)(window() & 1ull);
    rpos++;
    return bit;
  };

  uint32_t nfresh = 0;
  auto lip_set = [&](uint32_t x) {
    if (lane == 0)
      atomicOr(lip + (x >> 6), 1ull << (x & 63u));
  };
  auto found_pixel = [&](uint32_t idx, uint32_t sg) {   // decoder: sign, LSP_new
    if (lane == 0) {
      if (!sg)
        atomicAnd(sign + (idx >> 6), ~(1ull << (idx & 63u)));
      if (nfresh < N)
        fresh[nfresh] = idx;
    }
    nfresh++;
  };
  auto list_push = [&](uint32_t lev, uint32_t rlo, uint32_t rhi, int s) {
    const uint32_t idx = rdlane(vCnt, lev);
    const uint32_t slot = rdlane(vOff, lev) + idx;
    if (lev < b.nlists && slot < rdlane(vEnd, lev)) {
      if (lane == 0) {
        runs[slot] = (uint64_t)rlo | ((uint64_t)rhi << 32);
        if (ENC)
          sval[slot] = (int8_t)s;
      }
      wrlane(vCnt, lev, idx + 1);
    }
    else
      err = 2;
  };

  // encoder: largest msb inside each quadrant of a set (BR, BL, TR, TL as bytes 0..3), and the
  // signs of quadrants that are single pixels.  A set of at most 64 samples is loaded once --
  // lane e holds sample (cbx + e % cblx, cby + e / cblx) -- and everything below it is answered
  // from those registers; larger sets are scanned in memory.
  uint32_t scanKS = 0, scanKG = 0;
  uint32_t cbx = 0, cby = 0, cblx = 0, cbly = 0;   // cached block (cblx == 0: none)
  int vCM = -1;
  uint32_t vCX = 0, vCY = 0, vCS = 0;
  auto scan_children = [&](uint32_t px, uint32_t py, uint32_t plx, uint32_t ply) {
    const uint32_t alx = plx - plx / 2, aly = ply - ply / 2;
    const uint32_t area = plx * ply;
    int m0 = -1, m1 = -1, m2 = -1, m3 = -1;
    if (area <= 64) {
      const bool inside = cblx != 0 && px >= cbx && py >= cby && px + plx <= cbx + cblx && py + ply <= cby + cbly;
      if (!inside) {
        cbx = px;
        cby = py;
        cblx = plx;
        cbly = ply;
        vCM = -1;
        vCS = 0;
        vCX = px + lane % plx;
        vCY = py + lane / plx;
        if (lane < area) {
          const uint32_t idx = vCY * dx + vCX;
          vCM = msb[idx];
          vCS = (uint32_t)((sign[idx >> 6] >> (idx & 63u)) & 1ull);
        }
        else
          vCX = 0xffffffffu;   // (never inside any set)
      }
      const bool in = vCX >= px && vCX < px + plx && vCY >= py && vCY < py + ply;
      const bool right = vCX >= px + alx, low = vCY >= py + aly;
      m0 = wave_max(in && right && low ? vCM : -1);
      m1 = wave_max(in && !right && low ? vCM : -1);
      m2 = wave_max(in && right && !low ? vCM : -1);
      m3 = wave_max(in && !right && !low ? vCM : -1);
      // a quadrant of one pixel: BR at (alx, aly), BL at (0, aly), TR at (alx, 0), TL at (0, 0)
      const uint64_t sb = __ballot(vCS != 0);
      auto sign_at = [&](uint32_t x, uint32_t y) -> uint32_t {
        if (x >= px + plx || y >= py + ply)
          return 0u;
        const uint32_t e = (y - cby) * cblx + (x - cbx);
        return (uint32_t)((sb >> e) & 1ull);
      };
      scanKG = sign_at(px + alx, py + aly) | (sign_at(px, py + aly) << 1) | (sign_at(px + alx, py) << 2) |
               (sign_at(px, py) << 3);
    }
    else {
      for (uint32_t i = 0; i < area; i += 64) {
        const uint32_t e = i + lane;
        if (e < area) {
          const uint32_t x = e % plx, y = e / plx;
          const int m = msb[(py + y) * dx + px + x];
          const uint32_t q = (x >= alx ? 1u : 0u) | (y >= aly ? 2u : 0u);   // TL 0, TR 1, BL 2, BR 3
          if (q == 3)
            m0 = max(m0, m);
          else if (q == 2)
            m1 = max(m1, m);
          else if (q == 1)
            m2 = max(m2, m);
          else
            m3 = max(m3, m);
        }
      }
      m0 = wave_max(m0);
      m1 = wave_max(m1);
      m2 = wave_max(m2);
      m3 = wave_max(m3);
      scanKG = 0;
    }
    scanKS = rfl((uint32_t)(m0 & 0xff) | ((uint32_t)(m1 & 0xff) << 8) | ((uint32_t)(m2 & 0xff) << 16) |
                 ((uint32_t)(m3 & 0xff) << 24));
  };

  // ---- encoder: the whole expansion of a significant set of at most 64 samples, by the wave at
  //      once (cf. wave_tree in outlier.hip).  Lane e is sample (px + e % plx, py + e / plx).
  //      Depth by depth every lane moves into the quadrant that holds its sample; the lane of a
  //      node's top-left sample leads the node.  The nodes' largest msb come from LDS atomics, the
  //      bits every significant node's expansion takes (E) bottom-up, the places of the expansions
  //      top-down; the leaders set their children's test and sign bits in an LDS bit buffer, hand
  //      single pixels born insignificant to the LIP, and sets born insignificant to the lists in
  //      the order of their test bits, which is stream order.
  constexpr int kWB = 8;   // depths of a 64-sample set: at most 7
  __shared__ uint32_t wbRect[ENC ? kWB : 1][64], wbE[ENC ? kWB : 1][64], wbO[ENC ? kWB : 1][64];
  __shared__ uint32_t wbBorn[ENC ? kWB : 1][64];
  __shared__ int wbMax[ENC ? kWB : 1][64];
  __shared__ unsigned long long wbBits[ENC ? 64 : 1], wbRank[ENC ? 64 : 1];
  auto wave_block = [&](uint32_t px, uint32_t py, uint32_t plx, uint32_t ply, uint32_t plev, int p) -> bool {
    const uint32_t area = plx * ply;
    const bool mine = lane < area;
    const uint32_t ex = mine ? lane % plx : 0u, ey = mine ? lane / plx : 0u;
    int m = -1;
    uint32_t sg = 0;
    if (mine) {
      const uint32_t idx = (py + ey) * dx + px + ex;
      m = msb[idx];
      sg = (uint32_t)((sign[idx >> 6] >> (idx & 63u)) & 1ull);
    }
    const uint64_t sgnMask = __ballot(sg != 0);
    // ---- structure: my node at every depth (relative rectangle x | y << 8 | lx << 16 | ly << 24)
    uint32_t rx = 0, ry = 0, rlx = plx, rly = ply, D = 0;
    for (;; D++) {
      if (D >= (uint32_t)kWB)
        return false;
      wbRect[D][lane] = rx | (ry << 8) | (rlx << 16) | (rly << 24);
      wbMax[D][lane] = -1;
      wbO[D][lane] = 0;
      wbBorn[D][lane] = 0;
      wbE[D][lane] = 0;
      if (__ballot(mine && rlx * rly > 1) == 0)
        break;
      if (rlx * rly > 1) {
        const uint32_t dlx = rlx / 2, dly = rly / 2, alx = rlx - dlx, aly = rly - dly;
        const bool right = ex >= rx + alx, low = ey >= ry + aly;
        rx += right ? alx : 0u;
        rlx = right ? dlx : alx;
        ry += low ? aly : 0u;
        rly = low ? dly : aly;
      }
    }
    __syncthreads();
    for (uint32_t d = 0; d <= D; d++)
      if (mine) {
        const uint32_t r = wbRect[d][lane];
        atomicMax(&wbMax[d][((r >> 8) & 0xffu) * plx + (r & 0xffu)], m);
      }
    __syncthreads();
    // children of the node (rx, ry, rlx, rly) in the coder's order BR, BL, TR, TL
    auto child = [&](uint32_t r, int k, uint32_t& cx, uint32_t& cy, uint32_t& clx, uint32_t& cly) {
      const uint32_t x0 = r & 0xffu, y0 = (r >> 8) & 0xffu, lx = (r >> 16) & 0xffu, ly = r >> 24;
      const uint32_t dlx = lx / 2, dly = ly / 2, alx = lx - dlx, aly = ly - dly;
      cx = (k == 0 || k == 2) ? x0 + alx : x0;
      cy = (k <= 1) ? y0 + aly : y0;
      clx = (k == 0 || k == 2) ? dlx : alx;
      cly = (k <= 1) ? dly : aly;
    };
    auto leads = [&](uint32_t d, uint32_t& r) -> bool {   // a significant set led by this lane
      if (!mine)
        return false;
      r = wbRect[d][lane];
      const uint32_t lx = (r >> 16) & 0xffu, ly = r >> 24;
      return (r & 0xffu) == ex && ((r >> 8) & 0xffu) == ey && lx * ly > 1 && wbMax[d][lane] == p;
    };
    // ---- expansion sizes, bottom-up
    for (uint32_t d = D; d-- > 0;) {
      uint32_t r;
      if (leads(d, r)) {
        uint32_t E = 0, found = 0;
        for (int k = 0; k < 4; k++) {
          uint32_t cx, cy, clx, cly;
          child(r, k, cx, cy, clx, cly);
          if (clx == 0 || cly == 0)
            continue;
          const uint32_t q = cy * plx + cx;
          const bool sig = wbMax[d + 1][q] == p;
          E += (found || k != 3) ? 1u : 0u;
          if (sig) {
            E += (clx * cly == 1) ? 1u : wbE[d + 1][q];
            found = 1;
          }
        }
        wbE[d][lane] = E;
      }
      __syncthreads();
    }
    const uint32_t total = wbE[0][0];
    if (total + 64 > 64u * 64u)
      return false;
    for (uint32_t w = lane; w < (total + 63) / 64 + 1; w += 64)
      wbBits[w] = 0ull;
    if (lane == 0)
      wbO[0][0] = 1;   // (offset + 1; 0 = the node is not reached)
    __syncthreads();
    auto setbit = [&](uint32_t x) { atomicOr(&wbBits[x >> 6], 1ull << (x & 63u)); };
    // ---- places, bits and births, top-down
    for (uint32_t d = 0; d < D; d++) {
      uint32_t r;
      if (leads(d, r) && wbO[d][lane] != 0) {
        uint32_t o = wbO[d][lane] - 1, found = 0;
        for (int k = 0; k < 4; k++) {
          uint32_t cx, cy, clx, cly;
          child(r, k, cx, cy, clx, cly);
          if (clx == 0 || cly == 0)
            continue;
          const uint32_t q = cy * plx + cx;
          const bool sig = wbMax[d + 1][q] == p, pixel = clx * cly == 1;
          const bool coded = found || k != 3;
          if (coded) {
            if (sig)
              setbit(o);
            else if (!pixel)
              wbBorn[d + 1][q] = o + 1;   // (ranked by the place of its test bit)
            o++;
          }
          if (sig) {
            found = 1;
            if (pixel) {
              if ((sgnMask >> q) & 1ull)
                setbit(o);
              o++;
            }
            else {
              wbO[d + 1][q] = o + 1;
              o += wbE[d + 1][q];
            }
          }
          else if (pixel) {
            const uint32_t idx = (py + cy) * dx + px + cx;
            atomicOr(lip + (idx >> 6), 1ull << (idx & 63u));
          }
        }
      }
      for (uint32_t w = lane; w < (total + 63) / 64 + 1; w += 64)
        wbRank[w] = 0ull;
      __syncthreads();
      const uint32_t bo = mine ? wbBorn[d + 1][lane] : 0u;
      if (bo)
        atomicOr(&wbRank[(bo - 1) >> 6], 1ull << ((bo - 1) & 63u));
      __syncthreads();
      const uint64_t bm = __ballot(bo != 0);
      if (bm) {
        const uint32_t lvl = plev + d + 1;
        const uint32_t have = rdlane(vCnt, lvl), first = rdlane(vOff, lvl) + have;
        const uint32_t nbn = (uint32_t)__popcll(bm);
        if (lvl >= b.nlists || first + nbn > rdlane(vEnd, lvl))
          err = 2;
        else {
          if (bo) {
            uint32_t rank = (uint32_t)__popcll(wbRank[(bo - 1) >> 6] & low_mask((bo - 1) & 63u));
            for (uint32_t w = 0; w < ((bo - 1) >> 6); w++)
              rank += (uint32_t)__popcll(wbRank[w]);
            const uint32_t cr = wbRect[d + 1][lane];
            const uint32_t ax = px + (cr & 0xffu), ay = py + ((cr >> 8) & 0xffu);
            runs[first + rank] = (uint64_t)(ax | (ay << 16)) | ((uint64_t)(((cr >> 16) & 0xffu) | ((cr >> 24) << 16)) << 32);
            sval[first + rank] = (int8_t)wbMax[d + 1][lane];
          }
          wrlane(vCnt, lvl, have + nbn);
        }
      }
      __syncthreads();
    }
    // ---- the bit buffer goes into the stream
    flush_acc();
    for (uint32_t w = lane; w < (total + 63) / 64; w += 64) {
      const uint64_t v = wbBits[w];
      if (v) {
        const uint64_t at = wpos + (uint64_t)w * 64;
        const uint32_t sh = (uint32_t)(at & 63);
        atomicOr(words + (at >> 6), (unsigned long long)(v << sh));
        if (sh && (v >> (64 - sh)))
          atomicOr(words + (at >> 6) + 1, (unsigned long long)(v >> (64 - sh)));
      }
    }
    wpos += total;
    __syncthreads();
    return true;
  };

  // the recursion below one significant set (m_code_S, src/SPECK2D_INT.cpp:58-82): p = plane
  auto expand = [&](uint32_t rlo, uint32_t rhi, uint32_t lev, int p) {
    uint32_t sp = 1;
    wrlane(vRlo, 0, rlo);
    wrlane(vRhi, 0, rhi);
    wrlane(vT, 0, lev << 16);
    if (ENC) {
      if ((rhi & 0xffffu) * (rhi >> 16) <= 64 && (rhi & 0xffffu) * (rhi >> 16) >= b.wbMin &&
          wave_block(rlo & 0xffffu, rlo >> 16, rhi & 0xffffu, rhi >> 16, lev, p))
        return;
      scan_children(rlo & 0xffffu, rlo >> 16, rhi & 0xffffu, rhi >> 16);
      wrlane(vKS, 0, scanKS);
      wrlane(vKG, 0, scanKG);
    }
    while (sp > 0) {
      const uint32_t f = sp - 1;
      const uint32_t state = rdlane(vT, f);
      const uint32_t k = state & 0xffu, found = (state >> 8) & 0xffu, flev = state >> 16;
      if (k == 4) {
        sp--;
        continue;
      }
      const uint32_t plo = rdlane(vRlo, f), phi = rdlane(vRhi, f);
      const uint32_t px = plo & 0xffffu, py = plo >> 16, plx = phi & 0xffffu, ply = phi >> 16;
      const uint32_t dlx = plx / 2, dly = ply / 2, alx = plx - dlx, aly = ply - dly;
      // BR, BL, TR, TL
      const uint32_t cx = (k == 0 || k == 2) ? px + alx : px, cy = (k <= 1) ? py + aly : py;
      const uint32_t clx = (k == 0 || k == 2) ? dlx : alx, cly = (k <= 1) ? dly : aly;
      uint32_t nstate = (k + 1) | (found << 8) | (flev << 16);
      if (clx == 0 || cly == 0) {   // an empty quadrant is skipped (SPECK2D_INT.cpp:62-64)
        wrlane(vT, f, nstate);
        continue;
      }
      const bool coded = found != 0 || k != 3;   // (the top-left quadrant is never empty: it is the last)
      const int cs_ = ENC ? (int)(int8_t)(rdlane(vKS, f) >> (8 * k)) : 0;
      uint32_t sig = 1;
      if (clx == 1 && cly == 1) {   // a pixel (m_process_P)
        const uint32_t idx = cy * dx + cx;
        if (ENC) {
          if (coded) {
            sig = cs_ == p;
            put(sig);
          }
          if (sig)
            put((rdlane(vKG, f) >> k) & 1u);
        }
        else {
          if (coded)
            sig = get();
          if (sig)
            found_pixel(idx, get());
        }
        if (sig)
          nstate |= 1u << 8;
        else
          lip_set(idx);
        wrlane(vT, f, nstate);
      }
      else {                        // a set (m_process_S)
        if (ENC) {
          if (coded) {
            sig = cs_ == p;
            put(sig);
          }
        }
        else if (coded)
          sig = get();
        const uint32_t crlo = cx | (cy << 16), crhi = clx | (cly << 16);
        if (sig) {
          nstate |= 1u << 8;
          wrlane(vT, f, nstate);
          if (ENC && clx * cly <= 64 && clx * cly >= b.wbMin && wave_block(cx, cy, clx, cly, flev + 1, p))
            continue;   // (at most 64 samples: expanded by the wave at once)
          wrlane(vRlo, sp, crlo);
          wrlane(vRhi, sp, crhi);
          wrlane(vT, sp, (flev + 1) << 16);
          if (ENC) {
            scan_children(cx, cy, clx, cly);
            wrlane(vKS, sp, scanKS);
            wrlane(vKG, sp, scanKG);
          }
          sp++;
        }
        else {
          wrlane(vT, f, nstate);
          list_push(flev + 1, crlo, crhi, cs_);
        }
      }
    }
  };

  // src/SPECK2D_INT.cpp:188-218 : the root set (the coarsest approximation) and the type-I set
  uint32_t Isx = approx_len(dx, b.nxforms), Isy = approx_len(dy, b.nxforms), Ilev = b.nxforms;
  {
    int s = -1;
    if (ENC) {   // largest msb inside the root box
      int m = -1;
      const uint32_t area = Isx * Isy;
      for (uint32_t i = lane; i < area; i += 64)
        m = max(m, (int)msb[(i / Isx) * dx + (i % Isx)]);
      s = (int)rfl((uint32_t)wave_max(m));
    }
    list_push(b.nxforms, 0, Isx | (Isy << 16), s);
  }

  // decoder: the values found in plane p get their initial magnitude and join the LSP
  // (SPECK_INT.cpp:462-468; also when the sorting pass ended the decoding, :216-220)
  auto init_fresh = [&](int p) {
    __threadfence_block();
    const CT thr = (CT)1 << p;
    const CT init = thr + thr - thr / 2 - 1;
    const uint32_t lim = min(nfresh, N);
    for (uint32_t k = lane; k < lim; k += 64) {
      const uint32_t x = fresh[k];
      coef[x] = init;
      atomicOr(lsp + (x >> 6), 1ull << (x & 63u));
    }
    nfresh = 0;
  };

  uint64_t total_bits = 0;
  bool stopped = false;
  for (int p = nbp - 1; p >= 0 && !stopped; p--) {
    nfresh = 0;
    __threadfence_block();
    // ================= LIP pass (src/SPECK2D_INT.cpp:13-42) ======================================
    for (uint32_t wb = 0; wb < nw; wb += 64) {
      const uint32_t w = wb + lane;
      const uint64_t lw = w < nw ? lip[w] : 0ull;
      uint64_t nz = __ballot(lw != 0);
      if (nz == 0)
        continue;
      if (ENC) {
        uint64_t plo = 0, phi = 0;
        uint32_t len = 0;
        if (lw) {
          // the msb bytes of the word's 64 coefficients: which of them equal the plane
          const uint4* mp = reinterpret_cast<const uint4*>(msb + (size_t)w * 64);
          uint64_t eq = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint4 v = mp[q];
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
              for (int by = 0; by < 4; by++)
                eq |= (uint64_t)((int)(int8_t)(d[t] >> (8 * by)) == p) << (q * 16 + t * 4 + by);
          }
          const uint64_t sig = lw & eq;
          const uint64_t sgn = sign[w];
          uint64_t m = lw;
          while (m) {
            const uint32_t j = (uint32_t)__ffsll((long long)m) - 1u;
            m &= m - 1;
            if ((sig >> j) & 1ull) {
              if (len < 64)
                plo |= 1ull << len;
              else
                phi |= 1ull << (len - 64);
              len++;
              if ((sgn >> j) & 1ull) {
                if (len < 64)
                  plo |= 1ull << len;
                else
                  phi |= 1ull << (len - 64);
              }
            }
            len++;
          }
          if (sig)
            lip[w] = lw & ~sig;
        }
        flush_acc();
        const uint32_t inc = wave_inclusive_scan<uint32_t>(len);
        const uint32_t total = rdlane(inc, 63);
        if (plo | phi) {
          const uint64_t at = wpos + (inc - len);
          const uint32_t sh = (uint32_t)(at & 63);
          unsigned long long* dst = words + (at >> 6);
          const uint64_t a0 = plo << sh;
          const uint64_t a1 = (sh ? (plo >> (64 - sh)) : 0ull) | (phi << sh);
          const uint64_t a2 = sh ? (phi >> (64 - sh)) : 0ull;
          if (a0)
            atomicOr(dst, (unsigned long long)a0);
          if (a1)
            atomicOr(dst + 1, (unsigned long long)a1);
          if (a2)
            atomicOr(dst + 2, (unsigned long long)a2);
        }
        wpos += total;
      }
      else {
        while (nz) {
          const uint32_t l = (uint32_t)__ffsll((long long)nz) - 1u;
          nz &= nz - 1;
          const uint64_t wv = (uint64_t)rdlane((uint32_t)lw, l) | ((uint64_t)rdlane((uint32_t)(lw >> 32), l) << 32);
          uint64_t keep = wv, m = wv;
          while (m) {
            const uint32_t j = (uint32_t)__ffsll((long long)m) - 1u;
            m &= m - 1;
            if (get()) {
              found_pixel((wb + l) * 64u + j, get());
              keep &= ~(1ull << j);
            }
          }
          if (keep != wv && lane == 0)
            lip[wb + l] = keep;
        }
      }
    }

    // ================= type-S sets, smallest first (SPECK2D_INT.cpp:44-52) ======================
    for (uint32_t lev = b.nlists; lev-- > 0;) {
      const uint32_t n = rdlane(vCnt, lev);
      if (n == 0)
        continue;
      __threadfence_block();
      const uint32_t base = rdlane(vOff, lev);
      uint32_t wr = 0;
      for (uint32_t rd = 0; rd < n; rd += 64) {
        const uint32_t blockN = min(64u, n - rd);
        const bool valid = lane < blockN;
        const uint64_t myRun = valid ? runs[base + rd + lane] : 0ull;
        const int myS = (ENC && valid) ? (int)sval[base + rd + lane] : -1;
        const uint64_t sigmask = ENC ? __ballot(valid && myS == p) : 0ull;
        uint32_t i = 0;
        while (i < blockN) {
          uint32_t z;
          if (ENC) {
            const uint64_t rest = sigmask >> i;
            z = rest ? (uint32_t)__ffsll((long long)rest) - 1u : blockN - i;
          }
          else {
            const uint64_t win = window();
            z = win ? (uint32_t)__ffsll((long long)win) - 1u : 64u;
            z = min(z, blockN - i);
          }
          if (z) {
            if (lane >= i && lane < i + z && wr + (lane - i) != rd + lane) {
              runs[base + wr + (lane - i)] = myRun;
              if (ENC)
                sval[base + wr + (lane - i)] = (int8_t)myS;
            }
            wr += z;
            if (ENC)
              skip_zeros(z);
            else
              rpos += z;
            i += z;
          }
          if (i >= blockN)
            break;
          if (!ENC && (window() & 1ull) == 0)
            continue;
          if (ENC)
            put(1);
          else
            rpos++;
          const uint32_t rlo = rdlane((uint32_t)myRun, i), rhi = rdlane((uint32_t)(myRun >> 32), i);
          i++;
          expand(rlo, rhi, lev, p);
        }
      }
      wrlane(vCnt, lev, wr);
    }

    // ================= the type-I set (SPECK2D_INT.cpp:54,84-98,149-186) ========================
    {
      bool codedI = true;
      while (Ilev > 0) {
        uint32_t sig = 1;
        if (codedI) {
          if (ENC) {
            sig = (int)rdlane(vImax, Ilev) == p;
            put(sig);
          }
          else
            sig = get();
        }
        if (!sig)
          break;
        // m_partition_I: bottom-right, top-right, bottom-left subband of the level; I shrinks
        const uint32_t ax = approx_len(dx, Ilev), ay = approx_len(dy, Ilev);
        const uint32_t bx = approx_len(dx, Ilev - 1), by = approx_len(dy, Ilev - 1);
        const uint32_t dxl = bx - ax, dyl = by - ay;
        const uint32_t klev = Ilev;
        Isx += dxl;
        Isy += dyl;
        Ilev--;
        uint32_t found = 0;
        for (int k = 0; k < 3; k++) {
          const uint32_t kx = (k == 2) ? 0u : ax, ky = (k == 1) ? 0u : ay;
          const uint32_t klx = (k == 2) ? ax : dxl, kly = (k == 1) ? ay : dyl;
          if (klx == 0 || kly == 0)
            continue;
          const int s = ENC ? b.prep[1 + 3 * klev + k] : 0;
          uint32_t ksig;
          if (ENC) {
            ksig = s == p;
            put(ksig);
          }
          else
            ksig = get();
          const uint32_t rlo = kx | (ky << 16), rhi = klx | (kly << 16);
          if (ksig) {
            found = 1;
            expand(rlo, rhi, klev, p);   // (also right for a subband of one coefficient: the
          }                              //  reference splits it into one pixel, significant by inference)
          else
            list_push(klev, rlo, rhi, s);
        }
        codedI = found != 0;
      }
    }
    if (ENC ? (wpos >= budget) : (rpos >= avail)) {
      if (!ENC)
        init_fresh(p);
      stopped = true;
      break;
    }

    // ================= refinement pass (SPECK_INT.cpp:310-357 / 359-469) ========================
    __threadfence_block();
    if (ENC) {
      flush_acc();
      for (uint32_t wb = 0; wb < nw; wb += 64) {
        const uint32_t w = wb + lane;
        uint64_t in = 0, bits = 0;
        if (w < nw) {
          const uint4* mp = reinterpret_cast<const uint4*>(msb + (size_t)w * 64);
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint4 v = mp[q];
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
              for (int by = 0; by < 4; by++)
                in |= (uint64_t)((int)(int8_t)(d[t] >> (8 * by)) > p) << (q * 16 + t * 4 + by);
          }
          if ((size_t)w * 64 + 64 > N)
            in &= low_mask(N - w * 64);
          if (in) {   // the word's 64 coefficients in one sweep (the buffer is padded past N)
            const CT* cp = coef + (size_t)w * 64;
#pragma unroll
            for (int j = 0; j < 64; j++)
              bits |= (uint64_t)((cp[j] >> p) & (CT)1) << j;
            bits &= in;
          }
        }
        // compress the bits of the word under its mask
        uint64_t packed = 0;
        uint32_t cnt = 0;
        {
          uint64_t m = in;
          while (m) {
            const uint32_t j = (uint32_t)__ffsll((long long)m) - 1u;
            m &= m - 1;
            packed |= ((bits >> j) & 1ull) << cnt;
            cnt++;
          }
        }
        const uint32_t inc = wave_inclusive_scan<uint32_t>(cnt);
        if (packed) {
          const uint64_t at = wpos + (inc - cnt);
          const uint32_t sh = (uint32_t)(at & 63);
          atomicOr(words + (at >> 6), (unsigned long long)(packed << sh));
          if (sh && (packed >> (64 - sh)))
            atomicOr(words + (at >> 6) + 1, (unsigned long long)(packed >> (64 - sh)));
        }
        wpos += rdlane(inc, 63);
      }
      if (wpos >= budget) {
        stopped = true;
        break;
      }
    }
    else {
      const CT half = (CT)(((CT)1 << p) / 2);
      bool exhausted = false;
      for (uint32_t wb = 0; wb < nw && !exhausted; wb += 64) {
        const uint32_t w = wb + lane;
        const uint64_t sw = w < nw ? lsp[w] : 0ull;
        if (__ballot(sw != 0) == 0)
          continue;
        const uint32_t cnt = (uint32_t)__popcll(sw);
        const uint32_t inc = wave_inclusive_scan<uint32_t>(cnt);
        const uint64_t left = avail - rpos;   // bits that may still be read (SPECK_INT.cpp:388-389)
        if (cnt) {
          const uint64_t first = inc - cnt;
          if (first < left) {
            const uint64_t at = rpos + first;
            const uint32_t sh = (uint32_t)(at & 63);
            const uint64_t a = word_at(at >> 6), d = word_at((at >> 6) + 1);
            uint64_t bits = sh ? (a >> sh) | (d << (64 - sh)) : a;
            // the word's 64 coefficients in one sweep: sample j takes bit rank(j) of `bits`, as
            // long as the stream has not run out (the buffer is padded past N)
            CT* cp = coef + (size_t)w * 64;
            CT cv[64];
#pragma unroll
            for (int j = 0; j < 64; j++)
              cv[j] = cp[j];
            const uint64_t room = left - first;   // >= 1
#pragma unroll
            for (int j = 0; j < 64; j++) {
              const uint32_t rk = (uint32_t)__popcll(sw & low_mask((uint32_t)j));
              if (((sw >> j) & 1ull) && rk < room) {
                const bool one = (bits >> rk) & 1ull;
                if (p >= 1)
                  cv[j] = one ? cv[j] + half : cv[j] - half;
                else if (one)
                  cv[j] = cv[j] + 1;
              }
            }
#pragma unroll
            for (int j = 0; j < 64; j++)
              cp[j] = cv[j];
          }
        }
        const uint32_t total = rdlane(inc, 63);
        if ((uint64_t)total >= left) {
          rpos = avail;
          exhausted = true;
        }
        else
          rpos += total;
      }
    }
    if (!ENC) {
      init_fresh(p);
      if (rpos >= avail) {
        stopped = true;
        break;
      }
    }
  }

  if (ENC) {
    flush_acc();
    total_bits = wpos;
    if (lane == 0) {
      const uint64_t keep = min(total_bits, budget);
      const uint64_t payload = (keep + 7) / 8;
      cs.stream_len = 17 + 9 + payload;
      cs.nbp = nbp;
      cs.total_bits = total_bits;
      if (rate_mode)
        cs.need_retry = (!wide_pass && (9 + payload) * 8 < raw_budget) ? 1u : 0u;
      if (err)
        b.prep[1 + 3 * kS2MaxLevels] = (int)err;
    }
  }
  else if (err && lane == 0)
    b.dst[0].error = err;
}

}  // namespace

size_t speck2d_list_entries(Speck2dBufs& b)
{
  // level L holds sets of about (dx / 2^L) x (dy / 2^L): at most 4^L of them, never more than
  // there are coefficient pairs
  uint64_t off = 0;
  const uint64_t most = (uint64_t)b.N / 2 + 4;
  for (uint32_t l = 0; l <= (uint32_t)kS2MaxLevels; l++) {
    b.levelOff[l] = (uint32_t)std::min<uint64_t>(off, 0xffffffffull);
    if (l < b.nlists)
      off += l < 16 ? std::min<uint64_t>(3ull << (2 * l), most) : most;
  }
  return (size_t)off + 64;
}

int launch_speck2d_encode(hipStream_t st, const Speck2dBufs& b_, uint64_t raw_budget, bool rate_mode,
                          bool wide_pass)
{
  Speck2dBufs b = b_;
  static const uint32_t wbMin = tune_getenv("SPERR_HIP_WB_MIN") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_WB_MIN")) : 4u;
  b.wbMin = wbMin;
  uint64_t budget = ~0ull;
  if (raw_budget != 0) {  // SPECK_INT.cpp:48-58
    budget = raw_budget;
    while (budget % 8)
      budget++;
  }
  HIP_CHECK(hipMemsetAsync(b.prep, 0xff, (2 + 3 * kS2MaxLevels) * sizeof(int32_t), st));
  HIP_CHECK(hipMemsetAsync(b.lip, 0, (size_t)(b.nw + 2) * 8, st));
  HIP_CHECK(hipMemsetAsync(b.stream, 0, b.streamWords * 8, st));
  LAUNCH_K(k_s2_prep, dim3(std::min<uint32_t>(1024, (b.N + kThreads - 1) / kThreads)), dim3(kThreads),
           0, st, b);
  if (wide_pass)
    LAUNCH_K((k_speck2d<true, uint64_t>), dim3(1), dim3(64), 0, st, b, budget, raw_budget,
             rate_mode ? 1 : 0, 1);
  else
    LAUNCH_K((k_speck2d<true, uint32_t>), dim3(1), dim3(64), 0, st, b, budget, raw_budget,
             rate_mode ? 1 : 0, 0);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_speck2d_decode(hipStream_t st, const Speck2dBufs& b, bool wide_pass)
{
  HIP_CHECK(hipMemsetAsync(b.lip, 0, (size_t)(b.nw + 2) * 8, st));
  HIP_CHECK(hipMemsetAsync(b.lsp, 0, (size_t)(b.nw + 2) * 8, st));
  if (wide_pass)
    LAUNCH_K((k_speck2d<false, uint64_t>), dim3(1), dim3(64), 0, st, b, ~0ull, 0ull, 0, 1);
  else
    LAUNCH_K((k_speck2d<false, uint32_t>), dim3(1), dim3(64), 0, st, b, ~0ull, 0ull, 0, 0);
  HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace sperrhip
