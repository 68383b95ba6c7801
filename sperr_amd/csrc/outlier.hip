// outlier.hip -- point-wise error mode (CompMode::PWE): the values whose reconstruction error
// exceeds the tolerance are found on the device, quantised in units of the tolerance
// (src/Outlier_Coder.cpp:179-197) and coded by the reference's 1D set-partitioning coder
// (src/SPECK1D_INT.cpp, src/SPECK1D_INT_ENC.cpp, src/SPECK1D_INT_DEC.cpp) over the length-N array
// that is zero everywhere else.
//
// The array is sparse, so the coder never touches N values:
//   * encoder: the outliers are kept as sorted (position, magnitude, sign, msb) records; per bit
//     plane a position bitmask of the outliers at or above the threshold, with a popcount prefix,
//     turns "is the run [start, start + len) significant" into a difference of two counts, and
//     once a run holds at most 64 of them their positions sit in a register.
//   * decoder: refinement bits are stored as one dense bit plane per threshold (deposited under the
//     LSP bitmask); magnitudes are assembled at the end for the values that were found.
// One wavefront codes one chunk.  The bit stream of a set-partitioning coder is inherently ordered
// (every bit's position depends on all earlier decisions), so the set recursion is walked by the
// wave in lock step; the passes that the order allows are spread over the 64 lanes: runs of
// insignificant list entries are skipped 64 at a time, the encoder's LIP and refinement passes
// and the decoder's refinement pass place a whole word of results at once.
#include "outlier.h"

namespace sperrhip {
namespace {

__device__ __forceinline__ uint64_t low_mask(uint32_t n)   // n in [0, 64]
{
  return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
}

// ------------------------------------------------------------------------------------------
// detection (src/SPECK_FLT.cpp:470-476) and quantisation (src/Outlier_Coder.cpp:88-100,179-197)
// ------------------------------------------------------------------------------------------
template <typename T, int PASS>
__global__ void __launch_bounds__(kThreads)
k_outlier_scan(const T* __restrict__ vol, VolDesc vd, const ChunkGeom* geom, uint32_t cx,
               uint32_t cy, const double* __restrict__ vals, size_t valsStride,
               const CoderState* cst, double tol, OutlierBufs b)
{
  const uint32_t c = blockIdx.y;
  const CoderState& cs = cst[c];
  if (cs.is_const)
    return;
  OutlierChunk& oc = b.oc[c];
  if (PASS > 0 && oc.flagged == 0)
    return;
  const ChunkGeom g = geom[c];
  const double mean = cs.mean, inv = 1.0 / tol;
  const size_t vx = vd.dims[0], vy = vd.dims[1];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, nwave = kThreads / 64;
  const double* in = vals + c * valsStride;
  const unsigned long long widthMask = PASS > 0 ? oc.widthMask : 0ull;
  uint32_t cnt = 0;
  unsigned long long best = 0;
  for (uint32_t w = blockIdx.x * nwave + wave; w < b.nw; w += gridDim.x * nwave) {
    const uint32_t i = w * 64u + lane;
    bool flag = false;
    double diff = 0.0;
    // (pass 0 leaves the word's flags in maskGE, which nobody needs before the 1D coder clears it: the later passes
    //  read the volume and the reconstruction only for the samples that were flagged -- round 5)
    unsigned long long fw = ~0ull;
    if (PASS > 0) {
      fw = b.maskGE[c * b.wordStride + w];
      if (fw == 0ull) {   // (uniform) no outlier in the word
        if (PASS == 1 && lane == 0) {
          b.outPre[c * b.wordStride + w] = 0;
          b.signMask[c * b.wordStride + w] = 0;
        }
        continue;
      }
    }
    if (i < b.N && ((fw >> lane) & 1ull)) {
      const uint32_t x = i % cx, r = i / cx;
      const uint32_t y = r % cy, z = r / cy;
      // the conditioned input, as the first lifting pass computed it (Conditioner.cpp:46-50)
      const double orig =
          (double)vol[((size_t)(g.org[2] + z) * vy + (g.org[1] + y)) * vx + g.org[0] + x] - mean;
      diff = orig - in[i];
      flag = fabs(diff) > tol;
    }
    if (PASS == 0) {
      const unsigned long long fl = __ballot(flag);
      if (lane == 0)
        b.maskGE[c * b.wordStride + w] = fl;
      cnt += (uint32_t)__popcll(fl);
      if (flag) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(diff));
        best = key > best ? key : best;
      }
    }
    else {
      long long ll = 0;
      unsigned long long m = 0;
      if (flag) {
        ll = __double2ll_rn(diff * inv);
        m = (unsigned long long)(ll < 0 ? -ll : ll) & widthMask;
      }
      const bool f2 = flag && m != 0;
      const unsigned long long word = __ballot(f2);
      if (PASS == 1) {
        const unsigned long long sgw = __ballot(f2 && ll >= 0);
        if (lane == 0) {
          b.outPre[c * b.wordStride + w] = (uint32_t)__popcll(word);
          b.signMask[c * b.wordStride + w] = sgw;
        }
      }
      else if (f2) {
        const uint32_t k = b.outPre[c * b.wordStride + w] + (uint32_t)__popcll(word & low_mask(lane));
        if (k < b.kStride) {
          b.pos[c * b.kStride + k] = i;
          b.mag[c * b.kStride + k] = m;
          b.sgn[c * b.kStride + k] = ll >= 0 ? 1 : 0;
          b.msb[c * b.kStride + k] = (uint8_t)(63 - __clzll((long long)m));
        }
        best = m > best ? m : best;
      }
    }
  }
  if (PASS == 0) {
    if (lane == 0 && cnt)
      atomicAdd(&oc.flagged, cnt);
    if (best)
      atomicMax(&oc.maxErrKey, best);
  }
  else if (PASS == 2 && best)
    atomicMax(&oc.maxMag, best);
}

// per-word counts -> exclusive prefix, in place; [nw] and oc.count receive the total
__global__ void __launch_bounds__(1024) k_outlier_prefix(OutlierBufs b)
{
  const uint32_t c = blockIdx.x;
  OutlierChunk& oc = b.oc[c];
  if (oc.flagged == 0)
    return;
  __shared__ uint32_t sm[1024 / 64 + 1];
  uint32_t* pre = b.outPre + c * b.wordStride;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < b.nw; base += 1024) {
    const uint32_t w = base + threadIdx.x;
    const uint32_t v = w < b.nw ? pre[w] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<uint32_t>(v, sm, &total) + carry;
    if (w < b.nw)
      pre[w] = ex;
    carry += total;
  }
  if (threadIdx.x == 0) {
    pre[b.nw] = carry;
    oc.count = carry;
    if (carry > b.kStride)
      oc.error = 1;
  }
}

// ------------------------------------------------------------------------------------------
// SPECK1D, one wavefront per chunk
// ------------------------------------------------------------------------------------------
// The set recursion is strictly sequential, so it is written for the scalar unit: every value that
// steers it is wave-uniform (readfirstlane / readlane results), and the small indexed arrays it
// needs -- the recursion stack and the list lengths -- live in vector registers addressed by LANE
// (v_readlane / v_writelane with a uniform index), not in LDS or scratch.
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
// inclusive prefix sum over the wavefront with DPP moves (no LDS round trips)
__device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31
  return v;
}
__device__ __forceinline__ void wrlane(uint32_t& v, uint32_t l, uint32_t val)
{
  v = threadIdx.x == l ? val : v;   // (val and l are wave-uniform)
}

template <bool ENC>
__global__ void __launch_bounds__(64)
k_speck1d(OutlierBufs b)
{
  const uint32_t c = blockIdx.x;
  OutlierChunk& oc = b.oc[c];
  if (ENC) {
    if (oc.flagged == 0 || oc.error)
      return;
    if (oc.count == 0) {   // every magnitude fell to zero: {0 planes, 0 bits} (SPECK_INT.cpp:128-135)
      if (threadIdx.x == 0) {
        oc.nbp = 0;
        oc.total_bits = 0;
      }
      return;
    }
  }
  else if (!oc.has || oc.nbp == 0)
    return;
  const uint32_t lane = threadIdx.x;
  const uint32_t N = b.N, nw = b.nw;
  const uint32_t K = ENC ? rfl(oc.count) : 0u;
  const int nbp = (int)rfl((uint32_t)(ENC ? 64 - __clzll((long long)oc.maxMag) : oc.nbp));

  uint64_t* lip = b.lip + c * b.wordStride;
  uint64_t* runs = b.runs + c * b.runStride;
  unsigned long long* words = reinterpret_cast<unsigned long long*>(b.stream + c * b.streamStride);
  // encoder: the outliers (ascending position) and, per plane, those at or above the threshold
  const uint32_t* opos = b.pos + c * b.kStride;
  const uint64_t* omag = ENC ? b.mag + c * b.kStride : nullptr;
  const uint8_t* osgn = b.sgn + c * b.kStride;
  const uint8_t* omsb = ENC ? b.msb + c * b.kStride : nullptr;
  uint32_t* posGE = ENC ? b.posGE + c * b.kStride : nullptr;
  uint8_t* sgnGE = ENC ? b.sgnGE + c * b.kStride : nullptr;
  unsigned long long* maskGE = ENC ? reinterpret_cast<unsigned long long*>(b.maskGE + c * b.wordStride) : nullptr;
  unsigned long long* maskEQ = ENC ? reinterpret_cast<unsigned long long*>(b.maskEQ + c * b.wordStride) : nullptr;
  uint32_t* cpos = ENC ? b.cpos + c * b.wordStride : nullptr;
  const uint64_t* signMask = ENC ? b.signMask + c * b.wordStride : nullptr;
  // decoder
  uint64_t* lsp = ENC ? nullptr : b.lsp + c * b.wordStride;
  uint64_t* planeBits = ENC ? nullptr : b.planeBits + c * b.planeStride;
  uint32_t* fpos = b.pos + c * b.kStride;      // values found
  uint8_t* fmeta = b.sgn + c * b.kStride;

  // lane-indexed registers: list length / first slot / end slot of level `lane`; recursion stack
  uint32_t vCnt = 0;
  const uint32_t vOff = b.levelOff[min(lane, (uint32_t)kO1MaxLevels)];
  const uint32_t vEnd = b.levelOff[min(lane + 1u, (uint32_t)kO1MaxLevels)];
  uint32_t vS = 0, vL = 0, vA = 0, vM = 0, vB = 0, vT = 0;   // start, len, counts at start / mid / end, state
  uint32_t err = 0;

  // ---- bit writer (encoder): bits leave in order through `acc`; whole-word results of the
  //      lane-parallel passes are OR-ed straight into the zeroed stream
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
  // ---- bit reader (decoder): 64 words of the stream sit in a register pair, lane = word, so that
  //      the serial parse waits for memory once per 4032 bits instead of once per word
  uint64_t rpos = 0, vW = 0, cBase = 1ull << 62;   // (cBase: index of the word in lane 0; none yet)
  uint64_t cw0 = 0, cw1 = 0, cwi = 1ull << 62;      // the two words the window lies in, as scalars (none yet)
  auto window = [&]() -> uint64_t {   // the next 64 bits from rpos on
    const uint64_t wi = rpos >> 6;
    if (wi != cwi) {
      if (wi - cBase >= 63u) {
        cBase = wi;
        vW = wi + lane < b.streamStride ? words[wi + lane] : 0ull;
      }
      const uint32_t k = (uint32_t)(wi - cBase);
      cw0 = wi == cwi + 1 ? cw1 : (uint64_t)rdlane((uint32_t)vW, k) | ((uint64_t)rdlane((uint32_t)(vW >> 32), k) << 32);
      cw1 = (uint64_t)rdlane((uint32_t)vW, k + 1u) | ((uint64_t)rdlane((uint32_t)(vW >> 32), k + 1u) << 32);
      cwi = wi;
    }
    const uint32_t sh = (uint32_t)(rpos & 63);
    return sh ? (cw0 >> sh) | (cw1 << (64 - sh)) : cw0;
  };
  auto get = [&]() -> uint32_t {
    const uint32_t bit = (uint32_t)(window() & 1ull);
    rpos++;
    return bit;
  };

  // ---- encoder: number of outliers at or above the threshold before position x
  auto rank_mem = [&](uint32_t x) -> uint32_t {
    const uint32_t w = x >> 6;
    const uint32_t pre = cpos[w];
    const uint64_t m = maskGE[w];
    return rfl(pre + (uint32_t)__popcll(m & low_mask(x & 63u)));
  };
  auto rank_in = [&](uint32_t x, uint32_t a, uint32_t e) -> uint32_t {   // a, e: counts at the run's ends
    return e == a ? a : rank_mem(x);
  };

  uint32_t nfound = 0, lspDone = 0;   // decoder: values found so far / of them, already in the LSP mask
  auto lip_set = [&](uint32_t x) {
    if (lane == 0)
      atomicOr(reinterpret_cast<unsigned long long*>(lip) + (x >> 6), 1ull << (x & 63u));
  };
  // (the bookkeeping of level `lev` sits in lane `lev`: that lane does the append itself)
  uint32_t vErr = 0;
  auto list_push = [&](uint32_t lev, uint32_t start, uint32_t len) {
    if (lane == lev) {
      if (vOff + vCnt < vEnd) {
        runs[vOff + vCnt] = (uint64_t)start | ((uint64_t)len << 32);
        vCnt++;
      }
      else
        vErr = 2;   // list storage exhausted (cannot happen with the host's bounds)
    }
  };

  // ---- encoder: the whole expansion of a significant run, 64 of its outliers at a time, lane =
  //      outlier (all of them at or above this plane's threshold, ascending position).  The code
  //      below a run is one PATH per outlier: the closing '1' of the parked right half it lies in
  //      (not for the run's first outlier, whose run got its '1' from the list), one bit per level
  //      from there down ('1': it lies in the left half and the right one is parked; '0': the left
  //      half is born insignificant and the right one is significant without a bit), its sign, and a
  //      '0' for every parked right half on its way that the next outlier does not lie in.  A node
  //      that also holds the previous outlier was coded by an earlier path, so every lane finds its
  //      own bits from its position and its two neighbours alone; all lanes walk down from the run
  //      together, one level per round, and the halves born at that level -- at most one per lane,
  //      in lane order, which is stream order -- are appended to the level's list at once.
  auto expand_enc = [&](uint32_t ns, uint32_t nl, uint32_t nlev, uint32_t a, uint32_t e) {
    flush_acc();
    for (uint32_t k0 = a; k0 < e; k0 += 64) {
      const uint32_t k = k0 + lane;
      const bool mine = k < e;
      const uint32_t x = mine ? posGE[k] : 0u;
      const bool hasPrev = mine && k > a, hasNext = mine && k + 1 < e;
      const uint32_t prev = hasPrev ? posGE[k - 1] : 0u;
      const uint32_t next = hasNext ? posGE[k + 1] : 0u;
      uint64_t code = hasPrev ? 1ull : 0ull;
      uint32_t nbits = hasPrev ? 1u : 0u, nz = 0;
      uint32_t s0 = ns, l0 = nl;
      for (uint32_t d = 0;; d++) {
        const bool act = mine && l0 > 1;
        if (__ballot(act) == 0)
          break;
        const uint32_t lvl = nlev + d + 1;   // list level of the halves of this round's nodes
        const uint32_t h0 = l0 - l0 / 2, r0 = l0 / 2;
        const bool left = x < s0 + h0;
        const bool owned = !(hasPrev && prev >= s0);          // no earlier outlier of the run in this node
        const bool nextIn = hasNext && next < s0 + l0;         // the next outlier lies in this node
        if (act && owned) {
          code |= (uint64_t)(left ? 1u : 0u) << nbits;
          nbits++;
        }
        const bool closes = act && left && !nextIn;            // the parked right half gets its '0' after this path
        nz += closes ? 1u : 0u;
        const bool born = (act && owned && !left) || closes;
        const uint32_t bs = left ? s0 + h0 : s0, bl = left ? r0 : h0;
        if (born && bl == 1)
          atomicOr(reinterpret_cast<unsigned long long*>(lip) + (bs >> 6), 1ull << (bs & 63u));
        const uint64_t bm = __ballot(born && bl > 1);
        if (bm) {
          const uint32_t have = rdlane(vCnt, lvl), first = rdlane(vOff, lvl) + have;
          const uint32_t nbn = (uint32_t)__popcll(bm);
          if (lvl >= b.nlists || first + nbn > rdlane(vEnd, lvl))
            err = 2;
          else {
            if (born && bl > 1)
              runs[first + (uint32_t)__popcll(bm & low_mask(lane))] = (uint64_t)bs | ((uint64_t)bl << 32);
            wrlane(vCnt, lvl, have + nbn);
          }
        }
        if (act) {
          if (left)
            l0 = h0;
          else {
            s0 += h0;
            l0 = r0;
          }
        }
      }
      // sign, then the closing zeros
      if (mine) {
        code |= (uint64_t)sgnGE[k] << nbits;
        nbits++;
      }
      const uint32_t len = mine ? nbits + nz : 0u;   // (at most 1 + 31 + 1 + 31 bits)
      const uint32_t inc = wave_scan_dpp(len);
      if (mine && code) {
        const uint64_t at = wpos + (inc - len);
        const uint32_t sh = (uint32_t)(at & 63);
        atomicOr(words + (at >> 6), (unsigned long long)(code << sh));
        if (sh && (code >> (64 - sh)))
          atomicOr(words + (at >> 6) + 1, (unsigned long long)(code >> (64 - sh)));
      }
      wpos += rdlane(inc, 63);
    }
  };

  // ---- encoder: ALL significant entries of a block of 64 list entries at once.  Most significant runs
  //      hold a handful of outliers, so a block per run leaves most lanes idle: here the outliers of the
  //      block's significant entries fill the lanes in list order (a lane finds its entry by a search over
  //      the entries' running counts).  The '1' of a run's list test is the leading bit of its first
  //      outlier's code, like the closing '1' of the others; an insignificant entry is one position.
  auto expand_block = [&](uint64_t myRun, uint32_t myA, uint32_t myB, uint64_t sigmask, uint32_t blockN,
                          uint32_t lev) {
    flush_acc();
    const bool sig = ((sigmask >> lane) & 1ull) != 0;
    const uint32_t cnt = sig ? myB - myA : 0u;
    const uint32_t incl = wave_scan_dpp(cnt);
    const uint32_t T = rdlane(incl, 63);
    const uint32_t cstart = incl - cnt;
    const uint32_t insigBefore = (uint32_t)__popcll(~sigmask & low_mask(lane));
    const uint32_t nInsig = blockN - (uint32_t)__popcll(sigmask);
    uint64_t carry = 0;   // code bits of the outliers of the chunks before
    for (uint32_t c0 = 0; c0 < T; c0 += 64) {
      const uint32_t o = c0 + lane;
      const bool mine = o < T;
      uint32_t lo = 0, hi = 64;   // the first entry whose running count exceeds o
#pragma unroll
      for (int it = 0; it < 7; it++) {   // (65 possible answers)
        const uint32_t mid = min((lo + hi) >> 1, 63u);
        const uint32_t v = (uint32_t)__shfl((int)incl, (int)mid, 64);
        if (lo < hi) {
          if (v <= o)
            lo = mid + 1;
          else
            hi = mid;
        }
      }
      const int j = (int)min(lo, 63u);
      const uint32_t ja = (uint32_t)__shfl((int)myA, j, 64), jcs = (uint32_t)__shfl((int)cstart, j, 64),
                     jcnt = (uint32_t)__shfl((int)cnt, j, 64), jins = (uint32_t)__shfl((int)insigBefore, j, 64);
      const uint32_t js = (uint32_t)__shfl((int)(uint32_t)myRun, j, 64),
                     jl = (uint32_t)__shfl((int)(uint32_t)(myRun >> 32), j, 64);
      const uint32_t k = ja + (o - jcs);
      const uint32_t x = mine ? posGE[k] : 0u;
      const bool hasPrev = mine && o > jcs, hasNext = mine && o + 1 < jcs + jcnt;
      const uint32_t prev = hasPrev ? posGE[k - 1] : 0u;
      const uint32_t next = hasNext ? posGE[k + 1] : 0u;
      uint64_t code = mine ? 1ull : 0ull;
      uint32_t nbits = mine ? 1u : 0u, nz = 0;
      uint32_t s0 = js, l0 = mine ? jl : 0u;
      for (uint32_t d = 0;; d++) {
        const bool act = mine && l0 > 1;
        if (__ballot(act) == 0)
          break;
        const uint32_t lvl = lev + d + 1;
        const uint32_t h0 = l0 - l0 / 2, r0 = l0 / 2;
        const bool left = x < s0 + h0;
        const bool owned = !(hasPrev && prev >= s0);
        const bool nextIn = hasNext && next < s0 + l0;
        if (act && owned) {
          code |= (uint64_t)(left ? 1u : 0u) << nbits;
          nbits++;
        }
        const bool closes = act && left && !nextIn;
        nz += closes ? 1u : 0u;
        const bool born = (act && owned && !left) || closes;
        const uint32_t bs = left ? s0 + h0 : s0, bl = left ? r0 : h0;
        if (born && bl == 1)
          atomicOr(reinterpret_cast<unsigned long long*>(lip) + (bs >> 6), 1ull << (bs & 63u));
        const uint64_t bm = __ballot(born && bl > 1);
        if (bm) {
          const uint32_t have = rdlane(vCnt, lvl), first = rdlane(vOff, lvl) + have;
          const uint32_t nbn = (uint32_t)__popcll(bm);
          if (lvl >= b.nlists || first + nbn > rdlane(vEnd, lvl))
            err = 2;
          else {
            if (born && bl > 1)
              runs[first + (uint32_t)__popcll(bm & low_mask(lane))] = (uint64_t)bs | ((uint64_t)bl << 32);
            wrlane(vCnt, lvl, have + nbn);
          }
        }
        if (act) {
          if (left)
            l0 = h0;
          else {
            s0 += h0;
            l0 = r0;
          }
        }
      }
      if (mine) {
        code |= (uint64_t)sgnGE[k] << nbits;
        nbits++;
      }
      const uint32_t len = mine ? nbits + nz : 0u;
      const uint32_t inc = wave_scan_dpp(len);
      if (mine) {   // (the leading '1' is always there)
        const uint64_t at = wpos + jins + carry + (inc - len);
        const uint32_t sh = (uint32_t)(at & 63);
        atomicOr(words + (at >> 6), (unsigned long long)(code << sh));
        if (sh && (code >> (64 - sh)))
          atomicOr(words + (at >> 6) + 1, (unsigned long long)(code >> (64 - sh)));
      }
      carry += rdlane(inc, 63);
    }
    wpos += nInsig + carry;
  };

  // ---- decoder: the recursion below one significant run (m_code_S), written as a descent with a
  //      stack of the right halves that still wait for their test bit: a '1' goes down the left
  //      half and parks the right one, a '0' hands the left half to its list (or the LIP) and goes
  //      down the right half, which is then significant without a test bit
  auto found_pixel = [&](uint32_t idx, int p) {
    const uint32_t sg = get();
    if (nfound < b.kStride && lane == 0) {
      fpos[nfound] = idx;
      fmeta[nfound] = (uint8_t)((uint32_t)p | (sg << 7));
    }
    nfound++;
  };
  int curPlane = 0;
  // The code below a significant run is a sequence of PATHS: from a run, one bit per level ('1':
  // go into the left half and park the right one, '0': the left half is born insignificant and the
  // right one is significant without a bit) down to a single value and its sign; then one closing
  // bit per parked half, innermost first: '0' hands it to its list (or the LIP), '1' starts the next
  // path there (a parked single value: its sign follows).
  //
  // Round 3: the serial part, the CHAIN, only finds out where every path starts and ends.  A half is
  // known by its depth below the list entry and the way to it (R: bit j = "went right at depth j"),
  // its length in closed form ((L >> d) + (R mod 2^d < L mod 2^d), the rule of speck_tree.h), the
  // parked right halves by a mask of depths `m` -- so a path costs a few dozen scalar instructions on
  // the stream window (it ends at the first step whose chosen half is one value: step e - 1 or e for
  // a half of 2^e .. 2^(e+1) - 1 values) and leaves one RECORD in lane `nrec` of five registers.
  // What the paths mean for the lists, the LIP and the values found is worked out 64 records at a
  // time (`flush_paths`, lane = record): all lanes walk down from their list entry together, one
  // depth per round; at a depth a record has at most one half to hand over -- the left half it
  // passed by on the right during its own path, or the parked right half that one of its closing
  // zeros released -- and the halves of a round join that level's list in lane order, which is
  // stream order.  tests/model/speck_model.cpp::model_speck1d_decode_batched is the CPU model.
  // (Before: a whole path per step with lane = level, about 1400 cycles per path.)
  uint32_t rES = 0, rEL = 0, rR = 0, rMeta = 0, rClosed = 0, nrec = 0;
  // (records of chain_p2 below hold the path's own bits and its first depth instead of R and the meta word: `recRaw` marks
  //  them, flush_paths works R out for all of them at once; `rCarry`: R of the last record flushed)
  uint64_t recRaw = 0;
  uint32_t rCarry = 0;
#ifdef SPERR_1D_STAMPS
  long long tkLip = 0, tkLis = 0, tkChain = 0, tkFlush = 0, tkRef = 0, tkLsp = 0, tk0 = 0;
  uint32_t nPaths = 0, nEntries = 0, nFlush = 0;
#define STAMP_BEGIN() (tk0 = clock64())
#define STAMP_END(acc) (acc += clock64() - tk0)
#else
#define STAMP_BEGIN() ((void)0)
#define STAMP_END(acc) ((void)0)
#endif
  auto flush_paths = [&](uint32_t lev) {
    if (nrec == 0)
      return;
#ifdef SPERR_1D_STAMPS
    const long long tf0 = clock64();
    nFlush++;
    nPaths += nrec;
#endif
    const bool valid = lane < nrec;
    if (recRaw) {
      // A path from depth u of a run of 2^g values: its g - u bits (a '0' = went right) and its sign are rR's low bits.
      // R of a record = R of the record before it below depth u - 1, a 1 at depth u - 1, the path's bits from u on: the
      // records' (keep, value) pairs compose, so a scan over the lanes gives every R (a run's first record keeps nothing).
      const bool raw = ((recRaw >> lane) & 1ull) != 0;
      const uint32_t u0 = raw ? rMeta - 1u : 0u, g = 31u - (uint32_t)__clz((int)(rEL | 1u)), ns = raw ? g - u0 : 0u;
      uint32_t keep = 0, val = rR;
      if (raw) {
        keep = u0 ? (1u << (u0 - 1u)) - 1u : 0u;
        val = (u0 ? 1u << (u0 - 1u) : 0u) | ((~rR & ((1u << ns) - 1u)) << u0);
      }
      const uint32_t sg = (rR >> ns) & 1u;
      for (uint32_t off = 1; off < 64u; off <<= 1) {
        const uint32_t pk = (uint32_t)__shfl_up((int)keep, off, 64), pv = (uint32_t)__shfl_up((int)val, off, 64);
        if (lane >= off) {
          val = (pv & keep) | val;
          keep &= pk;
        }
      }
      val = (rCarry & keep) | val;
      if (raw) {
        rR = val;
        rMeta = u0 | (g << 8) | (sg << 16);
      }
      recRaw = 0;
    }
    rCarry = rdlane(rR, nrec - 1u);
    const uint32_t u = rMeta & 0xffu, tEnd = (rMeta >> 8) & 0xffu;
    uint32_t s = rES, l = rEL;
    for (uint32_t j = 0;; j++) {
      const bool act = valid && j < tEnd;
      if (__ballot(act) == 0)
        break;
      const uint32_t h0 = l - l / 2, r0 = l / 2;
      const bool right = ((rR >> j) & 1u) != 0;
      const bool born = act && (right ? j >= u : ((rClosed >> (30u - j)) & 1u) != 0);   // (depth j + 1, bit-reversed)
      const uint32_t bs = right ? s : s + h0, bl = right ? h0 : r0;
      if (born && bl == 1)
        atomicOr(reinterpret_cast<unsigned long long*>(lip) + (bs >> 6), 1ull << (bs & 63u));
      const uint64_t bm = __ballot(born && bl > 1);
      if (bm) {
        const uint32_t lvl = lev + j + 1u;
        if (lvl >= b.nlists)
          err = 2;
        else {
          const uint32_t have = rdlane(vCnt, lvl), first = rdlane(vOff, lvl) + have;
          const uint32_t nbn = (uint32_t)__popcll(bm);
          if (first + nbn > rdlane(vEnd, lvl))
            err = 2;   // list storage exhausted (cannot happen with the host's bounds)
          else {
            if (born && bl > 1)
              runs[first + (uint32_t)__popcll(bm & low_mask(lane))] = (uint64_t)bs | ((uint64_t)bl << 32);
            wrlane(vCnt, lvl, have + nbn);
          }
        }
      }
      if (act) {
        if (right) {
          s += h0;
          l = r0;
        }
        else
          l = h0;
      }
    }
    if (valid && nfound + lane < b.kStride) {
      fpos[nfound + lane] = s;
      fmeta[nfound + lane] = (uint8_t)((uint32_t)curPlane | (((rMeta >> 16) & 1u) << 7));
    }
    nfound += nrec;
    nrec = 0;
#ifdef SPERR_1D_STAMPS
    tkFlush += clock64() - tf0;
#endif
  };
  // (P2: the run's length is a power of two -- every run of a chunk of 2^k values -- so all halves are
  //  and a path from depth u is log2(el) - u steps; the parked depths are kept bit-REVERSED, `mr` bit
  //  31 - d for depth d, so that the innermost parked halves are the lowest set bits: closing z of them
  //  is z times x & (x - 1))
  auto expand_chain = [&](auto p2tag, uint32_t es, uint32_t el, uint32_t lev) {   // a run of at least two values
    constexpr bool P2 = decltype(p2tag)::value;
    const uint32_t lgEl = 31u - (uint32_t)__clz((int)el);
    uint32_t R = 0, mr = 0, u = 0, lo = el;
    for (;;) {
      const uint64_t peek = window();
      uint32_t nsteps;
      if (P2)
        nsteps = lgEl - u;
      else {
        nsteps = 0;
        if (lo > 1) {
          const uint32_t t0 = 30u - (uint32_t)__clz((int)lo), mk = (1u << t0) - 1u;
          const uint32_t lt = (lo >> t0) + (((~(uint32_t)peek & mk) < (lo & mk)) ? 1u : 0u);
          const uint32_t bt = (uint32_t)(peek >> t0) & 1u;
          nsteps = (lt == 2u || (lt == 3u && bt == 0u)) ? t0 + 1u : t0 + 2u;
        }
      }
      if (u + nsteps > 31u) {   // (cannot happen: N < 2^32)
        err = 2;
        return;
      }
      const uint32_t pm = (1u << nsteps) - 1u;
      R |= (~(uint32_t)peek & pm) << u;
      mr |= __brev((uint32_t)peek & pm) >> (u + 1u);   // a '1' at step j parks the right half of depth u + j + 1
      const uint32_t sg = (uint32_t)(peek >> nsteps) & 1u;
      const uint32_t cnt = (uint32_t)__popc(mr);
      const uint64_t cw = peek >> (nsteps + 1u);   // (at least 31 valid bits, cnt <= 31)
      const uint32_t z = min(cw ? (uint32_t)__ffsll((long long)cw) - 1u : 64u, cnt);
      uint32_t rest = mr;
      for (uint32_t k = 0; k < z; k++)   // the z innermost parked halves are born insignificant
        rest &= rest - 1u;
      wrlane(rES, nrec, es);
      wrlane(rEL, nrec, el);
      wrlane(rR, nrec, R);
      wrlane(rMeta, nrec, u | ((u + nsteps) << 8) | (sg << 16));
      wrlane(rClosed, nrec, mr ^ rest);
      mr = rest;
      nrec++;
      if (nrec == 64u)
        flush_paths(lev);
      rpos += nsteps + 1u + z;
      if (z == cnt)
        return;
      rpos++;   // (a '1': z < cnt <= 31, so the bit was inside the window)
      u = 32u - (uint32_t)__ffs((int)mr);
      mr &= mr - 1u;
      R = (R & ((1u << (u - 1u)) - 1u)) | (1u << (u - 1u));
      if (!P2) {
        const uint32_t um = (1u << u) - 1u;
        lo = (el >> u) + (((R & um) < (el & um)) ? 1u : 0u);
      }
    }
  };

  // The chain for a run of 2^g values, g <= 30, as one asm statement per run (a lone wavefront completes an instruction
  // every nine cycles or so: the compiler's 83 per path were 630 cycles): 36 instructions per path.  State: the two
  // stream words the window lies in (`cw0`, `cw1s` = the second one shifted left by one, so that a shift by 63 - sh never
  // is one by 64) and the bit offset `sh` into the first, `wk` = that word's lane in vW; the parked depths `mr`
  // (bit-reversed, as above) and `u1` = the path's first depth + 1.  A record is three lane writes: the path's bits with
  // its sign on top, the first depth, the closed halves; the run's start and length are filled in for all of its records
  // afterwards, R and the meta word by flush_paths.  Clearing the z lowest set bits of `mr` is three vector instructions
  // (lane i: does bit i have z set bits below it).  The statement comes back when the run is through (0), the record
  // registers are full (1), the stream words in vW are used up (2).
  // peek: s[88:89]; scratch: s[90:95], v[T].
  auto chain_p2 = [&](uint32_t es, uint32_t el, uint32_t lev) {
    const uint32_t g = 31u - (uint32_t)__clz((int)el);
    const uint32_t l2 = rfl(g + 2u), dm = rfl(((1u << g) - 1u) << (31u - g));
    const uint32_t lowm = lane < 32u ? (1u << lane) - 1u : 0xffffffffu;
    uint32_t mr = 0, u1 = 1, n0 = nrec;
    for (;;) {
      // the window's state from rpos
      uint64_t wi = rpos >> 6;
      if (wi - cBase >= 62u) {
        cBase = wi;
        vW = wi + lane < b.streamStride ? words[wi + lane] : 0ull;
      }
      uint32_t wk = rfl((uint32_t)(wi - cBase)), sh = rfl((uint32_t)(rpos & 63u)), st, nr = rfl(nrec);
      const uint32_t a0 = rdlane((uint32_t)vW, wk), a1 = rdlane((uint32_t)(vW >> 32), wk);
      const uint32_t b0 = rdlane((uint32_t)vW, wk + 1u), b1 = rdlane((uint32_t)(vW >> 32), wk + 1u);
      uint64_t c0 = (uint64_t)rfl(a0) | ((uint64_t)rfl(a1) << 32);
      uint64_t c1s = (uint64_t)rfl(b0 << 1) | ((uint64_t)rfl((b1 << 1) | (b0 >> 31)) << 32);
      uint32_t tv_;
      mr = rfl(mr);
      u1 = rfl(u1);
      asm volatile(
          "s_mov_b32 m0, %[nr]\n\t"
          "s_branch 0f\n\t"
          ".p2align 7\n\t"
          "0:\n\t"                                   // ---- a path
          "s_lshr_b64 s[88:89], %[cw0], %[sh]\n\t"
          "s_xor_b32 s90, %[sh], 63\n\t"
          "s_lshl_b64 s[90:91], %[cw1s], s90\n\t"
          "s_sub_u32 s92, %[l2], %[u1]\n\t"          // its steps + 1 (the sign)
          "s_or_b64 s[88:89], s[88:89], s[90:91]\n\t"  // the next 64 bits of the stream
          "s_bfm_b32 s93, s92, 0\n\t"
          "s_lshr_b64 s[90:91], s[88:89], s92\n\t"   // the closing bits
          "s_and_b32 s93, s88, s93\n\t"              // the path's bits, the sign on top
          "s_add_u32 %[sh], %[sh], s92\n\t"
          "s_brev_b32 s94, s93\n\t"
          "v_writelane_b32 %[rb], s93, m0\n\t"
          "s_lshr_b32 s94, s94, %[u1]\n\t"           // a '1' at step j parks the right half of depth u + j + 1
          "v_writelane_b32 %[ru], %[u1], m0\n\t"
          "s_and_b32 s94, s94, %[dm]\n\t"            // (not the sign)
          "s_ff1_i32_b64 s95, s[90:91]\n\t"          // zeros at the head of the closing bits
          "s_or_b32 %[mr], %[mr], s94\n\t"
          "s_bcnt1_i32_b32 s94, %[mr]\n\t"
          "v_and_b32 %[tv], %[mr], %[lowm]\n\t"
          "s_min_u32 s95, s95, s94\n\t"              // z: that many parked halves are born insignificant, innermost first
          "v_bcnt_u32_b32 %[tv], %[tv], 0\n\t"
          "s_add_u32 %[sh], %[sh], s95\n\t"
          "v_cmp_le_u32 vcc, s95, %[tv]\n\t"         // (bits of mr with z set bits below them stay)
          "s_andn2_b32 s93, %[mr], vcc_lo\n\t"
          "s_and_b32 %[mr], %[mr], vcc_lo\n\t"
          "v_writelane_b32 %[rc], s93, m0\n\t"
          "s_add_u32 m0, m0, 1\n\t"
          "s_cmp_eq_u32 s95, s94\n\t"
          "s_cbranch_scc1 8f\n\t"                    // all of them: the run is through
          "s_ff1_i32_b32 s93, %[mr]\n\t"             // the next path starts in the innermost half left (its '1')
          "s_add_u32 %[sh], %[sh], 1\n\t"
          "s_sub_u32 %[u1], 32, s93\n\t"
          "s_bitset0_b32 %[mr], s93\n\t"
          "s_cmp_eq_u32 m0, 64\n\t"
          "s_cbranch_scc1 7f\n\t"
          "s_cmp_lt_u32 %[sh], 64\n\t"
          "s_cbranch_scc1 0b\n\t"
          "s_sub_u32 %[sh], %[sh], 64\n\t"           // ---- the next stream word
          "s_add_u32 %[wk], %[wk], 1\n\t"
          "s_cmp_ge_u32 %[wk], 62\n\t"
          "s_cbranch_scc1 6f\n\t"
          "s_add_u32 s92, %[wk], 1\n\t"
          "v_readlane_b32 s90, %[wlo], %[wk]\n\t"
          "v_readlane_b32 s91, %[whi], %[wk]\n\t"
          "v_readlane_b32 s94, %[wlo], s92\n\t"
          "v_readlane_b32 s95, %[whi], s92\n\t"
          "s_mov_b64 %[cw0], s[90:91]\n\t"
          "s_lshl_b64 %[cw1s], s[94:95], 1\n\t"
          "s_branch 0b\n\t"
          "6:\n\t"
          "s_mov_b32 %[st], 2\n\t"
          "s_branch 9f\n\t"
          "7:\n\t"
          "s_mov_b32 %[st], 1\n\t"
          "s_branch 9f\n\t"
          "8:\n\t"
          "s_mov_b32 %[st], 0\n\t"
          "9:\n\t"
          "s_mov_b32 %[nr], m0\n\t"
          : [sh] "+s"(sh), [wk] "+s"(wk), [mr] "+s"(mr), [u1] "+s"(u1), [nr] "+s"(nr), [cw0] "+s"(c0), [cw1s] "+s"(c1s), [rb] "+v"(rR),
            [ru] "+v"(rMeta), [rc] "+v"(rClosed), [st] "=&s"(st), [tv] "=&v"(tv_)
          : [l2] "s"(l2), [dm] "s"(dm), [lowm] "v"(lowm), [wlo] "v"((uint32_t)vW), [whi] "v"((uint32_t)(vW >> 32))
          : "scc", "vcc", "m0", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95");
      rpos = (cBase + wk) * 64ull + sh;
      cwi = 1ull << 62;   // (window()'s two words are not the statement's)
      if (lane >= n0 && lane < nr) {   // the records just written belong to this run
        rES = es;
        rEL = el;
      }
      recRaw |= low_mask(nr) & ~low_mask(n0);
      nrec = nr;
      if (nrec == 64u) {   // (st 1, or the run's last record was the 64th)
        flush_paths(lev);
        n0 = 0;
      }
      if (st == 0u)
        return;
    }
  };

  // (runs of a single value on a list -- arrays of fewer than four values -- keep the serial walk: a
  //  descent with a stack of the right halves that still wait for their test bit)
  auto expand_serial = [&](uint32_t ns, uint32_t nl, uint32_t nlev) {
    uint32_t sp = 0;
    while (true) {
      const uint32_t h0 = nl - nl / 2, r0 = nl / 2;
      bool atPixel = false;
      if (get()) {
        wrlane(vS, sp, ns + h0);
        wrlane(vL, sp, r0);
        wrlane(vT, sp, nlev + 1);
        sp++;
        if (h0 == 1) {
          found_pixel(ns, curPlane);
          atPixel = true;
        }
        else {
          nl = h0;
          nlev++;
        }
      }
      else {
        if (h0 == 1)
          lip_set(ns);
        else
          list_push(nlev + 1, ns, h0);
        if (r0 == 1) {
          found_pixel(ns + h0, curPlane);
          atPixel = true;
        }
        else {
          ns += h0;
          nl = r0;
          nlev++;
        }
      }
      if (!atPixel)
        continue;
      // back up: the parked right halves, innermost first
      bool down = false;
      while (sp > 0) {
        sp--;
        const uint32_t rs = rdlane(vS, sp), rl = rdlane(vL, sp), rlev = rdlane(vT, sp);
        if (get()) {
          if (rl == 1) {
            found_pixel(rs, curPlane);
            continue;
          }
          ns = rs;
          nl = rl;
          nlev = rlev;
          down = true;
          break;
        }
        if (rl == 1)
          lip_set(rs);
        else
          list_push(rlev, rs, rl);
      }
      if (!down)
        break;
    }
  };

  // src/SPECK1D_INT.cpp:19-34 : the two halves of the array start on the list of level 1
  list_push(1, 0, N - N / 2);
  list_push(1, N - N / 2, N / 2);

  for (int p = nbp - 1; p >= 0; p--) {
    curPlane = p;
    if (ENC) {
      // ---- this plane's threshold structures: EQ = outliers whose msb is p, GE = at or above
      for (uint32_t w = lane; w < nw; w += 64)
        maskEQ[w] = 0ull;
      __threadfence_block();
      for (uint32_t kb = 0; kb < K; kb += 64) {
        const uint32_t k = kb + lane;
        if (k < K && (int)omsb[k] == p) {
          const uint32_t x = opos[k];
          atomicOr(maskEQ + (x >> 6), 1ull << (x & 63u));
          atomicOr(maskGE + (x >> 6), 1ull << (x & 63u));
        }
      }
      __threadfence_block();
      uint32_t run = 0;
      for (uint32_t wb = 0; wb < nw; wb += 512) {   // eight words per lane and round: the loads overlap
        const uint32_t w0 = wb + lane * 8u;
        uint32_t cw[8], sum = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          cw[j] = w0 + j < nw ? (uint32_t)__popcll(maskGE[w0 + j]) : 0u;
          sum += cw[j];
        }
        const uint32_t inc = wave_scan_dpp(sum);
        uint32_t at = run + inc - sum;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          if (w0 + j < nw)
            cpos[w0 + j] = at;
          at += cw[j];
        }
        run += rdlane(inc, 63);
      }
      if (lane == 0)
        cpos[nw] = run;
      run = 0;
      for (uint32_t kb = 0; kb < K; kb += 64) {
        const uint32_t k = kb + lane;
        const bool in = k < K && (int)omsb[k] >= p;
        const uint64_t bm = __ballot(in);
        if (in) {
          const uint32_t j = run + (uint32_t)__popcll(bm & low_mask(lane));
          posGE[j] = opos[k];
          sgnGE[j] = osgn[k];
        }
        run += (uint32_t)__popcll(bm);
      }
    }
    __threadfence_block();
    // ================= LIP pass (src/SPECK1D_INT_ENC.cpp:15-45, _DEC.cpp:15-45) =================
    STAMP_BEGIN();
    for (uint32_t wb8 = 0; wb8 < nw; wb8 += 512) {   // (eight blocks of 64 words are loaded at once)
      uint64_t lwv[8], any = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t wj = wb8 + (uint32_t)j * 64u + lane;
        lwv[j] = wj < nw ? lip[wj] : 0ull;
        any |= lwv[j];
      }
      if (__ballot(any != 0) == 0)
        continue;
#pragma unroll
      for (int jb = 0; jb < 8; jb++) {
      const uint32_t wb = wb8 + (uint32_t)jb * 64u;
      const uint32_t w = wb + lane;
      const uint64_t lw = lwv[jb];
      uint64_t nz = __ballot(lw != 0);
      if (nz == 0)
        continue;
      if (ENC) {
        // every lane codes the pixels of its own word: test bit, then the sign of a significant one
        uint64_t plo = 0, phi = 0;
        uint32_t len = 0;
        if (lw) {
          const uint64_t sig = lw & maskEQ[w];
          const uint64_t sgn = signMask[w];
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
        const uint32_t inc = wave_scan_dpp(len);
        const uint32_t total = rdlane(inc, 63);
        if (plo | phi) {
          const uint64_t at = wpos + (inc - len);
          const uint32_t sh = (uint32_t)(at & 63);
          unsigned long long* dst = words + (at >> 6);
          // 128 pattern bits shifted by sh span up to three words
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
              const uint32_t sg = get();
              if (nfound < b.kStride && lane == 0) {
                fpos[nfound] = (wb + l) * 64u + j;
                fmeta[nfound] = (uint8_t)((uint32_t)p | (sg << 7));
              }
              nfound++;
              keep &= ~(1ull << j);
            }
          }
          if (keep != wv && lane == 0)
            lip[wb + l] = keep;
        }
      }
      }
    }

    STAMP_END(tkLip);
    // ================= LIS pass, smallest sets first (ENC.cpp:47-56, DEC.cpp:47-54) =============
    STAMP_BEGIN();
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
        if (!ENC) {
          // entries that stay insignificant are counted off the stream window; a significant one
          // hands its paths to the chain; the block's kept entries are compacted at its end
          uint64_t sigm = 0;
          uint32_t i = 0;
          while (i < blockN) {
            const uint64_t win = window();
            const uint32_t z = win ? (uint32_t)__ffsll((long long)win) - 1u : 64u;
            if (z >= blockN - i) {
              rpos += blockN - i;
              break;
            }
            i += z;
            rpos += z + 1u;   // entry i is significant: its '1', then the recursion (m_code_S)
            sigm |= 1ull << i;
            const uint32_t es = rdlane((uint32_t)myRun, i), el = rdlane((uint32_t)(myRun >> 32), i);
            if (el >= 2) {
#ifdef SPERR_1D_STAMPS
              const long long tc0 = clock64();
              nEntries++;
#endif
              if ((el & (el - 1u)) == 0u && el <= (1u << 30))
                chain_p2(es, el, lev);
              else if ((el & (el - 1u)) == 0u)
                expand_chain(std::true_type{}, es, el, lev);
              else
                expand_chain(std::false_type{}, es, el, lev);
#ifdef SPERR_1D_STAMPS
              tkChain += clock64() - tc0;
#endif
            }
            else {
              flush_paths(lev);
              expand_serial(es, el, lev);
            }
            i++;
          }
          const uint64_t keepM = ~sigm & low_mask(blockN);
          if ((keepM >> lane) & 1ull) {
            const uint32_t dst = wr + (uint32_t)__popcll(keepM & low_mask(lane));
            if (dst != rd + lane)
              runs[base + dst] = myRun;
          }
          wr += (uint32_t)__popcll(keepM);
          continue;
        }
        uint32_t myA = 0, myB = 0;
        if (ENC && valid) {   // every lane tests its own entry: outliers at or above the threshold inside
          const uint32_t s0 = (uint32_t)myRun, e0 = s0 + (uint32_t)(myRun >> 32);
          myA = cpos[s0 >> 6] + (uint32_t)__popcll(maskGE[s0 >> 6] & low_mask(s0 & 63u));
          myB = e0 >= N ? cpos[nw] : cpos[e0 >> 6] + (uint32_t)__popcll(maskGE[e0 >> 6] & low_mask(e0 & 63u));
        }
        const uint64_t sigmask = ENC ? __ballot(valid && myB > myA) : 0ull;
        if (ENC && sigmask && __ballot(valid && myB > myA && (uint32_t)(myRun >> 32) < 2u) == 0) {
          // (runs of two and more values: all the block's significant entries at once)
          expand_block(myRun, myA, myB, sigmask, blockN, lev);
          const uint64_t keepM = ~sigmask & low_mask(blockN);
          if ((keepM >> lane) & 1ull) {
            const uint32_t dst = wr + (uint32_t)__popcll(keepM & low_mask(lane));
            if (dst != rd + lane)
              runs[base + dst] = myRun;
          }
          wr += (uint32_t)__popcll(keepM);
          continue;
        }
        uint32_t i = 0;
        while (i < blockN) {
          // entries that stay insignificant, up to the next significant one
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
            if (lane >= i && lane < i + z && wr + (lane - i) != rd + lane)
              runs[base + wr + (lane - i)] = myRun;
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
            continue;   // the run of zeros was cut by the 64-bit window: keep counting
          // entry i is significant: its '1', then the recursion (m_code_S)
          if (ENC)
            put(1);
          else {
            rpos++;
            {
              const uint32_t es = rdlane((uint32_t)myRun, i), el = rdlane((uint32_t)(myRun >> 32), i);
              expand_serial(es, el, lev);   // (unreachable: the decoder's block loop is above)
            }
            i++;
            continue;
          }
          uint32_t sp = 1;
          wrlane(vS, 0, rdlane((uint32_t)myRun, i));
          wrlane(vL, 0, rdlane((uint32_t)(myRun >> 32), i));
          wrlane(vT, 0, lev << 16);
          if (ENC) {
            const uint32_t a0 = rdlane(myA, i), e0 = rdlane(myB, i);
            wrlane(vA, 0, a0);
            wrlane(vB, 0, e0);
            if (rdlane((uint32_t)(myRun >> 32), i) >= 2) {
              expand_enc(rdlane((uint32_t)myRun, i), rdlane((uint32_t)(myRun >> 32), i), lev, a0, e0);
              sp = 0;   // (the whole expansion is done; runs of one value keep the serial walk below)
            }
          }
          i++;
          while (sp > 0) {
            const uint32_t f = sp - 1;
            const uint32_t state = rdlane(vT, f);
            const uint32_t k = state & 0xffu, found = (state >> 8) & 0xffu, flev = state >> 16;
            if (k == 2) {
              sp--;
              continue;
            }
            const uint32_t ps = rdlane(vS, f), pl = rdlane(vL, f);
            const uint32_t l0 = pl - pl / 2;
            uint32_t clo = 0, chi = 0;
            if (ENC) {
              const uint32_t fa = rdlane(vA, f), fb = rdlane(vB, f);
              uint32_t fm;
              if (k == 0) {
                fm = rank_in(ps + l0, fa, fb);
                wrlane(vM, f, fm);
              }
              else
                fm = rdlane(vM, f);
              clo = k == 0 ? fa : fm;
              chi = k == 0 ? fm : fb;
            }
            const uint32_t cs = k == 0 ? ps : ps + l0;
            const uint32_t cl = k == 0 ? l0 : pl / 2;
            const bool coded = k == 0 || found != 0;
            uint32_t sig = 1;
            uint32_t nstate = (k + 1) | (found << 8) | (flev << 16);
            if (cl == 1) {   // a pixel (m_process_P)
              if (ENC) {
                if (coded) {
                  sig = chi > clo;
                  put(sig);
                }
                if (sig)
                  put(rfl((uint32_t)sgnGE[clo]));
              }
              else {
                if (coded)
                  sig = get();
                if (sig) {
                  const uint32_t sg = get();
                  if (nfound < b.kStride && lane == 0) {
                    fpos[nfound] = cs;
                    fmeta[nfound] = (uint8_t)((uint32_t)p | (sg << 7));
                  }
                  nfound++;
                }
              }
              if (sig)
                nstate |= 1u << 8;
              else
                lip_set(cs);
              wrlane(vT, f, nstate);
            }
            else {           // a set (m_process_S)
              if (ENC) {
                if (coded) {
                  sig = chi > clo;
                  put(sig);
                }
              }
              else if (coded)
                sig = get();
              if (sig) {
                nstate |= 1u << 8;
                wrlane(vT, f, nstate);
                wrlane(vS, sp, cs);
                wrlane(vL, sp, cl);
                wrlane(vT, sp, (flev + 1) << 16);
                if (ENC) {
                  wrlane(vA, sp, clo);
                  wrlane(vB, sp, chi);
                }
                sp++;
              }
              else {
                wrlane(vT, f, nstate);
                list_push(flev + 1, cs, cl);
              }
            }
          }
        }
      }
      if (!ENC)
        flush_paths(lev);
      wrlane(vCnt, lev, wr);
    }

    STAMP_END(tkLis);
    // ================= refinement pass (SPECK_INT.cpp:310-357 / 359-469) ========================
    STAMP_BEGIN();
    if (ENC) {
      flush_acc();
      for (uint32_t kb = 0; kb < K; kb += 64) {
        const uint32_t k = kb + lane;
        const bool in = k < K && (int)omsb[k] > p;
        const uint32_t bit = in ? (uint32_t)((omag[k] >> p) & 1ull) : 0u;
        const uint64_t lm = __ballot(in);
        if (bit) {
          const uint64_t at = wpos + (uint32_t)__popcll(lm & low_mask(lane));
          atomicOr(words + (at >> 6), 1ull << (at & 63));
        }
        wpos += (uint32_t)__popcll(lm);
      }
    }
    else {
      __threadfence_block();
      uint64_t* pb = planeBits + (size_t)p * b.wordStride;
      for (uint32_t wb8 = 0; wb8 < nw; wb8 += 512) {
        uint64_t swv[8], any = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t wj = wb8 + (uint32_t)j * 64u + lane;
          swv[j] = wj < nw ? lsp[wj] : 0ull;
          any |= swv[j];
        }
        if (__ballot(any != 0) == 0)
          continue;
        // the stream words of all eight blocks are asked for before the first deposit (block after block,
        // each waited for its own loads: 1000 rounds of eight memory latencies per plane)
        uint64_t av[8], dv[8];
        uint32_t shv[8], cntv[8];
#pragma unroll
        for (int jb = 0; jb < 8; jb++) {
          const uint32_t cnt = (uint32_t)__popcll(swv[jb]);
          const uint32_t inc = wave_scan_dpp(cnt);
          const uint64_t at = rpos + (inc - cnt);
          cntv[jb] = cnt;
          shv[jb] = (uint32_t)(at & 63);
          av[jb] = cnt ? words[at >> 6] : 0ull;
          dv[jb] = cnt ? words[(at >> 6) + 1] : 0ull;
          rpos += rdlane(inc, 63);
        }
#pragma unroll
        for (int jb = 0; jb < 8; jb++) {
          if (cntv[jb]) {
            const uint32_t sh = shv[jb];
            uint64_t bits = sh ? (av[jb] >> sh) | (dv[jb] << (64 - sh)) : av[jb];
            uint64_t res = 0, m = swv[jb];   // deposit the next cnt bits under the mask
            while (m) {
              const uint32_t j = (uint32_t)__ffsll((long long)m) - 1u;
              m &= m - 1;
              res |= (bits & 1ull) << j;
              bits >>= 1;
            }
            pb[wb8 + (uint32_t)jb * 64u + lane] = res;
          }
        }
      }
      STAMP_END(tkRef);
      STAMP_BEGIN();
      // the values found in this plane join the LSP (SPECK_INT.cpp:462-468)
      __threadfence_block();
      const uint32_t lim = min(nfound, (uint32_t)b.kStride);
      for (uint32_t k = lspDone + lane; k < lim; k += 64) {
        const uint32_t x = fpos[k];
        atomicOr(reinterpret_cast<unsigned long long*>(lsp) + (x >> 6), 1ull << (x & 63u));
      }
      lspDone = lim;
      STAMP_END(tkLsp);
    }
  }
#ifdef SPERR_1D_STAMPS
  if (!ENC && lane == 0 && c == 0)
    printf("1D dec chunk0: lip %lld lis %lld (chain %lld incl flush %lld) ref %lld lsp %lld ticks; paths %u entries %u flushes %u found %u bits %llu\n",
           tkLip, tkLis, tkChain, tkFlush, tkRef, tkLsp, nPaths, nEntries, nFlush, nfound, (unsigned long long)rpos);
#endif

  if (__any(vErr != 0))
    err = 2;
  if (ENC) {
    flush_acc();
    if (lane == 0) {
      oc.nbp = nbp;
      oc.total_bits = wpos;
      if (err)
        oc.error = err;
    }
  }
  else {
    if (nfound > b.kStride || err) {
      if (lane == 0)
        oc.error = err ? err : 3;
      return;
    }
    if (lane == 0)
      oc.found = nfound;
  }
}

// The correctors of the values the 1D decoder found (src/Outlier_Coder.cpp:199-233): 1.1 tol for
// magnitude 1, (m - 0.25) tol above.  A kernel of its own: the decoder needs nothing but the outlier
// stream, so it runs beside the chunk's SPECK3D decoder and the inverse transform on another stream.
__global__ void __launch_bounds__(kThreads)
k_outlier_apply(OutlierBufs b, const CoderState* cst, double* vals, size_t valsStride)
{
  const uint32_t c = blockIdx.y;
  const OutlierChunk& oc = b.oc[c];
  if (!oc.has || oc.nbp == 0 || oc.error)
    return;
  const uint32_t nfound = oc.found;
  const uint32_t* fpos = b.pos + c * b.kStride;
  const uint8_t* fmeta = b.sgn + c * b.kStride;
  const uint64_t* planeBits = b.planeBits + c * b.planeStride;
  const double tol = cst[c].q / 1.5;
  double* out = vals + c * valsStride;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nfound; k += gridDim.x * blockDim.x) {
    const uint32_t x = fpos[k], meta = fmeta[k];
    const int pl = (int)(meta & 0x7f);
    unsigned long long m = 1ull << pl;
    for (int q = 0; q < pl; q++)
      m |= ((planeBits[(size_t)q * b.wordStride + (x >> 6)] >> (x & 63u)) & 1ull) << q;
    double e = m == 1 ? 1.1 : (double)m - 0.25;
    e *= tol * ((meta >> 7) ? 1.0 : -1.0);
    out[x] += e;
  }
}

// {u8 planes, u64 total_bits, payload bytes}
__global__ void __launch_bounds__(kThreads)
k_outlier_stream_out(OutlierBufs b, const uint32_t* gids, uint8_t* slots, const uint64_t* slotOff,
                     uint64_t* lens2)
{
  const uint32_t c = blockIdx.y;
  const OutlierChunk& oc = b.oc[c];
  const uint32_t g = gids[c];
  if (oc.flagged == 0) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
      lens2[g] = 0;
    return;
  }
  const uint64_t payload = (oc.total_bits + 7) / 8;
  uint8_t* out = slots + slotOff[c];   // (slot offsets are per chunk of the batch)
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    lens2[g] = 9 + payload;
    out[0] = (uint8_t)oc.nbp;
    memcpy(out + 1, &oc.total_bits, 8);
  }
  const uint64_t* w = b.stream + c * b.streamStride;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < payload;
       i += (uint64_t)gridDim.x * blockDim.x)
    out[9 + i] = (uint8_t)(w[i >> 3] >> (8 * (i & 7)));
}

// payload bytes of the container -> aligned, zero-padded words
__global__ void __launch_bounds__(kThreads)
k_outlier_stream_in(OutlierBufs b, const uint8_t* container)
{
  const uint32_t c = blockIdx.y;
  const OutlierChunk& oc = b.oc[c];
  if (!oc.has)
    return;
  const uint64_t nbytes = (oc.total_bits + 7) / 8;
  const uint8_t* in = container + oc.streamOff + 9;
  uint64_t* w = b.stream + c * b.streamStride;
  const uint64_t nwords = nbytes / 8 + 2;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords && i < b.streamStride;
       i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t v = 0;
    for (int k = 0; k < 8; k++) {
      const uint64_t bi = i * 8 + k;
      if (bi < nbytes)
        v |= (uint64_t)in[bi] << (8 * k);
    }
    w[i] = v;
  }
}

}  // namespace

template <typename T>
int launch_outlier_scan(hipStream_t st, int pass, const T* vol, VolDesc vd, const ChunkGeom* geom,
                        const uint32_t cdims[3], const double* vals, size_t valsStride,
                        const CoderState* cst, double tol, const OutlierBufs& b)
{
  const uint32_t blocks = capped_blocks((b.nw + 3) / 4, b.nchunks, kGridCapWide);
  const dim3 grid(blocks, b.nchunks);
  if (pass == 0)
    LAUNCH_K((k_outlier_scan<T, 0>), grid, dim3(kThreads), 0, st, vol, vd, geom, cdims[0], cdims[1],
             vals, valsStride, cst, tol, b);
  else if (pass == 1)
    LAUNCH_K((k_outlier_scan<T, 1>), grid, dim3(kThreads), 0, st, vol, vd, geom, cdims[0], cdims[1],
             vals, valsStride, cst, tol, b);
  else
    LAUNCH_K((k_outlier_scan<T, 2>), grid, dim3(kThreads), 0, st, vol, vd, geom, cdims[0], cdims[1],
             vals, valsStride, cst, tol, b);
  HIP_CHECK(hipGetLastError());
  return 0;
}
template int launch_outlier_scan<float>(hipStream_t, int, const float*, VolDesc, const ChunkGeom*,
                                        const uint32_t[3], const double*, size_t,
                                        const CoderState*, double, const OutlierBufs&);
template int launch_outlier_scan<double>(hipStream_t, int, const double*, VolDesc,
                                         const ChunkGeom*, const uint32_t[3], const double*, size_t,
                                         const CoderState*, double, const OutlierBufs&);

int launch_outlier_prefix(hipStream_t st, const OutlierBufs& b)
{
  LAUNCH_K(k_outlier_prefix, dim3(b.nchunks), dim3(1024), 0, st, b);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_speck1d_encode(hipStream_t st, const OutlierBufs& b)
{
  LAUNCH_K(k_speck1d<true>, dim3(b.nchunks), dim3(64), 0, st, b);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_outlier_stream_out(hipStream_t st, const OutlierBufs& b, const uint32_t* gids,
                              uint8_t* slots, const uint64_t* slotOff, uint64_t* lens2)
{
  const uint32_t blocks =
      capped_blocks((uint32_t)((b.streamStride * 8 + kThreads - 1) / kThreads), b.nchunks);
  LAUNCH_K(k_outlier_stream_out, dim3(blocks, b.nchunks), dim3(kThreads), 0, st, b, gids, slots,
           slotOff, lens2);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_speck1d_decode(hipStream_t st, const OutlierBufs& b, const uint8_t* container)
{
  const uint32_t blocks =
      capped_blocks((uint32_t)((b.streamStride + kThreads - 1) / kThreads), b.nchunks);
  LAUNCH_K(k_outlier_stream_in, dim3(blocks, b.nchunks), dim3(kThreads), 0, st, b, container);
  LAUNCH_K(k_speck1d<false>, dim3(b.nchunks), dim3(64), 0, st, b);
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_outlier_apply(hipStream_t st, const OutlierBufs& b, const CoderState* cst, double* vals,
                         size_t valsStride)
{
  const uint32_t blocks = capped_blocks((uint32_t)((b.kStride + kThreads - 1) / kThreads), b.nchunks);
  LAUNCH_K(k_outlier_apply, dim3(std::max(1u, std::min(blocks, 256u)), b.nchunks), dim3(kThreads), 0, st, b, cst,
           vals, valsStride);
  HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace sperrhip
