// outlier.hip -- point-wise error mode (CompMode::PWE): the values whose reconstruction error
// exceeds the tolerance are found on the device, quantised in units of the tolerance
// (src/Outlier_Coder.cpp:179-197) and coded by the reference's 1D set-partitioning coder
// (src/SPECK1D_INT.cpp, src/SPECK1D_INT_ENC.cpp, src/SPECK1D_INT_DEC.cpp) over the length-N array
// that is zero everywhere else.
//
// The array is sparse, so the coder never touches N values:
//   * encoder: the outliers are kept as sorted (position, magnitude, sign) triples; a run of the
//     array [start, start + len) maps to a range of outlier indices through a position bitmask with
//     a popcount prefix, and the msb of its largest magnitude is one lookup in a range-maximum
//     table.  A run is tested against the current threshold by comparing that msb with the plane.
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
    if (i < b.N) {
      const uint32_t x = i % cx, r = i / cx;
      const uint32_t y = r % cy, z = r / cy;
      // the conditioned input, as the first lifting pass computed it (Conditioner.cpp:46-50)
      const double orig =
          (double)vol[((size_t)(g.org[2] + z) * vy + (g.org[1] + y)) * vx + g.org[0] + x] - mean;
      diff = orig - in[i];
      flag = fabs(diff) > tol;
    }
    if (PASS == 0) {
      cnt += (uint32_t)__popcll(__ballot(flag));
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
        if (lane == 0) {
          b.outMask[c * b.wordStride + w] = word;
          b.outPre[c * b.wordStride + w] = (uint32_t)__popcll(word);
        }
      }
      else if (f2) {
        const uint32_t k = b.outPre[c * b.wordStride + w] + (uint32_t)__popcll(word & low_mask(lane));
        if (k < b.kStride) {
          b.pos[c * b.kStride + k] = i;
          b.mag[c * b.kStride + k] = m;
          b.sgn[c * b.kStride + k] = ll >= 0 ? 1 : 0;
          b.tbl[c * b.kStride * b.tblLevels + k] = (int8_t)(63 - __clzll((long long)m));
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

// tbl[j][i] = max msb over outliers [i, i + 2^j)
__global__ void __launch_bounds__(kThreads) k_outlier_rmq(OutlierBufs b, uint32_t j)
{
  const uint32_t c = blockIdx.y;
  const uint32_t K = b.oc[c].count;
  const uint32_t span = 1u << j;
  if (K < span)
    return;
  int8_t* t = b.tbl + c * b.kStride * b.tblLevels;
  const int8_t* prev = t + (size_t)(j - 1) * b.kStride;
  int8_t* cur = t + (size_t)j * b.kStride;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i + span <= K;
       i += gridDim.x * blockDim.x) {
    const int8_t a = prev[i], d = prev[i + span / 2];
    cur[i] = a > d ? a : d;
  }
}

// ------------------------------------------------------------------------------------------
// SPECK1D, one wavefront per chunk
// ------------------------------------------------------------------------------------------
template <bool ENC>
__global__ void __launch_bounds__(64)
k_speck1d(OutlierBufs b, const CoderState* cst, double* vals, size_t valsStride)
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
  const uint32_t K = ENC ? oc.count : 0u;
  const int nbp = ENC ? 64 - __clzll((long long)oc.maxMag) : oc.nbp;

  uint64_t* lip = b.lip + c * b.wordStride;
  uint64_t* runs = b.runs + c * b.runStride;
  uint64_t* rng = ENC ? b.rng + c * b.runStride : nullptr;
  int8_t* sval = ENC ? b.sval + c * b.runStride : nullptr;
  unsigned long long* words = reinterpret_cast<unsigned long long*>(b.stream + c * b.streamStride);
  const uint64_t* outMask = ENC ? b.outMask + c * b.wordStride : nullptr;
  const uint32_t* outPre = ENC ? b.outPre + c * b.wordStride : nullptr;
  const uint32_t* opos = b.pos + c * b.kStride;
  (void)opos;
  const uint64_t* omag = ENC ? b.mag + c * b.kStride : nullptr;
  const uint8_t* osgn = b.sgn + c * b.kStride;
  const int8_t* tbl = ENC ? b.tbl + c * b.kStride * b.tblLevels : nullptr;
  uint64_t* lsp = ENC ? nullptr : b.lsp + c * b.wordStride;
  uint64_t* planeBits = ENC ? nullptr : b.planeBits + c * b.planeStride;
  uint32_t* fpos = b.pos + c * b.kStride;      // decoder: values found
  uint8_t* fmeta = b.sgn + c * b.kStride;

  __shared__ uint32_t sh_n[kO1MaxLevels + 1];
  __shared__ uint32_t st_start[kO1MaxLevels + 2], st_len[kO1MaxLevels + 2];
  __shared__ uint32_t st_lo[kO1MaxLevels + 2], st_mid[kO1MaxLevels + 2], st_hi[kO1MaxLevels + 2];
  __shared__ uint32_t st_state[kO1MaxLevels + 2];   // next child | found << 8 | level << 16
  for (uint32_t i = lane; i <= (uint32_t)kO1MaxLevels; i += 64)
    sh_n[i] = 0;

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
  // ---- bit reader (decoder)
  uint64_t rpos = 0, w0 = 0, w1 = 0, cw = 1ull << 62;   // (cw: index of the word held in w0; none yet)
  auto window = [&]() -> uint64_t {
    const uint64_t wi = rpos >> 6;
    if (wi != cw) {
      w0 = (wi == cw + 1) ? w1 : words[wi];
      w1 = words[wi + 1];
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

  // ---- encoder geometry: outliers before position x; msb of the largest magnitude in [lo, hi)
  auto rank_of = [&](uint32_t x) -> uint32_t {
    if (x >= N)
      return K;
    const uint32_t w = x >> 6;
    return outPre[w] + (uint32_t)__popcll(outMask[w] & low_mask(x & 63u));
  };
  auto range_msb = [&](uint32_t lo, uint32_t hi) -> int {
    if (hi <= lo)
      return -1;
    const uint32_t j = 31u - (uint32_t)__clz((int)(hi - lo));
    const int8_t* t = tbl + (size_t)j * b.kStride;
    const int a = t[lo], d = t[hi - (1u << j)];
    return a > d ? a : d;
  };

  uint32_t nfound = 0, lspDone = 0;   // decoder: values found so far / of them, already in the LSP mask
  auto lip_set = [&](uint32_t x) {
    if (lane == 0)
      atomicOr(reinterpret_cast<unsigned long long*>(lip) + (x >> 6), 1ull << (x & 63u));
  };
  auto list_push = [&](uint32_t lev, uint32_t start, uint32_t len, uint32_t lo, uint32_t hi, int s) {
    const uint32_t idx = sh_n[lev];
    const uint32_t slot = b.levelOff[lev] + idx;
    if (slot < b.levelOff[lev + 1]) {
      if (lane == 0) {
        runs[slot] = (uint64_t)start | ((uint64_t)len << 32);
        if (ENC) {
          rng[slot] = (uint64_t)lo | ((uint64_t)hi << 32);
          sval[slot] = (int8_t)s;
        }
      }
      sh_n[lev] = idx + 1;
    }
    else
      oc.error = 2;   // list storage exhausted (cannot happen with the host's bounds)
  };

  // src/SPECK1D_INT.cpp:19-34 : the two halves of the array start on the list of level 1
  {
    const uint32_t l0 = N - N / 2;
    const uint32_t m = ENC ? rank_of(l0) : 0u;
    list_push(1, 0, l0, 0, m, ENC ? range_msb(0, m) : 0);
    list_push(1, l0, N / 2, m, K, ENC ? range_msb(m, K) : 0);
  }

  for (int p = nbp - 1; p >= 0; p--) {
    __threadfence_block();
    // ================= LIP pass (src/SPECK1D_INT_ENC.cpp:15-45, _DEC.cpp:15-45) =================
    for (uint32_t wb = 0; wb < nw; wb += 64) {
      const uint32_t w = wb + lane;
      const uint64_t lw = w < nw ? lip[w] : 0ull;
      uint64_t nz = __ballot(lw != 0);
      if (nz == 0)
        continue;
      if (ENC) {
        // every lane codes the pixels of its own word: test bit, then the sign of a significant one
        uint64_t plo = 0, phi = 0, keep = lw;
        uint32_t len = 0;
        if (lw) {
          const uint64_t om = outMask[w];
          const uint32_t pre = outPre[w];
          uint64_t m = lw;
          while (m) {
            const uint32_t j = (uint32_t)__ffsll((long long)m) - 1u;
            m &= m - 1;
            uint32_t sig = 0, sg = 0;
            if ((om >> j) & 1ull) {
              const uint32_t k = pre + (uint32_t)__popcll(om & low_mask(j));
              sig = tbl[k] == p;
              sg = osgn[k];
            }
            if (sig) {
              if (len < 64)
                plo |= 1ull << len;
              else
                phi |= 1ull << (len - 64);
              len++;
              if (sg) {
                if (len < 64)
                  plo |= 1ull << len;
                else
                  phi |= 1ull << (len - 64);
              }
              keep &= ~(1ull << j);
            }
            len++;
          }
          if (keep != lw)
            lip[w] = keep;
        }
        flush_acc();
        const uint32_t inc = wave_inclusive_scan<uint32_t>(len);
        const uint32_t total = __shfl(inc, 63, 64);
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
          const uint64_t wv = __shfl(lw, (int)l, 64);
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

    // ================= LIS pass, smallest sets first (ENC.cpp:47-56, DEC.cpp:47-54) =============
    for (uint32_t lev = b.nlists; lev-- > 0;) {
      const uint32_t n = sh_n[lev];
      if (n == 0)
        continue;
      __threadfence_block();
      const uint32_t base = b.levelOff[lev];
      uint32_t wr = 0;
      for (uint32_t rd = 0; rd < n; rd += 64) {
        const uint32_t blockN = min(64u, n - rd);
        const bool valid = lane < blockN;
        const uint64_t myRun = valid ? runs[base + rd + lane] : 0ull;
        const uint64_t myRng = (ENC && valid) ? rng[base + rd + lane] : 0ull;
        const int myS = (ENC && valid) ? (int)sval[base + rd + lane] : -1;
        const uint64_t sigmask = ENC ? __ballot(valid && myS == p) : 0ull;
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
            if (lane >= i && lane < i + z && wr + (lane - i) != rd + lane) {
              runs[base + wr + (lane - i)] = myRun;
              if (ENC) {
                rng[base + wr + (lane - i)] = myRng;
                sval[base + wr + (lane - i)] = (int8_t)myS;
              }
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
            continue;   // the run of zeros was cut by the 64-bit window: keep counting
          // entry i is significant: its '1', then the recursion (m_code_S)
          if (ENC)
            put(1);
          else
            rpos++;
          const uint64_t er = __shfl(myRun, (int)i, 64);
          const uint64_t eg = __shfl(myRng, (int)i, 64);
          i++;
          uint32_t sp = 1;
          st_start[0] = (uint32_t)er;
          st_len[0] = (uint32_t)(er >> 32);
          st_lo[0] = (uint32_t)eg;
          st_hi[0] = (uint32_t)(eg >> 32);
          st_state[0] = lev << 16;
          while (sp > 0) {
            const uint32_t f = sp - 1;
            const uint32_t state = st_state[f];
            const uint32_t k = state & 0xffu, found = (state >> 8) & 0xffu, flev = state >> 16;
            if (k == 2) {
              sp--;
              continue;
            }
            const uint32_t ps = st_start[f], pl = st_len[f];
            const uint32_t l0 = pl - pl / 2;
            if (ENC && k == 0)
              st_mid[f] = rank_of(ps + l0);
            const uint32_t cs = k == 0 ? ps : ps + l0;
            const uint32_t cl = k == 0 ? l0 : pl / 2;
            const uint32_t clo = ENC ? (k == 0 ? st_lo[f] : st_mid[f]) : 0u;
            const uint32_t chi = ENC ? (k == 0 ? st_mid[f] : st_hi[f]) : 0u;
            const bool coded = k == 0 || found != 0;
            uint32_t sig = 1;
            uint32_t nstate = (k + 1) | (found << 8) | (flev << 16);
            if (cl == 1) {   // a pixel (m_process_P)
              if (ENC) {
                const int s = chi > clo ? (int)tbl[clo] : -1;
                if (coded) {
                  sig = s == p;
                  put(sig);
                }
                if (sig)
                  put(osgn[clo]);
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
              st_state[f] = nstate;
            }
            else {           // a set (m_process_S)
              int s = 0;
              if (ENC) {
                s = range_msb(clo, chi);
                if (coded) {
                  sig = s == p;
                  put(sig);
                }
              }
              else if (coded)
                sig = get();
              if (sig) {
                nstate |= 1u << 8;
                st_state[f] = nstate;
                st_start[sp] = cs;
                st_len[sp] = cl;
                st_lo[sp] = clo;
                st_hi[sp] = chi;
                st_state[sp] = (flev + 1) << 16;
                sp++;
              }
              else {
                st_state[f] = nstate;
                list_push(flev + 1, cs, cl, clo, chi, s);
              }
            }
          }
        }
      }
      sh_n[lev] = wr;
    }

    // ================= refinement pass (SPECK_INT.cpp:310-357 / 359-469) ========================
    if (ENC) {
      flush_acc();
      for (uint32_t kb = 0; kb < K; kb += 64) {
        const uint32_t k = kb + lane;
        const bool in = k < K && (int)tbl[k] > p;
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
      for (uint32_t wb = 0; wb < nw; wb += 64) {
        const uint32_t w = wb + lane;
        const uint64_t sw = w < nw ? lsp[w] : 0ull;
        if (__ballot(sw != 0) == 0)
          continue;
        const uint32_t cnt = (uint32_t)__popcll(sw);
        const uint32_t inc = wave_inclusive_scan<uint32_t>(cnt);
        if (cnt) {
          const uint64_t at = rpos + (inc - cnt);
          const uint32_t sh = (uint32_t)(at & 63);
          const uint64_t a = words[at >> 6], d = words[(at >> 6) + 1];
          uint64_t bits = sh ? (a >> sh) | (d << (64 - sh)) : a;
          uint64_t res = 0, m = sw;   // deposit the next cnt bits under the mask
          while (m) {
            const uint32_t j = (uint32_t)__ffsll((long long)m) - 1u;
            m &= m - 1;
            res |= (bits & 1ull) << j;
            bits >>= 1;
          }
          pb[w] = res;
        }
        rpos += __shfl(inc, 63, 64);
      }
      // the values found in this plane join the LSP (SPECK_INT.cpp:462-468)
      __threadfence_block();
      const uint32_t lim = min(nfound, (uint32_t)b.kStride);
      for (uint32_t k = lspDone + lane; k < lim; k += 64) {
        const uint32_t x = fpos[k];
        atomicOr(reinterpret_cast<unsigned long long*>(lsp) + (x >> 6), 1ull << (x & 63u));
      }
      lspDone = lim;
    }
  }

  if (ENC) {
    flush_acc();
    if (lane == 0) {
      oc.nbp = nbp;
      oc.total_bits = wpos;
    }
  }
  else {
    if (nfound > b.kStride) {
      if (lane == 0)
        oc.error = 3;
      return;
    }
    if (lane == 0)
      oc.found = nfound;
    __threadfence_block();
    // correctors (src/Outlier_Coder.cpp:199-233): 1.1 tol for magnitude 1, (m - 0.25) tol above
    const double tol = cst[c].q / 1.5;
    double* out = vals + c * valsStride;
    for (uint32_t kb = 0; kb < nfound; kb += 64) {
      const uint32_t k = kb + lane;
      if (k >= nfound)
        break;
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

int launch_outlier_rmq(hipStream_t st, const OutlierBufs& b, uint32_t maxCount)
{
  for (uint32_t j = 1; j < b.tblLevels && (1u << j) <= maxCount; j++) {
    const uint32_t blocks = capped_blocks((maxCount + kThreads - 1) / kThreads, b.nchunks);
    LAUNCH_K(k_outlier_rmq, dim3(blocks, b.nchunks), dim3(kThreads), 0, st, b, j);
  }
  HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_speck1d_encode(hipStream_t st, const OutlierBufs& b)
{
  LAUNCH_K(k_speck1d<true>, dim3(b.nchunks), dim3(64), 0, st, b, (const CoderState*)nullptr,
           (double*)nullptr, (size_t)0);
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

int launch_speck1d_decode(hipStream_t st, const OutlierBufs& b, const uint8_t* container,
                          const CoderState* cst, double* vals, size_t valsStride)
{
  const uint32_t blocks =
      capped_blocks((uint32_t)((b.streamStride + kThreads - 1) / kThreads), b.nchunks);
  LAUNCH_K(k_outlier_stream_in, dim3(blocks, b.nchunks), dim3(kThreads), 0, st, b, container);
  LAUNCH_K(k_speck1d<false>, dim3(b.nchunks), dim3(64), 0, st, b, cst, vals, valsStride);
  HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace sperrhip
