// speck_dec.hip -- SPECK3D bit-plane set-partitioning DECODER as HIP kernels.
//
// Mirrors /root/reference/src/SPECK_INT.cpp:165-228,359-469, src/SPECK3D_INT.cpp:99-212 and
// src/SPECK3D_INT_DEC.cpp:8-49; tests/model/speck_model.cpp (model_speck3d_decode) is the CPU
// model of the phases below.  Per bit plane:
//
//   LIP scan     (k_dec_count/_scan, k_lip_words/_scan/_apply/_deposit)  data parallel: a token
//                starts at every bit preceded by an EVEN number of consecutive 1 bits (a 1 is
//                always followed by its sign bit), so token starts and token ranks come from
//                prefix sums; the results are written by token rank and picked up by the pixels
//                through the rank of each candidate in raster order;
//   LIS phase    (k_lis_l0/_l1/_hi; k_lis_mx, speck_mx.hip, for lists that mix set shapes)  what each bit means depends
//                on every earlier bit of the phase: one workgroup per chunk; chunks run
//                concurrently;
//   refinement   (k_ref_apply2)  the j-th significant pixel in raster order takes bit j.
//
// Bits past the available length read as zero (the reference zero-pads a truncated stream,
// SPECK_INT.cpp:95-105); the loop stops where the reference's does.
#include "speck_dec.h"
#include "bit_words.h"

namespace sperrhip {

using namespace spk;

#define DEC_ACTIVE_OR_RETURN(s, p)                               \
  if (!(s).active || (s).done || (int)(p) >= (s).nbp)            \
    return;

__device__ __forceinline__ uint64_t get64(const uint64_t* words, uint64_t pos)
{
  const uint64_t lo = words[pos >> 6];
  const int sh = (int)(pos & 63);
  return sh ? (lo >> sh) | (words[(pos >> 6) + 1] << (64 - sh)) : lo;
}

// ------------------------------------------------------------------------------------------
// k_dec_load: parse the chunk stream (17-byte conditioner header, 9-byte SPECK header), copy the
// payload into an aligned, zero-padded word buffer (SPECK_FLT.cpp:27-109, SPECK_INT.cpp:79-108)
// ------------------------------------------------------------------------------------------
__global__ void k_dec_header(DecBuffers b, const uint8_t* container, const uint64_t* chunkOff,
                             const uint64_t* chunkLen, const uint64_t* initLIS,
                             const uint32_t* initLen, int wide_pass)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.nchunks)
    return;
  DecState& s = b.st[c];
  CoderState& cs = b.cst[c];
  const uint8_t* p = container + chunkOff[c];
  const uint64_t len = chunkLen[c];
  s.active = 0;
  s.done = 0;
  s.refPlaneP1 = 0;
  s.refPartial = 0;
  s.nbp = 0;
  s.pos = 0;
  s.cur = 0;
  s.error = 0;
  cs.is_const = 0;
  cs.wide = 0;
  if (len < 17) {
    s.error = 1;
    return;
  }
  double d1, d2;
  memcpy(&d1, p + 1, 8);
  memcpy(&d2, p + 9, 8);
  if (p[0] & 0x01) {  // constant field: {flags, u64 nval, f64 value}
    cs.is_const = 1;
    cs.mean = d2;
    if (len != 17)
      s.error = 1;
    return;
  }
  cs.mean = d1;
  cs.q = d2;
  if (len < 17 + 9) {
    s.error = 1;
    return;
  }
  const int nbp = p[17];
  uint64_t total_bits;
  memcpy(&total_bits, p + 18, 8);
  uint64_t avail = (len - 26) * 8;
  if (avail > total_bits)
    avail = total_bits;
  s.nbp = nbp;
  s.avail = avail;
  s.total_bits = total_bits;
  s.payload = chunkOff[c] + 26;
  cs.wide = nbp > 32 ? 1u : 0u;   // SPECK_FLT.cpp:64-72 (uint8/16/32 all fit the 32-bit path)
  cs.nbp = nbp;
  cs.total_bits = total_bits;
  s.active = (nbp > 0 && (int)cs.wide == wide_pass) ? 1u : 0u;
  s.iPart = b.iLevels;
  for (uint32_t l = 0; l < b.tree.nlevels; l++) {
    s.listLen[0][l] = initLen[l];
    s.listLen[1][l] = 0;
    for (uint32_t k = 0; k < initLen[l]; k++)
      b.lis[0][c * b.lisStride + b.levelOff[l] + k] = initLIS[b.levelOff[l] + k];
  }
}

// byte-wise gather of the payload into 64-bit words; words past the payload stay zero
__global__ void __launch_bounds__(kThreads)
k_dec_load_words(DecBuffers b, const uint8_t* container)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  if (!s.active)
    return;
  const uint64_t nbytes = (s.avail + 7) / 8;
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w * 8 >= nbytes)
    return;
  const uint8_t* p = container + s.payload + w * 8;
  uint64_t v = 0;
  const int n = (int)min((uint64_t)8, nbytes - w * 8);
  for (int k = 0; k < n; k++)
    v |= (uint64_t)p[k] << (8 * k);
  b.stream[c * b.streamStride + w] = v;
}

// ------------------------------------------------------------------------------------------
// candidates of the two pixel passes.  Pixel state is three bitmasks (one bit per sample, the
// reference's LIP / LSP masks, src/SPECK_INT.cpp:120-125): born = tested at least once, sigOld =
// significant before this plane, sigNew = found significant during this plane.  A decoder tile is
// 256 mask words = 16384 samples.
// ------------------------------------------------------------------------------------------
constexpr int kDecTileWords = kThreads;

// What the leaf states (DecBuffers::leafState) contribute to raster mask word `wi`: every sample
// of a split leaf has been tested (born), some are significant (sig), some of those negative
// (neg).  Sample (x, y, z) is child (x & 1) + 2 (y & 1) + 4 (z & 1) of leaf (x/2, y/2, z/2), so
// the word takes two adjacent bits of each of its 32 leaves.
// `tag`: only when a leaf of the word's 32-leaf block split on plane tag - 1 (k_leaf_apply leaves
// 1 + the plane in leafDirty): what earlier planes did to the block was folded when they ended, and
// a block's 64 bytes of states are not worth reading again on every plane (0: whatever its tag).
__device__ __forceinline__ bool leaf_word(const DecBuffers& b, uint32_t c, uint32_t wi,
                                          uint64_t& born, uint64_t& sig, uint64_t& neg, uint32_t tag)
{
  born = sig = neg = 0;
  if (b.wordLeaf == nullptr)
    return false;
  const uint32_t wl = b.wordLeaf[wi];
  if (wl == 0xffffffffu)
    return false;
  if (tag && b.leafDirty[c * b.leafDirtyStride + (wl >> 5)] != (uint8_t)tag)
    return false;
  const uint32_t sel = (wl & 3u) * 2u;
  const uint4* q = reinterpret_cast<const uint4*>(b.leafState + c * b.leafStateStride + (wl & ~31u));
  uint32_t v[16];
  uint32_t any = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint4 t = q[k];
    v[4 * k] = t.x;
    v[4 * k + 1] = t.y;
    v[4 * k + 2] = t.z;
    v[4 * k + 3] = t.w;
    any |= t.x | t.y | t.z | t.w;
  }
  if (!any)
    return false;
  uint32_t bo[2] = {0, 0}, si[2] = {0, 0}, ne[2] = {0, 0};
#pragma unroll
  for (int k = 0; k < 16; k++) {   // v[k] holds leaves 2k (low half) and 2k + 1
    const int h = k >> 3, sh = (k & 7) * 4;
    const uint32_t a = v[k] & 0xffffu, d = v[k] >> 16;
    const uint32_t s4 = ((a >> sel) & 3u) | (((d >> sel) & 3u) << 2);
    const uint32_t n4 = ((a >> (8 + sel)) & 3u) | (((d >> (8 + sel)) & 3u) << 2);
    const uint32_t b4 = ((a & 0xffu) ? 3u : 0u) | ((d & 0xffu) ? 12u : 0u);
    si[h] |= s4 << sh;
    ne[h] |= n4 << sh;
    bo[h] |= b4 << sh;
  }
  born = (uint64_t)bo[0] | ((uint64_t)bo[1] << 32);
  sig = (uint64_t)si[0] | ((uint64_t)si[1] << 32);
  neg = (uint64_t)ne[0] | ((uint64_t)ne[1] << 32);
  return true;
}

// merges last plane's new significances (sigNew and the leaf states), then counts LIP candidates
// (born & ~sig) and refinement candidates (sig) per tile
__global__ void __launch_bounds__(kThreads) k_dec_count(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  __shared__ uint32_t sm[kThreads / 64 + 1];
  __shared__ uint32_t sh_need;
  const uint32_t nw = (b.tree.nvals + 63) / 64;
  // A tile none of whose samples has ever been tested holds nothing to count, and nothing to fold unless a
  // leaf over it split on the plane before: then its three mask words per thread are not read at all (round
  // 4: the dozen planes before the heavy ones touch the coarse subbands only and swept every mask of the
  // chunk, 85 us per launch of 32 chunks).  Only where every birth comes through the leaf states: a word
  // without a leaf mapping (the small subbands, whose events k_leaf_apply applies itself) keeps its tile in.
  // The test is made for ALL of the workgroup's tiles at once (round 5, second session: a workgroup of a large batch
  // walks eight tiles, and tile after tile the test was two dependent loads and a barrier -- 55 to 70 us a launch
  // on the planes that hold next to nothing).
  constexpr uint32_t kAhead = 8;
  for (uint32_t tile0 = blockIdx.x; tile0 < b.nPixTiles; tile0 += gridDim.x * kAhead) {
  uint32_t needMask = 0;
  const bool skipTest = b.tileBorn != nullptr && b.wordLeaf != nullptr;
  if (skipTest) {
    if (threadIdx.x == 0)
      sh_need = 0;
    uint32_t wl[kAhead], tb[kAhead];
#pragma unroll
    for (uint32_t q = 0; q < kAhead; q++) {
      const uint32_t tile = tile0 + q * gridDim.x;
      const uint32_t wi = tile * kDecTileWords + threadIdx.x;
      tb[q] = tile < b.nPixTiles ? (uint32_t)b.tileBorn[c * b.tileStride + tile] : 0u;
      wl[q] = (tile < b.nPixTiles && wi < nw) ? b.wordLeaf[wi] : 0xfffffffeu;   // (..fe: no word here)
    }
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t q = 0; q < kAhead; q++) {
      const uint32_t tile = tile0 + q * gridDim.x;
      if (tile >= b.nPixTiles)
        continue;
      bool need = tb[q] != 0;
      if (!need && wl[q] != 0xfffffffeu)
        need = wl[q] == 0xffffffffu || b.leafDirty[c * b.leafDirtyStride + (wl[q] >> 5)] == (uint8_t)(p + 2);
      mine |= need ? 1u << q : 0u;
    }
    __syncthreads();
    if (mine)
      atomicOr(&sh_need, mine);
    __syncthreads();
    needMask = sh_need;
  }
  for (uint32_t q = 0; q < kAhead; q++) {
  const uint32_t tile = tile0 + q * gridDim.x;
  if (tile >= b.nPixTiles)
    break;
  const uint32_t wi = tile * kDecTileWords + threadIdx.x;
  if (skipTest && !((needMask >> q) & 1u)) {
    if (threadIdx.x == 0) {
      b.tileLip[c * b.tileStride + tile] = 0;
      b.tileRef[c * b.tileStride + tile] = 0;
    }
    continue;
  }
  uint32_t v = 0;
  bool anyBorn = false;
  if (wi < nw) {
    uint64_t* so = b.sigOld + c * b.maskPixStride + wi;
    uint64_t* sn = b.sigNew + c * b.maskPixStride + wi;
    const uint64_t fresh = *sn;
    uint64_t sig = *so;
    uint64_t born = b.bornM[c * b.maskPixStride + wi];
    uint64_t lb = 0, ls = 0, ln = 0;
    // (idempotent: old leaf results are folded again -- but once all 64 samples of the word are born,
    //  every leaf over it has split and was folded by an earlier plane: their states are final)
    if (born != ~0ull && leaf_word(b, c, wi, lb, ls, ln, (uint32_t)p + 2u)) {   // leaves that split on plane p + 1
      if (lb & ~born) {
        born |= lb;
        b.bornM[c * b.maskPixStride + wi] = born;
      }
      if (ln) {
        const uint64_t sg = b.sign[c * b.signStride + wi];
        if (sg & ln)
          b.sign[c * b.signStride + wi] = sg & ~ln;
      }
    }
    if (fresh | (ls & ~sig)) {
      if (b.refPlanes) {
        // found on plane p + 1: bit p + 1 of their magnitudes.  The word of that plane holds the refinement
        // bits of the older samples when there were any (k_ref_deposit(p + 1) wrote it), else nothing yet
        const uint64_t found = (fresh | ls) & ~sig;
        if (found && (uint32_t)(p + 1) < b.refNPlanes) {
          uint64_t* pw = b.refPlanes + c * b.refPlaneStride + ref_plane_word((uint32_t)(p + 1), wi);
          *pw = sig ? (*pw | found) : found;
          if (!sig)
            b.wordTop[c * b.wordTopStride + wi] = (uint8_t)(p + 2);
        }
      }
      sig |= fresh | ls;
      *so = sig;
    }
    if (fresh)
      *sn = 0;
    const uint64_t lip = born & ~sig;
    v = (uint32_t)__popcll(lip) | ((uint32_t)__popcll(sig) << 16);
    anyBorn = born != 0;
  }
  if (b.tileBorn != nullptr) {
    const int anyB = __syncthreads_or(anyBorn ? 1 : 0);
    if (threadIdx.x == 0 && anyB)
      b.tileBorn[c * b.tileStride + tile] = 1;
  }
  // the tile's two counts: 256 words x 64 bits, both sums fit in 15 bits -- one reduction of the packed pair (two
  // block scans before, whose prefixes nobody used)
  uint32_t r = v;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1)
    r += (uint32_t)__shfl_xor((int)r, d, 64);
  __syncthreads();   // (sm: the round before)
  if ((threadIdx.x & 63) == 0)
    sm[threadIdx.x >> 6] = r;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < kThreads / 64; w++)
      tot += sm[w];
    b.tileLip[c * b.tileStride + tile] = tot & 0xffffu;
    b.tileRef[c * b.tileStride + tile] = tot >> 16;
  }
  }
  }
}

__global__ void __launch_bounds__(kThreads) k_dec_scan(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.x;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  __shared__ uint64_t sm[kThreads / 64 + 1];
  uint32_t* tl = b.tileLip + c * b.tileStride;
  uint32_t* tr = b.tileRef + c * b.tileStride;
  uint32_t* ol = b.tileLipOff + c * b.tileStride;
  uint32_t* orr = b.tileRefOff + c * b.tileStride;
  uint64_t carry = 0;  // low 32: lip, high 32: ref
  for (uint32_t base = 0; base < b.nPixTiles; base += kThreads) {
    const uint32_t i = base + threadIdx.x;
    const uint64_t v = i < b.nPixTiles ? ((uint64_t)tl[i] | ((uint64_t)tr[i] << 32)) : 0;
    uint64_t total;
    const uint64_t ex = block_exclusive_scan<uint64_t>(v, sm, &total) + carry;
    if (i < b.nPixTiles) {
      ol[i] = (uint32_t)ex;
      orr[i] = (uint32_t)(ex >> 32);
    }
    carry += total;
  }
  if (threadIdx.x == 0) {
    s.nLip = (uint32_t)carry;
    s.nRef = (uint32_t)(carry >> 32);
    s.lipStart = s.pos;
    s.l0Ticket = 0;
    s.l1Ticket = 0;
    s.l2Ticket = 0;
    s.hiTicket = 0;
    s.hiCompactDone = 0;
    s.bornCount = 0;
    s.leafCount = 0;
    for (int g = 0; g < 8; g++)
      s.hiBornCnt[g] = s.hiLeafCnt[g] = 0;
  }
}

// The k-th LIP candidate in raster order owns token k of the scan: take the results k_lip_apply
// left at bit k of lipSig / lipNeg.  One thread per mask word; it is the only writer of that word.
__global__ void __launch_bounds__(kThreads) k_lip_deposit(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  if (s.nLip == 0)
    return;
  __shared__ uint32_t sm[kThreads / 64 + 1];
  const uint32_t nw = (b.tree.nvals + 63) / 64;
  const uint64_t* rs = b.lipSig + c * b.lipResStride;
  const uint64_t* rn = b.lipNeg + c * b.lipResStride;
  for (uint32_t tile = blockIdx.x; tile < b.nPixTiles; tile += gridDim.x) {
    if (b.tileLip[c * b.tileStride + tile] == 0)
      continue;
    const uint32_t wi = tile * kDecTileWords + threadIdx.x;
    uint64_t lip = 0;
    if (wi < nw)
      lip = b.bornM[c * b.maskPixStride + wi] & ~b.sigOld[c * b.maskPixStride + wi];
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<uint32_t>((uint32_t)__popcll(lip), sm, &total) +
                        b.tileLipOff[c * b.tileStride + tile];
    const uint32_t n = (uint32_t)__popcll(lip);
    if (n == 0)
      continue;
    const uint32_t sh = ex & 63u;
    uint64_t vs = rs[ex >> 6] >> sh, vn = rn[ex >> 6] >> sh;
    if (sh && sh + n > 64) {
      vs |= rs[(ex >> 6) + 1] << (64 - sh);
      vn |= rn[(ex >> 6) + 1] << (64 - sh);
    }
    if (n < 64) {
      vs &= (1ull << n) - 1;
      vn &= (1ull << n) - 1;
    }
    if (vs == 0)
      continue;
    uint64_t outS = vs, outN = vn;   // candidate i of the word is the i-th set bit of `lip`
    spread_under_mask(lip, outS, outN);
    b.sigNew[c * b.maskPixStride + wi] |= outS;
    if (outN)
      b.sign[c * b.signStride + wi] &= ~outN;
  }
}

// ------------------------------------------------------------------------------------------
// LIP scan: token starts of 64 stream bits per thread
// ------------------------------------------------------------------------------------------
// Relative bit k of the phase is stream bit lipStart + k.  The phase has nLip tokens and is at
// most 2*nLip bits long; bit 2*nLip is examined too so that "where token #nLip would start" (=
// the length of the phase) always exists.
__device__ __forceinline__ uint64_t lip_word(const uint64_t* words, const DecState& s,
                                             uint64_t w, uint64_t nbits)
{
  // bits past `avail` are zero padding; bits past nbits are ignored by the caller
  const uint64_t pos = s.lipStart + w * 64;
  uint64_t v = get64(words, pos);
  if (pos + 64 > s.avail)
    v = pos >= s.avail ? 0 : (v & ((1ull << (s.avail - pos)) - 1));
  (void)nbits;
  return v;
}

// Round 5: the token ranks are a SEGMENTED scan.  k_lip_words takes the words a segment (kLipSeg = 2048 words, eight in a
// row per thread) at a time and leaves every word's rank inside its segment and the segment's token count;
// k_lip_scan scans the few hundred segment counts and looks for the phase's end inside ONE segment; k_lip_apply adds
// the segment's rank.  (k_lip_scan scanned every word before, one workgroup per chunk, a block scan per 2048 words:
// 50 rounds = 180 us in the late planes, 37 us a launch on average in a batch of two chunks.)
constexpr int kLipPer = 8;
constexpr int kLipSeg = kThreads * kLipPer;
static_assert(kLipSeg == 2048, "carve_dec sizes tokSegSum by it");
__global__ void __launch_bounds__(kThreads) k_lip_words(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  if (s.nLip == 0)
    return;
  __shared__ uint32_t sm[kThreads / 64 + 1];
  const uint64_t nbits = 2ull * s.nLip + 1;
  const uint64_t nwords = (nbits + 63) / 64;
  const uint32_t nseg = (uint32_t)((nwords + kLipSeg - 1) / kLipSeg);
  const uint64_t* words = b.stream + c * b.streamStride;
  for (uint32_t seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    const uint64_t w0 = (uint64_t)seg * kLipSeg + (uint64_t)threadIdx.x * kLipPer;
    uint32_t cntv[kLipPer], tsum = 0;
    // (the thread's words first, all loads in flight together: the parity below chains their token starts)
    uint64_t xw[kLipPer];
#pragma unroll
    for (int k = 0; k < kLipPer; k++)
      xw[k] = w0 + (uint64_t)k < nwords ? lip_word(words, s, w0 + (uint64_t)k, nbits) : 0ull;
    // parity of the run of 1s that ends right before the thread's first word
    uint32_t parity = 0;
    if (w0 < nwords)
      for (uint64_t back = w0; back > 0;) {
        back--;
        const uint64_t v = lip_word(words, s, back, nbits);
        if (v == ~0ull)
          continue;  // 64 more ones: parity unchanged, keep looking
        parity = (uint32_t)__clzll((long long)~v) & 1u;  // leading ones of the previous word
        break;
      }
#pragma unroll
    for (int k = 0; k < kLipPer; k++) {
      const uint64_t w = w0 + (uint64_t)k;
      cntv[k] = 0;
      if (w >= nwords)
        continue;
      const uint64_t x = xw[k];
      uint64_t starts = lip_token_starts(x, parity);   // (bit_words.h; a loop over the 64 bits before: 300 operations a word)
      if (w == nwords - 1 && (nbits & 63))
        starts &= (1ull << (nbits & 63)) - 1;
      b.tokMask[c * b.tokStride + w] = starts;
      cntv[k] = (uint32_t)__popcll(starts);
      b.tokCnt[c * b.tokStride + w] = cntv[k];
      tsum += cntv[k];
      if (w <= (uint64_t)(s.nLip - 1) / 64) {   // results by token rank, filled by k_lip_apply
        b.lipSig[c * b.lipResStride + w] = 0;
        b.lipNeg[c * b.lipResStride + w] = 0;
      }
    }
    uint32_t total;
    uint32_t ex = block_exclusive_scan<uint32_t>(tsum, sm, &total);
#pragma unroll
    for (int k = 0; k < kLipPer; k++) {
      const uint64_t w = w0 + (uint64_t)k;
      if (w < nwords)
        b.tokOff[c * b.tokStride + w] = ex;
      ex += cntv[k];
    }
    if (threadIdx.x == 0)
      b.tokSegSum[c * b.tokSegStride + seg] = total;
  }
}

__global__ void __launch_bounds__(kThreads) k_lip_scan(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.x;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  if (s.nLip == 0) {
    if (threadIdx.x == 0)
      s.lipBits = 0;
    return;
  }
  __shared__ uint32_t sm[kThreads / 64 + 1];
  __shared__ uint32_t sh_seg, sh_segBase;
  const uint64_t nbits = 2ull * s.nLip + 1;
  const uint32_t nwords = (uint32_t)((nbits + 63) / 64);
  const uint32_t nseg = (nwords + kLipSeg - 1) / kLipSeg;
  const uint32_t* segSum = b.tokSegSum + c * b.tokSegStride;
  uint32_t* segBase = b.tokSegBase + c * b.tokSegStride;
  const uint32_t nLip = s.nLip;
  if (threadIdx.x == 0)
    sh_seg = sh_segBase = 0;   // (the block scan's barriers come before anybody writes them)
  // the segments' ranks; the one token #nLip (0-based) starts in: the phase is as long as where it starts
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nseg; base += kThreads) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nseg ? segSum[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<uint32_t>(v, sm, &total) + carry;
    if (i < nseg) {
      segBase[i] = ex;
      if (ex <= nLip && nLip < ex + v) {
        sh_seg = i;
        sh_segBase = ex;
      }
    }
    carry += total;
  }
  __syncthreads();
  const uint32_t seg = sh_seg, sb = sh_segBase;
  const uint32_t* cnt = b.tokCnt + c * b.tokStride;
  const uint32_t* off = b.tokOff + c * b.tokStride;
  const uint64_t* mask = b.tokMask + c * b.tokStride;
#pragma unroll
  for (int k = 0; k < kLipPer; k++) {
    const uint32_t i = seg * kLipSeg + threadIdx.x * kLipPer + (uint32_t)k;
    if (i >= nwords)
      continue;
    const uint32_t ex = sb + off[i], v = cnt[i];
    if (ex <= nLip && nLip < ex + v) {
      uint64_t m = mask[i];
      for (uint32_t r = nLip - ex; r > 0; r--)
        m &= m - 1;
      s.lipBits = (uint64_t)i * 64 + (uint64_t)__ffsll((long long)m) - 1;
    }
  }
}

template <typename CT>
__global__ void __launch_bounds__(kThreads) k_lip_apply(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  if (s.nLip == 0)
    return;
  const uint64_t nbits = 2ull * s.nLip + 1;
  const uint64_t nwords = min((nbits + 63) / 64, s.lipBits / 64 + 1);   // the scan's real length
  const uint64_t* words = b.stream + c * b.streamStride;
  unsigned long long* rs = reinterpret_cast<unsigned long long*>(b.lipSig + c * b.lipResStride);
  unsigned long long* rn = reinterpret_cast<unsigned long long*>(b.lipNeg + c * b.lipResStride);
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords;
       w += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t x = lip_word(words, s, w, nbits);
    uint64_t sig = b.tokMask[c * b.tokStride + w];
    const uint32_t j = b.tokOff[c * b.tokStride + w] + b.tokSegBase[c * b.tokSegStride + (w >> 11)];   // (kLipSeg words a segment)
    if (j >= s.nLip)
      continue;
    const uint64_t nextbit = lip_word(words, s, w + 1, nbits) & 1ull;
    // the tokens that start in this word have consecutive ranks j, j + 1, ...: their results go
    // to bits j, j + 1, ... of lipSig (found significant) and lipNeg (and negative).  The
    // magnitude is not written here: k_ref_apply2 / the inverse quantiser give a newly significant
    // coefficient its value 1.5 * 2^plane - 1 (SPECK_INT.cpp:462-468) when they first touch it.
    // token i of the word: its bit, and the bit behind it (the sign of a '1')
    uint64_t runS = x, runN = ~((x >> 1) | (nextbit << 63));
    gather_under_mask(sig, runS, runN);
    const uint32_t left = s.nLip - j;   // (tokens past the phase's own do not count)
    if (left < 64)
      runS &= (1ull << left) - 1;
    runN &= runS;
    if (runS == 0)
      continue;
    const uint32_t sh = j & 63u;
    atomicOr(rs + (j >> 6), runS << sh);
    if (sh && (runS >> (64 - sh)))
      atomicOr(rs + (j >> 6) + 1, runS >> (64 - sh));
    if (runN) {
      atomicOr(rn + (j >> 6), runN << sh);
      if (sh && (runN >> (64 - sh)))
        atomicOr(rn + (j >> 6) + 1, runN >> (64 - sh));
    }
  }
}

// ------------------------------------------------------------------------------------------
// LIS phase: one wavefront per chunk walks the lists.  Control flow is wave-uniform (every lane
// carries the same walker state; frames live in LDS); the lanes are used for what is parallel
// inside the walk: copying runs of insignificant entries to the next list (one count-trailing-
// zeros per run instead of one step per entry), evaluating the up-to-8 children of a splitting
// set, and writing the pixel results.
// ------------------------------------------------------------------------------------------
struct WFrame {
  uint64_t packed;      // the splitting set
  uint32_t ridx[8];     // raster index of every single-sample child
  uint32_t base[3];     // child index = base + (ci >> axis & 1)
  uint16_t childGrid;
  uint8_t present;      // bit ci: child ci exists
  uint8_t pixel;        // bit ci: child ci is a single sample
  uint8_t next;         // next child slot to visit (0..8)
  uint8_t found;        // some earlier child was significant
  uint8_t last;         // slot of the last existing child
  uint8_t kidlev;       // LIS level of the children
  uint8_t sigmask;      // pixel children found significant
  uint8_t signmask;     // their sign bits
};

struct BitReader {      // wave-uniform sequential reader over a 64-word window held in a register
  const uint64_t* words;
  uint64_t nwords;      // words at or past this index read as zero (zero padding)
  uint64_t pos, cur, n1;
  // Lane i keeps stream word wbase + i.  A word is picked out of the window when the reader gets
  // to it; memory is touched once per 4096 bits.  (With a load per word the compiler wants the
  // loaded word in scalar registers at once and waits for it there -- and with it for every
  // atomic the walk has in flight, several times per 64 stream bits.)
  uint64_t win, wbase;
  __device__ __forceinline__ uint64_t word(uint64_t idx)
  {
    if (idx - wbase >= 64ull) {
      wbase = idx;
      const uint64_t k = idx + (threadIdx.x & 63u);
      win = k < nwords ? words[k] : 0ull;
    }
    return __shfl(win, (int)(idx - wbase), 64);
  }
  __device__ __forceinline__ void init(const uint64_t* w, uint64_t p, uint64_t avail)
  {
    words = w;
    nwords = (avail + 63) / 64;
    pos = p;
    wbase = p >> 6;
    const uint64_t k = wbase + (threadIdx.x & 63u);
    win = k < nwords ? words[k] : 0ull;
    cur = word(p >> 6);
    n1 = word((p >> 6) + 1);
  }
  __device__ __forceinline__ uint64_t peek64() const
  {
    const int sh = (int)(pos & 63);
    return sh ? (cur >> sh) | (n1 << (64 - sh)) : cur;
  }
  __device__ __forceinline__ void skip(uint32_t n)   // n <= 64
  {
    const uint64_t np = pos + n;
    if ((np >> 6) != (pos >> 6)) {
      cur = n1;
      n1 = word((np >> 6) + 1);
    }
    pos = np;
  }
  __device__ __forceinline__ uint32_t get1()
  {
    const uint32_t bit = (uint32_t)((cur >> (pos & 63)) & 1ull);
    skip(1);
    return bit;
  }
};

constexpr int kWalkGrids = 512;    // grid descriptors and interval-start entries k_lis_walk keeps in LDS
constexpr int kWalkTab = 12288;    // (a tree with more reads them from global memory)

template <typename CT>
__global__ void __launch_bounds__(64) k_lis_walk(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.x;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  __shared__ WFrame fr[kMaxDepth + 2];
  __shared__ uint32_t nextLen[kMaxLevels];
  // The tree's tables in LDS: every frame reads its grid, the grid's root and (for lengths that
  // are not powers of two) interval starts, each load depending on the one before -- from global
  // memory that is a chain of round trips per significant set.  (Measured: 3.2 -> 3.07 s for a 250^3
  // chunk together with the list window below; the walk is bound by its instruction count, about
  // 230 cycles per stream bit, not by these loads.)
  __shared__ Root sh_roots[kMaxRoots];
  __shared__ Grid sh_grids[kWalkGrids];
  __shared__ uint16_t sh_tab[kWalkTab];
  Tree t = b.tree;
  const int lane = threadIdx.x;
  if (t.nroots <= (uint32_t)kMaxRoots && t.ngrids <= (uint32_t)kWalkGrids) {
    for (uint32_t i = lane; i < t.nroots; i += 64)
      sh_roots[i] = b.tree.roots[i];
    for (uint32_t i = lane; i < t.ngrids; i += 64)
      sh_grids[i] = b.tree.grids[i];
    t.roots = sh_roots;
    t.grids = sh_grids;
  }
  if (b.treeTabLen <= (uint32_t)kWalkTab) {
    for (uint32_t i = lane; i < b.treeTabLen; i += 64)
      sh_tab[i] = b.tree.tab[i];
    t.tab = sh_tab;
  }
  const uint64_t* words = b.stream + c * b.streamStride;
  unsigned long long* bornM = reinterpret_cast<unsigned long long*>(b.bornM + c * b.maskPixStride);
  unsigned long long* sigNew = reinterpret_cast<unsigned long long*>(b.sigNew + c * b.maskPixStride);
  CT* coef = reinterpret_cast<CT*>(b.coef) + c * b.coefStride;
  unsigned long long* sign = reinterpret_cast<unsigned long long*>(b.sign + c * b.signStride);
  const CT thr = (CT)1 << p;
  const CT init = thr + thr - thr / 2 - 1;
  const uint32_t cur = s.cur, nx = cur ^ 1u;
  for (uint32_t l = lane; l < t.nlevels; l += 64)
    nextLen[l] = 0;
  __syncthreads();
  BitReader rd;
  rd.init(words, s.lipStart + s.lipBits, s.avail);

  // A significant set whose children are all single samples (most significant sets are such leaf
  // parents): at most 16 bits, read off the stream at once, decoded in straight-line code.  Nothing
  // can be entered, so no frame is set up and no barrier is needed; the set's parent goes on with
  // its next child.  Returns false (and reads nothing) for any other set.
  // (what a set's grid and root contribute stays in registers: consecutive sets mostly share them)
  uint32_t lastGrid = 0xffffffffu;
  Root rc = {};
  int gee[3] = {0, 0, 0};             // log2 extent of the children's grid per axis
  uint32_t gsa[3] = {0, 0, 0};        // 1: the axis still splits at this depth
  uint32_t gstep[3] = {0, 0, 0};      // power-of-two axis: length of a child interval
  uint32_t gtab[3] = {0, 0, 0};       // other axes: index of interval 0 in the staged table
  uint32_t gpow2 = 0;                 // bit a: axis a has a power-of-two length
  const bool tabStaged = t.tab == sh_tab;
  auto try_leaf = [&](const Node& nd) -> bool {
    if (nd.grid != lastGrid) {
      lastGrid = nd.grid;
      const Grid gc = t.grids[nd.grid];
      rc = t.roots[gc.root];
      gpow2 = 0;
      for (int a = 0; a < 3; a++) {
        gsa[a] = gc.depth < rc.D[a] ? 1u : 0u;
        gee[a] = (int)gc.e[a] + (int)gsa[a];
        const uint32_t L = rc.len[a];
        if ((L & (L - 1u)) == 0) {
          gpow2 |= 1u << a;
          gstep[a] = L >> gee[a];
          gtab[a] = 0;
        }
        else {
          gstep[a] = 0;
          gtab[a] = rc.tabOff[a] + tab_index(gee[a], 0);
        }
      }
    }
    const Root& r = rc;
    const int* ee = gee;
    uint32_t cnt = 1, idx[3];
    uint32_t ok = (uint32_t)lane < 8u ? 1u : 0u;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const uint32_t bit = ((uint32_t)lane >> a) & 1u;
      ok &= (bit & (gsa[a] ^ 1u)) ^ 1u;                 // the upper half only where the axis splits
      idx[a] = ((uint32_t)nd.i[a] << gsa[a]) + bit;
      cnt *= axis_len(r.len[a], ee[a], idx[a]);
    }
    cnt *= ok;
    const uint32_t present = (uint32_t)(__ballot(cnt > 0) & 0xffull);
    const uint32_t pixel = (uint32_t)(__ballot(cnt == 1) & 0xffull);
    if ((present & ~pixel) != 0u)
      return false;
    const uint64_t w = rd.peek64();
    const uint32_t last = 31u - (uint32_t)__clz((int)present);
    // (branch-free: a lone wavefront pays some twenty cycles for every taken branch)
    uint32_t y = 0, found = 0, sigmask = 0, signmask = 0;
#pragma unroll
    for (uint32_t ci = 0; ci < 8; ci++) {
      const uint32_t pres = (present >> ci) & 1u;
      const uint32_t coded = pres & (found | (uint32_t)(ci != last));
      const uint32_t sig = pres & (((uint32_t)(w >> y) & coded) | (coded ^ 1u));   // not coded: implied 1
      y += coded;
      found |= sig;
      sigmask |= sig << ci;
      signmask |= ((uint32_t)(w >> y) & sig) << ci;   // the sign follows a significant sample
      y += sig;
    }
    rd.skip(y);
    if (cnt == 1) {   // lanes 0..7 that have a sample
      uint32_t ridx;
      if (tabStaged) {   // pixel_raster with the interval starts read from LDS as LDS, no branch per axis
        uint32_t cc[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const uint32_t tv = sh_tab[min(gtab[a] + idx[a], (uint32_t)kWalkTab - 1u)];
          cc[a] = (uint32_t)r.org[a] + (((gpow2 >> a) & 1u) ? idx[a] * gstep[a] : tv);
        }
        ridx = (cc[2] * t.dims[1] + cc[1]) * t.dims[0] + cc[0];
      }
      else
        ridx = pixel_raster(t, r, ee, idx);
      atomicOr(bornM + (ridx >> 6), 1ull << (ridx & 63));
      if ((sigmask >> lane) & 1u) {
        atomicOr(sigNew + (ridx >> 6), 1ull << (ridx & 63));
        if (!((signmask >> lane) & 1u))
          atomicAnd(sign + (ridx >> 6), ~(1ull << (ridx & 63)));
      }
    }
    return true;
  };

  for (uint32_t l = t.nlevels; l-- > 0;) {
    const uint32_t n = s.listLen[cur][l];
    const uint64_t* list = b.lis[cur] + c * b.lisStride + b.levelOff[l];
    uint64_t* keep = b.lis[nx] + c * b.lisStride + b.levelOff[l];
    uint32_t nkeep = 0, e = 0;
    // lane i holds entry winBase + i of the list: 64 entries per (coalesced) load instead of one
    // dependent load per significant entry
    uint32_t winBase = 0;
    uint64_t win = (uint32_t)lane < n ? list[lane] : 0ull;
    while (e < n) {
      const uint64_t w = rd.peek64();
      uint32_t z = w ? (uint32_t)__ffsll((long long)w) - 1u : 64u;
      z = min(z, n - e);
      if (z) {  // a run of insignificant entries: they stay, in order
        if ((uint32_t)lane < z)
          keep[nkeep + lane] = list[e + lane];
        nkeep += z;
        e += z;
        rd.skip(z);
        continue;
      }
      rd.skip(1);  // the entry's '1'
      int sp = 0;
      if (e - winBase >= 64u) {
        winBase = e;
        win = e + (uint32_t)lane < n ? list[e + lane] : 0ull;
      }
      uint64_t enter = __shfl(win, (int)(e - winBase), 64);
      e++;
      if (try_leaf(unpack_node(enter)))
        continue;
      bool fresh = true;
      while (sp >= 0) {
        WFrame& f = fr[sp];
        if (fresh) {  // set up the frame of the set `enter`
          const Node nd = unpack_node(enter);
          const Grid& g = t.grids[nd.grid];
          const Root& r = t.roots[g.root];
          int ee[3];
          uint32_t base[3], splits = 0, cnt = 1, idx[3];
          bool valid = lane < 8;
          for (int a = 0; a < 3; a++) {
            const bool sa = g.depth < r.D[a];
            splits |= (sa ? 1u : 0u) << a;
            ee[a] = sa ? g.e[a] + 1 : g.e[a];
            base[a] = sa ? (uint32_t)nd.i[a] * 2u : (uint32_t)nd.i[a];
            const uint32_t bit = ((uint32_t)lane >> a) & 1u;
            if (bit && !sa)
              valid = false;
            idx[a] = base[a] + bit;
            cnt *= axis_len(r.len[a], ee[a], idx[a]);
          }
          if (!valid)
            cnt = 0;
          const uint32_t present = (uint32_t)(__ballot(cnt > 0) & 0xffull);
          const uint32_t pixel = (uint32_t)(__ballot(cnt == 1) & 0xffull);
          const NodeGeom q = node_geom(t, nd);
          const uint32_t kidlev =
              node_level(t, nd) + (q.len[0] > 1) + (q.len[1] > 1) + (q.len[2] > 1);
          if (cnt == 1)
            f.ridx[lane] = pixel_raster(t, r, ee, idx);
          if (lane == 0) {
            f.packed = enter;
            f.base[0] = base[0];
            f.base[1] = base[1];
            f.base[2] = base[2];
            f.childGrid = (uint16_t)(nd.grid + 1);
            f.present = (uint8_t)present;
            f.pixel = (uint8_t)pixel;
            f.next = 0;
            f.found = 0;
            f.last = (uint8_t)(31 - __clz((int)present));
            f.kidlev = (uint8_t)kidlev;
            f.sigmask = 0;
            f.signmask = 0;
          }
          __syncthreads();
          fresh = false;
        }
        // visit the children in order until one of them has to be entered
        uint32_t nextc = f.next, found = f.found, sigmask = f.sigmask, signmask = f.signmask;
        const uint32_t present = f.present, pixel = f.pixel, last = f.last;
        bool descend = false;
        while (nextc < 8) {
          const uint32_t ci = nextc++;
          if (!((present >> ci) & 1u))
            continue;
          const bool coded = found || ci != last;
          const uint32_t sig = coded ? rd.get1() : 1u;
          found |= sig;
          if ((pixel >> ci) & 1u) {
            if (sig) {
              sigmask |= 1u << ci;
              signmask |= rd.get1() << ci;
            }
          }
          else {
            Node kid;
            kid.grid = f.childGrid;
            kid.i[0] = (uint16_t)(f.base[0] + (ci & 1u));
            kid.i[1] = (uint16_t)(f.base[1] + ((ci >> 1) & 1u));
            kid.i[2] = (uint16_t)(f.base[2] + ((ci >> 2) & 1u));
            if (sig) {
              if (try_leaf(kid))
                continue;   // (handled in place: the frame stays in registers)
              enter = pack_node(kid);
              descend = true;
              break;
            }
            const uint32_t kl = f.kidlev;
            if (lane == 0) {
              b.lis[nx][c * b.lisStride + b.levelOff[kl] + nextLen[kl]] = pack_node(kid);
              nextLen[kl]++;
            }
          }
        }
        __syncthreads();
        if (lane == 0) {
          f.next = (uint8_t)nextc;
          f.found = (uint8_t)found;
          f.sigmask = (uint8_t)sigmask;
          f.signmask = (uint8_t)signmask;
        }
        __syncthreads();
        if (descend) {
          sp++;
          fresh = true;
          continue;
        }
        // all children visited: lanes 0..7 write the pixel results, then leave the frame
        if (lane < 8 && ((present & pixel) >> lane) & 1u) {
          const uint32_t ridx = f.ridx[lane];
          atomicOr(bornM + (ridx >> 6), 1ull << (ridx & 63));
          if ((sigmask >> lane) & 1u) {
            atomicOr(sigNew + (ridx >> 6), 1ull << (ridx & 63));
            if (!((signmask >> lane) & 1u))
              atomicAnd(sign + (ridx >> 6), ~(1ull << (ridx & 63)));
          }
        }
        __syncthreads();
        sp--;
      }
    }
    if (lane == 0)
      nextLen[l] += nkeep;  // nothing was appended to this level yet: survivors come first
    __syncthreads();
  }
  for (uint32_t l = lane; l < t.nlevels; l += 64)
    s.listLen[nx][l] = nextLen[l];
  if (lane == 0) {
    s.cur = nx;
    s.pos = rd.pos;
    s.nLeafEv = 0;
    s.lastPlane = p;
    if (rd.pos >= s.avail)  // SPECK_INT.cpp:200-201
      s.done = 1;
  }
}

// ------------------------------------------------------------------------------------------
// LIS phase, list of the smallest sets (2x2x2 leaf sets: class 0).  The sorting pass visits the
// lists from the smallest sets to the largest (SPECK_INT.cpp:317-327), so this list's code starts
// where the LIP scan ended and its entry count is known: the whole GPU decodes it before
// k_lis_l1 / k_lis_hi take the other lists.  An entry is '0', or '1' followed by the <= 16 bits of its
// eight pixels; nothing is born.  The stream is cut into blocks of kL0W bits handed out by a
// ticket counter; each block
//   * finds, for every bit position, the length of a token that would start there, and by pointer
//     jumping where a chain of tokens entering at that position leaves the block, how many tokens
//     it holds and how many of them are significant;
//   * waits for its predecessor's look-back word (entry offset, entries and significant entries so
//     far), publishes its own straight from those tables, and only then
//   * marks the tokens really on the chain and lets every thread handle its share: insignificant
//     entries are copied to the next list in order, significant ones become leaf events.
// A block is handed out only after all earlier ones, so a waiting block always waits for a
// workgroup that is running (or has seen the pass end).
// ------------------------------------------------------------------------------------------
// (k_lis_l0's tick counters cost registers the kernel does not have at two workgroups a compute unit: they are compiled
//  in with -DSPERR_HIP_L0_STAMPS=1 only, for tools/hi_stamps.py)
#ifndef SPERR_HIP_L0_STAMPS
#define SPERR_HIP_L0_STAMPS 0
#endif
constexpr int kTabLdsGrids = 288;   // grid descriptors the GPU-wide list kernels keep in LDS
constexpr int kL0W = 8192;
constexpr int kL0Threads = 1024;
// (k_lis_l1 with 512 threads: two workgroups of eight wavefronts a compute unit and 128 registers a thread, where 1024
//  threads at 64 registers spilled 38 of them: 98.3 -> 100.8 GB/s of decompression at 64 chunks, round 5)
constexpr int kL1Threads = 512;
constexpr int kL0Sub = kL0W / 64;
constexpr int kL0Stage = kL0W / 2;   // list entries of a block staged in LDS (hop64's 32 KB)
constexpr uint32_t kL0None = 0xffffffffu;
constexpr size_t kL0Smem = (size_t)(kL0W / 64 + 4) * 8 + (size_t)kL0W * (1 + 4 + 4);

// chain summary: tokens << 21 | significant tokens << 14 | position reached (block-relative)
__global__ void __launch_bounds__(kL0Threads) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_lis_l0(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  const uint32_t L = (uint32_t)b.l0Level;
  const uint32_t cur = s.cur, nx = cur ^ 1u;
  const uint32_t n = s.listLen[cur][L];
  if (n == 0)
    return;   // (k_lis_hi finds the list empty as well)
  extern __shared__ __attribute__((aligned(16))) char l0_smem[];
  uint64_t* wbits = reinterpret_cast<uint64_t*>(l0_smem);
  const uint32_t* w32 = reinterpret_cast<const uint32_t*>(l0_smem);
  uint32_t* hop64 = reinterpret_cast<uint32_t*>(l0_smem + (size_t)(kL0W / 64 + 4) * 8);
  uint32_t* hopW = hop64 + kL0W;     // later: the marks of the tokens on the chain
  uint8_t* U = reinterpret_cast<uint8_t*>(hopW + kL0W);
  const uint64_t* stg64 = reinterpret_cast<const uint64_t*>(hop64);   // (later: the block's list entries)
  __shared__ uint32_t memoX[32], memoC[32], memoS[32];
  __shared__ uint32_t entR[kL0W / 1024], entK[kL0W / 1024], entS[kL0W / 1024];
  __shared__ uint32_t blkE[kL0Sub], blkK[kL0Sub], blkS[kL0Sub];
  __shared__ uint32_t sh_ticket, sh_e, sh_rank, sh_sig, sh_last, sh_stop, sh_endpos, sh_endsig;

  const int tid = threadIdx.x;
  const uint32_t lane = (uint32_t)tid & 63u, wave = (uint32_t)tid >> 6;
  const Tree& t = b.tree;
  const uint64_t phase0 = s.lipStart + s.lipBits;
  const uint64_t* words = b.stream + c * b.streamStride;
  const uint64_t nwordsAvail = (s.avail + 63) / 64;
  const uint64_t* list = b.lis[cur] + c * b.lisStride + b.levelOff[L];
  uint64_t* keep = b.lis[nx] + c * b.lisStride + b.levelOff[L];
  uint64_t* leafEv = b.leafEv + c * b.leafStride;
  unsigned long long* flags = b.l0Flags + c * b.l0FlagStride;
  unsigned long long* tabs = b.l0Tab ? b.l0Tab + c * b.l0FlagStride * 17 : nullptr;   // the blocks' published memo tables
  __shared__ uint32_t sh_lbT[64][17];   // X << 24 | tokens << 10 | significant tokens
  const unsigned long long tag = (unsigned long long)(p + 1) << 56;

  for (;;) {
    if (tid == 0) {
      const bool over = __hip_atomic_load(&s.l0PlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                        p + 1;
      sh_ticket = over ? kL0None : atomicAdd(&s.l0Ticket, 1u);
    }
    __syncthreads();
    const uint32_t i = sh_ticket;
    if (i == kL0None || (size_t)i + 1 >= b.l0FlagStride)
      break;
    const bool l0stamps = SPERR_HIP_L0_STAMPS && b.lisStamps != nullptr && tid == 0 && c == 0;
    uint64_t l0t[6] = {l0stamps ? __builtin_readcyclecounter() : 0, 0, 0, 0, 0, 0};
    const uint64_t a = phase0 + (uint64_t)i * kL0W;
    const uint64_t w0 = a >> 6;
    const uint32_t q0 = (uint32_t)(a & 63);
    for (uint32_t k = tid; k < (uint32_t)(kL0W / 64 + 4); k += kL0Threads)
      wbits[k] = w0 + k < nwordsAvail ? words[w0 + k] : 0ull;
    __syncthreads();
    auto bit_at = [&](uint32_t r) -> uint32_t {
      const uint32_t q = r + q0;
      return (w32[q >> 5] >> (q & 31)) & 1u;
    };
    auto bits32 = [&](uint32_t r) -> uint32_t {
      const uint32_t q = r + q0, sh = q & 31;
      const uint32_t lo = w32[q >> 5], hi = w32[(q >> 5) + 1];
      return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    };
    // ---- token length at every position
    for (uint32_t r = tid; r < (uint32_t)kL0W; r += kL0Threads) {
      uint32_t len = 1;
      if (bit_at(r)) {
        const uint32_t v = bits32(r + 1);
        uint32_t y = 0, found = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) {
          const uint32_t bit = (v >> y) & 1u;
          found |= bit;
          y += 1u + bit;
        }
        const uint32_t bit = found ? (v >> y) & 1u : 1u;
        len = 1u + y + found + bit;
      }
      U[r] = (uint8_t)len;
    }
    __syncthreads();
    // ---- chains inside 64-position sub-blocks (lane = position)
    for (uint32_t sb = wave; sb < (uint32_t)kL0Sub; sb += kL0Threads / 64) {
      const uint32_t r = sb * 64 + lane, hEnd = (sb + 1) * 64;
      uint32_t v = (1u << 21) | (bit_at(r) << 14) | (r + U[r]);
      bool inb = (v & 0x3fffu) < hEnd;
      for (int it = 0; it < 6 && __any(inb); it++) {
        const uint32_t o = __shfl(v, (v & 0x3fffu) & 63u, 64);
        if (inb) {
          v = (v & ~0x3fffu) + o;
          inb = (v & 0x3fffu) < hEnd;
        }
      }
      hop64[r] = v;
      hopW[r] = v;
    }
    __syncthreads();
    // ---- widen a copy to 1024-position blocks (in place: any version read is a valid summary)
    for (uint32_t wide = 128; wide <= 1024; wide <<= 1) {
      for (uint32_t r = tid; r < (uint32_t)kL0W; r += kL0Threads) {
        const uint32_t v = hopW[r], e = v & 0x3fffu;
        if (e < (uint32_t)kL0W && e / wide == r / wide)
          hopW[r] = (v & ~0x3fffu) + hopW[e];
      }
      __syncthreads();
    }
    // ---- the whole block, for each of the 17 offsets a chain can enter at
    if (tid < 17) {
      uint32_t r = tid, cnt = 0, sg = 0;
      while (r < (uint32_t)kL0W) {
        const uint32_t v = hopW[r];
        cnt += v >> 21;
        sg += (v >> 14) & 0x7fu;
        r = v & 0x3fffu;
      }
      memoX[tid] = r - kL0W;
      memoC[tid] = cnt;
      memoS[tid] = sg;
    }
    __syncthreads();
    // ---- look back, publish.  The state a chain enters a block with is (offset, entries so far, significant so
    //      far); what a block does to it is its memo table: a function of the 17 offsets.  Until round 5 a block
    //      waited for its predecessor's state, applied its table and published -- a hand-over through L2 per block,
    //      about a microsecond each, the list's whole length in series (what bounded a batch of a few chunks).  Now
    //      a block PUBLISHES ITS TABLE as soon as it has it (tagged entries, no fence), and a block that looks back
    //      takes the nearest state that is out (64 blocks back at most: as many workgroups as a chunk has) and
    //      applies the tables of the blocks in between itself -- the decoupled look-back of a scan whose carry is a
    //      function, not a sum.  Blocks that wait resolve together instead of one after the other.
    if (l0stamps)
      l0t[1] = __builtin_readcyclecounter();
    if (b.l0Tab && tid < 17)
      __hip_atomic_store(tabs + (size_t)i * 17 + tid,
                         tag | ((unsigned long long)memoX[tid] << 50) | ((unsigned long long)memoC[tid] << 25) |
                             (unsigned long long)memoS[tid],
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
      uint32_t e = 0, rank = 0, sg = 0, stop = 0;
      if (i > 0) {
        uint32_t spins = 0;
        uint64_t spinT0 = 0;
        const int idx = (int)i - 1 - (int)lane;   // lane j looks at block i - 1 - j (-1: before the first block)
        for (;;) {
          unsigned long long f = 0;
          if (idx >= 0 && (lane == 0 || b.l0Tab))
            f = __hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool out = (idx >= 0 && (f >> 56) == (unsigned long long)(p + 1)) || (idx == -1 && b.l0Tab);
          const uint64_t om = __ballot(out);
          bool done = false;
          if (om) {
            const uint32_t j0 = (uint32_t)__ffsll((long long)om) - 1u;   // the nearest state that is out
            const unsigned long long f0 = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)f, (int)j0) |
                                          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(f >> 32), (int)j0) << 32);
            const bool virt = (int)i - 1 - (int)j0 < 0;
            if (!virt && ((f0 >> 55) & 1ull)) {
              stop = 1;   // the list ended in that block
              done = true;
            }
            else {
              e = virt ? 0u : (uint32_t)(f0 >> 50) & 31u;
              rank = virt ? 0u : (uint32_t)(f0 >> 25) & 0x1ffffffu;
              sg = virt ? 0u : (uint32_t)f0 & 0x1ffffffu;
              // the tables of the blocks in between: lane j < j0 loads block i - 1 - j's
              bool have = true;
              if (lane < j0) {
#pragma unroll
                for (int k = 0; k < 17; k++) {
                  const unsigned long long t = __hip_atomic_load(tabs + (size_t)idx * 17 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  have = have && (t >> 56) == (unsigned long long)(p + 1);
                  sh_lbT[lane][k] = ((uint32_t)(t >> 50) & 31u) << 24 | ((uint32_t)(t >> 25) & 0x3fffu) << 10 | ((uint32_t)t & 0x3ffu);
                }
              }
              if (__ballot(lane < j0 && !have) == 0ull) {
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
                for (int j = (int)j0 - 1; j >= 0; j--) {   // (uniform: every lane follows the chain)
                  const uint32_t t = sh_lbT[j][e];
                  const uint32_t cn = (t >> 10) & 0x3fffu;
                  if (rank + cn >= n) {   // the list ended in a block before this one (it says so itself)
                    stop = 1;
                    break;
                  }
                  rank += cn;
                  sg += t & 0x3ffu;
                  e = t >> 24;
                }
                done = true;
              }
            }
          }
          if (done)
            break;
          // (the end-of-pass marker is looked at now and then: the poll stays one round of loads long)
          if ((++spins & 15u) == 0 &&
              __hip_atomic_load(&s.l0PlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p + 1) {
            stop = 1;
            break;
          }
          if (spin_expired(spins, spinT0)) {   // (a minute of wall time: the device has stopped making progress)
            if (lane == 0) {
              s.error = kErrLookBackTimeout;
              __hip_atomic_store(&s.l0PlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            stop = 1;
            break;
          }
        }
      }
      if (tid == 0) {
      uint32_t last = 0;
      if (!stop) {
        if (rank + memoC[e] >= n) {   // the list ends inside this block
          last = 1;
          __hip_atomic_store(flags + i, tag | (1ull << 55), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&s.l0PlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else
          __hip_atomic_store(flags + i,
                             tag | ((unsigned long long)memoX[e] << 50) |
                                 ((unsigned long long)(rank + memoC[e]) << 25) |
                                 (unsigned long long)(sg + memoS[e]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sh_e = e;
      sh_rank = rank;
      sh_sig = sg;
      sh_last = last;
      sh_stop = stop;
      sh_endpos = 0;
      sh_endsig = 0;
      }
    }
    for (uint32_t k = tid; k < (uint32_t)kL0Sub; k += kL0Threads)
      blkE[k] = kL0None;
    if (tid < kL0W / 1024)
      entR[tid] = kL0None;
    __syncthreads();
    if (sh_stop)
      break;
    if (l0stamps)
      l0t[2] = __builtin_readcyclecounter();
    // ---- where the chain enters each 1024-block, then each sub-block
    if (tid == 0) {
      uint32_t r = sh_e, rk = 0, sg = 0;
      while (r < (uint32_t)kL0W) {
        entR[r >> 10] = r;
        entK[r >> 10] = rk;
        entS[r >> 10] = sg;
        const uint32_t v = hopW[r];
        rk += v >> 21;
        sg += (v >> 14) & 0x7fu;
        r = v & 0x3fffu;
      }
    }
    __syncthreads();
    if (tid < kL0W / 1024 && entR[tid] != kL0None) {
      uint32_t r = entR[tid], rk = entK[tid], sg = entS[tid];
      const uint32_t end = ((uint32_t)tid + 1) * 1024;
      while (r < end) {
        blkE[r >> 6] = r;
        blkK[r >> 6] = rk;
        blkS[r >> 6] = sg;
        const uint32_t v = hop64[r];
        rk += v >> 21;
        sg += (v >> 14) & 0x7fu;
        r = v & 0x3fffu;
      }
    }
    __syncthreads();
    if (l0stamps)
      l0t[3] = __builtin_readcyclecounter();
    const uint32_t rank0 = sh_rank, sig0 = sh_sig;
    const uint32_t nloc = n - rank0;   // entries the list still holds at the start of the block
    // ---- the block's list entries are one stretch of the list: they travel from HBM straight into LDS (hop64 is
    //      free now: 4096 entries; global_load_lds, no register holds them) while the marks are worked out.  Fetched
    //      where each token is handled they were eight loads in series per thread, two thousand cycles each (the
    //      registers to have them in flight together this kernel does not have: profiles/r5_l01_phases.txt).
    const uint32_t nstg = min(min(memoC[sh_e], nloc), (uint32_t)kL0Stage);
    {
      const uint32_t* list32 = reinterpret_cast<const uint32_t*>(list + rank0);
      for (uint32_t d0 = wave * 64u; d0 < 2u * nstg; d0 += (kL0Threads / 64) * 64u)
        __builtin_amdgcn_global_load_lds(list32 + min(d0 + lane, 2u * nstg - 1u), hop64 + d0, 4, 0, 0);
    }
    // ---- marks: (1 + entries before the token) | significant entries before it << 16, block-local
    for (uint32_t r = tid; r < (uint32_t)kL0W; r += kL0Threads)
      hopW[r] = 0;
    __syncthreads();
    if (tid < kL0Sub && blkE[tid] != kL0None) {
      uint32_t r = blkE[tid], rk = blkK[tid], sg = blkS[tid];
      const uint32_t end = ((uint32_t)tid + 1) * 64;
      bool did = false;
      while (r < end && rk < nloc) {
        hopW[r] = (rk + 1u) | (sg << 16);
        rk++;
        sg += bit_at(r);
        r += U[r];
        did = true;
      }
      if (did && rk == nloc) {   // this thread decoded the list's last entry
        sh_endpos = r;
        sh_endsig = sg;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the staged list entries have landed)
    __syncthreads();
    if (l0stamps)
      l0t[4] = __builtin_readcyclecounter();
    // ---- every thread handles the tokens that start at its positions (loading the list entries of several positions
    //      before any is used was measured: the registers it takes spill at this kernel's 64, 98.6 -> 97.2 GB/s)
    for (uint32_t r = tid; r < (uint32_t)kL0W; r += kL0Threads) {
      const uint32_t mk = hopW[r];
      if (mk == 0)
        continue;
      const uint32_t q = rank0 + (mk & 0xffffu) - 1u, sb = sig0 + (mk >> 16);
      const uint32_t li = (mk & 0xffffu) - 1u;
      const uint64_t ident = li < nstg ? stg64[li] : list[q];
      if (!bit_at(r)) {
        keep[q - sb] = ident;
        continue;
      }
      const uint32_t v = bits32(r + 1);
      uint32_t yy = 0, found = 0, sigm = 0, negm = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) {
        const uint32_t bit = (v >> yy) & 1u, sgn = (v >> (yy + 1)) & 1u;
        sigm |= bit << k;
        negm |= (bit & (sgn ^ 1u)) << k;
        found |= bit;
        yy += 1u + bit;
      }
      const uint32_t bit = found ? (v >> yy) & 1u : 1u;
      const uint32_t sgn = (v >> (yy + found)) & 1u;
      sigm |= bit << 7;
      negm |= (bit & (sgn ^ 1u)) << 7;
      const Node nd = unpack_node(ident);
      const Grid g = t.grids[nd.grid];
      const uint32_t fid = g.nodeOff + ((((uint32_t)nd.i[2] << g.e[1]) + nd.i[1]) << g.e[0]) + nd.i[0];
      if (sb < b.leafCap)
        leafEv[sb] = (uint64_t)fid | ((uint64_t)sigm << 32) | ((uint64_t)negm << 40);
    }
    if (l0stamps) {
      l0t[5] = __builtin_readcyclecounter();
      atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 48), 1ull);
      for (int k = 0; k < 5; k++)
        atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 49 + k), l0t[k + 1] - l0t[k]);
    }
    if (sh_last) {
      if (tid == 0) {
        s.l0End = a + sh_endpos;
        s.l0Sig = sig0 + sh_endsig;
        s.leafCount = sig0 + sh_endsig;
        s.listLen[nx][L] = n - (sig0 + sh_endsig);
      }
      break;
    }
    __syncthreads();   // LDS is reused by the next block
  }
}

// ------------------------------------------------------------------------------------------
// LIS phase, the next list: 4x4x4 sets (class 1) whose children are the 2x2x2 leaf sets.  Same
// scheme as k_lis_l0 with one more level inside a token: an entry is '0', or '1' followed by its
// eight children, each '0' (the child joins the list of the smallest sets: a birth, recorded
// with its stream position exactly like k_lis_hi does) or '1' + eight pixels (a leaf event).
// A token takes at most 1 + 7 * 17 + 17 = 137 bits, so the tables of a block cover kL1Ahead
// positions more than the block itself.  Births and leaf events get their slots per block: each
// token reserves block-local slots while it is counted, the block reserves its range with one
// atomic per counter, and a second sweep writes.
// ------------------------------------------------------------------------------------------
constexpr int kL1W = 4096;
constexpr int kL1Ahead = 192;
constexpr int kL1P = kL1W + kL1Ahead;           // positions with class-0 tables
constexpr int kL1MaxTok = 137;
constexpr int kL1Per = kL1W / kL1Threads;        // positions a thread looks at in a sweep
constexpr int kL1QCap = kL1W / 2;               // children of a block's significant tokens: eight of each of kL1W / 16 tokens
constexpr size_t kL1Smem = (size_t)(kL1P / 64 + 4) * 8 + (size_t)(kL1W + 4) * 4 + (size_t)kL1W * (4 + 1) +
                           (size_t)kL1P * 2;

// (eight waves per SIMD = two of these workgroups per CU: the phases are barrier- and latency-
//  bound, a second resident workgroup fills the gaps)
__global__ void __launch_bounds__(kL1Threads) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_lis_l1(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  const uint32_t L = (uint32_t)b.l1Level, L0 = (uint32_t)b.l0Level;
  const uint32_t cur = s.cur, nx = cur ^ 1u;
  const uint32_t n = s.listLen[cur][L];
  const bool l0done = s.l0PlaneP1 == p + 1;
  if (n == 0 || (!l0done && s.listLen[cur][L0] != 0))
    return;   // (k_lis_hi takes the list)
  extern __shared__ __attribute__((aligned(16))) char l1_smem[];
  uint64_t* wbits = reinterpret_cast<uint64_t*>(l1_smem);
  const uint32_t* w32 = reinterpret_cast<const uint32_t*>(l1_smem);
  uint32_t* hop64 = reinterpret_cast<uint32_t*>(l1_smem + (size_t)(kL1P / 64 + 4) * 8);
  uint32_t* hopW = hop64 + kL1W + 4;     // later: the marks of the tokens on the chain
  uint8_t* U1 = reinterpret_cast<uint8_t*>(hopW + kL1W);   // token length at every position
  uint8_t* U0 = U1 + kL1W;           // coded class-0 item: 1, or 1 + T0 of the next position
  uint8_t* T0 = U0 + kL1P;           // split of a class-0 set that starts here
  __shared__ uint32_t memoX[kL1MaxTok + 1], memoC[kL1MaxTok + 1], memoS[kL1MaxTok + 1];
  __shared__ uint32_t entR[kL1W / 1024], entK[kL1W / 1024], entS[kL1W / 1024];
  __shared__ uint32_t blkE[kL1W / 64], blkK[kL1W / 64], blkS[kL1W / 64];
  __shared__ uint32_t sh_ticket, sh_e, sh_rank, sh_sig, sh_last, sh_stop, sh_endpos, sh_endsig;
  __shared__ uint32_t sh_nb, sh_nl, sh_baseB, sh_baseL, sh_ntok;
  __shared__ uint16_t tokQ[kL1W / 16];   // the significant tokens of a block: 16 bits each and more
  __shared__ Grid sh_grids[kTabLdsGrids];   // (the launcher checks that the tree's grids fit)

  const int tid = threadIdx.x;
  const uint32_t lane = (uint32_t)tid & 63u, wave = (uint32_t)tid >> 6;
  const Tree& t = b.tree;
  for (uint32_t k = tid; k < t.ngrids; k += kL1Threads)
    sh_grids[k] = t.grids[k];
  const uint64_t phase0 = s.lipStart + s.lipBits;
  const uint64_t start0 = l0done ? s.l0End : phase0;
  const uint64_t* words = b.stream + c * b.streamStride;
  const uint64_t nwordsAvail = (s.avail + 63) / 64;
  const uint64_t* list = b.lis[cur] + c * b.lisStride + b.levelOff[L];
  uint64_t* keep = b.lis[nx] + c * b.lisStride + b.levelOff[L];
  uint64_t* leafEv = b.leafEv + c * b.leafStride;
  uint64_t* bornPacked = b.bornPacked + c * b.bornPitch;
  uint64_t* bornPosLev = b.bornPosLev + c * b.bornPitch;
  unsigned long long* flags = b.l1Flags + c * b.l0FlagStride;
  const unsigned long long tag = (unsigned long long)(p + 1) << 56;
  const uint64_t maskBits = (uint64_t)b.maskWords * 64;
  const uint32_t bornLev = b.levelClass[L].lev[0];
  const uint32_t bornSlot = b.levelSlot[bornLev];
  uint64_t* bornMask = b.mask + c * b.maskStride + (size_t)bornSlot * b.maskWords;

  for (;;) {
    if (tid == 0) {
      const bool over = __hip_atomic_load(&s.l1PlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                        p + 1;
      sh_ticket = over ? kL0None : atomicAdd(&s.l1Ticket, 1u);
      sh_nb = 0;
      sh_nl = 0;
      sh_ntok = 0;
    }
    __syncthreads();
    const uint32_t i = sh_ticket;
    if (i == kL0None || (size_t)i + 1 >= b.l0FlagStride)
      break;
    const bool l1stamps = b.lisStamps != nullptr && tid == 0 && c == 0;
    uint64_t l1t0 = l1stamps ? __builtin_readcyclecounter() : 0, l1t1 = 0, l1t2 = 0, l1t3 = 0, l1t4 = 0, l1t5 = 0;
    const uint64_t a = start0 + (uint64_t)i * kL1W;
    const uint64_t w0 = a >> 6;
    const uint32_t q0 = (uint32_t)(a & 63);
    for (uint32_t k = tid; k < (uint32_t)(kL1P / 64 + 4); k += kL1Threads)
      wbits[k] = w0 + k < nwordsAvail ? words[w0 + k] : 0ull;
    __syncthreads();
    auto bit_at = [&](uint32_t r) -> uint32_t {
      const uint32_t q = r + q0;
      return (w32[q >> 5] >> (q & 31)) & 1u;
    };
    auto bits32 = [&](uint32_t r) -> uint32_t {
      const uint32_t q = r + q0, sh = q & 31;
      const uint32_t lo = w32[q >> 5], hi = w32[(q >> 5) + 1];
      return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    };
    // ---- class 0: split length, then the coded item
    for (uint32_t r = tid; r < (uint32_t)kL1P; r += kL1Threads) {
      const uint32_t v = bits32(r);
      uint32_t y = 0, found = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) {
        const uint32_t bit = (v >> y) & 1u;
        found |= bit;
        y += 1u + bit;
      }
      const uint32_t bit = found ? (v >> y) & 1u : 1u;
      T0[r] = (uint8_t)(y + found + bit);
    }
    __syncthreads();
    for (uint32_t r = tid; r < (uint32_t)kL1P; r += kL1Threads)
      U0[r] = (uint8_t)((bit_at(r) && r + 1 < (uint32_t)kL1P) ? 1u + T0[r + 1] : 1u);
    __syncthreads();
    // ---- class 1: the token at every position of the block
    for (uint32_t r = tid; r < (uint32_t)kL1W; r += kL1Threads) {
      uint32_t len = 1;
      if (bit_at(r)) {
        uint32_t y = r + 1, found = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) {
          const uint32_t u = U0[y];
          found |= u - 1u;
          y += u;
        }
        y += found ? U0[y] : T0[y];
        len = y - r;
      }
      U1[r] = (uint8_t)len;
    }
    __syncthreads();
    // ---- chains inside 64-position sub-blocks (lane = position)
    for (uint32_t sb = wave; sb < (uint32_t)(kL1W / 64); sb += kL1Threads / 64) {
      const uint32_t r = sb * 64 + lane, hEnd = (sb + 1) * 64;
      uint32_t v = (1u << 21) | (bit_at(r) << 14) | (r + U1[r]);
      bool inb = (v & 0x3fffu) < hEnd;
      for (int it = 0; it < 6 && __any(inb); it++) {
        const uint32_t o = __shfl(v, (v & 0x3fffu) & 63u, 64);
        if (inb) {
          v = (v & ~0x3fffu) + o;
          inb = (v & 0x3fffu) < hEnd;
        }
      }
      hop64[r] = v;
      hopW[r] = v;
    }
    __syncthreads();
    for (uint32_t wide = 128; wide <= 1024; wide <<= 1) {
      for (uint32_t r = tid; r < (uint32_t)kL1W; r += kL1Threads) {
        const uint32_t v = hopW[r], e = v & 0x3fffu;
        if (e < (uint32_t)kL1W && e / wide == r / wide)
          hopW[r] = (v & ~0x3fffu) + hopW[e];
      }
      __syncthreads();
    }
    // ---- the whole block, for each offset a chain can enter at
    if (tid <= kL1MaxTok) {
      uint32_t r = tid, cnt = 0, sg = 0;
      while (r < (uint32_t)kL1W) {
        const uint32_t v = hopW[r];
        cnt += v >> 21;
        sg += (v >> 14) & 0x7fu;
        r = v & 0x3fffu;
      }
      memoX[tid] = r - kL1W;
      memoC[tid] = cnt;
      memoS[tid] = sg;
    }
    __syncthreads();
    // ---- look back, publish: tag | done << 55 | exit offset << 47 | entries << 23 | significant
    if (l1stamps)
      l1t1 = __builtin_readcyclecounter();
    if (tid == 0) {
      uint32_t e = 0, rank = 0, sg = 0, stop = 0, last = 0;
      if (i > 0) {
        unsigned long long f = 0;
        uint32_t spins = 0;
        uint64_t spinT0 = 0;
        for (;;) {
          f = __hip_atomic_load(flags + (i - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((f >> 56) == (unsigned long long)(p + 1))
            break;
          // (the end-of-pass marker is looked at now and then: the poll stays one load long)
          if ((++spins & 15u) == 0 &&
              __hip_atomic_load(&s.l1PlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p + 1) {
            stop = 1;
            break;
          }
          if (spin_expired(spins, spinT0)) {   // (a minute of wall time: the device has stopped making progress)
            s.error = kErrLookBackTimeout;
            __hip_atomic_store(&s.l1PlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stop = 1;
            break;
          }
        }
        if (!stop) {
          if ((f >> 55) & 1ull)
            stop = 1;
          else {
            e = (uint32_t)(f >> 47) & 0xffu;
            rank = (uint32_t)(f >> 23) & 0xffffffu;
            sg = (uint32_t)f & 0x7fffffu;
          }
        }
      }
      if (!stop) {
        if (rank + memoC[e] >= n) {   // the list ends inside this block
          last = 1;
          __hip_atomic_store(flags + i, tag | (1ull << 55), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&s.l1PlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else
          __hip_atomic_store(flags + i,
                             tag | ((unsigned long long)memoX[e] << 47) |
                                 ((unsigned long long)(rank + memoC[e]) << 23) |
                                 (unsigned long long)(sg + memoS[e]),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sh_e = e;
      sh_rank = rank;
      sh_sig = sg;
      sh_last = last;
      sh_stop = stop;
      sh_endpos = 0;
      sh_endsig = 0;
      if (l1stamps)
        l1t2 = __builtin_readcyclecounter();
    }
    for (uint32_t k = tid; k < (uint32_t)(kL1W / 64); k += kL1Threads)
      blkE[k] = kL0None;
    if (tid < kL1W / 1024)
      entR[tid] = kL0None;
    __syncthreads();
    if (sh_stop)
      break;
    // ---- where the chain enters each 1024-block, then each sub-block
    if (tid == 0) {
      uint32_t r = sh_e, rk = 0, sg = 0;
      while (r < (uint32_t)kL1W) {
        entR[r >> 10] = r;
        entK[r >> 10] = rk;
        entS[r >> 10] = sg;
        const uint32_t v = hopW[r];
        rk += v >> 21;
        sg += (v >> 14) & 0x7fu;
        r = v & 0x3fffu;
      }
    }
    __syncthreads();
    if (tid < kL1W / 1024 && entR[tid] != kL0None) {
      uint32_t r = entR[tid], rk = entK[tid], sg = entS[tid];
      const uint32_t end = ((uint32_t)tid + 1) * 1024;
      while (r < end) {
        blkE[r >> 6] = r;
        blkK[r >> 6] = rk;
        blkS[r >> 6] = sg;
        const uint32_t v = hop64[r];
        rk += v >> 21;
        sg += (v >> 14) & 0x7fu;
        r = v & 0x3fffu;
      }
    }
    __syncthreads();
    if (l1stamps)
      l1t3 = __builtin_readcyclecounter();
    // ---- marks: (1 + entries before the token) | significant entries before it << 16, block-local
    for (uint32_t r = tid; r < (uint32_t)kL1W; r += kL1Threads)
      hopW[r] = 0;
    __syncthreads();
    const uint32_t rank0 = sh_rank, sig0 = sh_sig;
    const uint32_t nloc = n - rank0;
    if (tid < kL1W / 64 && blkE[tid] != kL0None) {
      uint32_t r = blkE[tid], rk = blkK[tid], sg = blkS[tid];
      const uint32_t end = ((uint32_t)tid + 1) * 64;
      bool did = false;
      while (r < end && rk < nloc) {
        hopW[r] = (rk + 1u) | (sg << 16);
        rk++;
        sg += bit_at(r);
        r += U1[r];
        did = true;
      }
      if (did && rk == nloc) {
        sh_endpos = r;
        sh_endsig = sg;
      }
    }
    __syncthreads();
    if (l1stamps)
      l1t4 = __builtin_readcyclecounter();
    // ---- first sweep: insignificant entries stay, significant ones leave their list entry in their own words of
    //      hop64 (a significant token is 16 bits and more) and queue up.  The list entries of a thread's positions are
    //      loaded before any is used.
    {
      uint32_t mk4[kL1Per];
      uint64_t id4[kL1Per];
#pragma unroll
      for (int j = 0; j < kL1Per; j++)
        mk4[j] = hopW[(uint32_t)tid + (uint32_t)j * kL1Threads];
#pragma unroll
      for (int j = 0; j < kL1Per; j++)
        id4[j] = mk4[j] ? list[rank0 + (mk4[j] & 0xffffu) - 1u] : 0ull;
#pragma unroll
      for (int j = 0; j < kL1Per; j++) {
        const uint32_t r = (uint32_t)tid + (uint32_t)j * kL1Threads;
        const uint32_t mk = mk4[j];
        if (mk == 0)
          continue;
        const uint32_t q = rank0 + (mk & 0xffffu) - 1u, sb = sig0 + (mk >> 16);
        const uint64_t ident = id4[j];
        if (!bit_at(r)) {
          keep[q - sb] = ident;
          continue;
        }
        hop64[r + 1] = (uint32_t)ident;
        hop64[r + 2] = (uint32_t)(ident >> 32);
        tokQ[atomicAdd(&sh_ntok, 1u)] = (uint16_t)r;
      }
    }
    __syncthreads();
    // ---- the significant tokens, one per thread (a token every 40 positions or so: found where they lie, six lanes of
    //      a wavefront walked eight children each while the others waited -- 20 of a block's 67 thousand cycles):
    //      births and leaf events are counted, take block-local slots, and every child is queued under its slot --
    //      births from the front of Q (the marks' array: they are dead now), leaf events from its back:
    //        position of the token | child << 12 | (child's position - token's) << 15 | coded << 23
    uint32_t* Q = hopW;
    if ((uint32_t)tid < sh_ntok) {
      const uint32_t r = tokQ[tid];
      uint32_t y = r + 1, found = 0, nb = 0, nl = 0, kinds = 0, offs[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const bool coded = found || k != 7;
        const uint32_t u = coded ? U0[y] : T0[y];
        offs[k] = (y - r) | ((uint32_t)coded << 8);
        if (coded && u == 1) {
          const uint64_t rel = a + y - phase0;
          if (bornSlot != 0xff && rel < maskBits) {
            kinds |= 0x100u << k;
            nb++;
          }
        }
        else {
          found = 1;
          nl++;
          kinds |= 1u << k;
        }
        y += u;
      }
      uint32_t slotB = nb ? atomicAdd(&sh_nb, nb) : 0u;
      uint32_t slotL = atomicAdd(&sh_nl, nl);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t desc = r | ((uint32_t)k << 12) | (offs[k] << 15);
        if ((kinds >> (8 + k)) & 1u)
          Q[slotB++] = desc;
        else if ((kinds >> k) & 1u)
          Q[kL1QCap - 1 - slotL++] = desc;
      }
    }
    __syncthreads();
    if (tid == 0)     // (two threads: the two returning atomics are in flight together)
      sh_baseB = sh_nb ? atomicAdd(&s.bornCount, sh_nb) : 0u;
    if (tid == 64)
      sh_baseL = sh_nl ? atomicAdd(&s.leafCount, sh_nl) : 0u;
    __syncthreads();
    if (l1stamps)
      l1t5 = __builtin_readcyclecounter();
    // ---- second sweep, a child per thread: the births ...
    for (uint32_t i2 = tid; i2 < sh_nb; i2 += kL1Threads) {
      const uint32_t desc = Q[i2];
      const uint32_t r = desc & 0xfffu, k = (desc >> 12) & 7u, y = r + ((desc >> 15) & 0xffu);
      const Node nd = unpack_node((uint64_t)hop64[r + 1] | ((uint64_t)hop64[r + 2] << 32));
      const uint32_t cx = 2u * nd.i[0] + (k & 1u), cy = 2u * nd.i[1] + ((k >> 1) & 1u), cz = 2u * nd.i[2] + (k >> 2);
      const uint64_t rel = a + y - phase0;
      const uint32_t slotB = sh_baseB + i2;
      if (slotB < b.bornStride) {   // stays insignificant: joins the list of the smallest sets
        bornPacked[slotB] = ((uint64_t)(nd.grid + 1) << 48) | ((uint64_t)cz << 32) | ((uint64_t)cy << 16) | (uint64_t)cx;
        bornPosLev[slotB] = ((uint64_t)bornLev << 48) | rel;
        atomic_or64(bornMask + (rel >> 6), 1ull << (rel & 63));
      }
    }
    // ---- ... and the leaf events: a child that splits into its pixels
    for (uint32_t i2 = tid; i2 < sh_nl; i2 += kL1Threads) {
      const uint32_t desc = Q[kL1QCap - 1 - i2];
      const uint32_t r = desc & 0xfffu, k = (desc >> 12) & 7u, y = r + ((desc >> 15) & 0xffu);
      const bool coded = (desc >> 23) & 1u;
      const Node nd = unpack_node((uint64_t)hop64[r + 1] | ((uint64_t)hop64[r + 2] << 32));
      const uint32_t cx = 2u * nd.i[0] + (k & 1u), cy = 2u * nd.i[1] + ((k >> 1) & 1u), cz = 2u * nd.i[2] + (k >> 2);
      const Grid g1 = sh_grids[nd.grid + 1];   // the grid of the children
      const uint32_t slotL = sh_baseL + i2;
      const uint32_t v = bits32(coded ? y + 1 : y);
      uint32_t yy = 0, fnd = 0, sigm = 0, negm = 0;
#pragma unroll
      for (int j = 0; j < 7; j++) {
        const uint32_t bit = (v >> yy) & 1u, sgn = (v >> (yy + 1)) & 1u;
        sigm |= bit << j;
        negm |= (bit & (sgn ^ 1u)) << j;
        fnd |= bit;
        yy += 1u + bit;
      }
      const uint32_t bit = fnd ? (v >> yy) & 1u : 1u;
      const uint32_t sgn = (v >> (yy + fnd)) & 1u;
      sigm |= bit << 7;
      negm |= (bit & (sgn ^ 1u)) << 7;
      const uint32_t fid = g1.nodeOff + (((cz << g1.e[1]) + cy) << g1.e[0]) + cx;
      if (slotL < b.leafCap)
        leafEv[slotL] = (uint64_t)fid | ((uint64_t)sigm << 32) | ((uint64_t)negm << 40);
    }
    if (l1stamps) {
      const uint64_t now_ = __builtin_readcyclecounter();
      atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 56), 1ull);
      atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 57), l1t1 - l1t0);
      atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 58), l1t2 - l1t1);
      atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 59), now_ - l1t2);
      atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 60), l1t3 - l1t2);
      atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 61), l1t4 - l1t3);
      atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 62), l1t5 - l1t4);
      atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + 63), now_ - l1t5);
    }
    if (sh_last) {
      if (tid == 0) {
        s.l1End = a + sh_endpos;
        s.listLen[nx][L] = n - (sig0 + sh_endsig);
      }
      break;
    }
    __syncthreads();   // LDS is reused by the next block
  }
}

// ------------------------------------------------------------------------------------------
// LIS phase, the third list: 8x8x8 sets (class 2) whose children are the 4x4x4 sets (round 6).  The scheme of
// k_lis_l0 / k_lis_l1 with one more level inside a token: an entry is '0', or '1' followed by its eight
// children, each '0' (a birth into the 4x4x4 sets' list) or '1' + ITS eight children, each '0' (a birth into
// the list of the smallest sets) or '1' + eight pixels (a leaf event); the last child of a set none of whose
// siblings was significant carries no test bit (src/SPECK3D_INT.cpp:140-212).  A token takes at most
// 1 + 7 * 137 + 137 = 1097 bits, so a block's class-1 tables cover 1024 positions more than the block and its
// class-0 tables 128 more than those; a chain can enter a block at 1098 offsets, which is the size of its memo
// table.  Until round 6 this list was k_lis_hi's: in the heavy planes most of that kernel's bits (2.5 Mbit per
// 256^3 chunk in plane 14 of the bench volume), decoded through its general machinery -- regions handed from
// workgroup to workgroup over a chain of 15 K cycles each, class tables for any chain of classes, breadth-first
// expansion through global queues.  Here a block's hand-over is one look-back word, as in k_lis_l1.
//   sweep 1   the tokens on the chain: insignificant entries stay, significant ones leave their list entry in
//             their own words of hop64 (a significant token is 24 bits and more) and queue up;
//   sweep T   a significant token per thread: its children -- births counted, significant ones queued;
//   sweep C   a significant child per thread: its children -- births and leaf events counted;
//   the block reserves its birth and leaf-event slots with one atomic each, and sweeps T and C run again to write
//   (a thread's records: at most eight).
// ------------------------------------------------------------------------------------------
constexpr int kL2Threads = 512;
constexpr int kL2W = 4096;
constexpr int kL2MaxTok = 1097;
constexpr int kL2P1 = kL2W + 1024;              // positions with class-1 tables
constexpr int kL2P0 = kL2P1 + 128;              // positions with class-0 tables
constexpr int kL2Per = kL2W / kL2Threads;
constexpr int kL2TokCap = 256;                  // significant tokens that START in a block: 24 bits each and more
constexpr int kL2ChildCap = 512;                // their significant children: 16 bits each and more inside 4096 + 1097
constexpr size_t kL2Smem = (size_t)(kL2P0 / 64 + 4) * 8 + (size_t)(kL2W + 4) * 4 + (size_t)kL2W * (4 + 2) +
                           (size_t)kL2P1 * 2 + (size_t)kL2P0 * 2;

__global__ void __launch_bounds__(kL2Threads) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_lis_l2(DecBuffers b, int p, uint32_t minEntries)
{
  const uint32_t c = blockIdx.y;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  const uint32_t L = (uint32_t)b.l2Level, L1 = (uint32_t)b.l1Level, L0 = (uint32_t)b.l0Level;
  const uint32_t cur = s.cur, nx = cur ^ 1u;
  const uint32_t n = s.listLen[cur][L];
  const bool l0done = s.l0PlaneP1 == p + 1, l1done = s.l1PlaneP1 == p + 1;
  // (a short list stays with k_lis_hi: one block's tables -- 5 K positions, three classes -- are 35 us whatever it holds)
  if (n < minEntries || (!l0done && s.listLen[cur][L0] != 0) || (!l1done && s.listLen[cur][L1] != 0))
    return;   // (k_lis_hi takes the list)
  extern __shared__ __attribute__((aligned(16))) char l2_smem[];
  uint64_t* wbits = reinterpret_cast<uint64_t*>(l2_smem);
  const uint32_t* w32 = reinterpret_cast<const uint32_t*>(l2_smem);
  uint32_t* hop64 = reinterpret_cast<uint32_t*>(l2_smem + (size_t)(kL2P0 / 64 + 4) * 8);
  uint32_t* hopW = hop64 + kL2W + 4;     // later: the marks of the tokens on the chain, then the queues
  uint16_t* U2 = reinterpret_cast<uint16_t*>(hopW + kL2W);   // token length at every position of the block
  uint8_t* U1 = reinterpret_cast<uint8_t*>(U2 + kL2W);       // coded class-1 item: 1, or 1 + T1 of the next position
  uint8_t* T1 = U1 + kL2P1;          // split of a class-1 set that starts here
  uint8_t* U0 = T1 + kL2P1;          // the same for class 0
  uint8_t* T0 = U0 + kL2P0;
  __shared__ uint32_t memo[kL2MaxTok + 1];   // exit offset << 21 | entries << 8 | significant entries, per entry offset
  __shared__ uint32_t entR[kL2W / 1024], entK[kL2W / 1024], entS[kL2W / 1024];
  __shared__ uint32_t blkE[kL2W / 64], blkK[kL2W / 64], blkS[kL2W / 64];
  __shared__ uint32_t sh_ticket, sh_e, sh_rank, sh_sig, sh_last, sh_stop, sh_endpos, sh_endsig;
  __shared__ uint32_t sh_nb, sh_nl, sh_nc, sh_baseB, sh_baseL, sh_ntok;
  __shared__ uint16_t tokQ[kL2TokCap];
  __shared__ uint32_t tokSlot[kL2TokCap];    // first birth slot of a token's children (block-local)
  __shared__ Grid sh_grids[kTabLdsGrids];    // (the launcher checks that the tree's grids fit)

  const int tid = threadIdx.x;
  const uint32_t lane = (uint32_t)tid & 63u, wave = (uint32_t)tid >> 6;
  const Tree& t = b.tree;
  for (uint32_t k = tid; k < t.ngrids; k += kL2Threads)
    sh_grids[k] = t.grids[k];
  const uint64_t phase0 = s.lipStart + s.lipBits;
  const uint64_t start0 = l1done ? s.l1End : l0done ? s.l0End : phase0;
  const uint64_t* words = b.stream + c * b.streamStride;
  const uint64_t nwordsAvail = (s.avail + 63) / 64;
  const uint64_t* list = b.lis[cur] + c * b.lisStride + b.levelOff[L];
  uint64_t* keep = b.lis[nx] + c * b.lisStride + b.levelOff[L];
  uint64_t* leafEv = b.leafEv + c * b.leafStride;
  uint64_t* bornPacked = b.bornPacked + c * b.bornPitch;
  uint64_t* bornPosLev = b.bornPosLev + c * b.bornPitch;
  unsigned long long* flags = b.l2Flags + c * b.l0FlagStride;
  const unsigned long long tag = (unsigned long long)(p + 1) << 56;
  const uint64_t maskBits = (uint64_t)b.maskWords * 64;
  const uint32_t lev1 = b.levelClass[L].lev[1], lev0 = b.levelClass[L].lev[0];
  const uint32_t slot1 = b.levelSlot[lev1], slot0 = b.levelSlot[lev0];
  uint64_t* mask1 = b.mask + c * b.maskStride + (size_t)slot1 * b.maskWords;
  uint64_t* mask0 = b.mask + c * b.maskStride + (size_t)slot0 * b.maskWords;

  for (;;) {
    if (tid == 0) {
      const bool over = __hip_atomic_load(&s.l2PlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p + 1;
      sh_ticket = over ? kL0None : atomicAdd(&s.l2Ticket, 1u);
      sh_nb = 0;
      sh_nl = 0;
      sh_nc = 0;
      sh_ntok = 0;
    }
    __syncthreads();
    const uint32_t i = sh_ticket;
    if (i == kL0None || (size_t)i + 1 >= b.l0FlagStride)
      break;
    const uint64_t a = start0 + (uint64_t)i * kL2W;
    const uint64_t w0 = a >> 6;
    const uint32_t q0 = (uint32_t)(a & 63);
    for (uint32_t k = tid; k < (uint32_t)(kL2P0 / 64 + 4); k += kL2Threads)
      wbits[k] = w0 + k < nwordsAvail ? words[w0 + k] : 0ull;
    __syncthreads();
    auto bit_at = [&](uint32_t r) -> uint32_t {
      const uint32_t q = r + q0;
      return (w32[q >> 5] >> (q & 31)) & 1u;
    };
    auto bits32 = [&](uint32_t r) -> uint32_t {
      const uint32_t q = r + q0, sh = q & 31;
      const uint32_t lo = w32[q >> 5], hi = w32[(q >> 5) + 1];
      return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    };
    // ---- class 0: split length, then the coded item
    for (uint32_t r = tid; r < (uint32_t)kL2P0; r += kL2Threads) {
      const uint32_t v = bits32(r);
      uint32_t y = 0, found = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) {
        const uint32_t bit = (v >> y) & 1u;
        found |= bit;
        y += 1u + bit;
      }
      const uint32_t bit = found ? (v >> y) & 1u : 1u;
      T0[r] = (uint8_t)(y + found + bit);
    }
    __syncthreads();
    for (uint32_t r = tid; r < (uint32_t)kL2P0; r += kL2Threads)
      U0[r] = (uint8_t)((bit_at(r) && r + 1 < (uint32_t)kL2P0) ? 1u + T0[r + 1] : 1u);
    __syncthreads();
    // ---- class 1
    for (uint32_t r = tid; r < (uint32_t)kL2P1; r += kL2Threads) {
      uint32_t y = r, found = 0;
#pragma unroll
      for (int k = 0; k < 7; k++) {
        const uint32_t u = U0[y];
        found |= u - 1u;
        y += u;
      }
      y += found ? U0[y] : T0[y];
      T1[r] = (uint8_t)(y - r);
    }
    __syncthreads();
    for (uint32_t r = tid; r < (uint32_t)kL2P1; r += kL2Threads)
      U1[r] = (uint8_t)((bit_at(r) && r + 1 < (uint32_t)kL2P1) ? 1u + T1[r + 1] : 1u);
    __syncthreads();
    // ---- class 2: the token at every position of the block
    for (uint32_t r = tid; r < (uint32_t)kL2W; r += kL2Threads) {
      uint32_t len = 1;
      if (bit_at(r)) {
        uint32_t y = r + 1, found = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) {
          const uint32_t u = U1[y];
          found |= u - 1u;
          y += u;
        }
        y += found ? U1[y] : T1[y];
        len = y - r;
      }
      U2[r] = (uint16_t)len;
    }
    __syncthreads();
    // ---- chains inside 64-position sub-blocks (lane = position)
    for (uint32_t sb = wave; sb < (uint32_t)(kL2W / 64); sb += kL2Threads / 64) {
      const uint32_t r = sb * 64 + lane, hEnd = (sb + 1) * 64;
      uint32_t v = (1u << 21) | (bit_at(r) << 14) | (r + U2[r]);
      bool inb = (v & 0x3fffu) < hEnd;
      for (int it = 0; it < 6 && __any(inb); it++) {
        const uint32_t o = __shfl(v, (v & 0x3fffu) & 63u, 64);
        if (inb) {
          v = (v & ~0x3fffu) + o;
          inb = (v & 0x3fffu) < hEnd;
        }
      }
      hop64[r] = v;
      hopW[r] = v;
    }
    __syncthreads();
    for (uint32_t wide = 128; wide <= 1024; wide <<= 1) {
      for (uint32_t r = tid; r < (uint32_t)kL2W; r += kL2Threads) {
        const uint32_t v = hopW[r], e = v & 0x3fffu;
        if (e < (uint32_t)kL2W && e / wide == r / wide)
          hopW[r] = (v & ~0x3fffu) + hopW[e];
      }
      __syncthreads();
    }
    // ---- the whole block, for each offset a chain can enter at
    for (uint32_t e0 = tid; e0 <= (uint32_t)kL2MaxTok; e0 += kL2Threads) {
      uint32_t r = e0, cnt = 0, sg = 0;
      while (r < (uint32_t)kL2W) {
        const uint32_t v = hopW[r];
        cnt += v >> 21;
        sg += (v >> 14) & 0x7fu;
        r = v & 0x3fffu;
      }
      memo[e0] = ((r - kL2W) << 21) | (cnt << 8) | sg;   // (exit offset <= 1097, entries <= 4096, significant <= 171)
    }
    __syncthreads();
    // ---- look back, publish: tag | done << 55 | exit offset << 44 | entries << 22 | significant
    if (tid == 0) {
      uint32_t e = 0, rank = 0, sg = 0, stop = 0, last = 0;
      if (i > 0) {
        unsigned long long f = 0;
        uint32_t spins = 0;
        uint64_t spinT0 = 0;
        for (;;) {
          f = __hip_atomic_load(flags + (i - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((f >> 56) == (unsigned long long)(p + 1))
            break;
          // (the end-of-pass marker is looked at now and then: the poll stays one load long)
          if ((++spins & 15u) == 0 &&
              __hip_atomic_load(&s.l2PlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p + 1) {
            stop = 1;
            break;
          }
          if (spin_expired(spins, spinT0)) {   // (a minute of wall time: the device has stopped making progress)
            s.error = kErrLookBackTimeout;
            __hip_atomic_store(&s.l2PlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            stop = 1;
            break;
          }
        }
        if (!stop) {
          if ((f >> 55) & 1ull)
            stop = 1;
          else {
            e = (uint32_t)(f >> 44) & 0x7ffu;
            rank = (uint32_t)(f >> 22) & 0x3fffffu;
            sg = (uint32_t)f & 0x3fffffu;
          }
        }
      }
      if (!stop) {
        const uint32_t m = memo[e], mc = (m >> 8) & 0x1fffu;
        if (rank + mc >= n) {   // the list ends inside this block
          last = 1;
          __hip_atomic_store(flags + i, tag | (1ull << 55), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&s.l2PlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else
          __hip_atomic_store(flags + i,
                             tag | ((unsigned long long)(m >> 21) << 44) |
                                 ((unsigned long long)(rank + mc) << 22) |
                                 (unsigned long long)(sg + (m & 0xffu)),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      sh_e = e;
      sh_rank = rank;
      sh_sig = sg;
      sh_last = last;
      sh_stop = stop;
      sh_endpos = 0;
      sh_endsig = 0;
    }
    for (uint32_t k = tid; k < (uint32_t)(kL2W / 64); k += kL2Threads)
      blkE[k] = kL0None;
    if (tid < kL2W / 1024)
      entR[tid] = kL0None;
    __syncthreads();
    if (sh_stop)
      break;
    // ---- where the chain enters each 1024-block, then each sub-block
    if (tid == 0) {
      uint32_t r = sh_e, rk = 0, sg = 0;
      while (r < (uint32_t)kL2W) {
        entR[r >> 10] = r;
        entK[r >> 10] = rk;
        entS[r >> 10] = sg;
        const uint32_t v = hopW[r];
        rk += v >> 21;
        sg += (v >> 14) & 0x7fu;
        r = v & 0x3fffu;
      }
    }
    __syncthreads();
    if (tid < kL2W / 1024 && entR[tid] != kL0None) {
      uint32_t r = entR[tid], rk = entK[tid], sg = entS[tid];
      const uint32_t end = ((uint32_t)tid + 1) * 1024;
      while (r < end) {
        blkE[r >> 6] = r;
        blkK[r >> 6] = rk;
        blkS[r >> 6] = sg;
        const uint32_t v = hop64[r];
        rk += v >> 21;
        sg += (v >> 14) & 0x7fu;
        r = v & 0x3fffu;
      }
    }
    __syncthreads();
    // ---- marks: (1 + entries before the token) | significant entries before it << 16, block-local
    for (uint32_t r = tid; r < (uint32_t)kL2W; r += kL2Threads)
      hopW[r] = 0;
    __syncthreads();
    const uint32_t rank0 = sh_rank, sig0 = sh_sig;
    const uint32_t nloc = n - rank0;
    if (tid < kL2W / 64 && blkE[tid] != kL0None) {
      uint32_t r = blkE[tid], rk = blkK[tid], sg = blkS[tid];
      const uint32_t end = ((uint32_t)tid + 1) * 64;
      bool did = false;
      while (r < end && rk < nloc) {
        hopW[r] = (rk + 1u) | (sg << 16);
        rk++;
        sg += bit_at(r);
        r += U2[r];
        did = true;
      }
      if (did && rk == nloc) {
        sh_endpos = r;
        sh_endsig = sg;
      }
    }
    __syncthreads();
    // ---- sweep 1: insignificant entries stay, significant ones leave their list entry in hop64 and queue up
    {
      uint32_t mk4[kL2Per];
      uint64_t id4[kL2Per];
#pragma unroll
      for (int j = 0; j < kL2Per; j++)
        mk4[j] = hopW[(uint32_t)tid + (uint32_t)j * kL2Threads];
#pragma unroll
      for (int j = 0; j < kL2Per; j++)
        id4[j] = mk4[j] ? list[rank0 + (mk4[j] & 0xffffu) - 1u] : 0ull;
#pragma unroll
      for (int j = 0; j < kL2Per; j++) {
        const uint32_t r = (uint32_t)tid + (uint32_t)j * kL2Threads;
        const uint32_t mk = mk4[j];
        if (mk == 0)
          continue;
        const uint32_t q = rank0 + (mk & 0xffffu) - 1u, sb = sig0 + (mk >> 16);
        const uint64_t ident = id4[j];
        if (!bit_at(r)) {
          keep[q - sb] = ident;
          continue;
        }
        hop64[r + 1] = (uint32_t)ident;
        hop64[r + 2] = (uint32_t)(ident >> 32);
        const uint32_t ti = atomicAdd(&sh_ntok, 1u);
        if (ti < (uint32_t)kL2TokCap)
          tokQ[ti] = (uint16_t)r;
      }
    }
    __syncthreads();
    // ---- sweeps T and C, twice: first counting (write == 0), then -- the block's slots reserved -- writing.
    //      QC (the marks' array: they are dead now): a significant child as
    //        position of the token | child << 12 | (child's position - token's) << 15 | coded << 26
    //      QS: the child's first birth slot | first leaf slot << 16, block-local
    uint32_t* QC = hopW;
    uint32_t* QS = hopW + kL2ChildCap;
    const uint32_t ntok = min(sh_ntok, (uint32_t)kL2TokCap);
    for (int write = 0; write < 2; write++) {
      // sweep T: a significant token per thread
      if ((uint32_t)tid < ntok) {
        const uint32_t r = tokQ[tid];
        const Node nd = unpack_node((uint64_t)hop64[r + 1] | ((uint64_t)hop64[r + 2] << 32));
        uint32_t y = r + 1, found = 0, nb = 0, nsc = 0, kinds = 0, offs[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
          const bool coded = found || k != 7;
          const uint32_t u = coded ? U1[y] : T1[y];
          offs[k] = (y - r) | ((uint32_t)coded << 11);
          if (coded && u == 1) {
            const uint64_t rel = a + y - phase0;
            if (slot1 != 0xff && rel < maskBits) {
              kinds |= 0x100u << k;
              nb++;
            }
          }
          else {
            found = 1;
            nsc++;
            kinds |= 1u << k;
          }
          y += u;
        }
        if (!write) {
          tokSlot[tid] = nb ? atomicAdd(&sh_nb, nb) : 0u;
          uint32_t sc = atomicAdd(&sh_nc, nsc);
#pragma unroll
          for (int k = 0; k < 8; k++)
            if (((kinds >> k) & 1u) && sc < (uint32_t)kL2ChildCap)
              QC[sc++] = r | ((uint32_t)k << 12) | (offs[k] << 15);
        }
        else {
          uint32_t slotB = sh_baseB + tokSlot[tid];
#pragma unroll
          for (int k = 0; k < 8; k++)
            if ((kinds >> (8 + k)) & 1u) {
              const uint32_t cx = 2u * nd.i[0] + (uint32_t)(k & 1), cy = 2u * nd.i[1] + (uint32_t)((k >> 1) & 1),
                             cz = 2u * nd.i[2] + (uint32_t)(k >> 2);
              const uint64_t rel = a + r + (offs[k] & 0x7ffu) - phase0;
              if (slotB < b.bornStride) {   // stays insignificant: joins the list of the 4x4x4 sets
                bornPacked[slotB] = ((uint64_t)(nd.grid + 1) << 48) | ((uint64_t)cz << 32) | ((uint64_t)cy << 16) | (uint64_t)cx;
                bornPosLev[slotB] = ((uint64_t)lev1 << 48) | rel;
                atomic_or64(mask1 + (rel >> 6), 1ull << (rel & 63));
              }
              slotB++;
            }
        }
      }
      __syncthreads();   // (QC is complete)
      // sweep C: a significant child per thread
      const uint32_t nchild = min(sh_nc, (uint32_t)kL2ChildCap);
      for (uint32_t i2 = tid; i2 < nchild; i2 += kL2Threads) {
        const uint32_t desc = QC[i2];
        const uint32_t r = desc & 0xfffu, k = (desc >> 12) & 7u, y0 = r + ((desc >> 15) & 0x7ffu);
        const bool ccoded = (desc >> 26) & 1u;
        uint32_t y = ccoded ? y0 + 1 : y0, found = 0, nb = 0, nl = 0, kinds = 0, offs[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const bool coded = found || j != 7;
          const uint32_t u = coded ? U0[y] : T0[y];
          offs[j] = (y - y0) | ((uint32_t)coded << 8);
          if (coded && u == 1) {
            const uint64_t rel = a + y - phase0;
            if (slot0 != 0xff && rel < maskBits) {
              kinds |= 0x100u << j;
              nb++;
            }
          }
          else {
            found = 1;
            nl++;
            kinds |= 1u << j;
          }
          y += u;
        }
        if (!write) {
          const uint32_t sB = nb ? atomicAdd(&sh_nb, nb) : 0u;
          const uint32_t sL = atomicAdd(&sh_nl, nl);
          QS[i2] = sB | (sL << 16);
          continue;
        }
        const Node nd = unpack_node((uint64_t)hop64[r + 1] | ((uint64_t)hop64[r + 2] << 32));
        const uint32_t px = 2u * nd.i[0] + (k & 1u), py = 2u * nd.i[1] + ((k >> 1) & 1u), pz = 2u * nd.i[2] + (k >> 2);
        const Grid g2 = sh_grids[nd.grid + 2];   // the grid of the 2x2x2 sets
        uint32_t slotB = sh_baseB + (QS[i2] & 0xffffu), slotL = sh_baseL + (QS[i2] >> 16);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const uint32_t cx = 2u * px + (uint32_t)(j & 1), cy = 2u * py + (uint32_t)((j >> 1) & 1), cz = 2u * pz + (uint32_t)(j >> 2);
          const uint32_t yy0 = y0 + (offs[j] & 0xffu);
          if ((kinds >> (8 + j)) & 1u) {
            const uint64_t rel = a + yy0 - phase0;
            if (slotB < b.bornStride) {   // stays insignificant: joins the list of the smallest sets
              bornPacked[slotB] = ((uint64_t)(nd.grid + 2) << 48) | ((uint64_t)cz << 32) | ((uint64_t)cy << 16) | (uint64_t)cx;
              bornPosLev[slotB] = ((uint64_t)lev0 << 48) | rel;
              atomic_or64(mask0 + (rel >> 6), 1ull << (rel & 63));
            }
            slotB++;
          }
          else if ((kinds >> j) & 1u) {   // splits into its pixels: a leaf event
            const bool coded = (offs[j] >> 8) & 1u;
            const uint32_t v = bits32(coded ? yy0 + 1 : yy0);
            uint32_t yy = 0, fnd = 0, sigm = 0, negm = 0;
#pragma unroll
            for (int q = 0; q < 7; q++) {
              const uint32_t bit = (v >> yy) & 1u, sgn = (v >> (yy + 1)) & 1u;
              sigm |= bit << q;
              negm |= (bit & (sgn ^ 1u)) << q;
              fnd |= bit;
              yy += 1u + bit;
            }
            const uint32_t bit = fnd ? (v >> yy) & 1u : 1u;
            const uint32_t sgn = (v >> (yy + fnd)) & 1u;
            sigm |= bit << 7;
            negm |= (bit & (sgn ^ 1u)) << 7;
            const uint32_t fid = g2.nodeOff + (((cz << g2.e[1]) + cy) << g2.e[0]) + cx;
            if (slotL < b.leafCap)
              leafEv[slotL] = (uint64_t)fid | ((uint64_t)sigm << 32) | ((uint64_t)negm << 40);
            slotL++;
          }
        }
      }
      __syncthreads();
      if (!write) {
        if (tid == 0)     // (two threads: the two returning atomics are in flight together)
          sh_baseB = sh_nb ? atomicAdd(&s.bornCount, sh_nb) : 0u;
        if (tid == 64)
          sh_baseL = sh_nl ? atomicAdd(&s.leafCount, sh_nl) : 0u;
        __syncthreads();
      }
    }
    if (sh_ntok > (uint32_t)kL2TokCap || sh_nc > (uint32_t)kL2ChildCap)   // (cannot be: see the constants)
      s.error = 1;
    if (sh_last) {
      if (tid == 0) {
        s.l2End = a + sh_endpos;
        s.listLen[nx][L] = n - (sig0 + sh_endsig);
      }
      break;
    }
    __syncthreads();   // LDS is reused by the next block
  }
}

// ------------------------------------------------------------------------------------------
// LIS phase, table-driven (chunks whose LIS levels are all "regular", spk::LevelClass): what
// k_lis_hi below shares with its predecessor k_lis_tables (one 1024-thread workgroup per chunk,
// rounds 1-3; removed in round 4 -- a regular tree k_lis_hi cannot take goes to k_lis_mx).
// tests/model/speck_model.cpp::model_speck3d_decode_par is the CPU model of the method:
//   tables   T_j[x] = bits the split of a class-j set takes when it starts at bit x (kTInf when
//            that would leave the window): speculative, one thread per bit position and class;
//   hop      the list entries: '0' -> next bit, '1' -> 1 + T lookup; an entry that does not fit
//            the tables is walked into;
//   expand   every set that splits inside the window is one work item; a thread finds its
//            children with T_{j-1}, queues significant child sets for the next round and records
//            insignificant ones with their stream position.
// After the last level the recorded sets are ranked by position (popcount prefix of per-level
// position masks) and appended to the next lists; old entries are compacted in order.
// ------------------------------------------------------------------------------------------
constexpr int kTabThreads = 1024;
constexpr uint32_t kTInf = 0xffffu;


__device__ __forceinline__ Node reg_child(const Tree& t, const Node& nd, uint32_t ord, int ee[3],
                                          uint32_t idx[3])
{
  const Grid& g = t.grids[nd.grid];
  const Root& r = t.roots[g.root];
  Node c;
  c.grid = (uint16_t)(nd.grid + 1);
  int bit = 0;
  for (int a = 0; a < 3; a++) {
    if (g.depth < r.D[a]) {
      ee[a] = g.e[a] + 1;
      idx[a] = (uint32_t)nd.i[a] * 2u + ((ord >> bit) & 1u);
      bit++;
    }
    else {
      ee[a] = g.e[a];
      idx[a] = nd.i[a];
    }
    c.i[a] = (uint16_t)idx[a];
  }
  return c;
}

__device__ __forceinline__ uint64_t reg_child_packed(const Tree& t, const Node& nd, uint32_t ord)
{
  int ee[3];
  uint32_t idx[3];
  return pack_node(reg_child(t, nd, ord, ee, idx));
}

__device__ __forceinline__ uint32_t reg_child_raster(const Tree& t, const Node& nd, uint32_t ord)
{
  int ee[3];
  uint32_t idx[3];
  reg_child(t, nd, ord, ee, idx);
  return pixel_raster(t, t.roots[t.grids[nd.grid].root], ee, idx);
}

// Tables of one window (all positions are relative to the window start, 0..W):
//   T_j[r]  bits the split of a class-j set takes when it starts at r, or kTInf
//   U_j[r]  the CODED item of class j at r: bit 15 = its test bit, low 15 bits = code length
//           (1 for an insignificant item, 1 + T_j[r+1] otherwise), or kTInf when it is
//           significant but leaves the window
// ------------------------------------------------------------------------------------------
// LIS phase, lists of the 8x8x8 and larger sets, GPU-WIDE: the table method above (speculative
// tables per window, pointer jumping over the list entries, breadth-first expansion of the sets
// that split inside the window) with the windows at FIXED places, so that several workgroups per
// chunk work on one chunk's phase at a time and only a short hop stays on the serial chain.
//
//   * the phase's stream, from where k_lis_l1 ended, is cut into regions of W bits; region i is
//     handed out by a ticket counter (a region is handed out only after all earlier ones, so a
//     waiting workgroup always waits for a running one);
//   * OFF the chain a workgroup loads its region and builds the tables T_j / U_j of every class
//     the level needs (the level is taken from a hint the chain leaves behind; a wrong guess
//     costs a rebuild on the chain) and the pointer-jump table of the list's own class;
//   * ON the chain it takes its predecessor's state -- stream position, list level, entry index,
//     entries left, and the stack of sets being walked into (child ordinal + "found" bit per
//     frame, the list entry at the bottom) packed into four tagged words read and written with
//     relaxed agent-scope atomics, no fence -- resolves the hop through its region with the
//     tables (serial contexts, then the list entries by pointer jumping), and publishes the
//     state at its end;
//   * OFF the chain again it expands the sets that split inside the region.
// Births and leaf events take their slots from the chunk's global counters, one atomic per
// wavefront and round.  Old entries are compacted afterwards by k_lis_compact; k_lis_hi_end sets
// the chunk's state for the placement kernels.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kHiTagShift = 57;
constexpr unsigned long long kHiPayloadMask = (1ull << kHiTagShift) - 1ull;
constexpr int kHiFrames = 10;   // list level + the sets being walked into (chains of up to 9 classes)
#ifndef HI_FIRST_REGION
#define HI_FIRST_REGION 1024
#endif
constexpr uint32_t kHiFirstRegion = HI_FIRST_REGION;   // bits of a phase's first region; the next ones double up to the full size

// Lanes of ONE wavefront that talk through LDS: the hardware keeps a wavefront's LDS traffic in
// order, but the compiler reasons per thread (it may forward a lane's own earlier store to its
// later load and never look at memory), so every hand-over between lanes needs a fence.
#define HI_WAVE_SYNC()                                        \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");    \
    __builtin_amdgcn_wave_barrier();                          \
  } while (0)

template <typename CT>
__global__ void __launch_bounds__(kTabThreads) k_lis_hi(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  extern __shared__ __attribute__((aligned(16))) char tab_smem[];
  uint64_t* wbits = reinterpret_cast<uint64_t*>(tab_smem);
  const uint32_t* w32 = reinterpret_cast<const uint32_t*>(tab_smem);
  // Round 5, second session: the FIRST regions of a plane's phase are short -- 1024, 2048, 4096 bits, then the full size
  // (hi_region below: where a region starts is still a function of its number alone) --, so that a phase of a few
  // hundred bits, which is what the dozen light planes hold, builds tables for 1.5 K positions and not for 6.9 K
  // (26 us of the 55 to 90 the kernel took on such a plane).  A long phase pays three short regions for it.
  const uint32_t Wmax = b.hiW, TS = Wmax + 2;   // positions the tables have room for
  const uint32_t SRmax = Wmax - b.hiAhead;      // bits of a full region (the tables look hiAhead bits further)
  uint32_t W = Wmax, SR = SRmax;                // of the region at hand (set with its ticket)
  const int Kcap = (int)b.hiK;                     // classes the LDS tables have room for
  const uint32_t kWords = Wmax / 64 + 4;
  uint32_t* hop = reinterpret_cast<uint32_t*>(tab_smem + (size_t)kWords * 8);   // [W + 130]
  uint32_t* hop2 = b.hiHop2 ? hop + (Wmax + 130) : hop;                         // [Wmax + 130], or none (never built then)
  uint16_t* Tt = reinterpret_cast<uint16_t*>(hop + (b.hiHop2 ? 2 : 1) * (Wmax + 130));   // [Kcap - 1][TS]
  uint16_t* Uu = Tt + (size_t)(Kcap - 1) * TS;                                  // [Kcap][TS]
  constexpr int kBlk = kTabWMax / 64 + 4;
  __shared__ uint32_t blkEB[kBlk];
  __shared__ uint32_t sh_total, sh_stopped, sh_newr;
  // The serial chain keeps NO geometry (round 5).  A frame -- the children of a set being walked into -- is one word:
  // children left | next ordinal << 8 | found << 16 | children's class << 24.  What the walk steps over goes to a
  // log (per step: the frame, its first ordinal, and per child what became of it and where), and the sets' nodes
  // are worked out from the log AFTER the region's state is published (hi_replay): the list entry at the bottom
  // is known by its index in the list storage, the set a frame entered by its ordinal.  Before, every step
  // derived eight child nodes from LDS geometry tables, claimed birth slots and wrote queue items on the chain,
  // the look-back rebuilt the parents of the open frames, and an entry that left the region was a global load.
  constexpr int kHiLogCap = 32;
  // A list with no more entries left than this is gone through entry by entry (one table look-up each) when the
  // region has no pointer-jump table of its class, instead of all hands leaving the chain to build one (with one
  // table per region, round 5, the dozen short lists of every light plane cost 94 such builds per chunk); and the
  // region's one table is built for the first list from the hinted one on that is longer than this.
#ifndef HI_SERIAL
#define HI_SERIAL 48
#endif
  constexpr uint32_t kHiSerial = HI_SERIAL;
  __shared__ uint32_t sh_fr[kHiFrames + 2];
  __shared__ uint32_t sh_logHdr[kHiLogCap], sh_logBase[kHiLogCap], sh_logLane[kHiLogCap][8];
  __shared__ uint32_t sh_nlog, sh_baseIdx, sh_parInit, sh_baseIdx0, sh_depth0;
  __shared__ uint8_t sh_ord0[kHiFrames + 2];
  __shared__ uint64_t sh_par[kHiFrames + 2];
  __shared__ uint32_t sh_qn[3];
  __shared__ uint32_t sh_len[kMaxLevels], sh_lOff[kMaxLevels];
  __shared__ LevelClass sh_lc[kMaxLevels];
  __shared__ uint8_t sh_lslot[kMaxLevels];
  __shared__ uint64_t sh_serve;   // bit l: the tables in LDS serve level l's lists
  __shared__ uint32_t sh_segBorn, sh_segLeaf, sh_segBornEnd;   // filled slots of this workgroup's segments
  constexpr int kLdsRoots = 48, kLdsGrids = 288;
  __shared__ Root sh_roots[kLdsRoots];
  __shared__ Grid sh_grids[kLdsGrids];
  // chain state (thread 0 owns it; the others read it between barriers)
  __shared__ uint64_t sh_pos;
  __shared__ uint32_t sh_level, sh_depth, sh_e, sh_rem, sh_stop, sh_action, sh_ticket, sh_any;
  __shared__ unsigned long long sh_in[4];
  __shared__ int sh_tabLevel, sh_tabK, sh_hopTop[2], sh_tabFrom, sh_hintK;   // what the tables in LDS were built for
  __shared__ uint32_t sh_over, sh_haveState, sh_act2, sh_zn;
  __shared__ uint64_t sh_t2, sh_t3, sh_tacc[8];
#define HI_T(k)                                               \
  if (b.lisStamps && lane == 0) {                             \
    const uint64_t now_ = __builtin_readcyclecounter();       \
    sh_tacc[k] += now_ - tmark;                               \
    tmark = now_;                                             \
  }

  // (the tree's geometry tables are read from LDS: the serial hop derives child sets from them)
  Tree t = b.tree;
  t.roots = sh_roots;
  t.grids = sh_grids;
  const int tid = threadIdx.x;
  const uint32_t lane = (uint32_t)tid & 63u;
  for (uint32_t i = tid; i < t.nroots && i < (uint32_t)kLdsRoots; i += kTabThreads)
    sh_roots[i] = b.tree.roots[i];
  for (uint32_t i = tid; i < t.ngrids && i < (uint32_t)kLdsGrids; i += kTabThreads)
    sh_grids[i] = b.tree.grids[i];
  const uint32_t cur = s.cur;
  for (uint32_t l = tid; l < t.nlevels; l += kTabThreads) {
    sh_len[l] = s.listLen[cur][l];
    sh_lOff[l] = b.levelOff[l];
    sh_lc[l] = b.levelClass[l];
    sh_lslot[l] = b.levelSlot[l];
  }
  const uint64_t* words = b.stream + c * b.streamStride;
  const uint64_t nwordsAvail = (s.avail + 63) / 64;
  const uint64_t phase0 = s.lipStart + s.lipBits;
  const uint64_t maskBits = (uint64_t)b.maskWords * 64;
  uint64_t* qbase = b.queue + c * b.queueStride + (size_t)blockIdx.x * b.queueCap * 4;
  uint64_t* qbuf[2] = {qbase, qbase + b.queueCap * 2};
  uint64_t* bornPacked = b.bornPacked + c * b.bornPitch;
  uint64_t* bornPosLev = b.bornPosLev + c * b.bornPitch;
  uint64_t* sigbits = b.sigbits + c * b.sigbitsStride;
  uint64_t* leafEv = b.leafEv + c * b.leafStride;
  const uint64_t* lisCur = b.lis[cur] + c * b.lisStride;
  unsigned long long* flags = b.hiFlags + c * b.hiFlagStride;
  const unsigned long long tag = (unsigned long long)(p + 1) << kHiTagShift;
  const bool l0done = b.l0Level >= 0 && s.l0PlaneP1 == p + 1;
  const bool l1done = b.l1Level >= 0 && s.l1PlaneP1 == p + 1;
  const bool l2done = b.l2Level >= 0 && s.l2PlaneP1 == p + 1;
  const uint64_t S0 = l2done ? s.l2End : l1done ? s.l1End : l0done ? s.l0End : phase0;
  if (tid == 0) {
    sh_segBorn = sh_segLeaf = 0;
    sh_segBornEnd = 0xffffffffu;
  }
  __syncthreads();
  // (candidate positions of the class tables, build_tables: the fewest '0' siblings that can stand in front of a set
  //  whose test bit is implied -- one less than the smallest arity of any class of the tree: seven for octrees)
  if (tid < 64) {
    uint32_t zn = 31;
    if ((uint32_t)tid < t.nlevels && sh_lc[tid].regular)
      for (int j = 0; j < (int)sh_lc[tid].K && j < kMaxClasses; j++)
        zn = min(zn, (uint32_t)max((int)sh_lc[tid].arity[j], 1) - 1u);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1)
      zn = min(zn, (uint32_t)__shfl_xor((int)zn, d, 64));
    if (tid == 0)
      sh_zn = zn;
  }
  __syncthreads();

  // the next list after level `l` (exclusive) that holds entries and is this kernel's to decode
  auto next_level = [&](int l) -> int {
    for (l = l - 1; l >= 0; l--) {
      if ((l0done && l == b.l0Level) || (l1done && l == b.l1Level) || (l2done && l == b.l2Level))
        continue;
      if (sh_len[l] != 0)
        return l;
    }
    return -1;
  };

  uint32_t wq0 = 0;
  auto bit_at = [&](uint32_t r) -> uint32_t {
    const uint32_t q = r + wq0;
    return (w32[q >> 5] >> (q & 31)) & 1u;
  };
  auto bits32 = [&](uint32_t r) -> uint32_t {
    const uint32_t q = r + wq0, sh = q & 31;
    const uint32_t lo = w32[q >> 5], hi = w32[(q >> 5) + 1];
    return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
  };
  uint64_t a = 0;   // absolute bit of window position 0

  // ---- tables of classes [j0, j1) of level `lv`'s chain (see the section comment above)
  auto split_len = [&](const LevelClass& C, int j, uint32_t r) -> uint32_t {
    const int ar = C.arity[j];
    if (j == 0) {
      // a leaf parent's split is at most 16 bits and the bits past the region's end are loaded:
      // it is known wherever it starts (r <= W), so the chain never has to walk into one
      if (r > W)
        return kTInf;
      const uint32_t v = bits32(r);
      uint32_t y = 0, found = 0;
      if (ar == 8) {
#pragma unroll
        for (int i = 0; i < 7; i++) {
          const uint32_t bit = (v >> y) & 1u;
          found |= bit;
          y += 1u + bit;
        }
        const uint32_t bit = found ? (v >> y) & 1u : 1u;
        y += found + bit;
      }
      else {
        for (int i = 0; i < ar; i++) {
          const uint32_t coded = found | (uint32_t)(i + 1 != ar);
          const uint32_t bit = coded ? (v >> y) & 1u : 1u;
          y += coded;
          found |= bit;
          y += bit;
        }
      }
      return y;
    }
    const uint16_t* Up = Uu + (size_t)(j - 1) * TS;
    const uint16_t* Tp = Tt + (size_t)(j - 1) * TS;
    if (ar == 8) {
      uint32_t y = min(r, W + 1), fl = 0;
#pragma unroll
      for (int i = 0; i < 7; i++) {
        const uint32_t u = Up[y];
        fl |= u;
        y = min(y + (u & 0x7fffu), W + 1);
      }
      const uint32_t v = ((fl & 0x8000u) ? Up : Tp)[y];
      return (y > W || v == kTInf) ? kTInf : y + (v & 0x7fffu) - r;
    }
    uint32_t y = r, found = 0;
    for (int i = 0; i + 1 < ar; i++) {
      const uint32_t u = y <= W + 1 ? Up[y] : kTInf;
      if (u == kTInf)
        return kTInf;
      found |= u >> 15;
      y += u & 0x7fffu;
    }
    if (y > W + 1)
      return kTInf;
    uint32_t last;
    if (found) {
      const uint32_t u = Up[y];
      if (u == kTInf)
        return kTInf;
      last = u & 0x7fffu;
    }
    else {
      last = Tp[y];
      if (last == kTInf)
        return kTInf;
    }
    return y + last - r;
  };
  // all threads; T_j for j < j1 - 1 ... every class in [j0, j1) gets both T_j and U_j (the top
  // class of a level only needs U, but a later level of the same chain may sit on top of it)
  // (from: the first position anybody will look at -- a build in the middle of a region, when the chain has come to a
  //  list the tables do not serve, starts where the chain stands: half the positions on average)
  // Round 6 -- `candOnly` (the build at the head of a region): classes 1 and up are evaluated only where a split of a set
  // can START: behind a '1' (a coded significant item's split), or behind as many '0's as the smallest arity leaves in
  // front of an implied last child, or at the window's edges; every other T is "leaves the window" (never looked at: a
  // reader that did would walk into the set, slow and right), every U of a '0' bit is 1 without a look-up.  At 2 bits
  // per sample that is 35 to 45 % of the positions.  Round 5 tried the test WITHOUT compaction -- the other lanes sat
  // the eight-deep look-up chain out -- and lost: the tables cost their ds_read_u16 wavefront instructions, 12.5 cycles
  // each whatever the lanes hold (profiles/r5_knob_ab.txt).  Here every wavefront first lists the candidates of its
  // slice of the window (a ballot and a prefix per 64 positions, in the pointer-jump table's memory: that table is
  // built after these) and then runs the chains over the list, all lanes busy.
  auto build_tables = [&](int lv, int j0, int j1, uint32_t from = 0, bool candOnly = false) {
    const LevelClass& C = sh_lc[lv];
    if (candOnly) {
      uint16_t* cand = reinterpret_cast<uint16_t*>(hop);
      const uint32_t wv = (uint32_t)tid >> 6, npos = W + 2;
      const uint32_t slice = ((npos + kTabThreads / 64 - 1) / (kTabThreads / 64) + 63u) & ~63u;   // positions a wavefront lists
      const uint32_t zn = sh_zn, zmask = zn >= 32 ? 0xffffffffu : ((1u << zn) - 1u);
      uint32_t ncand = 0;
      for (uint32_t r0 = wv * slice; r0 < min(npos, (wv + 1) * slice); r0 += 64) {
        const uint32_t r = r0 + lane;
        bool is = false;
        if (r < npos)
          is = r <= zn || r >= W || bit_at(r - 1) != 0 || (bits32(r - zn) & zmask) == 0;
        const uint64_t m = __ballot(is);
        if (is)
          cand[wv * slice + ncand + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)r;
        ncand += (uint32_t)__popcll(m);
      }
      // every entry of every class once, two positions a store: T "leaves the window", U 1 behind a '0'
      for (uint32_t h = (uint32_t)tid; 2 * h < npos; h += kTabThreads) {
        const uint32_t r = 2 * h;
        const uint32_t u0 = (r >= W || bit_at(r)) ? kTInf : 1u, u1 = (r + 1 >= W || bit_at(r + 1)) ? kTInf : 1u;
        const uint32_t u = u0 | (u1 << 16);
        for (int j = j0; j < j1; j++) {
          reinterpret_cast<uint32_t*>(Uu + (size_t)j * TS)[h] = u;
          if (j < Kcap - 1)
            reinterpret_cast<uint32_t*>(Tt + (size_t)j * TS)[h] = kTInf | (kTInf << 16);
        }
      }
      __syncthreads();
      for (int j = j0; j < j1; j++) {
        uint16_t* Uj = Uu + (size_t)j * TS;
        uint16_t* Tj = j < Kcap - 1 ? Tt + (size_t)j * TS : nullptr;
        for (uint32_t i0 = lane; i0 < ncand; i0 += 4 * 64) {
          uint32_t rr[4], tl[4];
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint32_t i = i0 + (uint32_t)q * 64u;
            rr[q] = i < ncand ? (uint32_t)cand[wv * slice + i] : 0xffffffffu;
            tl[q] = rr[q] != 0xffffffffu ? split_len(C, j, rr[q]) : kTInf;
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const uint32_t r = rr[q];
            if (r == 0xffffffffu)
              continue;
            if (Tj)
              Tj[r] = (uint16_t)tl[q];
            if (r >= 1 && r - 1 < W && bit_at(r - 1))
              Uj[r - 1] = (uint16_t)(tl[q] == kTInf ? kTInf : (0x8000u | (1u + tl[q])));
          }
        }
        __syncthreads();
      }
      return;
    }
    for (int j = j0; j < j1; j++) {
      uint16_t* Uj = Uu + (size_t)j * TS;
      uint16_t* Tj = j < Kcap - 1 ? Tt + (size_t)j * TS : nullptr;
      auto coded = [&](uint32_t q, uint32_t tl) -> uint16_t {   // U_j[q] given T_j[q + 1]
        if (q >= W)
          return (uint16_t)kTInf;
        if (!bit_at(q))
          return (uint16_t)1;
        return (uint16_t)(tl == kTInf ? kTInf : (0x8000u | (1u + tl)));
      };
      // four independent positions per thread and pass: their LDS chains overlap
      for (uint32_t r0 = from + (uint32_t)tid; r0 <= W + 1; r0 += 4 * kTabThreads) {
        uint32_t tl[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t r = r0 + (uint32_t)q * kTabThreads;
          tl[q] = r <= W + 1 ? split_len(C, j, r) : kTInf;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const uint32_t r = r0 + (uint32_t)q * kTabThreads;
          if (r > W + 1)
            continue;
          if (Tj)
            Tj[r] = (uint16_t)tl[q];
          if (r >= 1)
            Uj[r - 1] = coded(r - 1, tl[q]);
          if (r == W + 1)
            Uj[W + 1] = (uint16_t)kTInf;
        }
      }
      __syncthreads();
    }
  };
  // all threads; pointer-jump table over the coded items of class `top`, from every position:
  // hop[h] for h = r + wq0 (64-bit blocks are stream words): cnt << 16 | stop << 15 | exit
  auto build_hop = [&](uint32_t* hop, int top, uint32_t from = 0) {
    const uint16_t* Utop = Uu + (size_t)top * TS;
    const uint32_t nblk = ((SR - 1 + wq0) >> 6) + 1;
    const uint32_t blk0 = (min(from, SR - 1) + wq0) >> 6;   // the block position `from` lies in
    const int32_t rbase = -(int32_t)wq0;
    const uint32_t wave = (uint32_t)tid >> 6;
    for (uint32_t bi = blk0 + wave; bi < nblk; bi += kTabThreads / 64) {
      const uint32_t h = bi * 64 + lane;
      const int32_t rs = (int32_t)h + rbase;
      const bool live = rs >= 0 && rs < (int32_t)SR;
      uint32_t v = 0x8000u;
      bool inb = false;
      const uint32_t hEnd = (bi + 1) * 64;
      if (live) {
        const uint32_t r = (uint32_t)rs;
        const uint32_t u = Utop[r];
        if (u == kTInf)
          v = 0x8000u | r;
        else {
          const uint32_t nr = r + (u & 0x7fffu);
          v = (1u << 16) | nr;
          inb = nr < SR && (uint32_t)((int32_t)nr - rbase) < hEnd;
        }
      }
      for (int it = 0; it < 6 && __any(inb); it++) {
        const uint32_t src = (uint32_t)((int32_t)(v & 0x7fffu) - rbase) & 63u;
        const uint32_t o = __shfl(v, src, 64);
        if (inb) {
          v = (v & 0xffff0000u) + o;
          const uint32_t np = v & 0x7fffu;
          inb = !(v & 0x8000u) && np < SR && (uint32_t)((int32_t)np - rbase) < hEnd;
        }
      }
      if (live)
        hop[h] = v;
    }
    __syncthreads();
    for (uint32_t wide = 128; wide <= 256; wide <<= 1) {
      for (uint32_t h = (blk0 & ~3u) * 64u + (uint32_t)tid; h < nblk * 64; h += kTabThreads) {   // (from a 256-position boundary on)
        const int32_t rs = (int32_t)h + rbase;
        if (rs < 0 || rs >= (int32_t)SR || (h >> 6) < blk0)
          continue;
        const uint32_t v = hop[h];
        if (v & 0x8000u)
          continue;
        const uint32_t er = v & 0x7fffu;
        if (er >= SR)
          continue;
        const uint32_t eh = (uint32_t)((int32_t)er - rbase);
        if (eh / wide != h / wide)
          continue;
        hop[h] = (v & 0xffff0000u) + hop[eh];
      }
      __syncthreads();
    }
  };
  // thread 0: do the tables in LDS (built for sh_tabLevel, classes [0, sh_tabK)) serve level lv?
  // (one lane per level: tables built for level tl with tk classes)
  auto serve_mask = [&](int tl, int tk) -> uint64_t {
    const uint32_t l = (uint32_t)tid & 63u;
    bool ok = l < t.nlevels && sh_len[l] != 0 && tl >= 0;
    if (ok) {
      const LevelClass& A = sh_lc[tl];
      const LevelClass& B = sh_lc[l];
      const int BK = min((int)B.K, Kcap);
      ok = BK <= tk;
      for (int j = 0; ok && j < BK; j++)
        ok = A.arity[j] == B.arity[j] && A.lev[j] == B.lev[j];
    }
    return __ballot(ok);
  };
  auto tables_serve = [&](int lv) -> bool {
    if (sh_tabLevel < 0)
      return false;
    const LevelClass& A = sh_lc[sh_tabLevel];
    const LevelClass& B = sh_lc[lv];
    const int BK = min((int)B.K, Kcap);
    if (BK > sh_tabK)
      return false;
    for (int j = 0; j < BK; j++)
      if (A.arity[j] != B.arity[j] || A.lev[j] != B.lev[j])
        return false;
    return true;
  };

  // births and leaf events of the expansion go to a segment of the chunk's arrays that is this
  // workgroup's alone (an LDS counter; the shared part, with its global counter, takes what does
  // not fit)
  const uint32_t segB0 = (uint32_t)b.bornStride + blockIdx.x * b.bornSeg;
  const uint32_t segL0 = b.leafCap + blockIdx.x * b.leafSeg;
  auto born_slots = [&](uint32_t n) -> uint32_t {
    if (n == 0)
      return 0;
    const uint32_t k = atomicAdd(&sh_segBorn, n);
    if (k + n <= b.bornSeg)
      return segB0 + k;
    atomicMin(&sh_segBornEnd, k);        // the segment is full from here on
    return atomicAdd(&s.bornCount, n);   // (slots at or past bornStride are dropped by write_born)
  };
  auto leaf_slot = [&]() -> uint32_t {
    const uint32_t k = atomicAdd(&sh_segLeaf, 1u);
    if (k < b.leafSeg)
      return segL0 + k;
    const uint32_t g = atomicAdd(&s.leafCount, 1u);
    return g < b.leafCap ? g : 0xffffffffu;
  };
  auto write_born = [&](uint32_t slot, uint32_t lev, uint64_t abs, uint64_t packed) {
    const uint64_t rel = abs - phase0;
    if (!(slot < b.bornStride || (slot >= segB0 && slot < segB0 + b.bornSeg)))
      return;   // (only a damaged stream asks for more slots than there are sets)
    bornPacked[slot] = packed;
    bornPosLev[slot] = ((uint64_t)lev << 48) | rel;
    atomic_or64(b.mask + c * b.maskStride + (size_t)sh_lslot[lev] * b.maskWords + (rel >> 6),
                1ull << (rel & 63));
  };
  auto born_counts = [&](uint32_t lev, uint64_t abs) -> bool {   // is this birth recorded at all
    return sh_lslot[lev] != 0xff && abs - phase0 < maskBits;
  };

  // all threads: expand what the queue holds, breadth first (tables of level sh_tabLevel's chain)
  auto expand_all = [&]() {
    const LevelClass& C = sh_lc[sh_tabLevel];
    for (int round = 0;; round++) {
      const uint32_t nin = sh_qn[round % 3];
      if (nin == 0)
        break;
      if (tid == 0 && b.lisStamps) {
        unsigned long long* o_ = reinterpret_cast<unsigned long long*>(b.lisStamps + (size_t)c * 64);
        atomicAdd(o_ + 44, 1ull);
        atomicAdd(o_ + 45, (unsigned long long)nin);
      }
      const uint64_t* qin = qbuf[round & 1];
      uint64_t* qout = qbuf[(round + 1) & 1];
      for (uint32_t i0 = 0; i0 < nin; i0 += kTabThreads) {
        const uint32_t i = i0 + tid;
        const bool have = i < nin;
        uint64_t ident = 0, meta = 0;
        if (have) {
          ident = qin[i * 2];
          meta = qin[i * 2 + 1];
        }
        const int cls = (int)((meta >> 1) & 0x7f);
        const uint32_t y0 = (uint32_t)(meta >> 8);
        const Node nd = unpack_node(have ? ((meta & 1ull) ? lisCur[ident] : ident) : 0ull);
        const int ar = have ? C.arity[cls] : 0;
        const Grid g = sh_grids[nd.grid];
        // ---- leaf parents: one event word each
        const bool isLeaf = have && cls == 0;
        uint32_t sigm = 0, negm = 0;
        if (isLeaf) {
          const uint32_t v = bits32(y0);
          uint32_t yy = 0, found = 0;
          if (ar == 8) {
#pragma unroll
            for (int k = 0; k < 7; k++) {
              const uint32_t bit = (v >> yy) & 1u, sgn = (v >> (yy + 1)) & 1u;
              sigm |= bit << k;
              negm |= (bit & (sgn ^ 1u)) << k;
              found |= bit;
              yy += 1u + bit;
            }
            const uint32_t bit = found ? (v >> yy) & 1u : 1u;
            const uint32_t sgn = (v >> (yy + found)) & 1u;
            sigm |= bit << 7;
            negm |= (bit & (sgn ^ 1u)) << 7;
          }
          else {
            for (int k = 0; k < ar; k++) {
              const uint32_t coded = found | (uint32_t)(k + 1 != ar);
              const uint32_t bit = coded ? (v >> yy) & 1u : 1u;
              yy += coded;
              const uint32_t sgn = (v >> yy) & 1u;
              sigm |= bit << k;
              negm |= (bit & (sgn ^ 1u)) << k;
              found |= bit;
              yy += bit;
            }
          }
        }
        // ---- other sets: children from the tables
        const bool isSet = have && cls > 0;
        const uint16_t* Up = Uu + (size_t)(isSet ? cls - 1 : 0) * TS;
        const uint32_t kidLev = isSet ? C.lev[cls - 1] : 0u;
        if (isLeaf) {
          const uint32_t fid = g.nodeOff + ((((uint32_t)nd.i[2] << g.e[1]) + nd.i[1]) << g.e[0]) + nd.i[0];
          const uint32_t slotL = leaf_slot();
          if (slotL != 0xffffffffu)
            leafEv[slotL] = (uint64_t)fid | ((uint64_t)sigm << 32) | ((uint64_t)negm << 40);
        }
        if (isSet) {
          const Root rt = sh_roots[g.root];
          uint32_t cbase[3], cshift[3];
          uint32_t nb = 0;
          for (int ax = 0; ax < 3; ax++) {
            if (g.depth < rt.D[ax]) {
              cbase[ax] = (uint32_t)nd.i[ax] * 2u;
              cshift[ax] = nb++;
            }
            else {
              cbase[ax] = nd.i[ax];
              cshift[ax] = 31;
            }
          }
          const uint64_t gridBits = (uint64_t)(nd.grid + 1) << 48;
          uint32_t y = y0, found = 0;
          for (int k = 0; k < ar; k++) {
            const uint64_t kid = gridBits |
                                 ((uint64_t)(cbase[2] | (((uint32_t)k >> cshift[2]) & 1u)) << 32) |
                                 ((uint64_t)(cbase[1] | (((uint32_t)k >> cshift[1]) & 1u)) << 16) |
                                 (uint64_t)(cbase[0] | (((uint32_t)k >> cshift[0]) & 1u));
            const bool coded = found || (k + 1 != ar);
            uint32_t start = y;
            if (coded) {
              const uint32_t u = Up[y];
              if (!(u & 0x8000u)) {
                if (born_counts(kidLev, a + y))
                  write_born(born_slots(1u), kidLev, a + y, kid);
                y += 1;
                continue;
              }
              start = y + 1;
              y += u & 0x7fffu;
            }
            found = 1;
            const uint32_t slot = atomicAdd(&sh_qn[(round + 1) % 3], 1u);
            if (slot < b.queueCap) {
              qout[slot * 2] = kid;
              qout[slot * 2 + 1] = ((uint64_t)start << 8) | ((uint64_t)(cls - 1) << 1);
            }
          }
        }
      }
      if (tid == 0)
        sh_qn[(round + 2) % 3] = 0;
      __syncthreads();
    }
    if (tid == 0)
      sh_qn[0] = sh_qn[1] = sh_qn[2] = 0;
    __syncthreads();
  };

  // classes built speculatively beyond the hinted list's own (SPERR_HIP_HI_EXTRA; the chain builds
  // what is missing when it gets further than that inside one region)
  const int kSpecExtra = (int)b.hiExtra;
  const bool hiCand = b.hiCand != 0;
  // what the chain (wavefront 0) can ask the whole workgroup for
  constexpr uint32_t kActDone = 0, kActTables = 1, kActHopTab = 2, kActFlushTables = 4;
  const uint32_t wave = (uint32_t)tid >> 6;

  for (;;) {
    // ---- ticket
    if (tid == 0) {
      const bool over = __hip_atomic_load(&s.hiPlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p + 1;
      sh_ticket = over ? kL0None : atomicAdd(&s.hiTicket, 1u);
      sh_any = 0;
      sh_over = 0;
      sh_nlog = 0;
      sh_parInit = 0;
      sh_haveState = 0;
      sh_stop = 0;
      for (int k = 0; k < 8; k++)
        sh_tacc[k] = 0;
      sh_qn[0] = sh_qn[1] = sh_qn[2] = 0;
    }
    __syncthreads();
    const uint32_t i = sh_ticket;
    if (i == kL0None || ((size_t)i + 1) * 4 > b.hiFlagStride)
      break;
    // diagnostics (thread 0, when b.lisStamps != nullptr): ticks per part of a region
    const bool stamps = b.lisStamps != nullptr && tid == 0;
    uint64_t st0 = stamps ? __builtin_readcyclecounter() : 0, st1 = 0;
    {   // region i: bits [a, a + SR) of the phase
      uint64_t off = 0;
      uint32_t sz = kHiFirstRegion, j = 0;
      for (; j < i && sz < SRmax; j++, sz <<= 1)
        off += sz;
      if (j < i)
        off += (uint64_t)(i - j) * SRmax;
      SR = min(sz, SRmax);
      if (j < i)
        SR = SRmax;
      W = SR + b.hiAhead;
      a = S0 + off;
    }
    wq0 = (uint32_t)(a & 63);
    {
      const uint64_t w0 = a >> 6;
      uint64_t any = 0;
      const uint32_t nWordsNow = W / 64 + 4;
      for (uint32_t k = tid; k < nWordsNow; k += kTabThreads) {
        const uint64_t idx = w0 + k;
        const uint64_t v = idx < nwordsAvail ? words[idx] : 0ull;
        wbits[k] = v;
        any |= v;
      }
      if (any)
        sh_any = 1;   // (benign race: everybody writes 1)
    }
    if (tid < 64) {
      // the level the chain was last seen in: its list's class and the next one get pointer-jump
      // tables; the class tables are built for the longest chain that continues this one upwards
      // (every level of a power-of-two cube), so that list changes inside the region find theirs.
      // The first wavefront, one lane per level (a loop over the levels on one lane is a
      // dependent LDS load per step).
      int hint = 0;
      uint32_t hintRem = 0;   // entries the hinted list had left when the chain was last seen (0: not known)
      {
        unsigned long long h64 = 0;
        if (lane == 0)
          h64 = __hip_atomic_load(reinterpret_cast<unsigned long long*>(&s.hiHint), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int hi32 = __shfl((int)(h64 >> 32), 0, 64);
        hintRem = (uint32_t)__shfl((int)(uint32_t)h64, 0, 64);
        hint = ((hi32 >> 8) != p + 1) ? -1 : (hi32 & 0xff);
      }
      if (hint < 0 || hint >= (int)t.nlevels || sh_len[hint] == 0) {
        hint = next_level((int)t.nlevels);
        hintRem = 0;
      }
      // A list that had more entries left than the region has bits cannot end inside it (an entry is a bit at
      // least): no class tables beyond its own chain then -- a quarter of the table work of the heavy planes,
      // whose lists hold 10^5 entries and whose regions the workgroups' table building bounds at 64 chunks
      // (several regions may have been handed out since the hint was left: each takes at most TS bits)
      const int specExtra = (hintRem > 8u * TS) ? 0 : kSpecExtra;
      const int lv_ = (int)lane;
      const bool inr = lv_ < (int)t.nlevels;
      // (class chains are compared up to the classes the tables have room for: the sets above that are
      //  walked into bit by bit, whatever their list -- round 3)
      const int Kl = inr ? min((int)sh_lc[lv_].K, Kcap) : 0;
      const bool usable = inr && sh_len[lv_] != 0 && sh_lc[lv_].regular;
      int best = hint;
      if (hint >= 0) {
        const int AK = min((int)sh_lc[hint].K, Kcap);
        int bestK = AK;
        uint64_t cm = __ballot(usable && lv_ < hint && Kl > AK && Kl <= AK + specExtra);
        while (cm) {   // from the level below the hint downwards, as long as the chains get longer
          const int lv = 63 - __builtin_clzll(cm);
          cm &= ~(1ull << lv);
          const int Kv = __shfl(Kl, lv, 64);
          if (Kv <= bestK)
            continue;
          // (lane j: class j of level lv against the hint's and the best level's)
          const int j = (int)lane;
          bool diff = false;
          if (j < AK)
            diff = sh_lc[hint].arity[j] != sh_lc[lv].arity[j] || sh_lc[hint].lev[j] != sh_lc[lv].lev[j];
          if (j < bestK)
            diff = diff || sh_lc[best].arity[j] != sh_lc[lv].arity[j] || sh_lc[best].lev[j] != sh_lc[lv].lev[j];
          if (__ballot(diff) == 0ull) {
            best = lv;
            bestK = Kv;
          }
        }
      }
      if (lane == 0) {
        sh_tabLevel = best;
        int hk = hint >= 0 ? (int)sh_lc[hint].K : 0;
        if (hint >= 0 && best >= 0) {
          // short lists are walked entry by entry: the table goes to the first longer list the class tables serve
          const int bestK = min((int)sh_lc[best].K, Kcap);
          for (int lv2 = hint; lv2 >= 0; lv2 = next_level(lv2)) {
            const int K2 = (int)sh_lc[lv2].K;
            if (K2 > bestK || !sh_lc[lv2].regular)
              break;
            if (sh_len[lv2] > kHiSerial) {
              hk = K2;
              break;
            }
          }
        }
        sh_hintK = hk;
      }
    }
    __syncthreads();
    uint64_t dbg0 = stamps ? __builtin_readcyclecounter() : 0, dbg1 = 0, dbg2 = 0;
    // ---- speculative tables (nothing to build over a region of zeros: there every item is one
    //      insignificant bit, which the chain handles without tables)
    const bool zeroRegion = sh_any == 0;
    if (!zeroRegion && sh_tabLevel >= 0) {
      const int lv = sh_tabLevel;
      const int K = min((int)sh_lc[lv].K, Kcap), KhT = sh_hintK, Kh = min(KhT, Kcap);
      build_tables(lv, 0, K, 0, hiCand);
      if (stamps) dbg1 = __builtin_readcyclecounter();
      const bool hop1 = KhT <= Kcap;   // the hinted list's entries (class KhT - 1) have a table
      if (hop1)
        build_hop(hop, Kh - 1);
      const bool hop2Now = hop1 && Kh < K && b.hiHop2;
      if (hop2Now)
        build_hop(hop2, Kh);
      if (stamps) dbg2 = __builtin_readcyclecounter();
      if (tid < 64) {   // which lists the tables serve: one lane per level
        const uint64_t m = serve_mask(lv, K);
        if (tid == 0) {
          sh_tabK = K;
          sh_hopTop[0] = hop1 ? Kh - 1 : -1;
          sh_hopTop[1] = hop2Now ? Kh : -1;
          sh_serve = m;
        }
      }
    }
    else if (tid == 0) {
      sh_tabLevel = -1;
      sh_tabK = 0;
      sh_hopTop[0] = sh_hopTop[1] = -1;
      sh_serve = 0;
    }
    if (stamps)
      st1 = __builtin_readcyclecounter();
    if (stamps && dbg2) {
      unsigned long long* o_ = reinterpret_cast<unsigned long long*>(b.lisStamps + (size_t)c * 64);
      atomicAdd(o_ + 40, (unsigned long long)(dbg0 - st0));    // load + decide
      atomicAdd(o_ + 41, (unsigned long long)(dbg1 - dbg0));   // class tables
      atomicAdd(o_ + 42, (unsigned long long)(dbg2 - dbg1));   // hop tables
      atomicAdd(o_ + 43, (unsigned long long)(st1 - dbg2));    // serve mask
    }
    __syncthreads();

    // ---- the chain: wavefront 0 alone (no workgroup barrier on the serial path); it comes back
    //      when it is through the region or needs all hands (tables it does not have)
    for (;;) {
      if (wave == 0) {
        if (!sh_haveState) {
          // look back (lanes 0..3 take one word each)
          if (lane < 4) {
            unsigned long long f = 0;
            if (i > 0) {
              uint32_t spins = 0;
        uint64_t spinT0 = 0;
              for (;;) {
                f = __hip_atomic_load(flags + (size_t)(i - 1) * 4 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((f >> kHiTagShift) == (unsigned long long)(p + 1))
                  break;
                if ((++spins & 15u) == 0 &&
                    __hip_atomic_load(&s.hiPlaneP1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p + 1) {
                  f = tag | (1ull << 56);   // the phase is over
                  break;
                }
                if (spin_expired(spins, spinT0)) {   // (a minute of wall time: the device has stopped making progress)
                  s.error = kErrLookBackTimeout;
                  __hip_atomic_store(&s.hiPlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  f = tag | (1ull << 56);
                  break;
                }
              }
            }
            sh_in[lane] = f;
          }
          HI_WAVE_SYNC();
          if (lane == 0) {
            if (b.lisStamps)
              sh_t2 = __builtin_readcyclecounter();
            uint32_t stop = 0;
            if (i == 0) {
              const int lv = next_level((int)t.nlevels);
              sh_level = lv < 0 ? 0u : (uint32_t)lv;
              sh_depth = 1;
              sh_e = 0;
              sh_rem = lv < 0 ? 0u : sh_len[lv];
              sh_pos = S0;
              sh_baseIdx = sh_baseIdx0 = 0;
              sh_depth0 = 1;
              if (lv < 0)
                stop = 2;   // nothing for this kernel to decode: the phase ends where it starts
            }
            else {
              const unsigned long long f0 = sh_in[0], f1 = sh_in[1], f2 = sh_in[2], f3 = sh_in[3];
              if (((f0 | f1 | f2 | f3) >> 56) & 1ull)
                stop = 1;
              else {
                sh_depth = (uint32_t)(f0 >> 52) & 15u;
                sh_level = (uint32_t)(f0 >> 46) & 63u;
                sh_pos = S0 + (f0 & ((1ull << 46) - 1ull));
                sh_e = (uint32_t)(f1 >> 28) & 0xfffffffu;
                sh_rem = (uint32_t)f1 & 0xfffffffu;
                sh_baseIdx = sh_baseIdx0 = (uint32_t)(f3 & kHiPayloadMask);
                // (a state that does not add up cannot be followed: give up loudly)
                if (sh_level >= t.nlevels || sh_depth == 0 || sh_depth > (uint32_t)kHiFrames ||
                    sh_e + sh_rem != sh_len[sh_level] || sh_pos < a ||
                    sh_depth > (uint32_t)sh_lc[sh_level].K) {
                  s.error = 1;
                  __hip_atomic_store(&s.hiPlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  stop = 1;
                  sh_depth = 1;
                  sh_level = 0;
                }
                // the frames of the sets being walked into: frame d holds the children of the set that frame
                // d - 1 entered (frame 1: of the list entry); their nodes are hi_replay's business
                const LevelClass& C = sh_lc[sh_level];
                sh_depth0 = sh_depth;
                for (uint32_t d = 1; d < sh_depth; d++) {
                  const uint32_t fr = (uint32_t)(f2 >> (5 * (d - 1))) & 31u;
                  const int pcls = (int)C.K - (int)d;   // class of the parent set
                  sh_fr[d] = (((uint32_t)C.arity[pcls] - (fr & 15u)) & 0xffu) | ((fr & 15u) << 8) | ((fr >> 4) << 16) |
                             ((uint32_t)(pcls - 1) << 24);
                  sh_ord0[d] = (uint8_t)(fr & 15u);
                }
              }
            }
            sh_stop = stop;
            sh_haveState = 1;
          }
          HI_WAVE_SYNC();
        }
        // (LDS traffic of one wavefront is in order: what lane 0 wrote above is what the lanes read)
        uint64_t tmark = b.lisStamps ? __builtin_readcyclecounter() : 0;
        // The entries a pointer-jump pass has stepped over are queued (and their significance bits set) by all
        // lanes, block by block (P3 below).  Nothing on the chain depends on that unless the list ENDS inside the
        // region (then P3 finds where): otherwise it is put off until the region's state is published -- the
        // successor's chain starts a few thousand cycles earlier -- or until the tables it reads are about to change.
        // wavefront 0, all lanes: the nodes, birth records and queue items of what the log holds (see sh_fr above)
        auto hi_replay = [&]() {
          if (!sh_parInit) {   // the sets the open frames of the region's start belong to
            if (lane == 0) {
              const uint32_t d0 = sh_depth0;
              if (d0 > 1) {
                uint64_t parent = lisCur[sh_baseIdx0];
                sh_par[1] = parent;
                for (uint32_t d = 1; d + 1 < d0; d++) {
                  parent = reg_child_packed(t, unpack_node(parent), (uint32_t)sh_ord0[d] - 1u);
                  sh_par[d + 1] = parent;
                }
              }
              sh_parInit = 1;
            }
            HI_WAVE_SYNC();
          }
          const uint32_t nlog = sh_nlog;
          for (uint32_t j = 0; j < nlog; j++) {
            const uint32_t hdr = sh_logHdr[j];
            if (hdr & 1u) {   // a list entry was entered: the set at the bottom of the stack from here on
              if (lane == 0) {
                const uint32_t ei = sh_logBase[j];
                sh_par[1] = lisCur[ei];
                atomic_or64(sigbits + (ei >> 6), 1ull << (ei & 63));
              }
              HI_WAVE_SYNC();
              continue;
            }
            const uint32_t fi = (hdr >> 4) & 15u, firstOrd = (hdr >> 8) & 15u, done = (hdr >> 12) & 15u;
            const uint32_t cls = (hdr >> 16) & 15u, lv = (hdr >> 20) & 63u;
            if (lane < done) {
              const uint32_t v = sh_logLane[j][lane];
              const uint32_t kind = v & 3u, at = v >> 2;
              if (kind) {
                const uint64_t kid = reg_child_packed(t, unpack_node(sh_par[fi]), firstOrd + lane);
                if (kind == 1) {
                  const uint32_t lev = sh_lc[lv].lev[cls];
                  if (born_counts(lev, a + at))
                    write_born(born_slots(1u), lev, a + at, kid);
                }
                else if (kind == 2) {
                  const uint32_t slot = atomicAdd(&sh_qn[0], 1u);
                  if (slot < b.queueCap) {
                    qbuf[0][slot * 2] = kid;
                    qbuf[0][slot * 2 + 1] = ((uint64_t)at << 8) | ((uint64_t)cls << 1);
                  }
                }
                else
                  sh_par[fi + 1] = kid;
              }
            }
            HI_WAVE_SYNC();
          }
          if (lane == 0)
            sh_nlog = 0;
          HI_WAVE_SYNC();
        };
        bool pend = false;
        const uint32_t* pendHp = nullptr;
        uint32_t pendK = 0, pendRemaining = 0, pendE0 = 0, pendLOff = 0;
        auto emit_entries = [&](const uint32_t* hp, uint32_t K, uint32_t remaining, uint32_t e0, uint32_t lOff) {
          const uint16_t* Utop = Uu + (size_t)(K - 1) * TS;
          const uint32_t nblk = ((SR - 1 + wq0) >> 6) + 1;
          const int32_t rbase = -(int32_t)wq0;
          for (uint32_t kb = lane; kb < nblk; kb += 64) {  // P3: the blocks emit their entries
            const uint32_t eb = blkEB[kb];
            if (eb == 0xffffffffu)
              continue;
            uint32_t r = eb & 0xffffu;
            const uint32_t base = eb >> 16;
            const uint32_t cn = hp[(uint32_t)((int32_t)r - rbase)] >> 16;
            const uint32_t lim_k = min(cn, remaining - base);
            for (uint32_t k = 0; k < lim_k; k++) {
              const uint32_t u = Utop[r];
              if (u & 0x8000u) {
                const uint32_t ei = lOff + e0 + base + k;   // index into the chunk's list storage
                const uint32_t slot = atomicAdd(&sh_qn[0], 1u);
                if (slot < b.queueCap) {
                  qbuf[0][slot * 2] = ei;
                  qbuf[0][slot * 2 + 1] = ((uint64_t)(r + 1) << 8) | ((uint64_t)(K - 1) << 1) | 1ull;
                }
                atomic_or64(sigbits + (ei >> 6), 1ull << (ei & 63));
              }
              r += u & 0x7fffu;
            }
            if (lim_k > 0 && base + lim_k == remaining)
              sh_newr = r;  // the list ended in this block (only one block satisfies this)
          }
        };
        // The chain's scalars -- position, list level, frames open, entry index, entries left -- are the same in
        // every lane and live in REGISTERS while the chain runs (until round 5: LDS words owned by lane 0, read and
        // written back around every step, a wave-wide fence each time); they go back to LDS where others look: at
        // the publish, before an all-hands table build, behind the chain.
        uint64_t cpos = sh_pos;
        uint32_t clevel = sh_level, cdepth = sh_depth, ce = sh_e, crem = sh_rem, cstop = sh_stop, cover = sh_over;
        for (;;) {
          uint32_t act = kActDone;
          for (; cstop == 0;) {   // (every lane, the same values)
            if (cpos >= a + SR)
              break;
            if (cdepth == 1 && crem == 0) {   // this list is through: the next one
              const int lv = next_level((int)clevel);
              if (lv < 0) {
                cover = 1;
                break;
              }
              clevel = (uint32_t)lv;
              ce = 0;
              crem = sh_len[lv];
            }
            if (zeroRegion && cdepth == 1) {
              // one '0' per entry, nothing splits: count them off
              const uint32_t z = min((uint32_t)(a + SR - cpos), crem);
              ce += z;
              crem -= z;
              cpos += z;
              continue;
            }
            const int lv = (int)clevel;
            const int K = sh_lc[lv].K;
            if (!((sh_serve >> lv) & 1ull)) {
              // chains that agree from the leaf class upwards share tables: classes are only added
              bool extend = sh_tabLevel >= 0;
              if (extend) {
                const LevelClass& A = sh_lc[sh_tabLevel];
                const LevelClass& B = sh_lc[lv];
                for (int j = 0; j < sh_tabK && j < (int)B.K; j++)
                  extend = extend && A.arity[j] == B.arity[j] && A.lev[j] == B.lev[j];
              }
              if (lane == 0)
                sh_tabFrom = extend ? sh_tabK : 0;
              act = (!extend && sh_tabLevel >= 0 && (sh_qn[0] != 0 || pend || sh_nlog != 0)) ? kActFlushTables : kActTables;   // (pend: entries of the tables in place still to be queued)
            }
            else if (cdepth > 1)
              act = 8;    // serial hop
            else if (K - 1 >= Kcap)
              act = 10;   // a list of sets the tables do not cover: entry by entry
            else if (sh_hopTop[0] == K - 1 || sh_hopTop[1] == K - 1)
              act = 9;    // list entries
            else if (crem <= kHiSerial)
              act = 11;   // a few entries without a pointer-jump table: one by one
            else
              act = kActHopTab;
            break;
          }
          HI_T(0);
          if (act < 8) {   // through the region, or all hands needed
            if (lane == 0) {
              sh_pos = cpos;   // (where the other wavefronts, and this one when it comes back, look)
              sh_level = clevel;
              sh_depth = cdepth;
              sh_e = ce;
              sh_rem = crem;
              sh_over = cover;
              if (act == kActDone && cstop != 1) {
                // ---- publish the state at the end of the region (or the end of the phase)
                const bool over = cstop == 2 || cover != 0;
                if (over) {
                  s.hiEnd = cpos;
                  for (int k = 0; k < 4; k++)
                    __hip_atomic_store(flags + (size_t)i * 4 + k, tag | (1ull << 56), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(&s.hiPlaneP1, p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                else {
                  unsigned long long fr = 0;
                  for (uint32_t d = 1; d < cdepth; d++) {
                    const uint32_t fw = sh_fr[d];
                    fr |= (unsigned long long)(((fw >> 8) & 15u) | (((fw >> 16) & 1u) << 4)) << (5 * (d - 1));
                  }
                  const unsigned long long f0 = tag | ((unsigned long long)cdepth << 52) |
                                                ((unsigned long long)clevel << 46) |
                                                (unsigned long long)(cpos - S0);
                  const unsigned long long f1 = tag | ((unsigned long long)ce << 28) | (unsigned long long)crem;
                  __hip_atomic_store(flags + (size_t)i * 4 + 0, f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(flags + (size_t)i * 4 + 1, f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(flags + (size_t)i * 4 + 2, tag | fr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(flags + (size_t)i * 4 + 3, tag | (unsigned long long)sh_baseIdx, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
                  __hip_atomic_store(reinterpret_cast<unsigned long long*>(&s.hiHint),
                                     ((unsigned long long)(((p + 1) << 8) | (int)clevel) << 32) | (unsigned long long)crem,
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sh_stop = over ? 3u : cstop;   // (3: published the end of the phase)
                if (b.lisStamps)
                  sh_t3 = __builtin_readcyclecounter();
              }
              else
                sh_stop = cstop;
              sh_action = act;
            }
            HI_T(1);
            if (pend) {   // (after the state is out; before the workgroup expands the queue or rebuilds tables)
              emit_entries(pendHp, pendK, pendRemaining, pendE0, pendLOff);
              pend = false;
              HI_T(4);
            }
            if (sh_nlog)
              hi_replay();
            break;
          }
          const LevelClass& C = sh_lc[clevel];
          const int K = C.K;
          if (act == 8) {
            // The sets being walked into (the table method's serial hop); an item that leaves the
            // tables is entered, so the chain always reaches the region's end.  The whole wavefront
            // runs this with uniform values: the children of a frame are stepped over with one table
            // look-up each; what became of child k and where goes to the log (lane k), the frame's word is
            // updated, and nothing else: no node, no slot, no queue item on the chain (hi_replay).
            uint32_t r = (uint32_t)(cpos - a);
            int depth = (int)cdepth;
            const uint32_t lvNow = clevel;
            while (depth > 1 && r < SR) {
              const uint32_t fw = sh_fr[depth - 1];
              if ((fw & 0xffu) == 0) {
                depth--;
                continue;
              }
              const int cls = (int)(fw >> 24);
              const uint32_t n = fw & 0xffu, ord0 = (fw >> 8) & 0xffu;
              uint32_t y = r, found = (fw >> 16) & 1u, done = 0;
              uint32_t myKind = 0, myPos = 0;   // 1: born at myPos, 2: splits from myPos on (queued), 3: entered
              bool enter = false;
              for (uint32_t k = 0; k < n && y < SR; k++) {
                const bool coded = found || (n - k) > 1;
                uint32_t kind, at;
                if (coded) {
                  // (children of a class without a table: their test bit decides, a significant one
                  //  is entered)
                  const uint32_t u = cls < Kcap ? (uint32_t)Uu[(size_t)cls * TS + y] : (bit_at(y) ? (uint32_t)kTInf : 1u);
                  if (u == 1) {
                    kind = 1;
                    at = y;
                    y += 1;
                  }
                  else {
                    found = 1;
                    at = y + 1;
                    if (u == kTInf) {
                      kind = 3;
                      y = at;
                    }
                    else {
                      kind = 2;
                      y += u & 0x7fffu;
                    }
                  }
                }
                else {
                  found = 1;
                  at = y;
                  const uint32_t len = cls < Kcap - 1 ? (uint32_t)Tt[(size_t)cls * TS + y] : (uint32_t)kTInf;
                  if (len == kTInf)
                    kind = 3;
                  else {
                    kind = 2;
                    y += len;
                  }
                }
                if (lane == k) {
                  myKind = kind;
                  myPos = at;
                }
                done = k + 1;
                if (kind == 3) {
                  enter = true;
                  break;
                }
              }
              if (enter && (cls <= 0 || depth >= kHiFrames)) {   // cannot happen
                if (lane == 0)
                  s.error = 1;
                r = SR;
                break;
              }
              uint32_t j = sh_nlog;
              if (j >= (uint32_t)kHiLogCap) {   // (the log is full: work it off here, for once on the chain)
                hi_replay();
                j = 0;
              }
              if (lane < 8u)
                sh_logLane[j][lane] = myKind | (myPos << 2);
              if (lane == 0) {
                sh_logHdr[j] = ((uint32_t)(depth - 1) << 4) | (ord0 << 8) | (done << 12) | ((uint32_t)cls << 16) | (lvNow << 20);
                sh_nlog = j + 1;
                sh_fr[depth - 1] = (n - done) | ((ord0 + done) << 8) | (found << 16) | ((uint32_t)cls << 24);
                if (enter)
                  sh_fr[depth] = (uint32_t)C.arity[cls] | ((uint32_t)(cls - 1) << 24);
              }
              HI_WAVE_SYNC();
              r = y;
              if (enter)
                depth++;
            }
            cpos = a + r;
            cdepth = (uint32_t)depth;
            HI_T(2);
            continue;
          }
          if (act == 10) {
            // A list whose sets are larger than the tables' classes (at most a few hundred entries per
            // chunk, each of which splits once in its life): '0' entries are counted off 32 at a time,
            // a '1' entry is walked into like an entry that leaves the region.
            if (sh_nlog >= (uint32_t)kHiLogCap)   // (room for the entry it may enter)
              hi_replay();
            uint32_t r = (uint32_t)(cpos - a), e = ce, rem = crem, depth = 1;
            if (lane == 0) {
              const uint32_t lOff = sh_lOff[clevel];
              while (rem > 0 && r < SR) {
                const uint32_t lim = min(rem, SR - r);
                const uint32_t v = bits32(r);
                const uint32_t z = v ? (uint32_t)__ffs((int)v) - 1u : 32u;
                if (z >= lim) {
                  e += lim;
                  rem -= lim;
                  r += lim;
                  break;
                }
                e += z;
                rem -= z;
                r += z;
                if (z == 32)
                  continue;
                const uint32_t ei = lOff + e;
                sh_fr[1] = (uint32_t)C.arity[K - 1] | ((uint32_t)(K - 2) << 24);
                sh_baseIdx = ei;   // (the entry itself is read, and its significance bit set, by hi_replay)
                sh_logHdr[sh_nlog] = 1u;
                sh_logBase[sh_nlog] = ei;
                sh_nlog++;
                e++;
                rem--;
                r += 1;   // its '1'
                depth = 2;
                break;
              }
            }
            cpos = a + (uint32_t)__builtin_amdgcn_readfirstlane((int)r);   // (lane 0's, in every lane)
            ce = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
            crem = (uint32_t)__builtin_amdgcn_readfirstlane((int)rem);
            cdepth = (uint32_t)__builtin_amdgcn_readfirstlane((int)depth);
            HI_WAVE_SYNC();
            HI_T(2);
            continue;
          }
          if (act == 11) {
            // The rest of a short list, entry by entry: an insignificant one is a bit, a significant one whose split
            // lies within the tables is queued like the pointer-jump pass queues it, one that leaves them is entered.
            if (sh_nlog >= (uint32_t)kHiLogCap)   // (room for the entry it may enter)
              hi_replay();
            uint32_t r = (uint32_t)(cpos - a), e = ce, rem = crem, depth = 1;
            if (lane == 0) {
              const uint16_t* Utop = Uu + (size_t)(K - 1) * TS;
              const uint32_t lOff = sh_lOff[clevel];
              while (rem > 0 && r < SR) {
                const uint32_t u = Utop[r];
                if (u == 1u) {
                  e++;
                  rem--;
                  r++;
                  continue;
                }
                const uint32_t ei = lOff + e;
                if (u == kTInf) {
                  sh_fr[1] = (uint32_t)C.arity[K - 1] | ((uint32_t)(K - 2) << 24);
                  sh_baseIdx = ei;
                  sh_logHdr[sh_nlog] = 1u;
                  sh_logBase[sh_nlog] = ei;
                  sh_nlog++;
                  e++;
                  rem--;
                  r += 1;   // its '1'
                  depth = 2;
                  break;
                }
                const uint32_t slot = atomicAdd(&sh_qn[0], 1u);
                if (slot < b.queueCap) {
                  qbuf[0][slot * 2] = ei;
                  qbuf[0][slot * 2 + 1] = ((uint64_t)(r + 1) << 8) | ((uint64_t)(K - 1) << 1) | 1ull;
                }
                atomic_or64(sigbits + (ei >> 6), 1ull << (ei & 63));
                e++;
                rem--;
                r += u & 0x7fffu;
              }
            }
            cpos = a + (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
            ce = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
            crem = (uint32_t)__builtin_amdgcn_readfirstlane((int)rem);
            cdepth = (uint32_t)__builtin_amdgcn_readfirstlane((int)depth);
            HI_WAVE_SYNC();
            HI_T(2);
            continue;
          }
          // ---- act == 9: the list entries from sh_pos on, by pointer jumping
          if (pend) {   // (this pass rewrites the blocks' entry points)
            emit_entries(pendHp, pendK, pendRemaining, pendE0, pendLOff);
            pend = false;
            HI_WAVE_SYNC();
          }
          const uint32_t* hp = sh_hopTop[0] == K - 1 ? hop : hop2;
          const uint32_t pr = (uint32_t)(cpos - a);
          const uint32_t remaining = crem, e0 = ce;
          const uint32_t lOff = sh_lOff[clevel];
          const uint32_t nblk = ((SR - 1 + wq0) >> 6) + 1;
          const int32_t rbase = -(int32_t)wq0;
          for (uint32_t k = lane; k < nblk; k += 64)
            blkEB[k] = 0xffffffffu;
          HI_WAVE_SYNC();
          if (lane == 0) {  // P2: walk the blocks
            uint32_t r = pr, total = 0, stopped = 0, newr = 0xffffffffu;
            while (true) {
              if (r >= SR) {
                newr = r;
                break;
              }
              const uint32_t h = (uint32_t)((int32_t)r - rbase);
              const uint32_t v = hp[h];
              const uint32_t cn = v >> 16;
              blkEB[h >> 6] = r | (total << 16);
              if (total + cn >= remaining) {
                total = remaining;
                break;
              }
              total += cn;
              if (v & 0x8000u) {
                stopped = 1;
                newr = v & 0x7fffu;
                break;
              }
              r = v & 0x7fffu;
            }
            sh_total = total;
            sh_stopped = stopped;
            sh_newr = newr;
          }
          HI_WAVE_SYNC();
          HI_T(3);
          if (sh_newr == 0xffffffffu) {   // the list ends inside the region: where, P3 finds out
            emit_entries(hp, (uint32_t)K, remaining, e0, lOff);
          }
          else {
            pend = true;
            pendHp = hp;
            pendK = (uint32_t)K;
            pendRemaining = remaining;
            pendE0 = e0;
            pendLOff = lOff;
          }
          HI_WAVE_SYNC();
          HI_T(4);
          if (sh_nlog >= (uint32_t)kHiLogCap)   // (room for the entry P4 may enter)
            hi_replay();
          {  // P4 (every lane: the same LDS words in, the same values out)
            const uint32_t total = sh_total;
            uint32_t e = e0 + total;
            uint32_t rem = remaining - total;
            uint32_t r = sh_newr;
            uint32_t depth = 1;
            if (r == 0xffffffffu || e + rem != sh_len[clevel]) {   // cannot happen
              if (lane == 0)
                s.error = 1;
              r = SR;
              rem = sh_len[clevel] - min(e, sh_len[clevel]);
            }
            else if (sh_stopped && rem > 0) {
              // the entry at r leaves the region: walk into it
              const uint32_t ei = lOff + e;
              if (lane == 0) {
                sh_fr[1] = (uint32_t)C.arity[K - 1] | ((uint32_t)(K - 2) << 24);
                sh_baseIdx = ei;   // (the entry itself is read, and its significance bit set, by hi_replay)
                sh_logHdr[sh_nlog] = 1u;
                sh_logBase[sh_nlog] = ei;
                sh_nlog++;
              }
              e++;
              rem--;
              r += 1;  // its '1'
              depth = 2;
            }
            crem = rem;
            ce = e;
            cpos = a + r;
            cdepth = depth;
          }
          HI_WAVE_SYNC();
          HI_T(5);
        }
      }
      __syncthreads();
      uint32_t act = sh_action;
      if (act == kActDone)
        break;
      if (act == kActFlushTables) {   // sets of the previous chain are queued: expand them with its
        expand_all();                 // tables before those are replaced
        act = kActTables;
      }
      if (act == kActTables) {
        const int lv = (int)sh_level;
        const int K = min((int)sh_lc[lv].K, Kcap);
        const int j0 = sh_tabFrom;
        build_tables(lv, j0, K, (uint32_t)min((uint64_t)(sh_pos - a), (uint64_t)SR));
        if (tid < 64) {
          const uint64_t m = serve_mask(lv, K);
          if (tid == 0) {
            sh_tabLevel = lv;
            sh_tabK = K;
            sh_hopTop[0] = sh_hopTop[1] = -1;
            sh_serve = m;
          }
        }
      }
      else {   // kActHopTab
        const int K = sh_lc[sh_level].K;
        build_hop(hop, K - 1, (uint32_t)min((uint64_t)(sh_pos - a), (uint64_t)SR));
        if (tid == 0)
          sh_hopTop[0] = K - 1;
      }
      if (tid == 0 && b.lisStamps)
        atomicAdd(reinterpret_cast<unsigned long long*>(b.lisStamps + (size_t)c * 64) + (act == kActTables ? 5 : 6), 1ull);
      __syncthreads();
    }
    const bool last = sh_stop == 1 || sh_stop == 3;
    if (sh_stop == 1)
      break;
    // ---- off the chain: the sets that split in this region
    if (sh_tabLevel >= 0)
      expand_all();
    if (stamps) {
      unsigned long long* out = reinterpret_cast<unsigned long long*>(b.lisStamps + (size_t)c * 64);
      const uint64_t st4 = __builtin_readcyclecounter();
      atomicAdd(out + 0, 1ull);
      atomicAdd(out + 1, st1 - st0);          // load + speculative tables
      atomicAdd(out + 2, sh_t2 - st1);        // look-back wait
      atomicAdd(out + 3, sh_t3 - sh_t2);      // on the chain
      atomicAdd(out + 4, st4 - sh_t3);        // expansion
      atomicAdd(out + 9, zeroRegion ? 1ull : 0ull);
      for (int k = 0; k < 6; k++)
        atomicAdd(out + 10 + k, (unsigned long long)sh_tacc[k]);
    }
    if (last)
      break;
    __syncthreads();   // LDS is reused by the next region
  }
  __syncthreads();
  if (tid == 0 && blockIdx.x < 8) {
    s.hiBornCnt[blockIdx.x] = min(min(sh_segBorn, sh_segBornEnd), b.bornSeg);
    s.hiLeafCnt[blockIdx.x] = min(sh_segLeaf, b.leafSeg);
  }
}

// Old entries of the lists k_lis_hi decoded that stayed insignificant keep their order
// (SPECK3D_INT.cpp:12-20): one workgroup per (level, chunk) compacts by the significance bits.
__global__ void __launch_bounds__(kTabThreads) k_lis_compact(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y, l = blockIdx.x;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  // (the GPU-wide kernels of the two smallest set sizes kept their own survivors)
  const bool skip = (b.l0Level >= 0 && s.l0PlaneP1 == p + 1 && (int)l == b.l0Level) ||
                    (b.l1Level >= 0 && s.l1PlaneP1 == p + 1 && (int)l == b.l1Level) ||
                    (b.l2Level >= 0 && s.l2PlaneP1 == p + 1 && (int)l == b.l2Level);
  __shared__ uint32_t sh_scan[kTabThreads / 64 + 1];
  __shared__ uint32_t sh_stay[kTabThreads], sh_ex[kTabThreads];
  const uint32_t cur = s.cur, nx = cur ^ 1u;
  // A phase that ran into the stream's end was the chunk's last (SPECK_INT.cpp:200-201): nobody reads its lists or
  // its significance bits again (the bits are cleared before every decode).  Round 6: without this test the planes
  // on which most chunks of a batch end -- 17 and 19 of the bench volume's 20 -- spent 1.1 and 1.4 ms here: a chunk
  // whose stream ends inside k_lis_l0's list leaves the 4x4x4 sets' list, 10^5 entries, to this kernel
  // (k_lis_l1 returns at once, k_lis_hi finds the end), one workgroup copying it for nothing.
  const bool ended = s.hiEnd >= s.avail;
  const uint32_t n = (skip || ended) ? 0u : s.listLen[cur][l];
  const uint32_t lOff = b.levelOff[l];
  const uint64_t* __restrict__ list = b.lis[cur] + c * b.lisStride;
  uint64_t* __restrict__ keep = b.lis[nx] + c * b.lisStride + lOff;
  uint64_t* sigbits = b.sigbits + c * b.sigbitsStride;
  const int tid = threadIdx.x;
  uint32_t carry = 0;
  // the level's bits start at bit lOff of the chunk's bit array
  for (uint32_t base = 0; base < n; base += kTabThreads * 32) {
    const uint32_t i0 = base + (uint32_t)tid * 32;
    uint32_t stay = 0;
    if (i0 < n) {
      const uint32_t gi = lOff + i0;
      const uint64_t lo = sigbits[gi >> 6], hi = sigbits[(gi >> 6) + 1];
      const uint32_t sh = gi & 63u;
      const uint32_t sig = (uint32_t)(sh ? (lo >> sh) | (hi << (64 - sh)) : lo);
      stay = ~sig;
      const uint32_t valid = n - i0;
      if (valid < 32)
        stay &= (1u << valid) - 1u;
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<uint32_t>((uint32_t)__popc(stay), sh_scan, &total) + carry;
    // the copy with all threads along the list (a thread copying its own 32 entries one after the
    // other made the lists of 10^4 entries a chain of dependent round trips: 0.15 ms per launch)
    __syncthreads();
    sh_stay[tid] = stay;
    sh_ex[tid] = ex;
    __syncthreads();
    const uint32_t nHere = min(n - base, (uint32_t)kTabThreads * 32u);
    // (eight entries per thread in flight: with a load behind every conditional store the loop was a chain of 32
    //  dependent round trips per 8192 entries, 90 us a round -- the 8x8x8 sets' list of plane 14 took 0.37 ms)
    for (uint32_t i0 = (uint32_t)tid; i0 < nHere; i0 += 8u * kTabThreads) {
      uint64_t v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t i = i0 + (uint32_t)k * kTabThreads;
        v[k] = i < nHere ? list[lOff + base + i] : 0ull;
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t i = i0 + (uint32_t)k * kTabThreads;
        if (i >= nHere)
          break;
        const uint32_t st = sh_stay[i >> 5], bit = i & 31u;
        if ((st >> bit) & 1u)
          keep[sh_ex[i >> 5] + (uint32_t)__popc(st & ((1u << bit) - 1u))] = v[k];
      }
    }
    carry += total;
  }
  __syncthreads();
  // leave the bits clean for the next plane (neighbouring levels share words: atomic)
  for (uint32_t i0 = (uint32_t)tid * 64; i0 < n + 64; i0 += kTabThreads * 64) {
    const uint32_t g0 = lOff + i0, g1 = min(lOff + n, g0 + 64);
    if (g0 >= lOff + n)
      break;
    // bits [g0, g1) span at most two words
    for (uint32_t g = g0; g < g1;) {
      const uint32_t w = g >> 6, upto = min(g1, (w + 1) * 64);
      const uint64_t m = ((upto - w * 64 == 64) ? ~0ull : ((1ull << (upto - w * 64)) - 1ull)) &
                         ~((1ull << (g & 63)) - 1ull);
      atomicAnd(reinterpret_cast<unsigned long long*>(sigbits + w), ~m);
      g = upto;
    }
  }
  if (tid == 0) {
    if (!skip)
      s.listLen[nx][l] = carry;
    // the workgroup that finishes last sets the chunk's state after the phase (what the one-workgroup table kernel's
    // last lines do): nobody reads the current lists any more.  (No fence in front of the count: the last workgroup
    // reads nothing the others of this launch wrote -- its inputs are k_lis_hi's, a kernel ago -- and what all of them
    // write is for the next kernel.  Rounds 2-4 had a __threadfence() here: an agent-scope release per workgroup,
    // 25 levels x 32 chunks of them per plane, each writing back its XCD's L2.)
    if (atomicAdd(&s.hiCompactDone, 1u) == gridDim.x - 1) {
      const uint64_t phase0 = s.lipStart + s.lipBits;
      const uint64_t maskBits = (uint64_t)b.maskWords * 64;
      const uint64_t end = s.hiEnd;
      s.cur = nx;
      s.pos = end;
      s.nLeafEv = min(s.leafCount, b.leafCap);
      s.bornCount = min(s.bornCount, (uint32_t)b.bornStride);
      s.lisPhaseBits = min(end - phase0, maskBits);
      s.lastPlane = p;
      if (end >= s.avail)  // SPECK_INT.cpp:200-201
        s.done = 1;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Placement of the sets born in the plane the list kernels just decoded: a set joins
// the list of its level behind the entries that survived, in the order of the stream positions at
// which the sets were born (= the reference's append order).  Every birth set one bit of its
// level's position mask; the rank of a birth is the number of mask bits before its own.
// ------------------------------------------------------------------------------------------
#define PLACE_ACTIVE_OR_RETURN(s, p)                                                   \
  if (!(s).active || (int)(p) >= (s).nbp || (s).lastPlane != (int)(p) || (s).done)     \
    return;   /* (after the stream has ended nobody reads the lists again) */

// one workgroup per (mask slot, chunk): popcount prefix of the slot's mask
__global__ void __launch_bounds__(kTabThreads) k_place_scan(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y, slot = blockIdx.x;
  DecState& s = b.st[c];
  PLACE_ACTIVE_OR_RETURN(s, p);
  __shared__ uint32_t sh_scan[kTabThreads / 64 + 1];
  const uint32_t pw = (uint32_t)((s.lisPhaseBits + 63) / 64);
  const uint64_t* mask = b.mask + c * b.maskStride + (size_t)slot * b.maskWords;
  uint32_t* pre = b.maskPrefix + c * b.prefStride + (size_t)slot * b.prefWords;   // (one word per four mask words)
  constexpr int kPer = 4;   // consecutive words per thread and round: one batch of loads per scan, one prefix word
  uint32_t carry = 0;
  for (uint32_t base = 0; base < pw; base += kTabThreads * kPer) {
    const uint32_t w0 = base + threadIdx.x * kPer;
    uint32_t v[kPer], tsum = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      v[k] = w0 + k < pw ? (uint32_t)__popcll(mask[w0 + k]) : 0u;
      tsum += v[k];
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<uint32_t>(tsum, sh_scan, &total) + carry;
    if (w0 < pw)
      pre[w0 / kPer] = ex;
    carry += total;
  }
  if (threadIdx.x == 0)
    s.slotBorn[slot] = carry;
}

__global__ void __launch_bounds__(kThreads) k_place_scatter(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  PLACE_ACTIVE_OR_RETURN(s, p);
  const uint32_t cur = s.cur;   // (the list kernels have already made the next lists current)
  const uint64_t* bornPacked = b.bornPacked + c * b.bornPitch;
  const uint64_t* bornPosLev = b.bornPosLev + c * b.bornPitch;
  // the shared part, then the segments the workgroups of k_lis_hi filled
  uint32_t segEnd[9];
  segEnd[0] = s.bornCount;
  for (int g = 0; g < 8; g++)
    segEnd[g + 1] = segEnd[g] + (g < (int)b.hiGroupsMax ? s.hiBornCnt[g] : 0u);
  for (uint32_t kk = blockIdx.x * blockDim.x + threadIdx.x; kk < segEnd[8];
       kk += gridDim.x * blockDim.x) {
    uint32_t k = kk;
    if (kk >= segEnd[0]) {
      int g = 0;
      while (kk >= segEnd[g + 1])
        g++;
      k = (uint32_t)b.bornStride + (uint32_t)g * b.bornSeg + (kk - segEnd[g]);
    }
    const uint64_t pl = bornPosLev[k];
    const uint32_t lev = (uint32_t)(pl >> 48);
    const uint64_t rel = pl & ((1ull << 48) - 1);
    const uint32_t wi = (uint32_t)(rel >> 6), slot = b.levelSlot[lev];
    const size_t mo = c * b.maskStride + (size_t)slot * b.maskWords + wi;
    uint32_t rank = b.maskPrefix[c * b.prefStride + (size_t)slot * b.prefWords + (wi >> 2)] +
                    (uint32_t)__popcll(b.mask[mo] & ((1ull << (rel & 63)) - 1ull));
    for (uint32_t j = 1; j <= (wi & 3u); j++)   // (the words of the group in front of this one: the same 32 bytes)
      rank += (uint32_t)__popcll(b.mask[mo - j]);
    b.lis[cur][c * b.lisStride + b.levelOff[lev] + s.listLen[cur][lev] + rank] = bornPacked[k];
  }
}

// Turns the leaf events of one plane into pixel-mask updates: born bits for all children, sigNew
// and sign bits for the significant ones.  One thread per event, atomics merged per mask word.
__global__ void __launch_bounds__(kThreads) k_leaf_apply(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  DecState& s = b.st[c];
  if (!s.active || (int)p >= s.nbp || s.lastPlane != p)
    return;   // (runs for the plane just decoded, also when that plane ended the stream)
  // the end of the placement (after k_place_scatter): the lists grow by what was placed, the birth
  // masks are left clean for the next plane (after the stream has ended nobody reads the lists)
  if (!s.done && b.nSlots) {
    const uint32_t pw = (uint32_t)((s.lisPhaseBits + 63) / 64);
    for (uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x; wi < pw; wi += gridDim.x * blockDim.x)
      for (uint32_t slot = 0; slot < b.nSlots; slot++)
        b.mask[c * b.maskStride + (size_t)slot * b.maskWords + wi] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0)
      for (uint32_t slot = 0; slot < b.nSlots; slot++)
        s.listLen[s.cur][b.slotLevel[slot]] += s.slotBorn[slot];
  }
  uint32_t segEnd[9];
  segEnd[0] = s.nLeafEv;
  for (int g = 0; g < 8; g++)
    segEnd[g + 1] = segEnd[g] + (g < (int)b.hiGroupsMax ? s.hiLeafCnt[g] : 0u);
  const uint32_t n = segEnd[8];
  const Tree& t = b.tree;
  unsigned long long* bornM = reinterpret_cast<unsigned long long*>(b.bornM + c * b.maskPixStride);
  unsigned long long* sigNew = reinterpret_cast<unsigned long long*>(b.sigNew + c * b.maskPixStride);
  unsigned long long* sign = reinterpret_cast<unsigned long long*>(b.sign + c * b.signStride);
  const uint64_t* leafEv = b.leafEv + c * b.leafStride;
  for (uint32_t kk = blockIdx.x * blockDim.x + threadIdx.x; kk < n; kk += gridDim.x * blockDim.x) {
    uint32_t k = kk;
    if (kk >= segEnd[0]) {
      int g = 0;
      while (kk >= segEnd[g + 1])
        g++;
      k = b.leafCap + (uint32_t)g * b.leafSeg + (kk - segEnd[g]);
    }
    const uint64_t ev = leafEv[k];
    const uint32_t sigm = (uint32_t)(ev >> 32) & 0xffu, negm = (uint32_t)(ev >> 40) & 0xffu;
    Node nd;
    node_from_flat(t, (uint32_t)ev, nd);
    const Grid g = t.grids[nd.grid];
    if (g.kind & kGridLeafWord) {   // folded into the masks by k_dec_count / k_dec_fold
      b.leafState[c * b.leafStateStride + (uint32_t)ev] = (uint16_t)(sigm | (negm << 8));
      b.leafDirty[c * b.leafDirtyStride + ((uint32_t)ev >> 5)] = (uint8_t)(p + 1);   // (every writer stores the same value)
      continue;
    }
    if (!(g.kind & kGridOct)) {   // any shape (events of k_lis_mx): the existing children by ordinal
      KidBox kb;
      kid_box(t, nd, kb);
      for (uint32_t q = 0; q < kb.nk; q++) {
        const uint32_t ridx = kid_pixel_raster(t, nd, kb, q);
        const unsigned long long m = 1ull << (ridx & 63);
        atomicOr(bornM + (ridx >> 6), m);
        if ((sigm >> q) & 1u)
          atomicOr(sigNew + (ridx >> 6), m);
        if ((negm >> q) & 1u)
          atomicAnd(sign + (ridx >> 6), ~m);
      }
      continue;
    }
    const Root rt = t.roots[g.root];
    uint32_t cbase[3], cshift[3], nb = 0;
    for (int ax = 0; ax < 3; ax++) {
      if (g.depth < rt.D[ax]) {
        cbase[ax] = (uint32_t)nd.i[ax] * 2u;
        cshift[ax] = nb++;
      }
      else {
        cbase[ax] = nd.i[ax];
        cshift[ax] = 31;
      }
    }
    const uint32_t ar = 1u << nb;
    uint32_t wcur = 0xffffffffu;
    uint64_t bornBits = 0, sigBits = 0, negBits = 0;
    for (uint32_t q = 0; q <= ar; q++) {
      uint32_t w = 0xfffffffeu, bitpos = 0;
      if (q < ar) {
        const uint32_t cx = rt.org[0] + (cbase[0] | ((q >> cshift[0]) & 1u));
        const uint32_t cy = rt.org[1] + (cbase[1] | ((q >> cshift[1]) & 1u));
        const uint32_t cz = rt.org[2] + (cbase[2] | ((q >> cshift[2]) & 1u));
        const uint32_t ridx = (cz * t.dims[1] + cy) * t.dims[0] + cx;
        w = ridx >> 6;
        bitpos = ridx & 63;
      }
      if (w != wcur) {  // flush the finished word (q == ar flushes the last one)
        if (wcur != 0xffffffffu) {
          atomicOr(bornM + wcur, bornBits);
          if (sigBits)
            atomicOr(sigNew + wcur, sigBits);
          if (negBits)
            atomicAnd(sign + wcur, ~negBits);
        }
        wcur = w;
        bornBits = sigBits = negBits = 0;
      }
      if (q < ar) {
        const uint64_t m = 1ull << bitpos;
        bornBits |= m;
        if ((sigm >> q) & 1u)
          sigBits |= m;
        if ((negm >> q) & 1u)
          negBits |= m;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// refinement: the j-th pixel that was significant before this plane takes bit pos + j
// ------------------------------------------------------------------------------------------
// One lane per CANDIDATE (round 3; rounds 1-2 had one lane per sample of a mask word, of which a few
// percent are significant in the fine subbands, where 7/8 of the samples are: 0.34 against 0.25 ms per
// launch of 32 chunks).  The significant samples of a tile are first listed in LDS in raster order
// (thread = mask word: it writes its word's set bits at the word's rank, 14 bits each), then candidate r
// of the tile -- lane r -- takes stream bit base + r and its coefficient: consecutive lanes read
// consecutive bits.  A workgroup of 256 threads strides over the tiles.
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_ref_apply2(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  __shared__ uint32_t sm[kThreads / 64 + 1];
  __shared__ uint16_t list[kDecTileWords * 64];   // word << 6 | bit of every candidate, raster order
  const uint32_t nw = (b.tree.nvals + 63) / 64;
  const uint64_t* words = b.stream + c * b.streamStride;
  CT* coef = reinterpret_cast<CT*>(b.coef) + c * b.coefStride;
  const CT thr = (CT)1 << p, half = thr / 2;
  const CT initPrev = thr * 2 + thr * 2 - thr - 1;   // 1.5 * (2 thr) - 1  (SPECK_INT.cpp:462-468)
  const uint64_t avail = s.avail, pos0 = s.pos;
  for (uint32_t tile = blockIdx.x; tile < b.nPixTiles; tile += gridDim.x) {
    if (b.tileRef[c * b.tileStride + tile] == 0)
      continue;   // (uniform: the whole workgroup reads the same word)
    const uint32_t w0 = tile * kDecTileWords, wi = w0 + threadIdx.x;
    uint64_t sig = wi < nw ? b.sigOld[c * b.maskPixStride + wi] : 0ull;
    uint32_t total;
    uint32_t o = block_exclusive_scan_lds<uint32_t>((uint32_t)__popcll(sig), sm, &total);
    const uint32_t wtag = threadIdx.x << 6;
    while (sig) {
      const uint32_t j = (uint32_t)__ffsll((long long)sig) - 1u;
      sig &= sig - 1;
      list[o++] = (uint16_t)(wtag | j);
    }
    LDS_ONLY_BARRIER();
    const uint64_t base = pos0 + (uint64_t)b.tileRefOff[c * b.tileStride + tile];
    // the pass stops the moment the stream is exhausted (SPECK_INT.cpp:388-389)
    const uint32_t lim = base >= avail ? 0u : (uint32_t)min((uint64_t)total, avail - base);
    constexpr int kB = 4;   // candidates per thread and round: their loads are issued together
    for (uint32_t r0 = threadIdx.x; r0 < lim; r0 += kThreads * kB) {
      uint32_t idx[kB];
      uint64_t sw[kB];
      CT cv[kB];
#pragma unroll
      for (int u = 0; u < kB; u++) {
        const uint32_t r = r0 + (uint32_t)u * kThreads;
        const bool act = r < lim;
        idx[u] = act ? w0 * 64u + list[r] : 0xffffffffu;
        const uint64_t at = base + r;
        sw[u] = act ? (words[at >> 6] >> (at & 63)) & 1ull : 0ull;
        cv[u] = act ? coef[idx[u]] : (CT)0;
      }
#pragma unroll
      for (int u = 0; u < kB; u++) {
        if (idx[u] == 0xffffffffu)
          continue;
        CT v2 = cv[u];
        if (v2 == 0)          // first touch: found significant on the previous plane (threshold 2*thr)
          v2 = initPrev;
        if (p >= 1)
          v2 = sw[u] ? v2 + half : v2 - half;
        else if (sw[u])
          v2 += 1;
        coef[idx[u]] = v2;
      }
    }
    LDS_ONLY_BARRIER();   // (the list is rewritten for the next tile)
  }
}

// ------------------------------------------------------------------------------------------
// Refinement through bit planes (DecBuffers::refPlanes, 32-bit coefficients).  k_ref_apply2 above updates 4
// bytes here and there in a 64 MB array on every plane: whole cache lines moved for a few samples each, 10 GB
// per step of the bench volume at under 1 TB/s.  k_ref_deposit puts plane p's bits under the significance
// mask instead -- thread = mask word: candidate i of the word is its i-th set bit, the word's candidates start
// at the tile's offset + the word's rank -- with one 8-byte store per word; k_dec_count adds the '1' of the
// plane a sample was found on; k_ref_assemble gathers every sample's bits after the last plane.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_ref_deposit(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  if ((uint32_t)p >= b.refNPlanes)
    return;
  __shared__ uint32_t sm[kThreads / 64 + 1];
  const uint32_t nw = (b.tree.nvals + 63) / 64;
  const uint64_t* words = b.stream + c * b.streamStride;
  uint64_t* planes = b.refPlanes + c * b.refPlaneStride;
  const uint64_t avail = s.avail, pos0 = s.pos;
  const bool partial = pos0 + (uint64_t)s.nRef > avail;   // the stream ends inside this pass
  for (uint32_t tile = blockIdx.x; tile < b.nPixTiles; tile += gridDim.x) {
    if (b.tileRef[c * b.tileStride + tile] == 0)
      continue;   // (uniform: the whole workgroup reads the same word)
    const uint32_t wi = tile * kDecTileWords + threadIdx.x;
    const uint64_t sig = wi < nw ? b.sigOld[c * b.maskPixStride + wi] : 0ull;
    const uint32_t cnt = (uint32_t)__popcll(sig);
    uint32_t total;
    const uint32_t o = block_exclusive_scan_lds<uint32_t>(cnt, sm, &total);
    if (cnt == 0)
      continue;
    const uint64_t at = pos0 + (uint64_t)b.tileRefOff[c * b.tileStride + tile] + o;
    // the pass stops the moment the stream is exhausted (SPECK_INT.cpp:388-389)
    const uint32_t n = at >= avail ? 0u : (uint32_t)min((uint64_t)cnt, avail - at);
    uint64_t res = 0, got = 0;   // the bits under the mask; the candidates that did get one
    if (n) {
      const uint64_t bits = get64(words, at);
      res = n < 64 ? bits & ((1ull << n) - 1) : bits;   // candidate k of the word is the k-th set bit of the mask
      got = n < 64 ? (1ull << n) - 1 : ~0ull;
      if (cnt != 64)   // (64: every sample of the word is a candidate, the bits as they come)
        spread_under_mask(sig, res, got);
    }
    planes[ref_plane_word((uint32_t)p, wi)] = res;   // (always: the word is valid from the plane of its first significant sample on)
    if (partial)
      b.refMask[c * b.maskPixStride + wi] = got;   // the candidates that did get a bit
  }
}

// Every coefficient, once: magnitude bits (the '1' of the plane p0 a sample was found on, the refinement
// bits below it) + 2^(q-1) - 1, q = the lowest plane the sample was refined on, p0 itself when none
// (1.5 * 2^p0 - 1, src/SPECK_INT.cpp:462-468; plane 0 adds the bare bit, :440-447).  A wavefront takes
// four mask words per round, lane = sample: the plane words are wave-uniform (scalar loads) and a sample's
// bit of one is its lane's bit -- a select and a shift-or per plane.  Samples that are not significant get
// their zero here: the coefficient array is not cleared beforehand.  Round 6: IN PLACE -- the plane words of a
// round's eight mask words are the 2 KB its 512 coefficients go to (DecBuffers::refPlanes); every plane word of
// the round is in registers (pv[]) before the round's first store, and no other wavefront touches the tile.
__global__ void __launch_bounds__(kThreads) k_ref_assemble(DecBuffers b)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  const uint32_t nw = (uint32_t)(b.coefStride / 64);   // (the padding of the array is written too)
  const uint32_t nwv = (b.tree.nvals + 63) / 64;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t* coef = reinterpret_cast<uint32_t*>(b.coef) + c * b.coefStride;
  constexpr uint32_t kW = 8;   // mask words per wavefront and round (wordTopStride is a multiple of 64)
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * kThreads + threadIdx.x) >> 6));
  const uint32_t nwave = (gridDim.x * kThreads) >> 6;
  if (!s.active || s.nbp == 0) {   // nothing was decoded: zeros
    for (uint32_t w = wave; w < nw; w += nwave)
      coef[(size_t)w * 64 + lane] = 0;
    return;
  }
  const int lastPlane = s.lastPlane, refPlane = s.refPlaneP1 - 1;
  const bool partial = s.refPartial != 0;
  // the planes whose words are there: those a refinement pass ran on (k_ref_deposit) -- k_dec_count's '1's of the
  // samples found on plane lastPlane + 1 went into a plane at or above the last of them
  const int pLow = refPlane >= 0 ? refPlane : lastPlane + 1;
  const int nbp = min(s.nbp, (int)b.refNPlanes);
  const uint64_t* planes = b.refPlanes + c * b.refPlaneStride;
  const uint64_t* sigOld = b.sigOld + c * b.maskPixStride;
  const uint64_t* sigNew = b.sigNew + c * b.maskPixStride;
  const uint64_t* refMask = b.refMask + c * b.maskPixStride;
  const uint8_t* wordTop = b.wordTop + c * b.wordTopStride;
  // the sign in bit 31 for the dequantising inverse passes where the chunk allows it (coef_scheme, speck_dec.h)
  const int scheme = b.coefSigned != 0 ? coef_scheme(s) : 0;
  const uint64_t* sign = b.sign + c * b.signStride;
  for (uint32_t w0 = wave * kW; w0 + kW <= nw; w0 += nwave * kW) {
    uint64_t so[kW], sn[kW], rm[kW], sg[kW];
    int top[kW], maxTop = 0;
    uint64_t any = 0;
    // (the eight words' tops as two aligned dwords: scalar loads like the mask words)
    const uint32_t tops[2] = {reinterpret_cast<const uint32_t*>(wordTop + w0)[0], reinterpret_cast<const uint32_t*>(wordTop + w0)[1]};
#pragma unroll
    for (uint32_t u = 0; u < kW; u++) {
      const uint32_t w = w0 + u;
      const bool in = w < nwv;
      so[u] = sigOld[w];   // (w < nw <= maskPixStride, nw a multiple of eight: carve_dec)
      sn[u] = sigNew[w];
      rm[u] = refMask[w];  // (read with the others whether the pass was cut short or not: no load behind a branch)
      sg[u] = sign[in ? w : 0u];
      if (!in)
        so[u] = sn[u] = 0;
      top[u] = so[u] ? min((int)((tops[u >> 2] >> (8 * (u & 3))) & 0xffu), nbp) : 0;
      maxTop = max(maxTop, top[u]);
      any |= so[u] | sn[u];
    }
    uint32_t M[kW];
#pragma unroll
    for (uint32_t u = 0; u < kW; u++)
      M[u] = 0;
    if (any && maxTop > pLow) {
      // The planes some word of the round has, from the highest down to the lowest plane that was refined: at most
      // 32, fetched in ONE go -- lane (plane slot << 3 | word) loads its 8-byte word of up to four planes (a plane
      // the word does not have is not read: zero), and a sample's bit of plane word (pl, u) is its lane's bit of
      // the word read back from lane (pl - pLow) % 8 * 8 + u.  (Eight scalar loads per plane, one plane after the
      // other, before: a round waited for as many load round trips as it had planes -- 1.75 ms per launch of 32
      // chunks.)
      const int nPl = maxTop - pLow;
      const uint32_t myU = lane & 7u, myPs = lane >> 3;
      int mytop = 0;
#pragma unroll
      for (uint32_t u = 0; u < kW; u++)
        mytop = myU == u ? top[u] : mytop;
      uint64_t pv[4];
#pragma unroll
      for (int g = 0; g < 4; g++) {
        pv[g] = 0ull;
        const int pl = pLow + g * 8 + (int)myPs;
        if (g * 8 < nPl)   // (uniform)
          pv[g] = pl < mytop ? planes[ref_plane_word((uint32_t)pl, w0 + myU)] : 0ull;   // (w0 is a multiple of eight: the round is one tile)
      }
      // word by word, the planes the word has (the kernel is bound by its vector instructions: a word without a
      // significant sample -- half of the bench volume's -- costs a scalar branch here and one below)
#pragma unroll
      for (uint32_t u = 0; u < kW; u++) {
        const int nu = top[u] - pLow;   // (uniform)
        if (nu <= 0)
          continue;
#pragma unroll
        for (int g = 3; g >= 0; g--) {
          if (g * 8 >= nu)
            continue;
          for (int ps = min(7, nu - 1 - g * 8); ps >= 0; ps--) {
            const int src = ps * 8 + (int)u;
            const uint64_t pw = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)pv[g], src) |
                                ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(pv[g] >> 32), src) << 32);
            M[u] = 2u * M[u] + (__builtin_amdgcn_inverse_ballot_w64(pw) ? 1u : 0u);
          }
        }
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < kW; u++) {
      const uint32_t w = w0 + u;
      if ((so[u] | sn[u]) == 0ull) {   // (uniform) no significant sample in the word
        coef[(size_t)w * 64 + lane] = 0;
        continue;
      }
      uint32_t v = 0;
      const bool isNew = ((sn[u] >> lane) & 1ull) != 0, isOld = ((so[u] >> lane) & 1ull) != 0;
      if (isNew || isOld) {
        uint32_t m = (isOld && maxTop > pLow) ? M[u] << pLow : 0u;
        if (isNew)
          m |= 1u << lastPlane;
        const int p0 = 31 - __clz((int)m);
        int q = p0;   // lowest plane the sample was refined on, the plane it was found on when none
        if (isOld && refPlane >= 0) {
          bool atRef = p0 > refPlane;
          if (atRef && partial)
            atRef = ((rm[u] >> lane) & 1ull) != 0;
          if (atRef)
            q = refPlane;
          else if (p0 > refPlane + 1)
            q = refPlane + 1;
        }
        v = m ? m + (q >= 1 ? (1u << (q - 1)) - 1u : 0u) : 0u;
        if (scheme)   // (a set bit of the sign array is "positive", SPECK_INT.cpp:174-175)
          v = (scheme == 2 ? v >> 1 : v) | (((sg[u] >> lane) & 1ull) ? 0u : 0x80000000u);
      }
      coef[(size_t)w * 64 + lane] = v;
    }
  }
}

// After the last plane: leaf results that no k_dec_count has folded yet are those of the last
// decoded plane -- they are "new" for k_dec_finish.
__global__ void __launch_bounds__(kThreads) k_dec_fold(DecBuffers b)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  if (!s.active || s.nbp == 0)
    return;
  const uint32_t nw = (b.tree.nvals + 63) / 64;
  const uint32_t wi = blockIdx.x * blockDim.x + threadIdx.x;
  if (wi >= nw)
    return;
  uint64_t lb, ls, ln;
  if (!leaf_word(b, c, wi, lb, ls, ln, (uint32_t)s.lastPlane + 1u))   // leaves that split on the last plane
    return;
  const uint64_t fresh = ls & ~b.sigOld[c * b.maskPixStride + wi];
  if (fresh)
    b.sigNew[c * b.maskPixStride + wi] |= fresh;
  if (ln) {
    const uint64_t sg = b.sign[c * b.signStride + wi];
    if (sg & ln)
      b.sign[c * b.signStride + wi] = sg & ~ln;
  }
}

// After the last plane: coefficients that became significant but were never refined still hold 0.
// Those found during the last decoded plane `pl` (sigNew) get 1.5 * 2^pl - 1, those found on the
// plane before (sigOld, untouched by a complete refinement pass) 1.5 * 2^(pl+1) - 1
// (SPECK_INT.cpp:216-220,462-468).
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_dec_finish(DecBuffers b)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  if (!s.active || s.nbp == 0)
    return;
  const uint32_t n = b.tree.nvals;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const uint64_t m = 1ull << (i & 63);
  const bool isNew = (b.sigNew[c * b.maskPixStride + (i >> 6)] & m) != 0;
  const bool isOld = (b.sigOld[c * b.maskPixStride + (i >> 6)] & m) != 0;
  if (!isNew && !isOld)
    return;
  CT* coef = reinterpret_cast<CT*>(b.coef) + c * b.coefStride;
  if (coef[i] != 0)
    return;
  const int pl = s.lastPlane + (isNew ? 0 : 1);
  const CT thr = (CT)1 << pl;
  coef[i] = thr + thr - thr / 2 - 1;
}

__global__ void k_dec_plane_end(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.nchunks)
    return;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  const uint64_t room = s.avail - s.pos;
  s.refPlaneP1 = p + 1;
  s.refPartial = (uint64_t)s.nRef > room ? 1u : 0u;
  s.pos += min((uint64_t)s.nRef, room);
  if (s.pos >= s.avail || p == 0)  // SPECK_INT.cpp:204-205
    s.done = 1;
}

// chunks of the batch that still have planes to decode
__global__ void __launch_bounds__(kThreads) k_dec_live(DecBuffers b, uint32_t* out)
{
  uint32_t live = 0;
  for (uint32_t c = threadIdx.x; c < b.nchunks; c += blockDim.x)
    live += (b.st[c].active && !b.st[c].done) ? 1u : 0u;
  live = (uint32_t)__syncthreads_count(live != 0);
  if (threadIdx.x == 0)
    *out = live;
}

// ------------------------------------------------------------------------------------------
int launch_speck_decode(hipStream_t stream, const DecBuffers& b, const DecPlanHost& plan,
                        const uint8_t* container, const uint64_t* d_chunkOff,
                        const uint64_t* d_chunkLen, bool wide_pass, int maxPlanes)
{
  const uint32_t nc = b.nchunks;
  const dim3 perChunk((nc + 63) / 64);
  LAUNCH_K(k_dec_header, perChunk, dim3(64), 0, stream, b, container, d_chunkOff,
                     d_chunkLen, plan.d_initLIS, plan.d_initLen, wide_pass ? 1 : 0);
  const uint32_t wordBlocks = (uint32_t)((b.streamStride + kThreads - 1) / kThreads);
  LAUNCH_K(k_dec_load_words, dim3(wordBlocks, nc), dim3(kThreads), 0, stream, b,
                     container);
  const uint32_t tokBlocks = (uint32_t)((b.tokStride + kThreads - 1) / kThreads);
  // (plan.gridDiv: the batch decodes beside others that hold most of the device -- grids that fill a quarter of it)
  const uint32_t gdiv = std::max<uint32_t>(1u, plan.gridDiv);
  // (workgroups of the tile sweeps -- k_dec_count, k_lip_deposit, k_ref_deposit -- over all chunks: a large batch's light
  //  planes pay for the dispatch of 16 K workgroups that find nothing to do; a quarter of them loop over four tiles
  //  each: decompression of 64 chunks 93.6 -> 94.2 GB/s in alternating runs, round 5.  A few chunks keep the wide grid)
  static const uint32_t tileCapEnv = tune_getenv("SPERR_HIP_TILE_GRID") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_TILE_GRID")) : 0u;
  // (round 6: 1536 for every batch -- 64 chunks in three sub-batches 119.4 -> 120.5 to 120.9 GB/s, 8 chunks 51.9 -> 52.3; 1024:
  //  118.3, 2560: 118.5; round 5 had 4096 from 16 chunks on and the wide grid below)
  const uint32_t tileCap = tileCapEnv ? tileCapEnv : 1536u;
  const uint32_t tokSegGrid = capped_blocks((uint32_t)(b.tokStride / kLipSeg + 1), nc, kGridCap / gdiv);   // (k_lip_words: a segment a workgroup)
  const uint32_t tokGrid = capped_blocks(tokBlocks, nc, kGridCap / gdiv), tileGrid = capped_blocks(b.nPixTiles, nc, tileCap / gdiv);
  // workgroups per chunk of the k_lis_l0 pass: about two per CU over all chunks
  static const uint32_t l0Total = tune_getenv("SPERR_HIP_L0_WGS") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_L0_WGS")) : 512u;
  static const uint32_t l01Cap = tune_getenv("SPERR_HIP_L01_CAP") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_L01_CAP")) : 64u;   // workgroups per chunk at most (round 2: 16 -- a batch of 8 chunks left most CUs idle)
  const uint32_t l0Groups = std::min<uint32_t>(l01Cap, std::max<uint32_t>(1, l0Total / nc));
  if (plan.tables && plan.l0) {
    if (set_max_dyn_lds(reinterpret_cast<const void*>(&k_lis_l0), (int)kL0Smem) ||
        set_max_dyn_lds(reinterpret_cast<const void*>(&k_lis_l1), (int)kL1Smem) ||
        set_max_dyn_lds(reinterpret_cast<const void*>(&k_lis_l2), (int)kL2Smem))
      return -1;
  }
  // (workgroups over all chunks; measured: 768 beats 512 and 256 on 64 chunks)
  static const uint32_t l1Total = tune_getenv("SPERR_HIP_L1_WGS") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_L1_WGS")) : 512u;   // (round 6, three sub-batches of 21 chunks: 512 / 768 with k_lis_hi at 128: 119.5 / 117.3 GB/s)
  const uint32_t l1Groups = std::min<uint32_t>(l01Cap, std::max<uint32_t>(1, l1Total / nc));
  // k_lis_l2: two workgroups a CU over all chunks (SPERR_HIP_L2_WGS in the diagnostics build)
  static const uint32_t l2Total = tune_getenv("SPERR_HIP_L2_WGS") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_L2_WGS")) : 512u;
  const uint32_t l2Groups = std::min<uint32_t>(l01Cap, std::max<uint32_t>(1, l2Total / nc));
  // (entries from which the list is k_lis_l2's; measured on MI355X, alternating runs: 1 / 512 / 2048 / 4096 / never give
  //  50.0 / 50.7 / 47.5 / 47.0 / 45.7-46.8 GB/s for 8 chunks and 113.4 / 113.1-114.0 / 112.8 / 113.5 / 113.8-114.0 for 64)
  static const uint32_t l2Min = tune_getenv("SPERR_HIP_L2_MIN") ? (uint32_t)std::max(1, atoi(tune_getenv("SPERR_HIP_L2_MIN"))) : 512u;
  const uint32_t placeGrid = capped_blocks((uint32_t)((b.bornStride + kThreads - 1) / kThreads), nc, kGridCap / gdiv);
  // workgroups per chunk of the k_lis_hi pass, at most what the queues were sized for
  // (SPERR_HIP_HI_WGS: the total over the batch's chunks; measured on MI355X with two sub-batches
  // of 32 chunks side by side: 96 / 128 / 160 / 192 / 224 / 256 / 384 workgroups per sub-batch give
  // 70.8 / 75.3 / 76.3 / 76.0 / 75.0 / 74.5 / 74.3 GB/s of decompression; again after the region
  // bookkeeping went to one lane per level: 96 / 128 / 160 / 192 / 256 give 73.4 / 77.5 / 78.5 / 77.9 / 75.5)
  static const uint32_t hiTotal = tune_getenv("SPERR_HIP_HI_WGS") ? (uint32_t)atoi(tune_getenv("SPERR_HIP_HI_WGS")) : 128u;   // (round 6: 128 -- with three sub-batches side by side
                                                                                  //  96 / 112 / 128 / 144 / 160 workgroups a sub-batch: 119.2 / 118.8 / 118.5 to 119.5 / 117.3 / 115.9 to 116.7 GB/s)
  const uint32_t hiGroups = std::min<uint32_t>(std::max<uint32_t>(1, b.hiGroupsMax), std::max<uint32_t>(1, hiTotal / nc));
  // workgroups per chunk of k_lis_mx: the walk of a chunk is serial, the rows and the expansion of its regions are
  // what the other workgroups are for (SPERR_HIP_MX_WGS: the total over the batch's chunks)
  static const uint32_t mxTotal = getenv("SPERR_HIP_MX_WGS") ? (uint32_t)atoi(getenv("SPERR_HIP_MX_WGS")) : 208u;
  const uint32_t mxGroups = std::min<uint32_t>(std::min<uint32_t>(8u, std::max<uint32_t>(1, b.hiGroupsMax)),
                                               plan.mxGroups ? plan.mxGroups : std::max<uint32_t>(1, mxTotal / nc));
  if (plan.mixed && prepare_lis_mx(b))
    return -1;
  if (plan.tables && plan.hi) {
    if (set_max_dyn_lds(reinterpret_cast<const void*>(&k_lis_hi<uint32_t>), (int)b.hiSmemBytes) ||
        set_max_dyn_lds(reinterpret_cast<const void*>(&k_lis_hi<uint64_t>), (int)b.hiSmemBytes))
      return -1;
  }
  // the kernels of a plane; CT: the integer type of the coefficients (the 64-bit pass of chunks whose
  // largest coefficient needs it, src/SPECK_FLT.cpp:324-337)
  // (LAUNCH_CT: the kernel under its own name -- k_lis_hi<uint32_t>, not k_lis_hi<CT> -- for the profiler)
#define LAUNCH_CT(kern, ...)                    \
  do {                                          \
    if constexpr (sizeof(CT) == 4)              \
      LAUNCH_K(kern<uint32_t>, __VA_ARGS__);    \
    else                                        \
      LAUNCH_K(kern<uint64_t>, __VA_ARGS__);    \
  } while (0)
  auto plane = [&](auto ct, int p) -> int {
    using CT = decltype(ct);
    LAUNCH_K(k_dec_count, dim3(tileGrid, nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_K(k_dec_scan, dim3(nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_K(k_lip_words, dim3(tokSegGrid, nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_K(k_lip_scan, dim3(nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_CT(k_lip_apply, dim3(tokGrid, nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_K(k_lip_deposit, dim3(tileGrid, nc), dim3(kThreads), 0, stream, b, p);
    if (plan.tables) {
      if (plan.l0)
        LAUNCH_K(k_lis_l0, dim3(l0Groups, nc), dim3(kL0Threads), kL0Smem, stream, b, p);
      if (plan.l1)
        LAUNCH_K(k_lis_l1, dim3(l1Groups, nc), dim3(kL1Threads), kL1Smem, stream, b, p);
      if (plan.l2)
        LAUNCH_K(k_lis_l2, dim3(l2Groups, nc), dim3(kL2Threads), kL2Smem, stream, b, p, l2Min);
      if (!plan.hi)
        return -1;   // (use_tables() implies use_lis_hi(): engine.hip)
      LAUNCH_CT(k_lis_hi, dim3(hiGroups, nc), dim3(kTabThreads), b.hiSmemBytes, stream, b, p);
      LAUNCH_K(k_lis_compact, dim3(b.tree.nlevels, nc), dim3(kTabThreads), 0, stream, b, p);
    }
    else if (plan.mixed) {
      if (launch_lis_mx(stream, b, p, mxGroups, b.lisStamps != nullptr))
        return -1;
      LAUNCH_K(k_lis_compact, dim3(b.tree.nlevels, nc), dim3(kTabThreads), 0, stream, b, p);
    }
    else
      LAUNCH_CT(k_lis_walk, dim3(nc), dim3(64), 0, stream, b, p);
    if (plan.tables || plan.mixed) {
      if (b.nSlots) {
        LAUNCH_K(k_place_scan, dim3(b.nSlots, nc), dim3(kTabThreads), 0, stream, b, p);
        LAUNCH_K(k_place_scatter, dim3(placeGrid, nc), dim3(kThreads), 0, stream, b, p);
      }
      LAUNCH_K(k_leaf_apply, dim3(capped_blocks(1024, nc, kGridCap / gdiv), nc), dim3(kThreads), 0, stream, b, p);
    }
    if (b.refPlanes && sizeof(CT) == 4)
      LAUNCH_K(k_ref_deposit, dim3(tileGrid, nc), dim3(kThreads), 0, stream, b, p);
    else
      LAUNCH_CT(k_ref_apply2, dim3(tileGrid, nc), dim3(kThreads), 0, stream, b, p);
    return 0;
  };
#undef LAUNCH_CT
  for (int p = maxPlanes - 1; p >= 0; p--) {
    if (wide_pass ? plane(uint64_t{}, p) : plane(uint32_t{}, p))
      return -1;
    LAUNCH_K(k_dec_plane_end, perChunk, dim3(64), 0, stream, b, p);
    // Have the chunks run out of bits?  Asked after 16 planes and then after every second one, and
    // answered ONE QUESTION LATE: the host waits for the answer to the previous question while the
    // planes launched since are still queued, so the device never runs dry (a wait for the stream
    // itself drained it twice or three times per batch).
    const int planesDone = maxPlanes - p;
    if (plan.d_live && p > 0 && planesDone >= 16 && planesDone % 2 == 0 && plan.h_live && plan.liveEv) {
      const int k = (planesDone - 16) / 2;
      if (k < kLiveSlots) {
        LAUNCH_K(k_dec_live, dim3(1), dim3(kThreads), 0, stream, b, plan.d_live + k);
        HIP_CHECK(hipMemcpyAsync(plan.h_live + k, plan.d_live + k, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipEventRecord(plan.liveEv[k], stream));
        if (k >= 1) {
          HIP_CHECK(hipEventSynchronize(plan.liveEv[k - 1]));
          if (plan.h_live[k - 1] == 0)
            break;   // (every launch below would return at once)
        }
      }
    }
  }
  {
    const uint32_t n = b.tree.nvals;
    if ((plan.tables || plan.mixed) && b.wordLeaf)
      LAUNCH_K(k_dec_fold, dim3(((n + 63) / 64 + kThreads - 1) / kThreads, nc), dim3(kThreads), 0,
               stream, b);
    if (b.refPlanes && !wide_pass)
      LAUNCH_K(k_ref_assemble, dim3(capped_blocks((uint32_t)((b.coefStride / 64 + 15) / 16), nc, kGridCapWide), nc),
               dim3(kThreads), 0, stream, b);
    if (!plan.skipFinish) {
      if (wide_pass)
        LAUNCH_K(k_dec_finish<uint64_t>, dim3((n + kThreads - 1) / kThreads, nc), dim3(kThreads), 0,
                 stream, b);
      else
        LAUNCH_K(k_dec_finish<uint32_t>, dim3((n + kThreads - 1) / kThreads, nc), dim3(kThreads), 0,
                 stream, b);
    }
  }
  HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace sperrhip
