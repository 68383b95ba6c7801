// speck_dec.hip -- SPECK3D bit-plane set-partitioning DECODER as HIP kernels.
//
// Mirrors /root/reference/src/SPECK_INT.cpp:165-228,359-469, src/SPECK3D_INT.cpp:99-212 and
// src/SPECK3D_INT_DEC.cpp:8-49; tests/model/speck_model.cpp (model_speck3d_decode) is the CPU
// model of the phases below.  Per bit plane:
//
//   LIP scan     (k_dec_count/_scan/_candlist, k_lip_words/_scan/_apply)  data parallel: a token
//                starts at every bit preceded by an EVEN number of consecutive 1 bits (a 1 is
//                always followed by its sign bit), so token starts, token ranks and the pixel
//                each token belongs to all come from prefix sums;
//   LIS phase    (k_lis_walk)  what each bit means depends on every earlier bit of the phase:
//                one wavefront per chunk walks the lists; chunks run concurrently;
//   refinement   (k_ref_apply)  the j-th significant pixel in raster order takes bit j.
//
// Bits past the available length read as zero (the reference zero-pads a truncated stream,
// SPECK_INT.cpp:95-105); the loop stops where the reference's does.
#include "speck_dec.h"

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
// candidates of the two pixel passes
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void dec_load4(const int8_t* a, uint32_t i0, uint32_t n, int v[4])
{
  if (i0 + 4 <= n) {
    const char4 q = *reinterpret_cast<const char4*>(a + i0);
    v[0] = q.x;
    v[1] = q.y;
    v[2] = q.z;
    v[3] = q.w;
  }
  else
    for (int k = 0; k < 4; k++)
      v[k] = (i0 + k < n) ? a[i0 + k] : -1;
}

__global__ void __launch_bounds__(kThreads) k_dec_count(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  __shared__ uint32_t sm[kThreads / 64 + 1];
  const uint32_t n = b.tree.nvals;
  const uint32_t i0 = blockIdx.x * kPixTile + threadIdx.x * 4;
  int born[4], sg[4];
  dec_load4(b.born + c * b.pixStride, i0, n, born);
  dec_load4(b.sigp + c * b.pixStride, i0, n, sg);
  uint32_t v = 0;
  for (int k = 0; k < 4; k++) {
    v += (born[k] > p && sg[k] < 0) ? 1u : 0u;
    v += (sg[k] > p) ? (1u << 16) : 0u;
  }
  uint32_t total;
  block_exclusive_scan<uint32_t>(v, sm, &total);
  if (threadIdx.x == 0) {
    b.tileLip[c * b.tileStride + blockIdx.x] = total & 0xffffu;
    b.tileRef[c * b.tileStride + blockIdx.x] = total >> 16;
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
  }
}

__global__ void __launch_bounds__(kThreads) k_dec_candlist(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  if (b.tileLip[c * b.tileStride + blockIdx.x] == 0)
    return;
  __shared__ uint32_t sm[kThreads / 64 + 1];
  const uint32_t n = b.tree.nvals;
  const uint32_t i0 = blockIdx.x * kPixTile + threadIdx.x * 4;
  int born[4], sg[4];
  dec_load4(b.born + c * b.pixStride, i0, n, born);
  dec_load4(b.sigp + c * b.pixStride, i0, n, sg);
  uint32_t v = 0;
  for (int k = 0; k < 4; k++)
    v += (born[k] > p && sg[k] < 0) ? 1u : 0u;
  uint32_t total;
  uint32_t ex = block_exclusive_scan<uint32_t>(v, sm, &total) +
                b.tileLipOff[c * b.tileStride + blockIdx.x];
  uint32_t* cand = b.cand + c * b.candStride;
  for (int k = 0; k < 4; k++)
    if (born[k] > p && sg[k] < 0)
      cand[ex++] = i0 + k;
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

__global__ void __launch_bounds__(kThreads) k_lip_words(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  const uint64_t nbits = 2ull * s.nLip + 1;
  const uint64_t nwords = (nbits + 63) / 64;
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s.nLip == 0 || w >= nwords)
    return;
  const uint64_t* words = b.stream + c * b.streamStride;
  // parity of the run of 1s that ends right before this word
  uint32_t parity = 0;
  for (uint64_t back = w; back > 0;) {
    back--;
    const uint64_t v = lip_word(words, s, back, nbits);
    if (v == ~0ull)
      continue;  // 64 more ones: parity unchanged, keep looking
    parity = (uint32_t)__clzll((long long)~v) & 1u;  // leading ones of the previous word
    break;
  }
  const uint64_t x = lip_word(words, s, w, nbits);
  uint64_t starts = 0;
  uint32_t ones = parity;  // only the parity matters
  for (int k = 0; k < 64; k++) {
    if ((ones & 1u) == 0)
      starts |= 1ull << k;
    ones = ((x >> k) & 1ull) ? ones + 1 : 0;
  }
  if (w == nwords - 1 && (nbits & 63))
    starts &= (1ull << (nbits & 63)) - 1;
  b.tokMask[c * b.tokStride + w] = starts;
  b.tokCnt[c * b.tokStride + w] = (uint32_t)__popcll(starts);
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
  const uint64_t nbits = 2ull * s.nLip + 1;
  const uint32_t nwords = (uint32_t)((nbits + 63) / 64);
  uint32_t* cnt = b.tokCnt + c * b.tokStride;
  uint32_t* off = b.tokOff + c * b.tokStride;
  const uint64_t* mask = b.tokMask + c * b.tokStride;
  uint32_t carry = 0;
  for (uint32_t base = 0; base < nwords; base += kThreads) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t v = i < nwords ? cnt[i] : 0;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<uint32_t>(v, sm, &total) + carry;
    if (i < nwords) {
      off[i] = ex;
      // token #nLip (0-based) starts in this word?  Then the phase is that many bits long.
      if (ex <= s.nLip && s.nLip < ex + v) {
        uint64_t m = mask[i];
        for (uint32_t r = s.nLip - ex; r > 0; r--)
          m &= m - 1;
        s.lipBits = (uint64_t)i * 64 + (uint64_t)__ffsll((long long)m) - 1;
      }
    }
    carry += total;
  }
}

template <typename CT>
__global__ void __launch_bounds__(kThreads) k_lip_apply(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  const uint64_t nbits = 2ull * s.nLip + 1;
  const uint64_t nwords = (nbits + 63) / 64;
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s.nLip == 0 || w >= nwords)
    return;
  const uint64_t* words = b.stream + c * b.streamStride;
  const uint64_t x = lip_word(words, s, w, nbits);
  uint64_t sig = b.tokMask[c * b.tokStride + w];
  uint32_t j = b.tokOff[c * b.tokStride + w];
  if (j >= s.nLip)
    return;
  const uint64_t nextbit = lip_word(words, s, w + 1, nbits) & 1ull;
  const uint32_t* cand = b.cand + c * b.candStride;
  int8_t* sigp = b.sigp + c * b.pixStride;
  CT* coef = reinterpret_cast<CT*>(b.coef) + c * b.coefStride;
  unsigned long long* sign = reinterpret_cast<unsigned long long*>(b.sign + c * b.signStride);
  const CT thr = (CT)1 << p;
  const CT init = thr + thr - thr / 2 - 1;   // SPECK_INT.cpp:462-468
  while (sig && j < s.nLip) {
    const int k = __ffsll((long long)sig) - 1;
    sig &= sig - 1;
    if ((x >> k) & 1ull) {
      const uint32_t pix = cand[j];
      sigp[pix] = (int8_t)p;
      coef[pix] = init;
      const uint64_t sb = k < 63 ? (x >> (k + 1)) & 1ull : nextbit;
      if (!sb)
        atomicAnd(sign + (pix >> 6), ~(1ull << (pix & 63)));
    }
    j++;
  }
}

// ------------------------------------------------------------------------------------------
// LIS phase: serial walk, one wavefront (lane 0) per chunk
// ------------------------------------------------------------------------------------------
struct WalkFrame {
  Node nd;
  Kids k;
  int j;
  bool found;
  uint32_t kidlev;
};

template <typename CT>
__global__ void __launch_bounds__(64) k_lis_walk(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.x;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  if (threadIdx.x != 0)
    return;
  const Tree& t = b.tree;
  const uint64_t* words = b.stream + c * b.streamStride;
  const uint64_t avail = s.avail;
  uint64_t pos = s.lipStart + s.lipBits;
  int8_t* born = b.born + c * b.pixStride;
  int8_t* sigp = b.sigp + c * b.pixStride;
  CT* coef = reinterpret_cast<CT*>(b.coef) + c * b.coefStride;
  uint64_t* sign = b.sign + c * b.signStride;
  const CT thr = (CT)1 << p;
  const CT init = thr + thr - thr / 2 - 1;
  const uint32_t cur = s.cur, nx = cur ^ 1u;
  uint32_t nextLen[kMaxLevels];
  for (uint32_t l = 0; l < t.nlevels; l++)
    nextLen[l] = 0;
  WalkFrame st[kMaxDepth + 1];
  for (uint32_t l = t.nlevels; l-- > 0;) {
    const uint32_t n = s.listLen[cur][l];
    const uint64_t* list = b.lis[cur] + c * b.lisStride + b.levelOff[l];
    for (uint32_t e = 0; e < n; e++) {
      const uint64_t packed = list[e];
      if (!get_bit(words, pos++, avail)) {
        b.lis[nx][c * b.lisStride + b.levelOff[l] + nextLen[l]++] = packed;
        continue;
      }
      int sp = 0;
      auto push = [&](const Node& nd) {
        WalkFrame& f = st[sp++];
        f.nd = nd;
        node_kids(t, nd, f.k);
        f.j = 0;
        f.found = false;
        const NodeGeom q = node_geom(t, nd);
        f.kidlev = node_level(t, nd) + (q.len[0] > 1) + (q.len[1] > 1) + (q.len[2] > 1);
      };
      push(unpack_node(packed));
      while (sp > 0) {
        WalkFrame& f = st[sp - 1];
        if (f.j == f.k.n) {
          sp--;
          continue;
        }
        const int j = f.j++;
        const bool coded = f.found || (j + 1 != f.k.n);
        const bool sig = coded ? (get_bit(words, pos++, avail) != 0) : true;
        if (sig)
          f.found = true;
        if (f.k.count[j] == 1) {
          const uint32_t ridx = kid_raster(t, f.nd, f.k, j);
          born[ridx] = (int8_t)p;
          if (sig) {
            sigp[ridx] = (int8_t)p;
            coef[ridx] = init;
            if (!get_bit(words, pos++, avail))
              sign[ridx >> 6] &= ~(1ull << (ridx & 63));
          }
        }
        else if (sig)
          push(kid_node(f.k, j));
        else
          b.lis[nx][c * b.lisStride + b.levelOff[f.kidlev] + nextLen[f.kidlev]++] =
              pack_node(kid_node(f.k, j));
      }
    }
  }
  for (uint32_t l = 0; l < t.nlevels; l++)
    s.listLen[nx][l] = nextLen[l];
  s.cur = nx;
  s.pos = pos;
  if (pos >= avail)  // SPECK_INT.cpp:200-201
    s.done = 1;
}

// ------------------------------------------------------------------------------------------
// refinement: the j-th pixel that was significant before this plane takes bit pos + j
// ------------------------------------------------------------------------------------------
template <typename CT>
__global__ void __launch_bounds__(kThreads) k_ref_apply(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.y;
  const DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  if (b.tileRef[c * b.tileStride + blockIdx.x] == 0)
    return;
  __shared__ uint32_t sm[kThreads / 64 + 1];
  const uint32_t n = b.tree.nvals;
  const uint32_t i0 = blockIdx.x * kPixTile + threadIdx.x * 4;
  int sg[4];
  dec_load4(b.sigp + c * b.pixStride, i0, n, sg);
  uint32_t v = 0;
  for (int k = 0; k < 4; k++)
    v += (sg[k] > p) ? 1u : 0u;
  uint32_t total;
  uint64_t j = block_exclusive_scan<uint32_t>(v, sm, &total) +
               (uint64_t)b.tileRefOff[c * b.tileStride + blockIdx.x];
  const uint64_t* words = b.stream + c * b.streamStride;
  CT* coef = reinterpret_cast<CT*>(b.coef) + c * b.coefStride;
  const CT thr = (CT)1 << p, half = thr / 2;
  for (int k = 0; k < 4; k++)
    if (sg[k] > p) {
      const uint64_t at = s.pos + j++;
      if (at >= s.avail)   // the pass stops the moment the stream is exhausted
        break;             // (SPECK_INT.cpp:388-389)
      const int bit = (int)((words[at >> 6] >> (at & 63)) & 1);
      CT v2 = coef[i0 + k];
      if (p >= 1)
        v2 = bit ? v2 + half : v2 - half;
      else if (bit)
        v2 += 1;
      coef[i0 + k] = v2;
    }
}

__global__ void k_dec_plane_end(DecBuffers b, int p)
{
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b.nchunks)
    return;
  DecState& s = b.st[c];
  DEC_ACTIVE_OR_RETURN(s, p);
  const uint64_t room = s.avail - s.pos;
  s.pos += min((uint64_t)s.nRef, room);
  if (s.pos >= s.avail || p == 0)  // SPECK_INT.cpp:204-205
    s.done = 1;
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
  for (int p = maxPlanes - 1; p >= 0; p--) {
    LAUNCH_K(k_dec_count, dim3(b.nPixTiles, nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_K(k_dec_scan, dim3(nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_K(k_dec_candlist, dim3(b.nPixTiles, nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_K(k_lip_words, dim3(tokBlocks, nc), dim3(kThreads), 0, stream, b, p);
    LAUNCH_K(k_lip_scan, dim3(nc), dim3(kThreads), 0, stream, b, p);
    if (wide_pass) {
      LAUNCH_K(k_lip_apply<uint64_t>, dim3(tokBlocks, nc), dim3(kThreads), 0, stream, b,
                         p);
      LAUNCH_K(k_lis_walk<uint64_t>, dim3(nc), dim3(64), 0, stream, b, p);
      LAUNCH_K(k_ref_apply<uint64_t>, dim3(b.nPixTiles, nc), dim3(kThreads), 0, stream,
                         b, p);
    }
    else {
      LAUNCH_K(k_lip_apply<uint32_t>, dim3(tokBlocks, nc), dim3(kThreads), 0, stream, b,
                         p);
      LAUNCH_K(k_lis_walk<uint32_t>, dim3(nc), dim3(64), 0, stream, b, p);
      LAUNCH_K(k_ref_apply<uint32_t>, dim3(b.nPixTiles, nc), dim3(kThreads), 0, stream,
                         b, p);
    }
    LAUNCH_K(k_dec_plane_end, perChunk, dim3(64), 0, stream, b, p);
  }
  HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace sperrhip
