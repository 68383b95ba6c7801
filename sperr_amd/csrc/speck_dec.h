// speck_dec.h -- buffers and launcher of the SPECK3D decoder (speck_dec.hip)
#ifndef SPERR_AMD_SPECK_DEC_H
#define SPERR_AMD_SPECK_DEC_H

#include "common.h"

namespace sperrhip {

struct DecState {
  uint32_t active;
  int32_t done;
  uint32_t error;
  int32_t nbp;
  uint32_t cur;
  uint32_t nLip, nRef;           // candidates of the current plane's pixel passes
  uint32_t nLeafEv;              // leaf events of the plane just decoded
  int32_t lastPlane;             // last plane whose sorting pass ran
  int32_t refPlaneP1;            // 1 + the last plane whose refinement pass ran (0: none yet)
  uint32_t refPartial;           //   ... and the stream ended inside that pass
  uint64_t pos;                  // next unread bit
  uint64_t avail;                // usable bits of the stream
  uint64_t total_bits;
  uint64_t payload;              // byte offset of the SPECK payload inside the container
  uint64_t lipStart, lipBits;    // LIP scan of the current plane
  // GPU-wide pass over the list of the smallest sets (k_lis_l0), which the LIS phase visits first
  uint32_t l0Ticket;             // next block of the pass to hand out
  int32_t l0PlaneP1;             // 1 + the plane whose list k_lis_l0 has decoded (0: none)
  uint32_t l0Sig;                // entries it found significant (= leaf events it wrote)
  uint32_t l0Pad;
  uint64_t l0End;                // first bit after the list's code
  // same for the next list (4x4x4 sets), k_lis_l1
  uint32_t l1Ticket;
  int32_t l1PlaneP1;
  uint64_t l1End;
  // ... and the one after it (8x8x8 sets), k_lis_l2 (round 6)
  uint32_t l2Ticket;
  int32_t l2PlaneP1;
  uint64_t l2End;
  // GPU-wide pass over the lists of the larger sets (k_lis_hi)
  uint32_t hiTicket;             // next region of the pass to hand out
  int32_t hiPlaneP1;             // 1 + the plane whose pass has ended (0: none)
  int32_t hiHintPad;
  uint64_t hiHint;               // (plane + 1) << 40 | list level the chain was last seen in << 32 | entries that list had left
  uint32_t hiCompactDone;        // workgroups of k_lis_compact that have finished (the last one ends the phase)
  uint64_t hiEnd;                // first bit after the phase
  uint64_t mxHint;               // k_lis_mx: (plane + 1) << 57 | list level << 48 | region << 28 | entries left, as the chain last published
  uint32_t hiBornCnt[8];         // births / leaf events in the workgroups' own segments
  uint32_t hiLeafCnt[8];
  uint32_t bornCount;            // sets born / leaf events written so far by the GPU-wide passes
  uint32_t leafCount;            //   of the current plane (bornCount: all births once the list kernels end)
  uint64_t lisPhaseBits;         // bits of the plane's LIS phase covered by the birth masks
  uint32_t iPart, iPad;          // 2D coder: part_level of what is left of the type-I set (0: nothing)
  uint32_t slotBorn[spk::kMaxLevels];   // births per mask slot of the plane (k_place_scan)
  uint32_t listLen[2][spk::kMaxLevels];
};

// How k_ref_assemble hands a chunk's finished 32-bit coefficients to the dequantising inverse passes when the host
// asks for it (DecBuffers::coefSigned, LiftFuse::coefSigned): with the SIGN IN BIT 31, so that those passes read
// neither the sign nor the mask words.  Fixed-rate mode quantises to the full range of uint32_t
// (src/SPECK_FLT.cpp:282-290), so a magnitude may need all 32 bits; but a decoded magnitude is
// m + 2^(q-1) - 1 with q the lowest plane the sample was refined on (src/SPECK_INT.cpp:440-468): odd whenever q >= 2.
//   1: at most 31 planes -- magnitude | sign << 31
//   2: 32 planes, every q of the chunk >= 2 (the stream ran out at plane 2 or above) -- magnitude >> 1 | sign << 31
//      (a non-zero magnitude is 2 t + 1)
//   0: neither: magnitudes as they are, sign and masks read as ever
// (sign: 1 = negative here; DecBuffers::sign has 1 = positive, src/SPECK_INT.cpp:174-175)
__host__ __device__ inline int coef_scheme(const DecState& s)
{
  if (s.nbp <= 31)
    return 1;
  const int refPlane = s.refPlaneP1 - 1;
  const int qmin = (refPlane >= 0 && refPlane < s.lastPlane) ? refPlane : s.lastPlane;
  return qmin >= 2 ? 2 : 0;
}
// the magnitude of such a word (scheme 1 or 2)
__host__ __device__ inline uint32_t coef_scheme_mag(uint32_t stored, bool two)
{
  const uint32_t t = stored & 0x7fffffffu;
  return (two && t) ? 2u * t + 1u : t;
}

// word `w` of plane `p` inside a chunk's plane storage (DecBuffers::refPlanes): see there
__host__ __device__ inline size_t ref_plane_word(uint32_t p, uint32_t w)
{
  return ((size_t)(w >> 3) << 8) + ((size_t)p << 3) + (size_t)(w & 7u);
}

struct DecBuffers {
  spk::Tree tree;
  uint32_t nchunks;
  CoderState* cst;
  DecState* st;
  uint64_t* stream;            // payload as aligned words, zero padded (+2 words of slack)
  size_t streamStride;
  uint64_t* bornM;             // pixel state bitmasks: tested at least once,
  uint64_t* sigOld;            //   significant before the current plane,
  uint64_t* sigNew;            //   found significant during the current plane
  size_t maskPixStride;
  void* coef;                  // uint32_t or uint64_t magnitudes being reconstructed
  size_t coefStride;
  // Refinement bit planes (32-bit coefficients; round 5, after an experiment of round 3): the refinement pass
  // does not touch the coefficients.  refPlanes[plane][word] holds, one bit per sample in raster order, bit
  // `plane` of every magnitude -- the '1' of the plane a sample was found on (added by k_dec_count) and the
  // refinement bits of the planes below (k_ref_deposit: one 8-byte store per mask word instead of a 4-byte
  // read-modify-write per candidate scattered over 64 MB).  A word of a plane is valid from the plane on which
  // the word's first sample became significant downwards: wordTop[word] = 1 + that plane (0: none), so nothing
  // is cleared beforehand and k_ref_assemble reads only the planes a word has.  k_ref_assemble writes every
  // coefficient once after the last plane (src/SPECK_INT.cpp:359-469: found at p0, refined down to q ->
  // magnitude bits + 2^(q-1) - 1).  refMask: the candidates that got a bit in a pass the stream's end cut
  // short.  nullptr: k_ref_apply2 updates the coefficients plane by plane (64-bit coefficients, the 2D walk).
  //
  // Round 6: the planes take NO memory of their own -- they live in the coefficient array, which nobody reads or
  // writes before k_ref_assemble: 32 planes x 8 bytes per mask word are exactly the 64 x 4 bytes of the word's
  // coefficients.  Layout (ref_plane_word below): tiles of eight mask words, a tile = 32 plane slots of eight
  // consecutive words = 2 KB = the coefficients of those eight mask words -- k_ref_deposit's eight neighbouring
  // threads store 64 contiguous bytes, and k_ref_assemble, whose wavefront takes exactly such a tile per round,
  // reads the tile's plane words (512 contiguous bytes per eight planes) into registers before it writes the
  // tile's 512 coefficients over them.  64 MB less per 256^3 chunk in flight (371 -> 307 MB).
  uint64_t* refPlanes;         // == coef (32-bit coefficients), or nullptr
  size_t refPlaneStride;       // 8-byte words per chunk (coefStride / 2)
  uint32_t refNPlanes;
  uint64_t* refMask;
  uint8_t* wordTop;
  size_t wordTopStride;
  uint32_t coefSigned;         // k_ref_assemble: the sign goes into the coefficient word where the chunk allows it (coef_scheme)
  uint64_t* sign;              // initialised to all ones (SPECK_INT.cpp:174-175)
  size_t signStride;
  uint64_t* lis[2];
  size_t lisStride;
  const uint32_t* levelOff;
  uint32_t nPixTiles;
  uint32_t* tileLip;
  uint32_t* tileRef;
  uint32_t* tileLipOff;
  uint32_t* tileRefOff;
  size_t tileStride;
  uint8_t* tileBorn;           // 1: some sample of the tile has been tested (k_dec_count skips the others); null: not kept
  uint64_t* lipSig;            // LIP scan results of the current plane by token rank: found
  uint64_t* lipNeg;            //   significant / and negative
  size_t lipResStride;
  uint64_t* tokMask;           // token starts of every 64-bit word of the LIP scan
  uint32_t* tokCnt;
  uint32_t* tokOff;            // ... rank of a word's first token INSIDE its segment of kLipSeg words (k_lip_words)
  size_t tokStride;
  uint32_t* tokSegSum;         // tokens per segment, and (k_lip_scan) the rank of a segment's first token
  uint32_t* tokSegBase;
  size_t tokSegStride;
  // table-driven LIS phase (regular shapes)
  const spk::LevelClass* levelClass;
  const uint8_t* levelSlot;    // level -> birth-mask slot (0xff: none)
  const uint8_t* slotLevel;
  uint32_t nSlots;
  uint32_t maskWords;
  uint64_t* mask;              // [slot][maskWords] one bit per stream position of the phase
  uint32_t* maskPrefix;        // popcount prefix of a slot's mask, one word per FOUR mask words (round 6: a word each before,
                               //   14 MiB per 256^3 chunk): rank = prefix of the group + the group's words in front + the bits in front
  uint32_t prefWords;          //   prefix words per slot ((maskWords + 3) / 4)
  size_t prefStride;           //   ... per chunk (nSlots * prefWords)
  size_t maskStride;
  uint64_t* bornPacked;
  uint64_t* bornPosLev;
  size_t bornStride;           // slots of the shared part (claimed with atomics on DecState::bornCount)
  size_t bornPitch;            // slots per chunk: the shared part + hiGroupsMax segments of bornSeg
  uint32_t bornSeg;            // k_lis_hi: every workgroup of a chunk fills a segment of its own
  uint32_t leafSeg;            //   (same for the leaf events, behind leafCap shared slots)
  uint64_t* queue;             // two work queues of queueCap items (2 words each)
  uint32_t queueCap;
  size_t queueStride;
  uint64_t* sigbits;           // significance bit of every old entry of the level being decoded
  size_t sigbitsStride;
  uint32_t treeTabLen;         // entries of tree.tab (k_lis_walk stages the tree's tables in LDS)
  uint64_t* leafEv;            // leaf-parent splits of one plane: node id | sig mask | neg mask
  uint32_t leafCap;
  size_t leafStride;
  // pixel results of the leaf sets of spk::kGridLeafWord grids, by flat node id: low byte =
  // children found significant when the leaf split (never 0 for a split leaf), high byte = the
  // negative ones.  wordLeaf[w] = flat id of the first of the 32 leaves under raster mask word w
  // (a multiple of 32) | (row parity selector / 2), or 0xffffffff; nullptr when no word qualifies.
  uint16_t* leafState;
  size_t leafStateStride;
  uint8_t* leafDirty;          // per block of 32 leaves: 1 + the plane on which one of them split last
  size_t leafDirtyStride;
  const uint32_t* wordLeaf;
  // k_lis_l0: one look-back word per block of kL0W stream bits (tagged with the plane)
  unsigned long long* l0Flags;
  size_t l0FlagStride;
  int32_t l0Level;             // LIS level it handles, or -1
  unsigned long long* l0Tab;   // 17 words per block: its memo table (exit offset, entries, significant entries per entry offset), tagged like the look-back words; null: blocks wait for their predecessor's state only
  unsigned long long* l1Flags; // same for k_lis_l1
  int32_t l1Level;
  unsigned long long* l2Flags; // and k_lis_l2 (blocks of 4096 bits like k_lis_l1's: l0FlagStride words per chunk)
  int32_t l2Level;             // LIS level of the 8x8x8 sets it handles, or -1
  uint64_t* lisStamps;         // diagnostics: 16 tick counters per chunk, or nullptr
  // k_lis_hi: four look-back words per region of hiW stream bits (tagged with the plane)
  unsigned long long* hiFlags;
  size_t hiFlagStride;         // words per chunk
  uint32_t hiW;                // region bits (tab_window of the shape's longest class chain)
  uint32_t hiK;                // classes the LDS tables have room for
  uint32_t hiSmemBytes;        // dynamic LDS given to k_lis_hi
  uint32_t hiHop2;             // the next class's pointer-jump table is built ahead of the chain (else on demand)
  uint32_t hiGroupsMax;        // workgroups per chunk the queues are sized for (<= 8)
  uint32_t hiExtra;            // classes built speculatively beyond the hinted list's own
  uint32_t hiCand;             // class tables only where a split can start (build_tables, round 6; SPERR_HIP_HI_CAND=0: everywhere)
  uint32_t hiAhead;            // bits of a region's tables past the region's end: items that start in
                               //   the region and end within them are not walked into
  const uint64_t* iRoots;         // 2D coder (spk::kTree2D): packed roots of the subbands the type-I set releases,
  uint32_t iLevels;               //   three per level from the coarsest on (~0: empty); iLevels: transform levels
  // k_lis_mx (speck_mx.hip; chunks whose lists mix set shapes, slices) -- fixed regions of mxS stream bits handed out by a ticket
  // counter (DecState::hiTicket), rows of sixteen columns over mxS + mxM bits built off the serial chain, the
  // walker's state handed from region to region through DecBuffers::hiFlags (kMxWordsPerRegion words each)
  const uint8_t* mxSlot;          // column of every shape class (0xff: none)
  const uint8_t* mxLevelGroup;    // per list level: dominant column group | highest group << 4
  uint32_t mxS, mxM;              // bits of a region, bits its rows look further
  uint32_t mxQ;                   // items per expansion queue
  uint32_t mxSmemBytes;
};

struct DecPlanHost {
  const uint64_t* d_initLIS;
  const uint32_t* d_initLen;
  bool tables;                 // every LIS level is regular: the table kernels (k_lis_l0 / _l1 / _hi)
  bool l0;                     // the level of the smallest sets is made of 2x2x2 leaf sets: k_lis_l0
  bool l1;                     // and the next one of 4x4x4 sets: k_lis_l1
  int maxK;                    // longest class chain (sizes the LDS tables)
  bool hi = false;             // the other lists GPU-wide (k_lis_hi) instead of one workgroup per chunk
  bool l2 = false;             // the list after k_lis_l1's, of 8x8x8 sets: k_lis_l2 (round 6)
  bool mixed = false;          // lists that mix set shapes: k_lis_mx (shape-class rows, several workgroups per chunk) instead of k_lis_walk
  uint32_t gridDiv = 1;        // the per-plane kernels' grid caps divided by this: a batch that decodes beside other shape groups
                               //   whose k_lis_mx workgroups hold most CUs (1000^3 in 256^3 chunks: 150 -> 146 ms with 4)
  uint32_t mxGroups = 0;       // workgroups per chunk of k_lis_mx (0: the launcher's own choice by the batch's size) --
                               //   the caller knows how many such chunks of OTHER shapes decode beside this batch
  bool skipFinish = false;     // the caller's inverse quantiser completes the coefficients
  // Fixed-rate streams run out of bits many planes above plane 0, and the launches of a plane that
  // holds no work still cost about 0.1 ms per batch.  With d_live set (kLiveSlots device words) the
  // launcher asks after 16 planes, and then after every second, how many chunks still decode -- a
  // small kernel, a copy into h_live (pinned) and an event -- and stops launching when the answer to
  // the PREVIOUS question is "none": the host waits for an event that lies two planes back in the
  // queue, never for the stream itself, so the device does not run dry.  Only for a caller whose
  // host thread may block (the other sub-batches of a call are enqueued by threads of their own).
  uint32_t* d_live = nullptr;
  uint32_t* h_live = nullptr;
  hipEvent_t* liveEv = nullptr;
};
constexpr int kLiveSlots = 24;

constexpr int kTabWMax = 28672;   // window bits: < 2^15 (table entries keep a flag in bit 15)
// LDS bytes per window bit for a level with chain length K: T_0..T_{K-2} and U_0..U_{K-1} (u16
// each), the hop word (u32) and the bit itself.  The window is the largest multiple of 1024 that
// fits.
__host__ __device__ inline uint32_t tab_window(int K, uint32_t smemBytes)
{
  const uint32_t perBit8 = 8u * (uint32_t)(2 * (2 * K - 1) + 4) + 1u;   // eighths of a byte
  const uint32_t fixed = 4 * 8 + 130 * 4 + (uint32_t)(2 * K) * 8 + 64;
  uint32_t w = (uint32_t)(((uint64_t)(smemBytes - fixed) * 8) / perBit8);
  w = w / 1024 * 1024;
  if (w > (uint32_t)kTabWMax)
    w = kTabWMax;
  return w;
}

// the same for k_lis_hi, which keeps two pointer-jump tables (the list's class and the next one)
// (hop2 == 0: one pointer-jump table only -- the second is built on demand into the first's place --, 4 bytes
//  per position less: 6912 instead of 5632 positions for a 256^3 chunk)
__host__ __device__ inline uint32_t hi_window(int K, uint32_t smemBytes, uint32_t hop2 = 1)
{
  const uint32_t perBit8 = 8u * (uint32_t)(2 * (2 * K - 1) + (hop2 ? 8 : 4)) + 1u;   // eighths of a byte
  const uint32_t fixed = 4 * 8 + 2 * 130 * 4 + (uint32_t)(2 * K) * 8 + 64;
  uint32_t w = (uint32_t)(((uint64_t)(smemBytes - fixed) * 8) / perBit8);
  w = w / 256 * 256;   // (any multiple of 64 works; round 2 took multiples of 1024: 4096 instead of 4352 bits for a 256^3 chunk)
  if (w > (uint32_t)kTabWMax)
    w = kTabWMax;
  return w;
}

// k_lis_mx (speck_mx.hip)
constexpr int kMxWordsPerRegion = 8;
constexpr uint32_t kMxS = 2048, kMxM = 768, kMxQ = 1024, kMxRing = 8192;
uint32_t mx_smem_bytes(uint32_t S, uint32_t M, uint32_t Q);
int prepare_lis_mx(const DecBuffers& b);
int launch_lis_mx(hipStream_t stream, const DecBuffers& b, int p, uint32_t groups, bool stamps);

int launch_speck_decode(hipStream_t stream, const DecBuffers& b, const DecPlanHost& plan,
                        const uint8_t* container, const uint64_t* d_chunkOff,
                        const uint64_t* d_chunkLen, bool wide_pass, int maxPlanes);

}  // namespace sperrhip
#endif
