// speck_model.cpp -- CPU model of the DATA-PARALLEL SPECK3D formulation used by the HIP kernels.
//
// TEST INFRASTRUCTURE.  Every "kernel" of sperr_amd/csrc/speck_kernels.hip is modelled here as a
// plain loop over its thread index, calling the same shared geometry / per-node logic
// (sperr_amd/csrc/speck_tree.h).  tests/test_speck_model.py checks the model bit for bit against
// the oracle, which pins the formulation (count -> scan -> emit, no serial traversal) on the CPU
// before any GPU time is spent.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../sperr_amd/csrc/speck_tree_host.hpp"
#include "../../sperr_amd/csrc/bit_words.h"

using namespace spk;

namespace {

struct BitSink {
  std::vector<uint64_t> w;
  uint64_t limit;  // bits at positions >= limit are dropped
  void put(uint64_t pos, int bit)
  {
    if (pos >= limit || !bit)
      return;
    w[pos >> 6] |= uint64_t(1) << (pos & 63);
  }
};

inline int msb_of(uint64_t v)
{
  return v ? 63 - __builtin_clzll(v) : -1;
}

struct Enc {
  HostTree ht;
  Tree t;
  std::vector<int8_t> M, msb, bplane;
  std::vector<uint32_t> E;
  std::vector<uint64_t> opos;
};

// "kernel" E1: one thread per node of every grid at depth d
void build_depth(Enc& s, uint32_t depth)
{
  const Tree& t = s.t;
  for (uint32_t gi = 0; gi < t.ngrids; gi++) {
    const Grid& g = t.grids[gi];
    if (g.depth != depth)
      continue;
    const uint32_t n = 1u << (g.e[0] + g.e[1] + g.e[2]);
    for (uint32_t local = 0; local < n; local++) {
      Node nd;
      const uint32_t id = g.nodeOff + local;
      if (!node_from_flat(t, id, nd))
        abort();
      const NodeGeom q = node_geom(t, nd);
      if (q.count == 0)
        continue;
      const Root& r = t.roots[g.root];
      const bool isset = q.count > 1 || g.depth == 0;
      if (!isset) {
        const int e[3] = {g.e[0], g.e[1], g.e[2]};
        const uint32_t ii[3] = {nd.i[0], nd.i[1], nd.i[2]};
        s.M[id] = s.msb[pixel_raster(t, r, e, ii)];
        continue;
      }
      Kids k;
      node_kids(t, nd, k);
      KidInfo ki;
      kids_info(t, nd, k, s.M.data(), s.E.data(), s.msb.data(), ki);
      int m = -1;
      for (int j = 0; j < k.n; j++)
        m = std::max<int>(m, ki.m[j]);
      s.M[id] = (int8_t)m;
      s.E[id] = m >= 0 ? split_bits(k, ki, m) : 0;
      for (int j = 0; j < k.n; j++)
        if (ki.pixel[j])
          s.bplane[kid_raster(t, nd, k, j)] = (int8_t)m;
    }
  }
}

struct Born {
  uint32_t lev;
  uint64_t pos;
  uint64_t packed;
};

// "kernel" K2: one thread per set node; acts only when the node splits at plane p
void emit_splits(Enc& s, int p, BitSink& out, const std::vector<uint64_t>& sign,
                 std::vector<Born>& born)
{
  const Tree& t = s.t;
  for (uint32_t id = 0; id < t.nnodes; id++) {
    Node nd;
    if (!node_from_flat(t, id, nd))
      continue;
    const NodeGeom q = node_geom(t, nd);
    const bool isset = q.count > 1 || (t.grids[nd.grid].depth == 0 && q.count == 1);
    if (!isset || s.M[id] != p)
      continue;
    // walk up to the list entry that started this split chain
    uint64_t off = 0;
    Node cur = nd;
    uint32_t curid = id;
    while (!node_is_root(t, cur)) {
      const Node par = node_parent(t, cur);
      const uint32_t pid = flat_id(t, par);
      if (s.M[pid] != p)
        break;
      Kids pk;
      node_kids(t, par, pk);
      KidInfo pki;
      kids_info(t, par, pk, s.M.data(), s.E.data(), s.msb.data(), pki);
      int which = -1;
      for (int j = 0; j < pk.n; j++)
        if (pk.idx[j][0] == cur.i[0] && pk.idx[j][1] == cur.i[1] && pk.idx[j][2] == cur.i[2])
          which = j;
      if (which < 0)
        abort();
      bool coded;
      off += kid_offset(pk, pki, p, which, coded);
      off += coded ? 1 : 0;
      cur = par;
      curid = pid;
    }
    uint64_t pos = s.opos[curid] + 1 + off;  // first bit of this node's split

    Kids k;
    node_kids(t, nd, k);
    KidInfo ki;
    kids_info(t, nd, k, s.M.data(), s.E.data(), s.msb.data(), ki);
    const uint32_t kidlev = kid_level(t, nd, q);
    bool found = false;
    for (int j = 0; j < k.n; j++) {
      const bool coded = found || (j + 1 != k.n);
      const bool sig = coded ? (ki.m[j] == p) : true;
      if (coded)
        out.put(pos++, sig);
      if (sig) {
        found = true;
        if (ki.pixel[j]) {
          const uint32_t ridx = kid_raster(t, nd, k, j);
          out.put(pos++, (int)((sign[ridx >> 6] >> (ridx & 63)) & 1));
        }
        else
          pos += ki.e[j];
      }
      else if (!ki.pixel[j])
        born.push_back({kidlev, pos - 1, pack_node(kid_node(k, j))});
    }
  }
}

}  // namespace

extern "C" {

// Same contract as orc_speck3d_encode (oracle/sperr_oracle.h)
static int model_encode_impl(const uint64_t* coeffs, const uint64_t* signs, const size_t dims[3],
                             size_t budget_bits, uint8_t** stream, size_t* stream_len, bool twoD)
{
  Enc s;
  s.ht = build_tree(dims[0], dims[1], dims[2], twoD);
  s.t = s.ht.view();
  const Tree& t = s.t;
  const size_t N = t.nvals;
  uint64_t budget = budget_bits ? budget_bits : ~uint64_t(0);
  if (budget_bits)
    while (budget % 8)
      budget++;

  s.msb.resize(N);
  for (size_t i = 0; i < N; i++)
    s.msb[i] = (int8_t)msb_of(coeffs[i]);
  s.M.assign(t.nnodes, -1);
  s.E.assign(t.nnodes, 0);
  s.opos.assign(t.nnodes, 0);
  s.bplane.assign(N, -1);
  for (uint32_t d = t.maxDepth; d-- > 0;)
    build_depth(s, d);

  int nbp = 0;
  for (uint32_t r = 0; r < t.nroots; r++)
    nbp = std::max<int>(nbp, s.M[t.grids[t.roots[r].gridFirst].nodeOff] + 1);

  // census (kernel E2): bits per plane of the LIP scans and refinement passes
  std::vector<uint64_t> lipTot(64, 0), refTot(64, 0);
  for (size_t i = 0; i < N; i++) {
    const int m = s.msb[i], b = s.bplane[i];
    for (int p = 0; p < b && p < nbp; p++)
      if (p >= m) {
        lipTot[p] += 1;           // one test bit per LIP scan while insignificant
        if (p == m)
          lipTot[p] += 1;         // sign bit on the scan that finds it
      }
    for (int p = 0; p < m; p++)
      refTot[p] += 1;             // one refinement bit on every later plane
  }

  std::vector<std::vector<uint64_t>> lis(s.ht.initLIS), next(t.nlevels);
  uint32_t iPart = s.ht.iLevels;   // 2D: part_level of what is left of the type-I set
  std::vector<uint64_t> baseLIP(64, 0), baseLIS(64, 0), baseREF(64, 0);
  std::vector<char> didREF(64, 0);
  uint64_t pos = 0;
  int plast = nbp;  // lowest plane whose sorting pass ran
  BitSink out;
  out.limit = budget;
  // upper bound of the stream length is unknown before the plane loop; grow as needed
  std::vector<Born> born;
  std::vector<std::pair<int, std::vector<Born>>> unused;
  // first pass over planes: positions (K1) + set-split emission (K2)
  // the sink must be large enough: worst case every node coded every plane; size lazily
  out.w.assign(1, 0);
  auto ensure = [&](uint64_t bits) {
    uint64_t lim = std::min<uint64_t>(bits, budget);
    if ((lim + 127) / 64 > out.w.size())
      out.w.resize((lim + 127) / 64, 0);
  };
  for (int p = nbp - 1; p >= 0; p--) {
    plast = p;
    baseLIP[p] = pos;
    pos += lipTot[p];
    baseLIS[p] = pos;
    // kernel K1: positions of the list entries, list compaction
    for (uint32_t l = t.nlevels; l-- > 0;) {
      next[l].clear();
      for (uint64_t packed : lis[l]) {
        const Node nd = unpack_node(packed);
        const uint32_t id = flat_id(t, nd);
        if (s.M[id] == p) {
          s.opos[id] = pos;
          pos += 1 + s.E[id];
        }
        else {
          pos += 1;
          next[l].push_back(packed);
        }
      }
    }
    ensure(pos);
    // list entries' own test bits
    for (uint32_t l = 0; l < t.nlevels; l++)
      for (uint64_t packed : lis[l]) {
        const uint32_t id = flat_id(t, unpack_node(packed));
        if (s.M[id] == p)
          out.put(s.opos[id], 1);
      }
    born.clear();
    // 2D coder (k_enc_iphase): the type-I set at the end of the sorting pass (SPECK2D_INT.cpp:44-98).
    // Its test bit; when it is significant the three subbands of its level: a significant one splits on
    // the spot (its split is written by emit_splits like that of a list entry), an insignificant one
    // joins the list of its level; then what is left of it, implied when none of the three was
    if (twoD) {
      bool need = true;
      while (iPart > 0) {
        if (need) {
          int mi = -1;
          for (size_t k = (size_t)(s.ht.iLevels - iPart) * 3; k < (size_t)s.ht.iLevels * 3; k++)
            if (s.ht.iRoots[k] != HostTree::kNoRoot)
              mi = std::max<int>(mi, s.M[flat_id(t, unpack_node(s.ht.iRoots[k]))]);
          ensure(pos + 1);
          out.put(pos++, mi >= p);
          if (mi < p)
            break;
        }
        int counter = 0;
        for (int j = 0; j < 3; j++) {
          const uint64_t root = s.ht.iRoots[(size_t)(s.ht.iLevels - iPart) * 3 + j];
          if (root == HostTree::kNoRoot)
            continue;
          const uint32_t id = flat_id(t, unpack_node(root));
          if (s.M[id] >= p) {
            ensure(pos + 2 + s.E[id]);
            out.put(pos, 1);
            s.opos[id] = pos;
            pos += 1 + s.E[id];
            counter++;
          }
          else {
            ensure(pos + 1);
            out.put(pos, 0);
            born.push_back({iPart, pos, root});
            pos++;
          }
        }
        iPart--;
        need = counter != 0;
      }
      ensure(pos);
    }
    // kernel K2 + K2b
    emit_splits(s, p, out, std::vector<uint64_t>(signs, signs + (N + 63) / 64), born);
    std::stable_sort(born.begin(), born.end(), [](const Born& a, const Born& b) {
      return a.lev != b.lev ? a.lev < b.lev : a.pos < b.pos;
    });
    for (const Born& b : born)
      next[b.lev].push_back(b.packed);
    lis.swap(next);
    if (pos >= budget)
      break;
    baseREF[p] = pos;
    pos += refTot[p];
    didREF[p] = 1;
    if (pos >= budget)
      break;
  }
  const uint64_t total_bits = nbp ? pos : 0;
  ensure(total_bits);

  // kernel E5: LIP-scan and refinement bits, raster order
  for (int p = nbp - 1; p >= plast && nbp; p--) {
    uint64_t lp = baseLIP[p], rp = baseREF[p];
    for (size_t i = 0; i < N; i++) {
      const int m = s.msb[i], b = s.bplane[i];
      if (b > p && p >= m) {
        out.put(lp++, m == p);
        if (m == p)
          out.put(lp++, (int)((signs[i >> 6] >> (i & 63)) & 1));
      }
      if (didREF[p] && m > p)
        out.put(rp++, (int)((coeffs[i] >> p) & 1));
    }
  }

  const uint64_t keep = std::min<uint64_t>(total_bits, budget);
  const size_t nbytes = (size_t)((keep + 7) / 8);
  uint8_t* o = (uint8_t*)calloc(9 + nbytes + 8, 1);
  o[0] = (uint8_t)nbp;
  memcpy(o + 1, &total_bits, 8);
  memcpy(o + 9, out.w.data(), nbytes);
  *stream = o;
  *stream_len = 9 + nbytes;
  return 0;
}

int model_speck3d_encode(const uint64_t* coeffs, const uint64_t* signs, const size_t dims[3],
                         size_t budget_bits, uint8_t** stream, size_t* stream_len)
{
  return model_encode_impl(coeffs, signs, dims, budget_bits, stream, stream_len, false);
}

// the 2D coder's stream of a slice (dims = {x, y, 1}) from the same formulation on the 2D forest
int model_speck2d_encode(const uint64_t* coeffs, const uint64_t* signs, const size_t dims[3],
                         size_t budget_bits, uint8_t** stream, size_t* stream_len)
{
  return model_encode_impl(coeffs, signs, dims, budget_bits, stream, stream_len, true);
}


// ------------------------------------------------------------------------------------------
// Decoder model.  Per plane: (D1) LIP scan parsed with the run-parity rule, (D2) LIS phase as a
// serial walk (one thread per chunk in the kernel), (D3) refinement as a gather.
// ------------------------------------------------------------------------------------------
namespace {

struct BitSrc {
  const uint8_t* p;
  uint64_t avail;  // bits past `avail` read as zero (zero padding of a truncated stream)
  int get(uint64_t pos) const { return pos < avail ? (p[pos >> 3] >> (pos & 7)) & 1 : 0; }
};

}  // namespace

// Same contract as orc_speck3d_decode (oracle/sperr_oracle.h)
int model_speck3d_decode(const uint8_t* stream, size_t len, const size_t dims[3], uint64_t* coef,
                         uint64_t* sign)
{
  HostTree ht = build_tree(dims[0], dims[1], dims[2]);
  const Tree t = ht.view();
  const size_t N = t.nvals;
  const int nbp = stream[0];
  uint64_t total_bits;
  memcpy(&total_bits, stream + 1, 8);
  uint64_t avail = (uint64_t)(len - 9) * 8;
  if (avail > total_bits)
    avail = total_bits;
  BitSrc in{stream + 9, avail};

  std::vector<int8_t> born(N, -1), sigp(N, -1);
  memset(coef, 0, N * sizeof(uint64_t));
  memset(sign, 0xff, ((N + 63) / 64) * 8);
  auto set_sign = [&](uint32_t i, int b) {
    if (b)
      sign[i >> 6] |= uint64_t(1) << (i & 63);
    else
      sign[i >> 6] &= ~(uint64_t(1) << (i & 63));
  };
  std::vector<std::vector<uint64_t>> lis(ht.initLIS), next(t.nlevels);
  uint64_t pos = 0;
  for (int p = nbp - 1; p >= 0; p--) {
    const uint64_t thr = uint64_t(1) << p;
    const uint64_t init = thr + thr - thr / 2 - 1;
    // ---- D1: LIP scan.  candidates in raster order; token k starts where the number of
    //      consecutive 1 bits right before it is even.
    std::vector<uint32_t> cand;
    for (size_t i = 0; i < N; i++)
      if (born[i] > p && sigp[i] < 0)
        cand.push_back((uint32_t)i);
    {
      size_t j = 0;
      uint64_t k = 0, ones = 0;
      for (; j < cand.size(); k++) {
        const int b = in.get(pos + k);
        if ((ones & 1) == 0) {  // token start
          if (b) {
            sigp[cand[j]] = (int8_t)p;
            coef[cand[j]] = init;
            set_sign(cand[j], in.get(pos + k + 1));
          }
          j++;
        }
        ones = b ? ones + 1 : 0;
      }
      // the phase ends where token #cand.size() would start
      if (ones & 1)
        k++;  // the last token was "1 s": its sign bit is consumed too
      pos += k;
    }
    // ---- D2: LIS phase, serial
    for (uint32_t l = 0; l < t.nlevels; l++)
      next[l].clear();
    struct Frame {
      Node nd;
    };
    for (uint32_t l = t.nlevels; l-- > 0;) {
      for (uint64_t packed : lis[l]) {
        if (!in.get(pos++)) {
          next[l].push_back(packed);
          continue;
        }
        // significant: split depth-first with an explicit stack of (node, next child)
        struct Item {
          Node nd;
          Kids k;
          int j;
          bool found;
          uint32_t kidlev;
        };
        std::vector<Item> st;
        auto push = [&](const Node& nd) {
          Item it;
          it.nd = nd;
          node_kids(t, nd, it.k);
          it.j = 0;
          it.found = false;
          const NodeGeom q = node_geom(t, nd);
          it.kidlev = node_level(t, nd) + (q.len[0] > 1) + (q.len[1] > 1) + (q.len[2] > 1);
          st.push_back(it);
        };
        push(unpack_node(packed));
        while (!st.empty()) {
          Item& it = st.back();
          if (it.j == it.k.n) {
            st.pop_back();
            continue;
          }
          const int j = it.j++;
          const bool coded = it.found || (j + 1 != it.k.n);
          const bool sig = coded ? in.get(pos++) : true;
          const bool pixel = it.k.count[j] == 1;
          if (sig)
            it.found = true;
          if (pixel) {
            const uint32_t ridx = kid_raster(t, it.nd, it.k, j);
            born[ridx] = (int8_t)p;
            if (sig) {
              sigp[ridx] = (int8_t)p;
              coef[ridx] = init;
              set_sign(ridx, in.get(pos++));
            }
          }
          else if (sig) {
            const Node kid = kid_node(it.k, j);
            push(kid);  // invalidates `it`
          }
          else
            next[it.kidlev].push_back(pack_node(kid_node(it.k, j)));
        }
      }
    }
    lis.swap(next);
    if (pos >= avail)
      break;
    // ---- D3: refinement, j-th significant pixel (raster order) takes bit pos + j
    {
      const uint64_t half = thr / 2;
      uint64_t j = 0;
      for (size_t i = 0; i < N && pos + j < avail; i++)
        if (sigp[i] > p) {
          const int b = in.get(pos + j);
          j++;
          if (thr >= 2)
            coef[i] = b ? coef[i] + half : coef[i] - half;
          else if (b)
            coef[i]++;
        }
      pos += j;
    }
    if (pos >= avail)
      break;
  }
  return 0;
}


// ------------------------------------------------------------------------------------------
// Model of the PARALLEL LIS-phase decoder (kernel k_lis_tables in speck_dec.hip).
//
// For a list whose entries all have the same power-of-two shape ("regular" level) the code of an
// entry is '0' or '1' + the split of a class-(K-1) set; a class-j set has arity[j] children of
// class j-1 (class 0: children are pixels).  For a window of W stream bits the kernel computes,
// for EVERY bit position x and every class j, T_j[x] = number of bits the split of a class-j set
// would take if it started at x (INF when it would leave the window).  That is speculative and
// embarrassingly parallel.  A cheap serial walk then only hops over entries (1 + T lookups), and
// every set that splits inside the window is expanded by an independent thread that locates its
// children with T_{j-1}.  Sets that stay insignificant are collected with their stream position;
// their order in the next list is the order of those positions.
// ------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t T_INF = 0xffffffffu;

struct ParCtx {
  int cls;          // class of the items of this context
  int remaining;    // items left
  bool found;       // an earlier sibling was significant
  bool top;         // base context: list entries, always coded
  Node parent;      // parent node of the items (unused for top)
  int nextOrdinal;  // ordinal of the next child
};

struct WorkItem {
  Node nd;
  int cls;
  uint64_t pos;     // first bit of the node's split
};

// child `ord` (x-fastest over the axes that split) of a regular node
Node regular_child(const Tree& t, const Node& nd, int ord)
{
  const Grid& g = t.grids[nd.grid];
  const Root& r = t.roots[g.root];
  Node c;
  c.grid = (uint16_t)(nd.grid + 1);
  int bit = 0;
  for (int a = 0; a < 3; a++) {
    if (g.depth < r.D[a]) {
      c.i[a] = (uint16_t)(nd.i[a] * 2 + ((ord >> bit) & 1));
      bit++;
    }
    else
      c.i[a] = nd.i[a];
  }
  return c;
}

uint32_t regular_child_raster(const Tree& t, const Node& nd, int ord)
{
  const Grid& g = t.grids[nd.grid];
  const Root& r = t.roots[g.root];
  int e[3];
  uint32_t idx[3];
  int bit = 0;
  for (int a = 0; a < 3; a++) {
    if (g.depth < r.D[a]) {
      e[a] = g.e[a] + 1;
      idx[a] = nd.i[a] * 2u + ((ord >> bit) & 1);
      bit++;
    }
    else {
      e[a] = g.e[a];
      idx[a] = nd.i[a];
    }
  }
  return pixel_raster(t, r, e, idx);
}

}  // namespace

int g_model_window = 256;  // window size in bits (tests shrink it to stress the boundaries)

// same contract as model_speck3d_decode; regular levels go through the table-driven path
int model_speck3d_decode_par(const uint8_t* stream, size_t len, const size_t dims[3],
                             uint64_t* coef, uint64_t* sign, int window)
{
  if (window > 0)
    g_model_window = window;
  HostTree ht = build_tree(dims[0], dims[1], dims[2]);
  const Tree t = ht.view();
  const std::vector<LevelClass>& lc = ht.levelClass;
  const size_t N = t.nvals;
  const int nbp = stream[0];
  uint64_t total_bits;
  memcpy(&total_bits, stream + 1, 8);
  uint64_t avail = (uint64_t)(len - 9) * 8;
  if (avail > total_bits)
    avail = total_bits;
  BitSrc in{stream + 9, avail};
  std::vector<int8_t> born(N, -1), sigp(N, -1);
  memset(coef, 0, N * sizeof(uint64_t));
  memset(sign, 0xff, ((N + 63) / 64) * 8);
  auto set_sign = [&](uint32_t i, int b) {
    if (b)
      sign[i >> 6] |= uint64_t(1) << (i & 63);
    else
      sign[i >> 6] &= ~(uint64_t(1) << (i & 63));
  };
  std::vector<std::vector<uint64_t>> lis(ht.initLIS), next(t.nlevels);
  uint64_t pos = 0;
  for (int p = nbp - 1; p >= 0; p--) {
    const uint64_t thr = uint64_t(1) << p;
    const uint64_t init = thr + thr - thr / 2 - 1;
    // ---- D1 (as in model_speck3d_decode)
    {
      std::vector<uint32_t> cand;
      for (size_t i = 0; i < N; i++)
        if (born[i] > p && sigp[i] < 0)
          cand.push_back((uint32_t)i);
      size_t j = 0;
      uint64_t k = 0, ones = 0;
      for (; j < cand.size(); k++) {
        const int b = in.get(pos + k);
        if ((ones & 1) == 0) {
          if (b) {
            sigp[cand[j]] = (int8_t)p;
            coef[cand[j]] = init;
            set_sign(cand[j], in.get(pos + k + 1));
          }
          j++;
        }
        ones = b ? ones + 1 : 0;
      }
      if (ones & 1)
        k++;
      pos += k;
    }
    // ---- D2, table-driven
    struct BornRec {
      uint32_t lev;
      uint64_t pos;
      uint64_t packed;
    };
    std::vector<BornRec> bornv;
    for (uint32_t l = 0; l < t.nlevels; l++)
      next[l].clear();
    auto pixel_event = [&](uint32_t ridx, bool sig, uint64_t signpos) {
      born[ridx] = (int8_t)p;
      if (sig) {
        sigp[ridx] = (int8_t)p;
        coef[ridx] = init;
        set_sign(ridx, in.get(signpos));
      }
    };
    for (uint32_t l = t.nlevels; l-- > 0;) {
      const size_t n = lis[l].size();
      if (n == 0)
        continue;
      if (!lc[l].regular) {
        // fallback: serial walk of this level (identical to model_speck3d_decode's D2)
        for (uint64_t packed : lis[l]) {
          if (!in.get(pos++)) {
            next[l].push_back(packed);
            continue;
          }
          struct Item {
            Node nd;
            Kids k;
            int j;
            bool found;
            uint32_t kidlev;
          };
          std::vector<Item> st;
          auto push = [&](const Node& nd) {
            Item it;
            it.nd = nd;
            node_kids(t, nd, it.k);
            it.j = 0;
            it.found = false;
            const NodeGeom q = node_geom(t, nd);
            it.kidlev = node_level(t, nd) + (q.len[0] > 1) + (q.len[1] > 1) + (q.len[2] > 1);
            st.push_back(it);
          };
          push(unpack_node(packed));
          while (!st.empty()) {
            Item& it = st.back();
            if (it.j == it.k.n) {
              st.pop_back();
              continue;
            }
            const int j = it.j++;
            const bool coded = it.found || (j + 1 != it.k.n);
            const bool sig = coded ? in.get(pos++) : true;
            if (sig)
              it.found = true;
            if (it.k.count[j] == 1) {
              const uint32_t ridx = kid_raster(t, it.nd, it.k, j);
              pixel_event(ridx, sig, pos);
              if (sig)
                pos++;
            }
            else if (sig)
              push(kid_node(it.k, j));
            else
              bornv.push_back({it.kidlev, pos - 1, pack_node(kid_node(it.k, j))});
          }
        }
        continue;
      }
      const LevelClass& C = lc[l];
      const int K = C.K;
      const uint64_t W = (uint64_t)g_model_window;
      // walker state
      std::vector<ParCtx> ctx;
      ctx.push_back({K - 1, (int)n, false, true, Node{}, 0});
      size_t e = 0;  // next list entry
      while (!ctx.empty()) {
        // ---- window [a, a+W): tables (parallel in the kernel: one thread per position)
        const uint64_t a = pos, lim = pos + W;
        std::vector<std::vector<uint32_t>> T(K, std::vector<uint32_t>(W + 1, T_INF));
        for (int j = 0; j < K; j++)
          for (uint64_t x = a; x <= lim; x++) {
            uint64_t y = x;
            bool found = false, ok = true;
            for (int i = 0; i < C.arity[j] && ok; i++) {
              const bool coded = found || (i + 1 != C.arity[j]);
              int b = 1;
              if (coded) {
                if (y >= lim) {
                  ok = false;
                  break;
                }
                b = in.get(y++);
              }
              if (!b)
                continue;
              found = true;
              if (j == 0) {  // pixel child: sign bit
                if (y >= lim) {
                  ok = false;
                  break;
                }
                y++;
              }
              else {
                if (y > lim || T[j - 1][y - a] == T_INF) {
                  ok = false;
                  break;
                }
                y += T[j - 1][y - a];
              }
            }
            if (ok)
              T[j][x - a] = (uint32_t)(y - x);
          }
        // ---- hop, part S: contexts below the list (an entry larger than a window is being
        //      walked into) are handled by one thread, serially
        std::vector<WorkItem> queue;
        bool window_full = false;
        while (ctx.size() > 1 && !window_full) {
          ParCtx& c = ctx.back();
          if (c.remaining == 0) {
            ctx.pop_back();
            continue;
          }
          const bool coded = c.found || c.remaining > 1;
          uint64_t x = pos;
          int b = 1;
          if (coded) {
            if (x >= lim) {
              window_full = true;
              break;
            }
            b = in.get(x);
            x++;
          }
          const bool is_pixel = c.cls < 0;
          if (is_pixel) {
            if (b && x >= lim) {
              window_full = true;
              break;
            }
            pixel_event(regular_child_raster(t, c.parent, c.nextOrdinal), b != 0, x);
            if (b) {
              x++;
              c.found = true;
            }
            c.remaining--;
            c.nextOrdinal++;
            pos = x;
            continue;
          }
          const Node nd = regular_child(t, c.parent, c.nextOrdinal);
          if (!b) {
            bornv.push_back({C.lev[c.cls], x - 1, pack_node(nd)});
            c.remaining--;
            c.nextOrdinal++;
            pos = x;
            continue;
          }
          const uint32_t tl = (x <= lim) ? T[c.cls][x - a] : T_INF;
          if (tl == T_INF && x - (coded ? 1 : 0) != a) {
            window_full = true;
            break;
          }
          const int cls = c.cls;
          c.found = true;
          c.remaining--;
          c.nextOrdinal++;
          if (tl != T_INF) {
            queue.push_back({nd, cls, x});
            pos = x + tl;
          }
          else {
            pos = x;
            ctx.push_back({cls - 1, C.arity[cls], false, false, nd, 0});
          }
        }
        // ---- hop, part P: the list itself.  64-bit blocks aligned to absolute word
        //      boundaries; per block a backward memo gives, for every position, where the chain
        //      leaves the block and how many entries it passes; one thread then walks the
        //      blocks; the blocks emit their entries in parallel.
        if (ctx.size() == 1 && !window_full && ctx[0].remaining > 0 && pos < lim) {
          ParCtx& c = ctx[0];
          const uint64_t B0 = pos >> 6, B1 = (lim - 1) >> 6;
          const size_t nblk = (size_t)(B1 - B0 + 1);
          std::vector<uint64_t> exitp((size_t)nblk * 64, 0);
          std::vector<uint32_t> cnt((size_t)nblk * 64, 0);
          std::vector<char> stop((size_t)nblk * 64, 0);
          for (size_t bi = 0; bi < nblk; bi++)        // parallel over blocks
            for (int o = 63; o >= 0; o--) {
              const uint64_t x = (B0 + bi) * 64 + o;
              const size_t k = bi * 64 + o;
              if (x < pos || x >= lim)
                continue;
              uint64_t nx;
              if (!in.get(x))
                nx = x + 1;
              else {
                const uint32_t tl = (x + 1 <= lim) ? T[K - 1][x + 1 - a] : T_INF;
                if (tl == T_INF) {
                  stop[k] = 1;
                  exitp[k] = x;
                  cnt[k] = 0;
                  continue;
                }
                nx = x + 1 + tl;
              }
              if (nx >= (B0 + bi + 1) * 64 || nx >= lim) {
                exitp[k] = nx;
                cnt[k] = 1;
                stop[k] = 0;
              }
              else {
                const size_t kn = (size_t)(nx - B0 * 64);
                exitp[k] = exitp[kn];
                cnt[k] = cnt[kn] + 1;
                stop[k] = stop[kn];
              }
            }
          // one thread: walk the blocks
          std::vector<uint64_t> entry(nblk, ~0ull);
          std::vector<uint32_t> base(nblk, 0), limit(nblk, 0);
          uint64_t x = pos, newpos = pos;
          uint32_t total = 0;
          bool stopped = false;
          const uint32_t remaining = (uint32_t)c.remaining;
          while (true) {
            if (x >= lim) {
              newpos = x;
              break;
            }
            const size_t bi = (size_t)((x >> 6) - B0), k = (size_t)(x - B0 * 64);
            entry[bi] = x;
            base[bi] = total;
            if (total + cnt[k] >= remaining) {
              limit[bi] = remaining - total;  // the list ends inside this block
              total = remaining;
              newpos = ~0ull;                 // set by the block below
              break;
            }
            limit[bi] = cnt[k];
            total += cnt[k];
            if (stop[k]) {
              stopped = true;
              newpos = exitp[k];
              break;
            }
            x = exitp[k];
          }
          // parallel over blocks: emit
          for (size_t bi = 0; bi < nblk; bi++) {
            if (entry[bi] == ~0ull)
              continue;
            uint64_t y = entry[bi];
            for (uint32_t k = 0; k < limit[bi]; k++) {
              const size_t ei = e + base[bi] + k;
              if (!in.get(y))
                y += 1;  // stays in the list (collected in order below)
              else {
                const uint32_t tl = T[K - 1][y + 1 - a];
                queue.push_back({unpack_node(lis[l][ei]), K - 1, y + 1});
                y += 1 + tl;
              }
              (void)ei;
            }
            if (newpos == ~0ull && base[bi] + limit[bi] == remaining && limit[bi] > 0)
              newpos = y;
          }
          // survivors keep list order: redo the placeholders sequentially (the kernel uses a
          // significance bitmap + compaction)
          {
            // significance per entry in order
            std::vector<char> sigv(total, 0);
            for (size_t bi = 0; bi < nblk; bi++) {
              if (entry[bi] == ~0ull)
                continue;
              uint64_t y = entry[bi];
              for (uint32_t k = 0; k < limit[bi]; k++) {
                const int bb = in.get(y);
                sigv[base[bi] + k] = (char)bb;
                y += bb ? 1 + T[K - 1][y + 1 - a] : 1;
              }
            }
            for (uint32_t k = 0; k < total; k++)
              if (!sigv[k])
                next[l].push_back(lis[l][e + k]);
          }
          e += total;
          c.remaining -= (int)total;
          if (total > 0)
            c.found = true;
          if (newpos == ~0ull)
            abort();
          const bool progressed = newpos != a;
          pos = newpos;
          if (stopped && !progressed) {
            // the entry at the very start of the window does not fit: walk into it
            const Node nd = unpack_node(lis[l][e]);
            e++;
            c.remaining--;
            pos = pos + 1;  // its '1'
            ctx.push_back({K - 2, C.arity[K - 1], false, false, nd, 0});
          }
        }
        if (!ctx.empty() && ctx.size() == 1 && ctx[0].remaining == 0)
          ctx.pop_back();
        // ---- expand everything that was queued (parallel BFS in the kernel)
        while (!queue.empty()) {
          std::vector<WorkItem> nq;
          for (const WorkItem& w : queue) {
            uint64_t y = w.pos;
            bool found = false;
            const int ar = C.arity[w.cls];
            for (int i = 0; i < ar; i++) {
              const bool coded = found || (i + 1 != ar);
              int b = 1;
              if (coded)
                b = in.get(y++);
              if (w.cls == 0) {
                const uint32_t ridx = regular_child_raster(t, w.nd, i);
                pixel_event(ridx, b != 0, y);
                if (b) {
                  y++;
                  found = true;
                }
              }
              else {
                const Node kid = regular_child(t, w.nd, i);
                if (b) {
                  found = true;
                  nq.push_back({kid, w.cls - 1, y});
                  y += T[w.cls - 1][y - a];
                }
                else
                  bornv.push_back({C.lev[w.cls - 1], y - 1, pack_node(kid)});
              }
            }
          }
          queue.swap(nq);
        }
      }
    }
    std::stable_sort(bornv.begin(), bornv.end(), [](const BornRec& x, const BornRec& y) {
      return x.lev != y.lev ? x.lev < y.lev : x.pos < y.pos;
    });
    for (const BornRec& b : bornv)
      next[b.lev].push_back(b.packed);
    lis.swap(next);
    if (pos >= avail)
      break;
    // ---- D3
    {
      const uint64_t half = thr / 2;
      uint64_t j = 0;
      for (size_t i = 0; i < N && pos + j < avail; i++)
        if (sigp[i] > p) {
          const int b = in.get(pos + j);
          j++;
          if (thr >= 2)
            coef[i] = b ? coef[i] + half : coef[i] - half;
          else if (b)
            coef[i]++;
        }
      pos += j;
    }
    if (pos >= avail)
      break;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// Model of the LIS-phase decoder for chunks whose lists mix set shapes (kernel k_lis_mx, and before it k_lis_mixed, in
// speck_dec.hip).  The code of a set depends on its extents only (spk::ShapeCls): per window of W
// stream bits the kernel builds, for every class that has a table slot and EVERY bit position, the
// length of a split of that class starting there (speculative, parallel); ONE thread then walks
// the list -- runs of '0' entries counted off the stream, a significant entry of a class with a
// table hopped over with one look-up (its class read from the entry), any other set walked into
// child by child with the tables of its children -- and every set that was hopped over is expanded
// by an independent thread.  Births are ranked by stream position as in the regular model.
// ------------------------------------------------------------------------------------------
static int model_decode_mixed_impl(const uint8_t* stream, size_t len, const size_t dims[3],
                                   uint64_t* coef, uint64_t* sign, int window, int hmax, bool twoD)
{
  HostTree ht = build_tree(dims[0], dims[1], dims[2], twoD);
  build_classes(ht, hmax, 4, hmax >= 2 ? 0.0 : 0.02);
  if (ht.cls.empty())
    return -2;
  const Tree t = ht.view();
  const std::vector<ShapeCls>& cls = ht.cls;
  uint64_t W = window > 0 ? (uint64_t)window : 256;
  if (W < (uint64_t)ht.slotMaxT + 2)
    W = (uint64_t)ht.slotMaxT + 2;
  const size_t N = t.nvals;
  const int nbp = stream[0];
  uint64_t total_bits;
  memcpy(&total_bits, stream + 1, 8);
  uint64_t avail = (uint64_t)(len - 9) * 8;
  if (avail > total_bits)
    avail = total_bits;
  BitSrc in{stream + 9, avail};
  std::vector<int8_t> born(N, -1), sigp(N, -1);
  memset(coef, 0, N * sizeof(uint64_t));
  memset(sign, 0xff, ((N + 63) / 64) * 8);
  auto set_sign = [&](uint32_t i, int b) {
    if (b)
      sign[i >> 6] |= uint64_t(1) << (i & 63);
    else
      sign[i >> 6] &= ~(uint64_t(1) << (i & 63));
  };
  std::vector<std::vector<uint64_t>> lis(ht.initLIS), next(t.nlevels);
  uint64_t pos = 0;
  uint32_t iPart = ht.iLevels;   // 2D: part_level of the type-I set that is left (0: none)
  for (int p = nbp - 1; p >= 0; p--) {
    const uint64_t thr = uint64_t(1) << p;
    const uint64_t init = thr + thr - thr / 2 - 1;
    {  // ---- D1 (as in model_speck3d_decode)
      std::vector<uint32_t> cand;
      for (size_t i = 0; i < N; i++)
        if (born[i] > p && sigp[i] < 0)
          cand.push_back((uint32_t)i);
      size_t j = 0;
      uint64_t k = 0, ones = 0;
      for (; j < cand.size(); k++) {
        const int b = in.get(pos + k);
        if ((ones & 1) == 0) {
          if (b) {
            sigp[cand[j]] = (int8_t)p;
            coef[cand[j]] = init;
            set_sign(cand[j], in.get(pos + k + 1));
          }
          j++;
        }
        ones = b ? ones + 1 : 0;
      }
      if (ones & 1)
        k++;
      pos += k;
    }
    // ---- D2
    struct BornRec {
      uint32_t lev;
      uint64_t pos;
      uint64_t packed;
    };
    std::vector<BornRec> bornv;
    for (uint32_t l = 0; l < t.nlevels; l++)
      next[l].clear();
    auto pixel_event = [&](uint32_t ridx, bool sig, uint64_t signpos) {
      born[ridx] = (int8_t)p;
      if (sig) {
        sigp[ridx] = (int8_t)p;
        coef[ridx] = init;
        set_sign(ridx, in.get(signpos));
      }
    };
    struct Ctx {
      Node parent;
      uint32_t pc;
      KidBox kb;
      uint32_t next;
      bool found;
    };
    struct Item {
      Node nd;
      uint32_t c;
      uint64_t y;   // first bit of the split, window-relative
    };
    // one list (a level's, or a single subband the type-I set releases): sigv[k] = entry k was significant
    auto run_list = [&](const std::vector<uint64_t>& curList, std::vector<char>& sigv) {
      const size_t n = curList.size();
      sigv.assign(n, 0);
      std::vector<Ctx> ctx;
      size_t e = 0, rem = n;
      bool listDone = false;
      auto push_ctx = [&](const Node& nd, uint32_t c) {
        Ctx cx;
        cx.parent = nd;
        cx.pc = c;
        kid_box(t, nd, cx.kb);
        cx.next = 0;
        cx.found = false;
        if (cx.kb.nk != cls[c].nk)
          abort();
        ctx.push_back(cx);
      };
      while (!listDone) {
        const uint64_t a = pos;
        // ---- tables of the window (parallel in the kernel: one thread per position and slot)
        std::vector<std::vector<uint32_t>> T(12, std::vector<uint32_t>(W + 2, T_INF));
        for (uint32_t ci = 0; ci < cls.size(); ci++) {   // (classes are numbered children first)
          if (cls[ci].slot == 0xff)
            continue;
          const uint32_t s = cls[ci].slot;
          const ShapeCls& C = cls[ci];
          for (uint64_t x = 0; x <= W; x++) {
            uint64_t y = x;
            bool found = false, ok = true;
            for (uint32_t k = 0; k < C.nk && ok; k++) {
              const bool coded = found || (k + 1 != C.nk);
              int b = 1;
              if (coded)
                b = in.get(a + y++);
              if (!b)
                continue;
              found = true;
              if (C.kid[k] == kClsPixel)
                y++;
              else {
                const uint32_t tl = y <= W ? T[cls[C.kid[k]].slot][y] : T_INF;
                if (tl == T_INF)
                  ok = false;
                else
                  y += tl;
              }
            }
            if (ok && y <= W)
              T[s][x] = (uint32_t)(y - x);
          }
        }
        // ---- one thread: the walk
        std::vector<Item> queue;
        uint64_t r = 0;
        while (true) {
          if (!ctx.empty()) {
            Ctx& cx = ctx.back();
            const ShapeCls& C = cls[cx.pc];
            if (cx.next == C.nk) {
              ctx.pop_back();
              continue;
            }
            const uint32_t k = cx.next, kc = C.kid[k];
            const bool coded = cx.found || (k + 1 != C.nk);
            uint64_t x = r;
            int b = 1;
            if (coded) {
              if (x >= W)
                break;
              b = in.get(a + x);
              x++;
            }
            if (kc == kClsPixel) {
              if (b && x >= W)
                break;
              pixel_event(kid_pixel_raster(t, cx.parent, cx.kb, k), b != 0, a + x);
              if (b) {
                x++;
                cx.found = true;
              }
              cx.next++;
              r = x;
              continue;
            }
            if (!b) {
              bornv.push_back({cx.kb.kidlev, a + x - 1, kid_packed(cx.kb, k)});
              cx.next++;
              r = x;
              continue;
            }
            const Node kid = unpack_node(kid_packed(cx.kb, k));
            if (cls[kc].slot != 0xff) {
              const uint32_t tl = x <= W ? T[cls[kc].slot][x] : T_INF;
              if (tl == T_INF) {
                if (r == 0)
                  abort();   // (the window is longer than any split that has a table)
                break;
              }
              queue.push_back({kid, kc, x});
              cx.found = true;
              cx.next++;
              r = x + tl;
            }
            else {
              cx.found = true;
              cx.next++;
              r = x;
              push_ctx(kid, kc);
            }
            continue;
          }
          // the list itself
          if (rem == 0) {
            listDone = true;
            break;
          }
          if (r >= W)
            break;
          if (!in.get(a + r)) {
            r++;
            e++;
            rem--;
            continue;
          }
          const Node nd = unpack_node(curList[e]);
          const uint32_t c = node_cls(t, nd);
          if (c == kClsPixel)
            abort();
          const uint64_t x = r + 1;
          if (cls[c].slot != 0xff) {
            const uint32_t tl = x <= W ? T[cls[c].slot][x] : T_INF;
            if (tl == T_INF) {
              if (r == 0)
                abort();
              break;
            }
            queue.push_back({nd, c, x});
            r = x + tl;
          }
          else {
            r = x;
            push_ctx(nd, c);
          }
          sigv[e] = 1;
          e++;
          rem--;
        }
        pos = a + r;
        // ---- expansion of what was hopped over (parallel, breadth first in the kernel)
        while (!queue.empty()) {
          std::vector<Item> nq;
          for (const Item& w : queue) {
            const ShapeCls& C = cls[w.c];
            KidBox kb;
            kid_box(t, w.nd, kb);
            if (kb.nk != C.nk)
              abort();
            uint64_t y = w.y;
            bool found = false;
            for (uint32_t k = 0; k < C.nk; k++) {
              const bool coded = found || (k + 1 != C.nk);
              int b = 1;
              if (coded)
                b = in.get(a + y++);
              if (C.kid[k] == kClsPixel) {
                pixel_event(kid_pixel_raster(t, w.nd, kb, k), b != 0, a + y);
                if (b) {
                  y++;
                  found = true;
                }
              }
              else if (b) {
                found = true;
                nq.push_back({unpack_node(kid_packed(kb, k)), C.kid[k], y});
                y += T[cls[C.kid[k]].slot][y];
              }
              else
                bornv.push_back({kb.kidlev, a + y - 1, kid_packed(kb, k)});
            }
          }
          queue.swap(nq);
        }
      }
    };
    for (uint32_t l = t.nlevels; l-- > 0;) {
      if (lis[l].empty())
        continue;
      std::vector<char> sigv;
      run_list(lis[l], sigv);
      for (size_t k = 0; k < lis[l].size(); k++)
        if (!sigv[k])
          next[l].push_back(lis[l][k]);
    }
    // 2D: the type-I set, tested at the end of every sorting pass (SPECK2D_INT.cpp:44-98): when it
    // is significant the three subbands of its level are tested (always with a bit) and join the
    // lists or split at once, then the rest of it is tested (implied when none of the three was)
    if (twoD) {
      bool need = true;
      while (iPart > 0) {
        if (need && !in.get(pos++))
          break;
        int counter = 0;
        for (int j = 0; j < 3; j++) {
          const uint64_t root = ht.iRoots[(size_t)(ht.iLevels - iPart) * 3 + j];
          if (root == HostTree::kNoRoot)
            continue;
          const uint64_t at = pos;   // its test bit
          std::vector<uint64_t> one(1, root);
          std::vector<char> sv;
          run_list(one, sv);
          if (sv[0])
            counter++;
          else
            bornv.push_back({iPart, at, root});
        }
        iPart--;
        need = counter != 0;
      }
    }
    std::stable_sort(bornv.begin(), bornv.end(), [](const BornRec& x, const BornRec& y) {
      return x.lev != y.lev ? x.lev < y.lev : x.pos < y.pos;
    });
    for (const BornRec& b : bornv)
      next[b.lev].push_back(b.packed);
    lis.swap(next);
    if (pos >= avail)
      break;
    {  // ---- D3
      const uint64_t half = thr / 2;
      uint64_t j = 0;
      for (size_t i = 0; i < N && pos + j < avail; i++)
        if (sigp[i] > p) {
          const int b = in.get(pos + j);
          j++;
          if (thr >= 2)
            coef[i] = b ? coef[i] + half : coef[i] - half;
          else if (b)
            coef[i]++;
        }
      pos += j;
    }
    if (pos >= avail)
      break;
  }
  return 0;
}

int model_speck3d_decode_mixed(const uint8_t* stream, size_t len, const size_t dims[3],
                               uint64_t* coef, uint64_t* sign, int window, int hmax)
{
  return model_decode_mixed_impl(stream, len, dims, coef, sign, window, hmax, false);
}

// the 2D coder's streams (SPECK2D_INT) with the same machinery: dims[2] = 1
int model_speck2d_decode_mixed(const uint8_t* stream, size_t len, const size_t dims[3],
                               uint64_t* coef, uint64_t* sign, int window, int hmax)
{
  return model_decode_mixed_impl(stream, len, dims, coef, sign, window, hmax, true);
}

// every set node against its shape class (tests/test_speck_model.py::test_shape_classes_are_consistent)
static int check_classes_impl(const size_t dims[3], bool twoD)
{
  HostTree ht = build_tree(dims[0], dims[1], dims[2], twoD);
  if (ht.cls.empty())
    return -2;
  const Tree t = ht.view();
  for (uint32_t gi = 0; gi < ht.grids.size(); gi++) {
    const Grid& g = ht.grids[gi];
    Node n;
    n.grid = (uint16_t)gi;
    for (uint32_t z = 0; z < (1u << g.e[2]); z++)
      for (uint32_t y = 0; y < (1u << g.e[1]); y++)
        for (uint32_t x = 0; x < (1u << g.e[0]); x++) {
          n.i[0] = (uint16_t)x;
          n.i[1] = (uint16_t)y;
          n.i[2] = (uint16_t)z;
          const NodeGeom q = node_geom(t, n);
          const bool isSet = q.count > 1 || (g.depth == 0 && q.count == 1);
          if (!isSet)
            continue;
          const uint32_t ci = node_cls(t, n);
          if (ci == kClsPixel || ci >= ht.cls.size())
            return 1;
          const ShapeCls& c = ht.cls[ci];
          Kids k;
          node_kids(t, n, k);
          KidBox kb;
          kid_box(t, n, kb);
          if ((int)c.nk != k.n || kb.nk != c.nk)
            return 2;
          for (int j = 0; j < k.n; j++) {
            uint32_t idx[3];
            kid_index(kb, (uint32_t)j, idx);
            if (idx[0] != k.idx[j][0] || idx[1] != k.idx[j][1] || idx[2] != k.idx[j][2])
              return 3;
            if (k.count[j] == 1) {
              if (c.kid[j] != kClsPixel)
                return 4;
              if (kid_pixel_raster(t, n, kb, (uint32_t)j) != kid_raster(t, n, k, j))
                return 5;
            }
            else {
              const Node kid = kid_node(k, j);
              if (node_cls(t, kid) != c.kid[j] || pack_node(kid) != kid_packed(kb, (uint32_t)j))
                return 6;
              if (node_level(t, kid) != kb.kidlev)
                return 7;
              if (c.slot != 0xff && ht.cls[c.kid[j]].slot == 0xff)
                return 8;
            }
          }
          if (c.h == 0 && (c.nk == 2 || c.nk == 4 || c.nk == 8) && c.slot != (c.nk == 2 ? 1 : c.nk == 4 ? 2 : 3))
            return 9;
          if (c.slot != 0xff && (c.slot >> 2) != c.h)
            return 10;
        }
  }
  return 0;
}


// ------------------------------------------------------------------------------------------
// SPECK1D (the coder of the outlier list, src/SPECK1D_INT*.cpp) in the formulations of
// sperr_amd/csrc/outlier.hip: the ENCODER gives every outlier of a significant run one path of its
// own, found from its position and its two neighbours (expand_enc); the DECODER parses a whole path
// per step, the run lengths along it in closed form from the bits (expand_dec).  Arrays of at least
// four values (shorter ones keep the serial walk in the kernel).
// ------------------------------------------------------------------------------------------
namespace {
struct Bits1D {
  std::vector<uint64_t> w;
  uint64_t n = 0;
  void put(int b)
  {
    if ((n >> 6) >= w.size())
      w.resize((n >> 6) + 2, 0);
    if (b)
      w[n >> 6] |= 1ull << (n & 63);
    n++;
  }
  void zeros(uint64_t k)
  {
    n += k;
    if ((n >> 6) >= w.size())
      w.resize((n >> 6) + 2, 0);
  }
};
struct Run1D {
  uint32_t s, l;
};
}  // namespace

int model_speck1d_encode(const uint64_t* coef, const uint64_t* signmask, size_t n, uint8_t** stream,
                         size_t* stream_len)
{
  if (n < 4)
    return -1;
  std::vector<uint32_t> pos;
  std::vector<int> msb;
  for (size_t i = 0; i < n; i++)
    if (coef[i]) {
      pos.push_back((uint32_t)i);
      msb.push_back(msb_of(coef[i]));
    }
  int nbp = 0;
  for (int m : msb)
    nbp = std::max(nbp, m + 1);
  const uint32_t nlists = (uint32_t)num_of_partitions(n) + 1;
  std::vector<std::vector<Run1D>> lists(nlists + 2);
  lists[1].push_back({0u, (uint32_t)(n - n / 2)});
  lists[1].push_back({(uint32_t)(n - n / 2), (uint32_t)(n / 2)});
  std::vector<char> lip(n, 0);
  Bits1D out;
  auto sign_of = [&](uint32_t x) { return (int)((signmask[x >> 6] >> (x & 63)) & 1); };
  for (int p = nbp - 1; p >= 0; p--) {
    std::vector<uint32_t> ge;   // positions of the outliers at or above the threshold, ascending
    for (size_t k = 0; k < pos.size(); k++)
      if (msb[k] >= p)
        ge.push_back(pos[k]);
    // LIP pass
    for (size_t x = 0; x < n; x++)
      if (lip[x]) {
        const bool sig = coef[x] && msb_of(coef[x]) == p;
        out.put(sig);
        if (sig) {
          out.put(sign_of((uint32_t)x));
          lip[x] = 0;
        }
      }
    // LIS pass, smallest sets first
    for (uint32_t lev = nlists; lev-- > 0;) {
      std::vector<Run1D> cur;
      cur.swap(lists[lev]);
      for (const Run1D& r : cur) {
        const size_t a = std::lower_bound(ge.begin(), ge.end(), r.s) - ge.begin();
        const size_t e = std::lower_bound(ge.begin(), ge.end(), r.s + r.l) - ge.begin();
        if (e == a) {
          out.put(0);
          lists[lev].push_back(r);   // (kept entries stay in front of what is born into this level later)
          continue;
        }
        out.put(1);
        if (r.l < 2)
          return -2;
        // ---- expand_enc: one path per outlier of the run
        for (size_t k = a; k < e; k++) {
          const uint32_t x = ge[k];
          const bool hasPrev = k > a, hasNext = k + 1 < e;
          const uint32_t prev = hasPrev ? ge[k - 1] : 0, next = hasNext ? ge[k + 1] : 0;
          if (hasPrev)
            out.put(1);   // the closing '1' of the parked half this outlier lies in
          uint32_t s0 = r.s, l0 = r.l, nz = 0;
          for (uint32_t d = 0; l0 > 1; d++) {
            const uint32_t lvl = lev + d + 1, h0 = l0 - l0 / 2, r0 = l0 / 2;
            const bool left = x < s0 + h0;
            const bool owned = !(hasPrev && prev >= s0);
            const bool nextIn = hasNext && next < s0 + l0;
            if (owned)
              out.put(left);
            const bool closes = left && !nextIn;
            nz += closes;
            if ((owned && !left) || closes) {
              const uint32_t bs = left ? s0 + h0 : s0, bl = left ? r0 : h0;
              if (bl == 1)
                lip[bs] = 1;
              else {
                if (lvl >= lists.size())
                  return -3;
                lists[lvl].push_back({bs, bl});
              }
            }
            if (left)
              l0 = h0;
            else {
              s0 += h0;
              l0 = r0;
            }
          }
          out.put(sign_of(x));
          out.zeros(nz);
        }
      }
    }
    // refinement pass: the values found on earlier planes, in position order
    for (size_t k = 0; k < pos.size(); k++)
      if (msb[k] > p)
        out.put((int)((coef[pos[k]] >> p) & 1));
  }
  const uint64_t total = nbp ? out.n : 0;
  const size_t nbytes = (size_t)((total + 7) / 8);
  out.w.resize(nbytes / 8 + 2, 0);
  uint8_t* o = (uint8_t*)calloc(9 + nbytes + 8, 1);
  o[0] = (uint8_t)nbp;
  memcpy(o + 1, &total, 8);
  memcpy(o + 9, out.w.data(), nbytes);
  *stream = o;
  *stream_len = 9 + nbytes;
  return 0;
}

int model_speck1d_decode(const uint8_t* stream, size_t len, size_t n, uint64_t* coef, uint64_t* signmask)
{
  if (n < 4 || len < 9)
    return -1;
  const int nbp = stream[0];
  uint64_t total;
  memcpy(&total, stream + 1, 8);
  std::vector<uint64_t> w((len - 9) / 8 + 3, 0);
  memcpy(w.data(), stream + 9, len - 9);
  uint64_t rpos = 0;
  auto window = [&]() -> uint64_t {
    const uint64_t wi = rpos >> 6;
    const uint32_t sh = (uint32_t)(rpos & 63);
    const uint64_t w0 = wi < w.size() ? w[wi] : 0, w1 = wi + 1 < w.size() ? w[wi + 1] : 0;
    return sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
  };
  auto get = [&]() -> int {
    const int b = (int)(window() & 1);
    rpos++;
    return b;
  };
  const uint32_t nlists = (uint32_t)num_of_partitions(n) + 1;
  std::vector<std::vector<Run1D>> lists(nlists + 2);
  lists[1].push_back({0u, (uint32_t)(n - n / 2)});
  lists[1].push_back({(uint32_t)(n - n / 2), (uint32_t)(n / 2)});
  std::vector<char> lip(n, 0), lsp(n, 0);
  for (size_t i = 0; i < n; i++)
    coef[i] = 0;
  for (size_t i = 0; i < (n + 63) / 64; i++)
    signmask[i] = 0;
  std::vector<uint32_t> foundNow;
  auto found = [&](uint32_t x, int p, int sg) {
    coef[x] = 1ull << p;
    if (sg)
      signmask[x >> 6] |= 1ull << (x & 63);
    foundNow.push_back(x);
  };
  auto born = [&](uint32_t lvl, uint32_t bs, uint32_t bl) -> int {
    if (bl == 1)
      lip[bs] = 1;
    else {
      if (lvl >= lists.size())
        return -3;
      lists[lvl].push_back({bs, bl});
    }
    return 0;
  };
  for (int p = nbp - 1; p >= 0; p--) {
    foundNow.clear();
    for (size_t x = 0; x < n; x++)
      if (lip[x] && get()) {
        found((uint32_t)x, p, get());
        lip[x] = 0;
      }
    for (uint32_t lev = nlists; lev-- > 0;) {
      std::vector<Run1D> cur;
      cur.swap(lists[lev]);
      for (const Run1D& r : cur) {
        if (!get()) {
          lists[lev].push_back(r);
          continue;
        }
        if (r.l < 2)
          return -2;
        // ---- expand_dec: a whole path per step; parked right halves by level
        uint32_t ns = r.s, nl = r.l, nlev = lev;
        uint64_t parked = 0;
        uint32_t pS[64] = {0}, pL[64] = {0};
        for (;;) {
          const uint64_t peek = window();
          // run length at step t in closed form: the interval at depth t that the complemented bits,
          // reversed, index
          uint32_t T = 64;
          uint32_t lt[33], st[33];
          uint32_t acc = ns;
          for (uint32_t t = 0; t < 32; t++) {
            const uint32_t mk = (1u << t) - 1u;
            lt[t] = (nl >> t) + (((~(uint32_t)peek & mk) < (nl & mk)) ? 1u : 0u);
            const uint32_t bt = (uint32_t)(peek >> t) & 1u, h0 = lt[t] - lt[t] / 2, r0 = lt[t] / 2;
            st[t] = acc;
            if (lt[t] > 1 && (bt ? h0 : r0) == 1u) {
              T = t;
              break;
            }
            acc += bt ? 0u : h0;
          }
          if (T == 64)
            return -4;
          for (uint32_t t = 0; t <= T; t++) {
            const uint32_t bt = (uint32_t)(peek >> t) & 1u, h0 = lt[t] - lt[t] / 2, r0 = lt[t] / 2;
            const uint32_t lvl = nlev + 1 + t;
            if (!bt) {
              if (born(lvl, st[t], h0))
                return -3;
            }
            else {
              pS[lvl] = st[t] + h0;
              pL[lvl] = r0;
              parked |= 1ull << lvl;
            }
          }
          {
            const uint32_t bt = (uint32_t)(peek >> T) & 1u, h0 = lt[T] - lt[T] / 2;
            found(bt ? st[T] : st[T] + h0, p, (int)((peek >> (T + 1)) & 1));
          }
          rpos += T + 2;
          bool nextPath = false;
          while (parked) {
            const uint64_t cw = window();
            const uint32_t cnt = (uint32_t)__builtin_popcountll(parked);
            const uint32_t z = std::min<uint32_t>(cw ? (uint32_t)__builtin_ctzll(cw) : 64u, cnt);
            for (uint32_t k = 0; k < z; k++) {   // the z innermost parked halves are born insignificant
              const uint32_t top = 63u - (uint32_t)__builtin_clzll(parked);
              parked &= ~(1ull << top);
              if (born(top, pS[top], pL[top]))
                return -3;
            }
            rpos += z;
            if (z == cnt)
              break;
            rpos++;
            const uint32_t top = 63u - (uint32_t)__builtin_clzll(parked);
            parked &= ~(1ull << top);
            if (pL[top] == 1) {
              found(pS[top], p, get());
              continue;
            }
            ns = pS[top];
            nl = pL[top];
            nlev = top;
            nextPath = true;
            break;
          }
          if (!nextPath)
            break;
        }
      }
    }
    // refinement: the values found on earlier planes, in position order
    for (size_t x = 0; x < n; x++)
      if (lsp[x] && get())
        coef[x] |= 1ull << p;
    for (uint32_t x : foundNow)
      lsp[x] = 1;
  }
  (void)total;
  return 0;
}

// The 1D decoder as k_speck1d<false> runs it since round 3: the serial part (the CHAIN) only finds
// out where every path starts and ends -- a handful of integer operations on the stream window, the
// set being descended known by its depth below the list entry and the bits of the way to it
// (R: bit j = "went right at depth j"), the parked right halves by a mask of depths -- and leaves
// one record per path; what the path means for the lists, the LIP and the values found is worked
// out for `batch` records at a time, one depth per round (the kernel: lane = record), the halves
// born at a depth joining that level's list in record order, which is stream order.
int model_speck1d_decode_batched(const uint8_t* stream, size_t len, size_t n, uint64_t* coef,
                                 uint64_t* signmask, uint32_t batch)
{
  if (n < 4 || len < 9 || batch == 0 || batch > 64)
    return -1;
  const int nbp = stream[0];
  std::vector<uint64_t> w((len - 9) / 8 + 3, 0);
  memcpy(w.data(), stream + 9, len - 9);
  uint64_t rpos = 0;
  auto window = [&]() -> uint64_t {
    const uint64_t wi = rpos >> 6;
    const uint32_t sh = (uint32_t)(rpos & 63);
    const uint64_t w0 = wi < w.size() ? w[wi] : 0, w1 = wi + 1 < w.size() ? w[wi + 1] : 0;
    return sh ? (w0 >> sh) | (w1 << (64 - sh)) : w0;
  };
  auto get = [&]() -> int {
    const int b = (int)(window() & 1);
    rpos++;
    return b;
  };
  const uint32_t nlists = (uint32_t)num_of_partitions(n) + 1;
  std::vector<std::vector<Run1D>> lists(nlists + 2);
  lists[1].push_back({0u, (uint32_t)(n - n / 2)});
  lists[1].push_back({(uint32_t)(n - n / 2), (uint32_t)(n / 2)});
  std::vector<char> lip(n, 0), lsp(n, 0);
  for (size_t i = 0; i < n; i++)
    coef[i] = 0;
  for (size_t i = 0; i < (n + 63) / 64; i++)
    signmask[i] = 0;
  std::vector<uint32_t> foundNow;
  auto found = [&](uint32_t x, int p, int sg) {
    coef[x] = 1ull << p;
    if (sg)
      signmask[x >> 6] |= 1ull << (x & 63);
    foundNow.push_back(x);
  };
  struct Rec {
    uint32_t es, el, R, u, tEnd, sg, closed, raw;   // raw: the path's own bits with its sign on top (what chain_p2's records hold)
  };
  std::vector<Rec> recs;
  int rc = 0;
  uint32_t rCarry = 0;   // R of the last record of the batch before
  auto low = [](uint32_t k) -> uint32_t { return k >= 32 ? ~0u : (1u << k) - 1u; };
  for (int p = nbp - 1; p >= 0 && rc == 0; p--) {
    foundNow.clear();
    for (size_t x = 0; x < n; x++)
      if (lip[x] && get()) {
        found((uint32_t)x, p, get());
        lip[x] = 0;
      }
    for (uint32_t lev = nlists; lev-- > 0 && rc == 0;) {
      auto flush = [&]() {
        // k_speck1d's flush_paths works R out of the raw records of runs of 2^g values (sperr_amd/csrc/outlier.hip,
        // chain_p2): a record is "keep R below depth u - 1, set bit u - 1, my bits from u on"; the (keep, value) pairs
        // compose, so log2(batch) steps over the lanes give every R.  Checked here against the R the chain carried along.
        {
          const size_t nr = recs.size();
          std::vector<uint32_t> keep(nr), val(nr);
          for (size_t k = 0; k < nr; k++) {
            const Rec& r = recs[k];
            if ((r.el & (r.el - 1u)) == 0u) {
              const uint32_t ns = r.tEnd - r.u;
              keep[k] = r.u ? low(r.u - 1u) : 0u;
              val[k] = (r.u ? 1u << (r.u - 1u) : 0u) | ((~r.raw & low(ns)) << r.u);
              if (((r.raw >> ns) & 1u) != r.sg)
                rc = -8;
            }
            else {
              keep[k] = 0;
              val[k] = r.R;
            }
          }
          for (size_t off = 1; off < nr; off <<= 1) {
            std::vector<uint32_t> k2 = keep, v2 = val;
            for (size_t k = off; k < nr; k++) {
              v2[k] = (val[k - off] & keep[k]) | val[k];
              k2[k] = keep[k] & keep[k - off];
            }
            keep.swap(k2);
            val.swap(v2);
          }
          for (size_t k = 0; k < nr; k++)
            if (((rCarry & keep[k]) | val[k]) != recs[k].R)
              rc = -7;
          if (nr)
            rCarry = recs[nr - 1].R;
        }
        // one depth per round, all records of the batch side by side
        std::vector<uint32_t> s(recs.size()), l(recs.size());
        for (size_t k = 0; k < recs.size(); k++) {
          s[k] = recs[k].es;
          l[k] = recs[k].el;
        }
        for (uint32_t j = 0;; j++) {
          bool any = false;
          for (size_t k = 0; k < recs.size(); k++) {
            const Rec& r = recs[k];
            if (j >= r.tEnd)
              continue;
            any = true;
            const uint32_t h0 = l[k] - l[k] / 2, r0 = l[k] / 2;
            const bool right = (r.R >> j) & 1u;
            const bool born = right ? j >= r.u : ((r.closed >> (j + 1)) & 1u) != 0;
            const uint32_t bs = right ? s[k] : s[k] + h0, bl = right ? h0 : r0;
            if (born) {
              if (bl == 1)
                lip[bs] = 1;
              else if (lev + j + 1 >= lists.size())
                rc = -3;
              else
                lists[lev + j + 1].push_back({bs, bl});
            }
            if (right) {
              s[k] += h0;
              l[k] = r0;
            }
            else
              l[k] = h0;
          }
          if (!any)
            break;
        }
        for (size_t k = 0; k < recs.size(); k++) {
          if (l[k] != 1)
            rc = -5;
          found(s[k], p, (int)recs[k].sg);
        }
        recs.clear();
      };
      std::vector<Run1D> cur;
      cur.swap(lists[lev]);
      for (const Run1D& r : cur) {
        if (!get()) {
          lists[lev].push_back(r);
          continue;
        }
        if (r.l < 2)
          return -2;
        // ---- the chain: integer work on the window only
        uint32_t R = 0, m = 0, u = 0, lo = r.l;
        for (;;) {
          const uint64_t peek = window();
          uint32_t nsteps = 0;
          if (lo > 1) {
            const uint32_t e = 31u - (uint32_t)__builtin_clz(lo), t0 = e - 1u, mk = low(t0);
            const uint32_t lt = (lo >> t0) + (((~(uint32_t)peek & mk) < (lo & mk)) ? 1u : 0u);
            const uint32_t bt = (uint32_t)(peek >> t0) & 1u;
            nsteps = (lt == 2u || (lt == 3u && bt == 0u)) ? t0 + 1u : t0 + 2u;
          }
          if (u + nsteps > 31u)
            return -4;
          const uint32_t pm = low(nsteps);
          R |= (~(uint32_t)peek & pm) << u;
          m |= ((uint32_t)peek & pm) << (u + 1u);
          const uint32_t sg = (uint32_t)(peek >> nsteps) & 1u;
          const uint32_t cnt = (uint32_t)__builtin_popcount(m);
          const uint64_t cw = peek >> (nsteps + 1u);   // (at least 31 valid bits: cnt <= 31)
          const uint32_t z = std::min<uint32_t>(cw ? (uint32_t)__builtin_ctzll(cw) : 64u, cnt);
          uint32_t closed = 0;
          for (uint32_t k = 0; k < z; k++) {
            const uint32_t top = 31u - (uint32_t)__builtin_clz(m);
            m &= ~(1u << top);
            closed |= 1u << top;
          }
          recs.push_back({r.s, r.l, R, u, u + nsteps, sg, closed, (uint32_t)peek & low(nsteps + 1u)});
          if (recs.size() == batch)
            flush();
          rpos += nsteps + 1u + z;
          if (z == cnt)
            break;
          rpos++;
          u = 31u - (uint32_t)__builtin_clz(m);
          m &= ~(1u << u);
          R = (R & low(u - 1u)) | (1u << (u - 1u));
          lo = (r.l >> u) + (((R & low(u)) < (r.l & low(u))) ? 1u : 0u);
        }
      }
      if (!recs.empty())
        flush();
    }
    for (size_t x = 0; x < n; x++)
      if (lsp[x] && get())
        coef[x] |= 1ull << p;
    for (uint32_t x : foundNow)
      lsp[x] = 1;
  }
  return rc;
}

int model_check_classes(const size_t dims[3])
{
  return check_classes_impl(dims, false);
}

// The columns of k_lis_mx (spk::build_mx_columns, speck_mx.hip): every column is one class's alone, the leaf parents sit
// in columns 1..3, a class with a column has columns for all its children -- and those columns are LOWER (the rows of a
// region are built column after column) --, columns are numbered by steps above the leaf parents, and every list level
// names two different groups of four.  Returns 0, or which rule broke; *ncols: columns handed out.
int model_check_mx_columns(const size_t dims[3], int twoD, int* ncols)
{
  HostTree ht = build_tree(dims[0], dims[1], dims[2], twoD != 0);
  if (ht.cls.empty())
    return -2;
  if (ht.mxSlot.size() != ht.cls.size() || ht.mxLevelGroup.size() != ht.nlevels)
    return 1;
  int used[16] = {0};
  int n = 0;
  for (size_t i = 0; i < ht.cls.size(); i++) {
    const ShapeCls& c = ht.cls[i];
    const uint8_t sl = ht.mxSlot[i];
    if (c.h == 0 && (c.nk == 2 || c.nk == 4 || c.nk == 8)) {
      if (sl != (c.nk == 2 ? 1 : c.nk == 4 ? 2 : 3))
        return 2;
      continue;
    }
    if (sl == 0xff)
      continue;
    if (sl == 16) {   // (the one column beside the sixteen: up to four steps up, children with columns)
      if (c.h == 0 || c.h > 4 || c.maxT >= 0x7000u)
        return 4;
      for (int k = 0; k < c.nk; k++)
        if (c.kid[k] != kClsPixel && ht.mxSlot[c.kid[k]] == 0xff)
          return 5;
      continue;
    }
    if (sl < 4 || sl >= 16 || used[sl]++)
      return 3;
    n++;
    if (c.h == 0 || c.h > 3 || c.maxT >= 0x7000u)
      return 4;
    for (int k = 0; k < c.nk; k++)
      if (c.kid[k] != kClsPixel && (ht.mxSlot[c.kid[k]] == 0xff || ht.mxSlot[c.kid[k]] >= sl))
        return 5;
  }
  for (int col = 5; col < 16; col++)
    if (used[col] && !used[col - 1])
      return 6;   // (columns are handed out without gaps)
  for (size_t i = 0; i < ht.cls.size(); i++)
    for (size_t j = 0; j < ht.cls.size(); j++)
      if (ht.mxSlot[i] >= 4 && ht.mxSlot[i] < 16 && ht.mxSlot[j] >= 4 && ht.mxSlot[j] < 16 &&
          ht.cls[i].h < ht.cls[j].h && ht.mxSlot[i] > ht.mxSlot[j])
        return 7;
  for (uint32_t l = 0; l < ht.nlevels; l++) {
    const uint32_t ga = ht.mxLevelGroup[l] & 3u, gb = (ht.mxLevelGroup[l] >> 2) & 3u;
    if (ga == gb || (ht.mxLevelGroup[l] >> 7))
      return 8;
  }
  if (ncols)
    *ncols = n;
  return 0;
}

// the same for the 2D coder's forest of a slice (dims = {x, y, 1})
int model_check_classes_2d(const size_t dims[3])
{
  return check_classes_impl(dims, true);
}


// ---- the word-level steps of the decoder's pixel passes (sperr_amd/csrc/bit_words.h) against bit-by-bit loops.
// Returns the number of words that differ (0: all equal); `seed` picks the pseudo-random words, a third of them
// sparse, a third dense, with runs of ones that cross the word's ends.
static uint64_t bw_rng(uint64_t& st)
{
  st ^= st << 13;
  st ^= st >> 7;
  st ^= st << 17;
  return st;
}
int model_check_bit_words(uint64_t seed, int n)
{
  uint64_t st = seed * 0x9e3779b97f4a7c15ull + 1;
  int bad = 0;
  for (int t = 0; t < n; t++) {
    uint64_t m = bw_rng(st), x0 = bw_rng(st), x1 = bw_rng(st);
    if (t % 3 == 1)
      m &= bw_rng(st) & bw_rng(st);
    if (t % 3 == 2)
      m |= bw_rng(st) | bw_rng(st);
    if (t % 97 == 0)
      m = t % 2 ? ~0ull : 0ull;
    // spread: bit i of x goes to the i-th set bit of m
    {
      uint64_t a0 = x0, a1 = x1, r0 = 0, r1 = 0;
      sperrhip::spread_under_mask(m, a0, a1);
      int k = 0;
      for (int b = 0; b < 64; b++)
        if ((m >> b) & 1ull) {
          r0 |= ((x0 >> k) & 1ull) << b;
          r1 |= ((x1 >> k) & 1ull) << b;
          k++;
        }
      bad += (a0 != r0) + (a1 != r1);
    }
    // gather: the bits of x at the set bits of m, packed
    {
      uint64_t a0 = x0, a1 = x1, r0 = 0, r1 = 0;
      sperrhip::gather_under_mask(m, a0, a1);
      int k = 0;
      for (int b = 0; b < 64; b++)
        if ((m >> b) & 1ull) {
          r0 |= ((x0 >> b) & 1ull) << k;
          r1 |= ((x1 >> b) & 1ull) << k;
          k++;
        }
      bad += (a0 != r0) + (a1 != r1);
    }
    // token starts of a word of the LIP scan, both parities
    for (uint32_t par = 0; par < 2; par++) {
      const uint64_t x = t % 5 == 0 ? (x0 | x1 | m) : (t % 7 == 0 ? ~0ull : x0);
      uint32_t p = par;
      const uint64_t got = sperrhip::lip_token_starts(x, p);
      uint64_t want = 0;
      uint32_t ones = par;
      for (int q = 0; q < 64; q++) {
        if ((ones & 1u) == 0)
          want |= 1ull << q;
        ones = ((x >> q) & 1ull) ? ones + 1 : 0;
      }
      bad += (got != want) + (p != (ones & 1u));
    }
  }
  return bad;
}

}  // extern "C"
