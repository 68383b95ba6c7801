/*
 * sperr_oracle.c -- plain-C restatement of the NCAR/SPERR 3D chunk pipeline.
 *
 * TEST INFRASTRUCTURE ONLY (see sperr_oracle.h).  Written from the reference's behaviour; every
 * function cites the reference file:line it follows.  Compile with -ffp-contract=off: every
 * fused multiply-add the canonical reference build performs is spelled out as fma() below
 * (SURVEY.md Appendix B), everything else rounds separately.
 */
#include "sperr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* geometry                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* src/sperr_helper.cpp:36-49 : >=9 samples per level, at most 6 levels */
size_t orc_num_of_xforms(size_t len)
{
  size_t n = 0;
  while (len >= 9) {
    n++;
    len -= len / 2;
  }
  return n < 6 ? n : 6;
}

/* src/sperr_helper.cpp:51-68 */
int orc_can_use_dyadic(const size_t dims[3], size_t* levels)
{
  if (dims[2] < 2 || dims[1] < 2)
    return 0;
  size_t xy = orc_num_of_xforms(dims[0] < dims[1] ? dims[0] : dims[1]);
  size_t z = orc_num_of_xforms(dims[2]);
  if (xy == z || (xy >= 5 && z >= 5)) {
    *levels = xy < z ? xy : z;
    return 1;
  }
  return 0;
}

/* src/sperr_helper.cpp:136-146 */
void orc_approx_detail_len(size_t orig_len, size_t lev, size_t* approx, size_t* detail)
{
  size_t lo = orig_len, hi = 0;
  for (size_t i = 0; i < lev; i++) {
    hi = lo / 2;
    lo -= hi;
  }
  *approx = lo;
  *detail = hi;
}

/* src/sperr_helper.cpp:125-134 */
size_t orc_num_of_partitions(size_t len)
{
  size_t n = 0;
  while (len > 1) {
    n++;
    len -= len / 2;
  }
  return n;
}

/* src/sperr_helper.cpp:542-592 : x fastest; a remainder longer than half a chunk is its own
 * segment, a shorter one is merged into the last segment. */
size_t orc_chunk_volume(const size_t vol[3], const size_t chunk[3], size_t (*out)[6], size_t cap)
{
  size_t nseg[3];
  for (int a = 0; a < 3; a++) {
    nseg[a] = vol[a] / chunk[a];
    if (vol[a] % chunk[a] > chunk[a] / 2)
      nseg[a]++;
    if (nseg[a] == 0)
      nseg[a] = 1;
  }
  size_t total = nseg[0] * nseg[1] * nseg[2];
  if (!out)
    return total;
  size_t k = 0;
  for (size_t z = 0; z < nseg[2]; z++)
    for (size_t y = 0; y < nseg[1]; y++)
      for (size_t x = 0; x < nseg[0]; x++) {
        if (k >= cap)
          return total;
        size_t idx[3] = {x, y, z};
        for (int a = 0; a < 3; a++) {
          size_t beg = idx[a] * chunk[a];
          size_t end = (idx[a] + 1 == nseg[a]) ? vol[a] : beg + chunk[a];
          out[k][2 * a] = beg;
          out[k][2 * a + 1] = end - beg;
        }
        k++;
      }
  return total;
}

/* ------------------------------------------------------------------------------------------ */
/* conditioner                                                                                */
/* ------------------------------------------------------------------------------------------ */

/* src/Conditioner.cpp:137-163 */
static size_t condi_num_strides(size_t len)
{
  const size_t dflt = 2048;
  if (len % dflt == 0)
    return dflt;
  for (size_t n = dflt; n <= 32768; n++)
    if (len % n == 0)
      return n;
  size_t n = dflt;
  while (len % n != 0)
    n--;
  return n;
}

/* src/Conditioner.cpp:119-135 : strictly sequential sums, order is part of the result */
static double condi_mean(const double* buf, size_t len)
{
  const size_t ns = condi_num_strides(len);
  const size_t ssz = len / ns;
  double total = 0.0;
  for (size_t s = 0; s < ns; s++) {
    double acc = 0.0;
    const double* p = buf + s * ssz;
    for (size_t i = 0; i < ssz; i++)
      acc += p[i];
    total += acc / (double)ssz;
  }
  return total / (double)ns;
}

/* src/Conditioner.cpp:10-64 ; flag byte per pack_8_booleans (src/sperr_helper.cpp:262-273):
 * bool[0] -> 0x80 (mean subtracted), bool[7] -> 0x01 (constant field) */
int orc_condition(double* buf, size_t n, uint8_t header[17])
{
  memset(header, 0, 17);
  int constant = 1;
  for (size_t i = 1; i < n; i++)
    if (buf[i] != buf[0]) {
      constant = 0;
      break;
    }
  if (constant) {
    header[0] = 0x81;
    uint64_t nval = n;
    memcpy(header + 1, &nval, 8);
    memcpy(header + 9, &buf[0], 8);
    return 1;
  }
  const double mean = condi_mean(buf, n);
  for (size_t i = 0; i < n; i++)
    buf[i] -= mean;
  header[0] = 0x80;
  memcpy(header + 1, &mean, 8);
  return 0;
}

/* src/Conditioner.cpp:66-96 (constant case fills; caller sizes buf from dims) */
void orc_inverse_condition(double* buf, size_t n, const uint8_t header[17])
{
  if (header[0] & 0x01) {
    double v;
    memcpy(&v, header + 9, 8);
    for (size_t i = 0; i < n; i++)
      buf[i] = v;
    return;
  }
  double mean;
  memcpy(&mean, header + 1, 8);
  for (size_t i = 0; i < n; i++)
    buf[i] += mean;
}

/* ------------------------------------------------------------------------------------------ */
/* CDF 9/7                                                                                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct {
  double alpha, beta, gamma, delta, eps, inv_eps;
} cdf_consts;

/* include/CDF97.h:136-147 */
static cdf_consts cdf_make_consts(void)
{
  const double h[5] = {0.602949018236, 0.266864118443, -0.078223266529, -0.016864118443,
                       0.026748757411};
  const double r0 = h[0] - 2.0 * h[4] * h[1] / h[3];
  const double r1 = h[2] - h[4] - h[4] * h[1] / h[3];
  const double s0 = h[1] - h[3] - h[3] * r0 / r1;
  const double t0 = h[0] - 2.0 * (h[2] - h[4]);
  cdf_consts c;
  c.alpha = h[4] / h[3];
  c.beta = h[3] / r1;
  c.gamma = r1 / s0;
  c.delta = s0 / t0;
  c.eps = sqrt(2.0) * t0;
  c.inv_eps = 1.0 / c.eps;
  (void)r0;
  return c;
}

/* One lifting step on the "odd" half: odd[i] = fma(k, even[i] + even[i+1], odd[i]) with the
 * right neighbour of the last odd sample mirrored (src/CDF97.cpp:606-608,617-619,652-654,
 * 663-665).  For even len the last even sample is its own neighbour, for odd len it is the
 * extra even sample. */
static void lift_odd(double* even, double* odd, size_t even_len, size_t odd_len, double k)
{
  for (size_t i = 0; i + 1 < odd_len; i++)
    odd[i] = fma(k, even[i] + even[i + 1], odd[i]);
  odd[odd_len - 1] = fma(k, even[odd_len - 1] + even[even_len - 1], odd[odd_len - 1]);
}

/* One lifting step on the "even" half (src/CDF97.cpp:611-614,657-660): the first sample sees
 * odd[0] twice (2k * odd[0], the doubling is exact), the last one odd[even_len-2] and
 * odd[odd_len-1] (the same sample twice when len is odd). */
static void lift_even(double* even, const double* odd, size_t even_len, size_t odd_len, double k)
{
  even[0] = fma(k + k, odd[0], even[0]);
  for (size_t i = 1; i + 1 < even_len; i++)
    even[i] = fma(k, odd[i - 1] + odd[i], even[i]);
  even[even_len - 1] = fma(k, odd[even_len - 2] + odd[odd_len - 1], even[even_len - 1]);
}

/* src/CDF97.cpp:598-631 on a line laid out [even | odd] */
static void cdf_analysis(double* sig, size_t len, const cdf_consts* c)
{
  const size_t even_len = len - len / 2, odd_len = len / 2;
  double* even = sig;
  double* odd = sig + even_len;
  lift_odd(even, odd, even_len, odd_len, c->alpha);
  lift_even(even, odd, even_len, odd_len, c->beta);
  lift_odd(even, odd, even_len, odd_len, c->gamma);
  /* even = EPS * (even + DELTA*(..)) : inner fma, outer multiply rounds separately */
  lift_even(even, odd, even_len, odd_len, c->delta);
  for (size_t i = 0; i < even_len; i++)
    even[i] = c->eps * even[i];
  for (size_t i = 0; i < odd_len; i++)
    odd[i] = (-c->inv_eps) * odd[i];
}

/* src/CDF97.cpp:633-666 */
static void cdf_synthesis(double* sig, size_t len, const cdf_consts* c)
{
  const size_t even_len = len - len / 2, odd_len = len / 2;
  double* even = sig;
  double* odd = sig + even_len;
  for (size_t i = 0; i < odd_len; i++)
    odd[i] = (-c->eps) * odd[i];
  /* even = even*INV_EPS - DELTA*(..) : the DELTA product rounds first, then one fma */
  even[0] = fma(even[0], c->inv_eps, -((c->delta + c->delta) * odd[0]));
  for (size_t i = 1; i + 1 < even_len; i++)
    even[i] = fma(even[i], c->inv_eps, -(c->delta * (odd[i - 1] + odd[i])));
  even[even_len - 1] =
      fma(even[even_len - 1], c->inv_eps, -(c->delta * (odd[even_len - 2] + odd[odd_len - 1])));
  lift_odd(even, odd, even_len, odd_len, -c->gamma);
  lift_even(even, odd, even_len, odd_len, -c->beta);
  lift_odd(even, odd, even_len, odd_len, -c->alpha);
}

/* Transform every line along `axis` inside the box [0,region) of a dims-shaped volume.
 * Forward: de-interleave (src/CDF97.cpp:476-519), analyse, store as [approx | detail].
 * Inverse: synthesise, re-interleave (src/CDF97.cpp:521-564). */
static void cdf_lift_axis(double* buf, const size_t dims[3], int axis, const size_t region[3],
                          int forward, const cdf_consts* c, double* tmp)
{
  const size_t stride[3] = {1, dims[0], dims[0] * dims[1]};
  const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
  const size_t len = region[axis], st = stride[axis];
  const size_t even_len = len - len / 2;
  for (size_t j = 0; j < region[a2]; j++)
    for (size_t i = 0; i < region[a1]; i++) {
      double* line = buf + i * stride[a1] + j * stride[a2];
      if (forward) {
        for (size_t k = 0; k < len; k++)
          tmp[(k & 1) ? even_len + k / 2 : k / 2] = line[k * st];
        cdf_analysis(tmp, len, c);
        for (size_t k = 0; k < len; k++)
          line[k * st] = tmp[k];
      }
      else {
        for (size_t k = 0; k < len; k++)
          tmp[k] = line[k * st];
        cdf_synthesis(tmp, len, c);
        for (size_t k = 0; k < len; k++)
          line[k * st] = tmp[(k & 1) ? even_len + k / 2 : k / 2];
      }
    }
}

/* src/CDF97.cpp:132-139,170-225,284-292,387-429 */
void orc_dwt3d(double* buf, const size_t dims[3])
{
  const cdf_consts c = cdf_make_consts();
  size_t maxlen = dims[0] > dims[1] ? dims[0] : dims[1];
  if (dims[2] > maxlen)
    maxlen = dims[2];
  double* tmp = (double*)malloc(maxlen * sizeof(double));
  size_t levels, d;
  if (orc_can_use_dyadic(dims, &levels)) {
    for (size_t lev = 0; lev < levels; lev++) {
      size_t region[3];
      for (int a = 0; a < 3; a++)
        orc_approx_detail_len(dims[a], lev, &region[a], &d);
      cdf_lift_axis(buf, dims, 0, region, 1, &c, tmp); /* X rows, then Y columns, per plane */
      cdf_lift_axis(buf, dims, 1, region, 1, &c, tmp);
      cdf_lift_axis(buf, dims, 2, region, 1, &c, tmp); /* then Z columns */
    }
  }
  else { /* wavelet packet: all Z levels first, then all XY levels on every plane */
    const size_t nz = orc_num_of_xforms(dims[2]);
    const size_t nxy = orc_num_of_xforms(dims[0] < dims[1] ? dims[0] : dims[1]);
    for (size_t lev = 0; lev < nz; lev++) {
      size_t region[3] = {dims[0], dims[1], 0};
      orc_approx_detail_len(dims[2], lev, &region[2], &d);
      cdf_lift_axis(buf, dims, 2, region, 1, &c, tmp);
    }
    for (size_t lev = 0; lev < nxy; lev++) {
      size_t region[3] = {0, 0, dims[2]};
      orc_approx_detail_len(dims[0], lev, &region[0], &d);
      orc_approx_detail_len(dims[1], lev, &region[1], &d);
      cdf_lift_axis(buf, dims, 0, region, 1, &c, tmp);
      cdf_lift_axis(buf, dims, 1, region, 1, &c, tmp);
    }
  }
  free(tmp);
}

/* src/sperr_helper.cpp:70-97 : the coarsened resolutions of a dyadic chunk, coarsest first */
size_t orc_coarsened_resolutions(const size_t dims[3], size_t (*res)[3], size_t cap)
{
  size_t levels;
  if (!orc_can_use_dyadic(dims, &levels))
    return 0;
  for (size_t lev = levels; lev > 0; lev--) {
    const size_t h = levels - lev;
    if (res && h < cap)
      for (int a = 0; a < 3; a++) {
        size_t d;
        orc_approx_detail_len(dims[a], lev, &res[h][a], &d);
      }
  }
  return levels;
}

/* src/sperr_helper.cpp:86-95 : the coarsened resolutions of a slice (dims[2] == 1), coarsest
 * first: one per level of dwt2d */
size_t orc_coarsened_resolutions_2d(const size_t dims[3], size_t (*res)[3], size_t cap)
{
  const size_t levels = orc_num_of_xforms(dims[0] < dims[1] ? dims[0] : dims[1]);
  for (size_t lev = levels; lev > 0; lev--) {
    const size_t h = levels - lev;
    if (res && h < cap) {
      size_t d;
      orc_approx_detail_len(dims[0], lev, &res[h][0], &d);
      orc_approx_detail_len(dims[1], lev, &res[h][1], &d);
      res[h][2] = 1;
    }
  }
  return levels;
}

static void idwt3d_impl(double* buf, const size_t dims[3], double** hier);

/* src/CDF97.cpp:141-148,227-302,366-385,431-474 */
void orc_idwt3d(double* buf, const size_t dims[3])
{
  idwt3d_impl(buf, dims, NULL);
}

/* src/CDF97.cpp:150-168 : before every level of the inverse transform the approximation corner
 * (the volume at that coarser resolution, unscaled) is copied out, coarsest first */
void orc_idwt3d_multi_res(double* buf, const size_t dims[3], double** hier)
{
  idwt3d_impl(buf, dims, hier);
}

static void idwt3d_impl(double* buf, const size_t dims[3], double** hier)
{
  const cdf_consts c = cdf_make_consts();
  size_t maxlen = dims[0] > dims[1] ? dims[0] : dims[1];
  if (dims[2] > maxlen)
    maxlen = dims[2];
  double* tmp = (double*)malloc(maxlen * sizeof(double));
  size_t levels, d;
  if (orc_can_use_dyadic(dims, &levels)) {
    for (size_t lev = levels; lev > 0; lev--) {
      size_t region[3];
      if (hier) { /* m_sub_volume, src/CDF97.cpp:581-593 */
        size_t sub[3];
        for (int a = 0; a < 3; a++)
          orc_approx_detail_len(dims[a], lev, &sub[a], &d);
        double* dst = hier[levels - lev];
        for (size_t z = 0; z < sub[2]; z++)
          for (size_t y = 0; y < sub[1]; y++) {
            memcpy(dst, buf + (z * dims[1] + y) * dims[0], sub[0] * sizeof(double));
            dst += sub[0];
          }
      }
      for (int a = 0; a < 3; a++)
        orc_approx_detail_len(dims[a], lev - 1, &region[a], &d);
      cdf_lift_axis(buf, dims, 2, region, 0, &c, tmp);
      cdf_lift_axis(buf, dims, 1, region, 0, &c, tmp);
      cdf_lift_axis(buf, dims, 0, region, 0, &c, tmp);
    }
  }
  else {
    const size_t nz = orc_num_of_xforms(dims[2]);
    const size_t nxy = orc_num_of_xforms(dims[0] < dims[1] ? dims[0] : dims[1]);
    for (size_t lev = nxy; lev > 0; lev--) {
      size_t region[3] = {0, 0, dims[2]};
      if (hier && dims[2] == 1) { /* idwt2d_multi_res + m_sub_slice, src/CDF97.cpp:114-130,569-579 */
        size_t sx, sy;
        orc_approx_detail_len(dims[0], lev, &sx, &d);
        orc_approx_detail_len(dims[1], lev, &sy, &d);
        double* dst = hier[nxy - lev];
        for (size_t y = 0; y < sy; y++)
          memcpy(dst + y * sx, buf + y * dims[0], sx * sizeof(double));
      }
      orc_approx_detail_len(dims[0], lev - 1, &region[0], &d);
      orc_approx_detail_len(dims[1], lev - 1, &region[1], &d);
      cdf_lift_axis(buf, dims, 1, region, 0, &c, tmp);
      cdf_lift_axis(buf, dims, 0, region, 0, &c, tmp);
    }
    for (size_t lev = nz; lev > 0; lev--) {
      size_t region[3] = {dims[0], dims[1], 0};
      orc_approx_detail_len(dims[2], lev - 1, &region[2], &d);
      cdf_lift_axis(buf, dims, 2, region, 0, &c, tmp);
    }
  }
  free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* quantiser                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* src/SPECK_FLT.cpp:311-371 : llrint is round-half-even in the default rounding mode; the sign
 * bit is (ll >= 0), so zeros carry a set bit. */
int orc_quantize(const double* vals, size_t n, double q, uint64_t* coeffs, uint64_t* signs,
                 int* width)
{
  double maxabs = 0.0;
  for (size_t i = 0; i < n; i++) {
    double a = fabs(vals[i]);
    if (!(a <= maxabs))
      maxabs = a; /* NaN would stick; the reference returns FE_Invalid in that case */
  }
  if (!(maxabs / q < 9.2e18)) /* llrint would raise FE_INVALID (SPECK_FLT.cpp:323-327) */
    return 7;                 /* RTNType::FE_Invalid (include/sperr_helper.h:54-64) */
  const long long maxll = llrint(maxabs / q);
  *width = maxll <= 0xff ? 1 : maxll <= 0xffff ? 2 : maxll <= 0xffffffffLL ? 4 : 8;
  const double inv = 1.0 / q;
  memset(signs, 0, ((n + 63) / 64) * sizeof(uint64_t));
  for (size_t i = 0; i < n; i++) {
    const long long ll = llrint(vals[i] * inv);
    if (ll >= 0)
      signs[i >> 6] |= (uint64_t)1 << (i & 63);
    coeffs[i] = (uint64_t)(ll < 0 ? -ll : ll);
  }
  return 0;
}

/* src/SPECK_FLT.cpp:373-399 : (q * c) * (+-1.0), left to right */
void orc_inv_quantize(const uint64_t* coeffs, const uint64_t* signs, size_t n, double q,
                      double* vals)
{
  for (size_t i = 0; i < n; i++) {
    const double s = ((signs[i >> 6] >> (i & 63)) & 1) ? 1.0 : -1.0;
    vals[i] = q * (double)coeffs[i] * s;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* bit containers                                                                             */
/* ------------------------------------------------------------------------------------------ */

/* src/Bitstream.cpp:74-85,111-173 : LSB-first inside little-endian 64-bit words */
typedef struct {
  uint64_t* w;
  size_t cap_words;
  size_t pos; /* next bit to write / read */
} bitfifo;

static void fifo_put(bitfifo* f, int bit)
{
  const size_t wi = f->pos >> 6;
  if (wi >= f->cap_words) {
    size_t ncap = f->cap_words ? f->cap_words * 2 : 64;
    f->w = (uint64_t*)realloc(f->w, ncap * sizeof(uint64_t));
    memset(f->w + f->cap_words, 0, (ncap - f->cap_words) * sizeof(uint64_t));
    f->cap_words = ncap;
  }
  f->w[wi] |= (uint64_t)(bit & 1) << (f->pos & 63);
  f->pos++;
}

static int fifo_get(bitfifo* f)
{
  const int b = (int)((f->w[f->pos >> 6] >> (f->pos & 63)) & 1);
  f->pos++;
  return b;
}

static int mask_get(const uint64_t* m, size_t i)
{
  return (int)((m[i >> 6] >> (i & 63)) & 1);
}
static void mask_set(uint64_t* m, size_t i)
{
  m[i >> 6] |= (uint64_t)1 << (i & 63);
}
static void mask_clr(uint64_t* m, size_t i)
{
  m[i >> 6] &= ~((uint64_t)1 << (i & 63));
}

/* ------------------------------------------------------------------------------------------ */
/* SPECK3D                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* include/SPECK3D_INT.h:10-22 */
typedef struct {
  uint64_t first; /* position of the set's first sample in the depth-first sample order */
  uint32_t org[3], len[3];
} box3;

typedef struct {
  box3* v;
  size_t n, cap;
} boxlist;

static void list_push(boxlist* l, const box3* b)
{
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 16;
    l->v = (box3*)realloc(l->v, l->cap * sizeof(box3));
  }
  l->v[l->n++] = *b;
}

static size_t box_count(const box3* b)
{
  return (size_t)b->len[0] * b->len[1] * b->len[2];
}

/* Octree split (src/SPECK3D_INT.cpp:214-326): each axis into (len - len/2, len/2); children in
 * x-fastest order; depth-first sample offsets accumulate in that order.  Returns the number of
 * axes that really split (the LIS level increment). */
static int box_split(const box3* s, box3 kid[8])
{
  uint32_t part[3][2], off[3][2];
  int inc = 0;
  for (int a = 0; a < 3; a++) {
    part[a][1] = s->len[a] / 2;
    part[a][0] = s->len[a] - part[a][1];
    off[a][0] = s->org[a];
    off[a][1] = s->org[a] + part[a][0];
    inc += part[a][1] != 0;
  }
  uint64_t first = s->first;
  for (int k = 0; k < 8; k++) {
    const int h[3] = {k & 1, (k >> 1) & 1, (k >> 2) & 1};
    for (int a = 0; a < 3; a++) {
      kid[k].org[a] = off[a][h[a]];
      kid[k].len[a] = part[a][h[a]];
    }
    kid[k].first = first;
    first += box_count(&kid[k]);
  }
  return inc;
}

/* Initialisation-only splits (src/SPECK3D_INT.cpp:328-427) */
static int box_split_xy(const box3* s, box3 kid[4])
{
  box3 tmp = *s, all[8];
  tmp.len[2] = 1; /* split as if one plane thick, then restore the z extent */
  int inc = box_split(&tmp, all);
  for (int k = 0; k < 4; k++) {
    kid[k] = all[k];
    kid[k].org[2] = s->org[2];
    kid[k].len[2] = s->len[2];
  }
  return inc;
}
static int box_split_z(const box3* s, box3 kid[2])
{
  kid[0] = *s;
  kid[1] = *s;
  kid[1].len[2] = s->len[2] / 2;
  kid[0].len[2] = s->len[2] - kid[1].len[2];
  kid[1].org[2] = s->org[2] + kid[0].len[2];
  return kid[1].len[2] != 0;
}

typedef struct {
  size_t dims[3], n;
  int encoding;
  uint64_t* coef; /* encoder: working copy, decoder: output */
  uint64_t* sign; /* encoder: input (const), decoder: output */
  int8_t* msb;    /* encoder only: msb of every coefficient, depth-first order */
  uint64_t *lip, *lsp;
  size_t *fresh, nfresh, capfresh; /* LSP_new */
  boxlist* lis;
  size_t nlis;
  bitfifo bits;
  uint64_t thr;
  int plane; /* msb of thr */
} speck;

static size_t raster_of(const speck* s, const uint32_t p[3])
{
  return ((size_t)p[2] * s->dims[1] + p[1]) * s->dims[0] + p[0];
}

static void fresh_push(speck* s, size_t idx)
{
  if (s->nfresh == s->capfresh) {
    s->capfresh = s->capfresh ? s->capfresh * 2 : 1024;
    s->fresh = (size_t*)realloc(s->fresh, s->capfresh * sizeof(size_t));
  }
  s->fresh[s->nfresh++] = idx;
}

/* src/SPECK3D_INT_ENC.cpp:8-139 : write msb positions in depth-first order */
static void deposit_msb(speck* s, const box3* b)
{
  const size_t cnt = box_count(b);
  if (cnt == 0)
    return;
  if (cnt == 1) {
    const uint64_t v = s->coef[raster_of(s, b->org)];
    s->msb[b->first] = v ? (int8_t)(63 - __builtin_clzll(v)) : -1; /* sperr_helper.cpp:645-658 */
    return;
  }
  box3 kid[8];
  box_split(b, kid);
  for (int k = 0; k < 8; k++)
    deposit_msb(s, &kid[k]);
}

/* src/SPECK3D_INT.cpp:22-97 (+ ENC.cpp:141-159 for the encoder's depth-first numbering) */
static void speck_init_lists(speck* s)
{
  s->nlis = 1 + orc_num_of_partitions(s->dims[0]) + orc_num_of_partitions(s->dims[1]) +
            orc_num_of_partitions(s->dims[2]);
  s->lis = (boxlist*)calloc(s->nlis, sizeof(boxlist));
  box3 big = {0, {0, 0, 0}, {(uint32_t)s->dims[0], (uint32_t)s->dims[1], (uint32_t)s->dims[2]}};
  size_t lev = 0, dy;
  box3 kid[8];
  if (orc_can_use_dyadic(s->dims, &dy)) {
    for (size_t i = 0; i < dy; i++) {
      lev += box_split(&big, kid);
      for (int k = 1; k < 8; k++)
        list_push(&s->lis[lev], &kid[k]);
      big = kid[0];
    }
  }
  else {
    const size_t nxy = orc_num_of_xforms(s->dims[0] < s->dims[1] ? s->dims[0] : s->dims[1]);
    const size_t nz = orc_num_of_xforms(s->dims[2]);
    size_t xf = 0;
    for (; xf < nxy && xf < nz; xf++) {
      lev += box_split(&big, kid);
      for (int k = 1; k < 8; k++)
        list_push(&s->lis[lev], &kid[k]);
      big = kid[0];
    }
    for (; xf < nxy; xf++) {
      lev += box_split_xy(&big, kid);
      for (int k = 1; k < 4; k++)
        list_push(&s->lis[lev], &kid[k]);
      big = kid[0];
    }
    for (; xf < nz; xf++) {
      lev += box_split_z(&big, kid);
      list_push(&s->lis[lev], &kid[1]);
      big = kid[0];
    }
  }
  /* the remaining low-pass box goes to the FRONT of its list (src/SPECK3D_INT.cpp:93) */
  boxlist* l = &s->lis[lev];
  list_push(l, &big);
  memmove(l->v + 1, l->v, (l->n - 1) * sizeof(box3));
  l->v[0] = big;

  if (s->encoding) {
    uint64_t first = 0;
    for (size_t t = s->nlis; t-- > 0;)
      for (size_t i = 0; i < s->lis[t].n; i++) {
        box3* b = &s->lis[t].v[i];
        b->first = first;
        deposit_msb(s, b);
        first += box_count(b);
      }
  }
}

static void speck_test_box(speck* s, size_t lev, size_t idx, int* counter, int coded);

/* pixel with known position (src/SPECK3D_INT_ENC.cpp:182-199, _DEC.cpp:24-37) */
static void speck_test_pixel(speck* s, size_t ridx, uint64_t first, int* counter, int coded)
{
  int sig = 1;
  if (coded) {
    if (s->encoding) {
      sig = s->msb[first] >= s->plane;
      fifo_put(&s->bits, sig);
    }
    else
      sig = fifo_get(&s->bits);
  }
  if (sig) {
    (*counter)++;
    if (s->encoding)
      fifo_put(&s->bits, mask_get(s->sign, ridx));
    else if (fifo_get(&s->bits))
      mask_set(s->sign, ridx);
    else
      mask_clr(s->sign, ridx);
    fresh_push(s, ridx);
    mask_clr(s->lip, ridx);
  }
}

/* src/SPECK3D_INT.cpp:140-212 : the last non-empty child is not coded when none of its
 * siblings turned out significant.  (The reference's 2x2x2 fast path is the same rule.) */
static void speck_split_box(speck* s, size_t lev, size_t idx)
{
  const box3 parent = s->lis[lev].v[idx];
  box3 kid[8];
  const size_t next = lev + (size_t)box_split(&parent, kid);
  int nkid = 0;
  for (int k = 0; k < 8; k++)
    if (box_count(&kid[k]))
      kid[nkid++] = kid[k];
  int found = 0;
  for (int k = 0; k < nkid; k++) {
    const int coded = found != 0 || k + 1 != nkid;
    if (box_count(&kid[k]) == 1) {
      const size_t ridx = raster_of(s, kid[k].org);
      mask_set(s->lip, ridx);
      speck_test_pixel(s, ridx, kid[k].first, &found, coded);
    }
    else {
      list_push(&s->lis[next], &kid[k]);
      speck_test_box(s, next, s->lis[next].n - 1, &found, coded);
    }
  }
}

/* src/SPECK3D_INT_ENC.cpp:161-180, _DEC.cpp:8-22 */
static void speck_test_box(speck* s, size_t lev, size_t idx, int* counter, int coded)
{
  int sig = 1;
  if (coded) {
    if (s->encoding) {
      const box3* b = &s->lis[lev].v[idx];
      const int8_t* m = s->msb + b->first;
      const size_t cnt = box_count(b);
      sig = 0;
      for (size_t i = 0; i < cnt; i++)
        if (m[i] >= s->plane) {
          sig = 1;
          break;
        }
      fifo_put(&s->bits, sig);
    }
    else
      sig = fifo_get(&s->bits);
  }
  if (sig) {
    (*counter)++;
    speck_split_box(s, lev, idx);
    s->lis[lev].v[idx].len[0] = 0; /* make_empty */
  }
}

/* src/SPECK3D_INT.cpp:99-138 */
static void speck_sorting_pass(speck* s)
{
  for (size_t i = 0; i < s->n; i++) { /* LIP scan in raster order */
    if ((i & 63) == 0 && s->lip[i >> 6] == 0) {
      i += 63;
      continue;
    }
    if (!mask_get(s->lip, i))
      continue;
    int sig;
    if (s->encoding) {
      sig = s->coef[i] >= s->thr;
      fifo_put(&s->bits, sig);
    }
    else
      sig = fifo_get(&s->bits);
    if (sig) {
      if (s->encoding)
        fifo_put(&s->bits, mask_get(s->sign, i));
      else if (fifo_get(&s->bits))
        mask_set(s->sign, i);
      else
        mask_clr(s->sign, i);
      fresh_push(s, i);
      mask_clr(s->lip, i);
    }
  }
  for (size_t lev = s->nlis; lev-- > 0;)
    for (size_t i = 0; i < s->lis[lev].n; i++) {
      int dummy = 0;
      speck_test_box(s, lev, i, &dummy, 1);
    }
}

/* src/SPECK3D_INT.cpp:12-20 */
static void speck_clean_lists(speck* s)
{
  for (size_t lev = 0; lev < s->nlis; lev++) {
    boxlist* l = &s->lis[lev];
    size_t k = 0;
    for (size_t i = 0; i < l->n; i++)
      if (box_count(&l->v[i]))
        l->v[k++] = l->v[i];
    l->n = k;
  }
}

static void speck_alloc(speck* s, const size_t dims[3], int encoding)
{
  memset(s, 0, sizeof(*s));
  memcpy(s->dims, dims, sizeof(s->dims));
  s->n = dims[0] * dims[1] * dims[2];
  s->encoding = encoding;
  const size_t words = (s->n + 63) / 64;
  s->lip = (uint64_t*)calloc(words, 8);
  s->lsp = (uint64_t*)calloc(words, 8);
}

static void speck_free(speck* s)
{
  for (size_t i = 0; i < s->nlis; i++)
    free(s->lis[i].v);
  free(s->lis);
  free(s->lip);
  free(s->lsp);
  free(s->fresh);
  free(s->msb);
  free(s->bits.w);
}

/* src/SPECK_INT.cpp:48-58 */
static size_t round_budget(size_t budget)
{
  if (budget == 0)
    return (size_t)-1;
  while (budget % 8)
    budget++;
  return budget;
}

/* src/SPECK_INT.cpp:110-163,284-357 */
int orc_speck3d_encode(const uint64_t* coeffs, const uint64_t* signs, const size_t dims[3],
                       size_t budget_bits, uint8_t** stream, size_t* stream_len)
{
  speck s;
  speck_alloc(&s, dims, 1);
  const size_t budget = round_budget(budget_bits);
  s.coef = (uint64_t*)malloc(s.n * sizeof(uint64_t));
  memcpy(s.coef, coeffs, s.n * sizeof(uint64_t));
  s.sign = (uint64_t*)signs;
  s.msb = (int8_t*)malloc(s.n);
  speck_init_lists(&s);

  uint64_t maxc = 0;
  for (size_t i = 0; i < s.n; i++)
    if (s.coef[i] > maxc)
      maxc = s.coef[i];
  uint8_t nplanes = 0;
  uint64_t total_bits = 0;
  if (maxc) {
    nplanes = 1;
    s.thr = 1;
    while (maxc - s.thr >= s.thr) {
      s.thr *= 2;
      nplanes++;
    }
    for (uint8_t p = 0; p < nplanes; p++) {
      s.plane = 63 - __builtin_clzll(s.thr);
      speck_sorting_pass(&s);
      if (s.bits.pos >= budget)
        break;
      /* refinement (src/SPECK_INT.cpp:310-357, ENC.cpp:220-227) */
      for (size_t i = 0; i < s.n; i++)
        if (mask_get(s.lsp, i)) {
          const int b = s.coef[i] >= s.thr;
          if (b)
            s.coef[i] -= s.thr;
          fifo_put(&s.bits, b);
        }
      for (size_t k = 0; k < s.nfresh; k++) {
        s.coef[s.fresh[k]] -= s.thr;
        mask_set(s.lsp, s.fresh[k]);
      }
      s.nfresh = 0;
      if (s.bits.pos >= budget)
        break;
      s.thr /= 2;
      speck_clean_lists(&s);
    }
    total_bits = s.bits.pos;
  }

  /* header {u8 planes, u64 total_bits} + first min(budget,total) bits (SPECK_INT.cpp:264-308) */
  const size_t keep = total_bits < budget ? (size_t)total_bits : budget;
  const size_t nbytes = (keep + 7) / 8;
  uint8_t* out = (uint8_t*)calloc(9 + nbytes + 8, 1);
  out[0] = nplanes;
  memcpy(out + 1, &total_bits, 8);
  if (nbytes)
    memcpy(out + 9, s.bits.w, nbytes);
  if (keep % 8) /* bits past `keep` inside the last byte can only exist when keep==budget, */
    out[9 + nbytes - 1] &= (uint8_t)((1u << (keep % 8)) - 1); /* which is a multiple of 8   */
  *stream = out;
  *stream_len = 9 + nbytes;
  free(s.coef);
  speck_free(&s);
  return 0;
}

/* src/SPECK_INT.cpp:79-108,165-228,359-469 */
int orc_speck3d_decode(const uint8_t* stream, size_t len, const size_t dims[3], uint64_t* coeffs,
                       uint64_t* signs)
{
  if (len < 9)
    return 1;
  speck s;
  speck_alloc(&s, dims, 0);
  const uint8_t nplanes = stream[0];
  uint64_t total_bits;
  memcpy(&total_bits, stream + 1, 8);
  uint64_t avail = (uint64_t)(len - 9) * 8;
  if (avail > total_bits)
    avail = total_bits;
  /* zero-padded bit buffer of total_bits (SPECK_INT.cpp:95-105) */
  s.bits.cap_words = (size_t)(total_bits / 64 + 2);
  s.bits.w = (uint64_t*)calloc(s.bits.cap_words, 8);
  memcpy(s.bits.w, stream + 9, (size_t)((avail + 7) / 8));

  s.coef = coeffs;
  s.sign = signs;
  memset(coeffs, 0, s.n * sizeof(uint64_t));
  memset(signs, 0xff, ((s.n + 63) / 64) * 8);
  speck_init_lists(&s);

  if (nplanes) {
    s.thr = (uint64_t)1 << (nplanes - 1);
    for (uint8_t p = 0; p < nplanes; p++) {
      speck_sorting_pass(&s);
      if (s.bits.pos >= avail)
        break;
      /* refinement with per-bit exhaustion test (SPECK_INT.cpp:374-451) */
      const uint64_t half = s.thr / 2;
      int exhausted = 0;
      for (size_t i = 0; i < s.n && !exhausted; i++) {
        if ((i & 63) == 0 && s.lsp[i >> 6] == 0) {
          i += 63;
          continue;
        }
        if (!mask_get(s.lsp, i))
          continue;
        const int b = fifo_get(&s.bits);
        if (s.thr >= 2)
          s.coef[i] = b ? s.coef[i] + half : s.coef[i] - half;
        else if (b)
          s.coef[i]++;
        if (s.bits.pos == avail)
          exhausted = 1;
      }
      const uint64_t init = s.thr + s.thr - s.thr / 2 - 1; /* SPECK_INT.cpp:462-468 */
      for (size_t k = 0; k < s.nfresh; k++) {
        s.coef[s.fresh[k]] = init;
        mask_set(s.lsp, s.fresh[k]);
      }
      s.nfresh = 0;
      if (s.bits.pos >= avail)
        break;
      s.thr /= 2;
      speck_clean_lists(&s);
    }
    if (s.nfresh) { /* loop left right after a sorting pass (SPECK_INT.cpp:216-220) */
      const uint64_t init = s.thr + s.thr - s.thr / 2 - 1;
      for (size_t k = 0; k < s.nfresh; k++)
        s.coef[s.fresh[k]] = init;
    }
  }
  speck_free(&s);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* SPECK1D (the coder of the outlier list)                                                    */
/* ------------------------------------------------------------------------------------------ */

/* include/SPECK1D_INT.h:10-42 : a run of the array; its LIS level is the number of splits */
typedef struct {
  uint64_t start, len;
} run1;

typedef struct {
  run1* v;
  size_t n, cap;
} runlist;

typedef struct {
  size_t n;
  int encoding;
  uint64_t* coef; /* encoder: working copy, decoder: output */
  uint64_t* sign;
  uint32_t* nzpre; /* encoder only: non-zero values before each 64-value block (skips empty runs) */
  uint64_t *lip, *lsp;
  size_t *fresh, nfresh, capfresh;
  runlist* lis;
  size_t nlis;
  bitfifo bits;
  uint64_t thr;
} speck1;

static void run_push(runlist* l, run1 r)
{
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 16;
    l->v = (run1*)realloc(l->v, l->cap * sizeof(run1));
  }
  l->v[l->n++] = r;
}

static void fresh1_push(speck1* s, size_t idx)
{
  if (s->nfresh == s->capfresh) {
    s->capfresh = s->capfresh ? s->capfresh * 2 : 1024;
    s->fresh = (size_t*)realloc(s->fresh, s->capfresh * sizeof(size_t));
  }
  s->fresh[s->nfresh++] = idx;
}

/* src/SPECK1D_INT_ENC.cpp:165-178 (any value of the run at or above the threshold) */
static int run_significant(const speck1* s, run1 r)
{
  const uint64_t end = r.start + r.len;
  /* whole 64-blocks without a non-zero value cannot hold a significant one */
  uint64_t i = r.start;
  while (i < end) {
    if ((i & 63) == 0 && i + 64 <= end && s->nzpre[(i >> 6) + 1] == s->nzpre[i >> 6]) {
      i += 64;
      continue;
    }
    if (s->coef[i] >= s->thr)
      return 1;
    i++;
  }
  return 0;
}

static void speck1_test_run(speck1* s, size_t lev, size_t idx, int* counter, int coded);

/* src/SPECK1D_INT_ENC.cpp:97-118, _DEC.cpp:72-87 */
static void speck1_test_pixel(speck1* s, size_t idx, int* counter, int coded)
{
  int sig = 1;
  if (s->encoding) {
    if (coded) { /* (an uncoded pixel is significant by inference) */
      sig = s->coef[idx] >= s->thr;
      fifo_put(&s->bits, sig);
    }
  }
  else if (coded)
    sig = fifo_get(&s->bits);
  if (sig) {
    (*counter)++;
    if (s->encoding) {
      fifo_put(&s->bits, mask_get(s->sign, idx));
      s->coef[idx] -= s->thr;
    }
    else if (fifo_get(&s->bits))
      mask_set(s->sign, idx);
    else
      mask_clr(s->sign, idx);
    fresh1_push(s, idx);
    mask_clr(s->lip, idx);
  }
}

/* src/SPECK1D_INT.cpp:36-56 + _ENC.cpp:120-163, _DEC.cpp:89-124 : halves (len - len/2, len/2),
 * one level deeper; the second half is not coded when the first turned out insignificant */
static void speck1_split_run(speck1* s, size_t lev, size_t idx)
{
  const run1 parent = s->lis[lev].v[idx];
  run1 kid[2];
  kid[0].start = parent.start;
  kid[0].len = parent.len - parent.len / 2;
  kid[1].start = parent.start + kid[0].len;
  kid[1].len = parent.len / 2;
  int found = 0;
  for (int k = 0; k < 2; k++) {
    const int coded = k == 0 || found != 0;
    if (kid[k].len == 1) {
      mask_set(s->lip, kid[k].start);
      speck1_test_pixel(s, kid[k].start, &found, coded);
    }
    else {
      run_push(&s->lis[lev + 1], kid[k]);
      speck1_test_run(s, lev + 1, s->lis[lev + 1].n - 1, &found, coded);
    }
  }
}

/* src/SPECK1D_INT_ENC.cpp:58-95, _DEC.cpp:56-70 */
static void speck1_test_run(speck1* s, size_t lev, size_t idx, int* counter, int coded)
{
  int sig = 1;
  if (s->encoding) {
    sig = coded ? run_significant(s, s->lis[lev].v[idx]) : 1;
    if (coded)
      fifo_put(&s->bits, sig);
  }
  else if (coded)
    sig = fifo_get(&s->bits);
  if (sig) {
    (*counter)++;
    speck1_split_run(s, lev, idx);
    s->lis[lev].v[idx].len = 0;
  }
}

/* src/SPECK1D_INT_ENC.cpp:12-56, _DEC.cpp:12-54 */
static void speck1_sorting_pass(speck1* s)
{
  for (size_t i = 0; i < s->n; i++) {
    if ((i & 63) == 0 && s->lip[i >> 6] == 0) {
      i += 63;
      continue;
    }
    if (!mask_get(s->lip, i))
      continue;
    int dummy = 0;
    speck1_test_pixel(s, i, &dummy, 1);
  }
  for (size_t lev = s->nlis; lev-- > 0;)
    for (size_t i = 0; i < s->lis[lev].n; i++) {
      int dummy = 0;
      speck1_test_run(s, lev, i, &dummy, 1);
    }
}

static void speck1_alloc(speck1* s, size_t n, int encoding)
{
  memset(s, 0, sizeof(*s));
  s->n = n;
  s->encoding = encoding;
  const size_t words = (n + 63) / 64;
  s->lip = (uint64_t*)calloc(words, 8);
  s->lsp = (uint64_t*)calloc(words, 8);
  /* src/SPECK1D_INT.cpp:19-34 : the two halves of the array, on the list of level 1 */
  s->nlis = orc_num_of_partitions(n) + 1;
  s->lis = (runlist*)calloc(s->nlis + 1, sizeof(runlist));
  run1 whole = {0, n}, kid[2];
  kid[0].start = 0;
  kid[0].len = whole.len - whole.len / 2;
  kid[1].start = kid[0].len;
  kid[1].len = whole.len / 2;
  run_push(&s->lis[1], kid[0]);
  run_push(&s->lis[1], kid[1]);
}

static void speck1_free(speck1* s)
{
  for (size_t i = 0; i <= s->nlis; i++)
    free(s->lis[i].v);
  free(s->lis);
  free(s->lip);
  free(s->lsp);
  free(s->fresh);
  free(s->nzpre);
  free(s->bits.w);
}

static void speck1_clean_lists(speck1* s)
{
  for (size_t lev = 0; lev < s->nlis; lev++) {
    runlist* l = &s->lis[lev];
    size_t k = 0;
    for (size_t i = 0; i < l->n; i++)
      if (l->v[i].len)
        l->v[k++] = l->v[i];
    l->n = k;
  }
}

/* src/SPECK_INT.cpp:110-163 with the 1D passes; no bit budget (the outlier coder never sets one) */
int orc_speck1d_encode(const uint64_t* coeffs, const uint64_t* signs, size_t n, uint8_t** stream,
                       size_t* stream_len)
{
  speck1 s;
  speck1_alloc(&s, n, 1);
  s.coef = (uint64_t*)malloc(n * sizeof(uint64_t));
  memcpy(s.coef, coeffs, n * sizeof(uint64_t));
  s.sign = (uint64_t*)signs;
  s.nzpre = (uint32_t*)calloc(n / 64 + 2, sizeof(uint32_t));
  uint64_t maxc = 0;
  for (size_t i = 0; i < n; i++) {
    if (s.coef[i] > maxc)
      maxc = s.coef[i];
    if (s.coef[i])
      s.nzpre[(i >> 6) + 1]++;
  }
  for (size_t b = 0; b < n / 64 + 1; b++)
    s.nzpre[b + 1] += s.nzpre[b];
  uint8_t nplanes = 0;
  uint64_t total_bits = 0;
  if (maxc) {
    nplanes = 1;
    s.thr = 1;
    while (maxc - s.thr >= s.thr) {
      s.thr *= 2;
      nplanes++;
    }
    for (uint8_t p = 0; p < nplanes; p++) {
      speck1_sorting_pass(&s);
      for (size_t i = 0; i < n; i++) { /* src/SPECK_INT.cpp:310-357 */
        if ((i & 63) == 0 && s.lsp[i >> 6] == 0) {
          i += 63;
          continue;
        }
        if (!mask_get(s.lsp, i))
          continue;
        const int b = s.coef[i] >= s.thr;
        if (b)
          s.coef[i] -= s.thr;
        fifo_put(&s.bits, b);
      }
      for (size_t k = 0; k < s.nfresh; k++)
        mask_set(s.lsp, s.fresh[k]);
      s.nfresh = 0;
      s.thr /= 2;
      speck1_clean_lists(&s);
    }
    total_bits = s.bits.pos;
  }
  const size_t nbytes = (size_t)((total_bits + 7) / 8);
  uint8_t* out = (uint8_t*)calloc(9 + nbytes + 8, 1);
  out[0] = nplanes;
  memcpy(out + 1, &total_bits, 8);
  if (nbytes)
    memcpy(out + 9, s.bits.w, nbytes);
  *stream = out;
  *stream_len = 9 + nbytes;
  free(s.coef);
  speck1_free(&s);
  return 0;
}

/* src/SPECK_INT.cpp:165-228 with the 1D passes; the stream is complete (a partial outlier stream
 * is discarded by the caller, src/SPECK_FLT.cpp:88-103) */
int orc_speck1d_decode(const uint8_t* stream, size_t len, size_t n, uint64_t* coeffs,
                       uint64_t* signs)
{
  if (len < 9)
    return 1;
  speck1 s;
  speck1_alloc(&s, n, 0);
  const uint8_t nplanes = stream[0];
  uint64_t total_bits;
  memcpy(&total_bits, stream + 1, 8);
  if ((uint64_t)(len - 9) * 8 < total_bits) {
    speck1_free(&s);
    return 1;
  }
  s.bits.cap_words = (size_t)(total_bits / 64 + 2);
  s.bits.w = (uint64_t*)calloc(s.bits.cap_words, 8);
  memcpy(s.bits.w, stream + 9, (size_t)((total_bits + 7) / 8));
  s.coef = coeffs;
  s.sign = signs;
  memset(coeffs, 0, n * sizeof(uint64_t));
  memset(signs, 0xff, ((n + 63) / 64) * 8);
  if (nplanes) {
    s.thr = (uint64_t)1 << (nplanes - 1);
    for (uint8_t p = 0; p < nplanes; p++) {
      speck1_sorting_pass(&s);
      if (s.bits.pos >= total_bits)
        break;
      const uint64_t half = s.thr / 2;
      int exhausted = 0;
      for (size_t i = 0; i < n && !exhausted; i++) {
        if ((i & 63) == 0 && s.lsp[i >> 6] == 0) {
          i += 63;
          continue;
        }
        if (!mask_get(s.lsp, i))
          continue;
        const int b = fifo_get(&s.bits);
        if (s.thr >= 2)
          s.coef[i] = b ? s.coef[i] + half : s.coef[i] - half;
        else if (b)
          s.coef[i]++;
        if (s.bits.pos == total_bits)
          exhausted = 1;
      }
      const uint64_t init = s.thr + s.thr - s.thr / 2 - 1;
      for (size_t k = 0; k < s.nfresh; k++) {
        s.coef[s.fresh[k]] = init;
        mask_set(s.lsp, s.fresh[k]);
      }
      s.nfresh = 0;
      if (s.bits.pos >= total_bits)
        break;
      s.thr /= 2;
      speck1_clean_lists(&s);
    }
    if (s.nfresh) {
      const uint64_t init = s.thr + s.thr - s.thr / 2 - 1;
      for (size_t k = 0; k < s.nfresh; k++)
        s.coef[s.fresh[k]] = init;
    }
  }
  speck1_free(&s);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* SPECK2D (slices)                                                                           */
/* ------------------------------------------------------------------------------------------ */

/* include/SPECK2D_INT.h:12-31 */
typedef struct {
  uint32_t sx, sy, lx, ly;
  uint16_t lev;
} rect2;

typedef struct {
  rect2* v;
  size_t n, cap;
} rectlist;

typedef struct {
  size_t dx, dy, n;
  int encoding;
  uint64_t* coef;
  uint64_t* sign;
  int8_t* msb; /* encoder: msb of every coefficient, raster order (SPECK2D_INT_ENC.cpp:101-107) */
  uint64_t *lip, *lsp;
  size_t *fresh, nfresh, capfresh;
  rectlist* lis;
  size_t nlis;
  rect2 I; /* the type-I set: everything outside the box [0,sx) x [0,sy); lev == 0: empty */
  bitfifo bits;
  uint64_t thr;
  int plane;
} speck2;

static void rect_push(rectlist* l, rect2 r)
{
  if (l->n == l->cap) {
    l->cap = l->cap ? l->cap * 2 : 16;
    l->v = (rect2*)realloc(l->v, l->cap * sizeof(rect2));
  }
  l->v[l->n++] = r;
}

static void fresh2_push(speck2* s, size_t idx)
{
  if (s->nfresh == s->capfresh) {
    s->capfresh = s->capfresh ? s->capfresh * 2 : 1024;
    s->fresh = (size_t*)realloc(s->fresh, s->capfresh * sizeof(size_t));
  }
  s->fresh[s->nfresh++] = idx;
}

static int rect_any_significant(const speck2* s, uint32_t x0, uint32_t x1, uint32_t y0, uint32_t y1)
{
  for (uint32_t y = y0; y < y1; y++) {
    const int8_t* row = s->msb + (size_t)y * s->dx;
    for (uint32_t x = x0; x < x1; x++)
      if (row[x] >= s->plane)
        return 1;
  }
  return 0;
}

static void speck2_test_set(speck2* s, size_t lev, size_t idx, int* counter, int coded);

/* src/SPECK2D_INT_ENC.cpp:29-46, _DEC.cpp:24-39 */
static void speck2_test_pixel(speck2* s, size_t idx, int* counter, int coded)
{
  int sig = 1;
  if (coded) {
    if (s->encoding) {
      sig = s->msb[idx] >= s->plane;
      fifo_put(&s->bits, sig);
    }
    else
      sig = fifo_get(&s->bits);
  }
  if (sig) {
    (*counter)++;
    if (s->encoding)
      fifo_put(&s->bits, mask_get(s->sign, idx));
    else if (fifo_get(&s->bits))
      mask_set(s->sign, idx);
    else
      mask_clr(s->sign, idx);
    fresh2_push(s, idx);
    mask_clr(s->lip, idx);
  }
}

/* src/SPECK2D_INT.cpp:58-82,104-147 : quadrants in the order bottom-right, bottom-left,
 * top-right, top-left; the last non-empty one is not coded when none before it was significant */
static void speck2_split_set(speck2* s, size_t lev, size_t idx)
{
  const rect2 p = s->lis[lev].v[idx];
  const uint32_t dlx = p.lx / 2, dly = p.ly / 2, alx = p.lx - dlx, aly = p.ly - dly;
  rect2 kid[4] = {{p.sx + alx, p.sy + aly, dlx, dly, (uint16_t)(p.lev + 1)},
                  {p.sx, p.sy + aly, alx, dly, (uint16_t)(p.lev + 1)},
                  {p.sx + alx, p.sy, dlx, aly, (uint16_t)(p.lev + 1)},
                  {p.sx, p.sy, alx, aly, (uint16_t)(p.lev + 1)}};
  int nkid = 0;
  for (int k = 0; k < 4; k++)
    if (kid[k].lx && kid[k].ly)
      kid[nkid++] = kid[k];
  int found = 0;
  for (int k = 0; k < nkid; k++) {
    const int coded = found != 0 || k + 1 != nkid;
    if (kid[k].lx == 1 && kid[k].ly == 1) {
      const size_t pix = (size_t)kid[k].sy * s->dx + kid[k].sx;
      mask_set(s->lip, pix);
      speck2_test_pixel(s, pix, &found, coded);
    }
    else {
      rect_push(&s->lis[kid[k].lev], kid[k]);
      speck2_test_set(s, kid[k].lev, s->lis[kid[k].lev].n - 1, &found, coded);
    }
  }
}

/* src/SPECK2D_INT_ENC.cpp:7-27, _DEC.cpp:6-22 */
static void speck2_test_set(speck2* s, size_t lev, size_t idx, int* counter, int coded)
{
  int sig = 1;
  if (coded) {
    if (s->encoding) {
      const rect2* r = &s->lis[lev].v[idx];
      sig = rect_any_significant(s, r->sx, r->sx + r->lx, r->sy, r->sy + r->ly);
      fifo_put(&s->bits, sig);
    }
    else
      sig = fifo_get(&s->bits);
  }
  if (sig) {
    (*counter)++;
    speck2_split_set(s, lev, idx);
    s->lis[lev].v[idx].lx = 0; /* make_empty */
  }
}

static void speck2_test_I(speck2* s, int coded);

/* src/SPECK2D_INT.cpp:84-98,149-186 : the three subbands of the level join the lists (always
 * coded), I shrinks; the new I is significant by inference when none of them was */
static void speck2_code_I(speck2* s)
{
  size_t ax, dxl, ay, dyl;
  orc_approx_detail_len(s->dx, s->I.lev, &ax, &dxl);
  orc_approx_detail_len(s->dy, s->I.lev, &ay, &dyl);
  const rect2 kid[3] = {{(uint32_t)ax, (uint32_t)ay, (uint32_t)dxl, (uint32_t)dyl, s->I.lev},
                        {(uint32_t)ax, 0, (uint32_t)dxl, (uint32_t)ay, s->I.lev},
                        {0, (uint32_t)ay, (uint32_t)ax, (uint32_t)dyl, s->I.lev}};
  s->I.sx += (uint32_t)dxl;
  s->I.sy += (uint32_t)dyl;
  s->I.lev--;
  int found = 0;
  for (int k = 0; k < 3; k++)
    if (kid[k].lx && kid[k].ly) {
      rect_push(&s->lis[kid[k].lev], kid[k]);
      speck2_test_set(s, kid[k].lev, s->lis[kid[k].lev].n - 1, &found, 1);
    }
  speck2_test_I(s, found != 0);
}

/* src/SPECK2D_INT_ENC.cpp:48-62,78-99, _DEC.cpp:41-51 */
static void speck2_test_I(speck2* s, int coded)
{
  if (s->I.lev == 0)
    return;
  int sig = 1;
  if (coded) {
    if (s->encoding) {
      sig = rect_any_significant(s, 0, (uint32_t)s->dx, s->I.sy, (uint32_t)s->dy) ||
            rect_any_significant(s, s->I.sx, (uint32_t)s->dx, 0, s->I.sy);
      fifo_put(&s->bits, sig);
    }
    else
      sig = fifo_get(&s->bits);
  }
  if (sig)
    speck2_code_I(s);
}

/* src/SPECK2D_INT.cpp:10-56 */
static void speck2_sorting_pass(speck2* s)
{
  for (size_t i = 0; i < s->n; i++) {
    if ((i & 63) == 0 && s->lip[i >> 6] == 0) {
      i += 63;
      continue;
    }
    if (!mask_get(s->lip, i))
      continue;
    int dummy = 0;
    speck2_test_pixel(s, i, &dummy, 1);
  }
  for (size_t lev = s->nlis; lev-- > 0;)
    for (size_t i = 0; i < s->lis[lev].n; i++) {
      int dummy = 0;
      speck2_test_set(s, lev, i, &dummy, 1);
    }
  speck2_test_I(s, 1);
}

/* src/SPECK2D_INT.cpp:188-218 */
static void speck2_alloc(speck2* s, size_t dx, size_t dy, int encoding)
{
  memset(s, 0, sizeof(*s));
  s->dx = dx;
  s->dy = dy;
  s->n = dx * dy;
  s->encoding = encoding;
  const size_t words = (s->n + 63) / 64;
  s->lip = (uint64_t*)calloc(words, 8);
  s->lsp = (uint64_t*)calloc(words, 8);
  s->nlis = orc_num_of_partitions(dx > dy ? dx : dy) + 1;
  s->lis = (rectlist*)calloc(s->nlis + 1, sizeof(rectlist));
  const size_t nx = orc_num_of_xforms(dx < dy ? dx : dy);
  size_t ax, ay, d;
  orc_approx_detail_len(dx, nx, &ax, &d);
  orc_approx_detail_len(dy, nx, &ay, &d);
  const rect2 root = {0, 0, (uint32_t)ax, (uint32_t)ay, (uint16_t)nx};
  rect_push(&s->lis[nx], root);
  s->I.sx = (uint32_t)ax;
  s->I.sy = (uint32_t)ay;
  s->I.lx = (uint32_t)dx;
  s->I.ly = (uint32_t)dy;
  s->I.lev = (uint16_t)nx;
}

static void speck2_free(speck2* s)
{
  for (size_t i = 0; i <= s->nlis; i++)
    free(s->lis[i].v);
  free(s->lis);
  free(s->lip);
  free(s->lsp);
  free(s->fresh);
  free(s->msb);
  free(s->bits.w);
}

static void speck2_clean_lists(speck2* s)
{
  for (size_t lev = 0; lev < s->nlis; lev++) {
    rectlist* l = &s->lis[lev];
    size_t k = 0;
    for (size_t i = 0; i < l->n; i++)
      if (l->v[i].lx && l->v[i].ly)
        l->v[k++] = l->v[i];
    l->n = k;
  }
}

/* src/SPECK_INT.cpp:110-163,284-357 with the 2D passes */
int orc_speck2d_encode(const uint64_t* coeffs, const uint64_t* signs, size_t dx, size_t dy,
                       size_t budget_bits, uint8_t** stream, size_t* stream_len)
{
  speck2 s;
  speck2_alloc(&s, dx, dy, 1);
  const size_t budget = round_budget(budget_bits);
  s.coef = (uint64_t*)malloc(s.n * sizeof(uint64_t));
  memcpy(s.coef, coeffs, s.n * sizeof(uint64_t));
  s.sign = (uint64_t*)signs;
  s.msb = (int8_t*)malloc(s.n);
  uint64_t maxc = 0;
  for (size_t i = 0; i < s.n; i++) {
    if (s.coef[i] > maxc)
      maxc = s.coef[i];
    s.msb[i] = s.coef[i] ? (int8_t)(63 - __builtin_clzll(s.coef[i])) : -1;
  }
  uint8_t nplanes = 0;
  uint64_t total_bits = 0;
  if (maxc) {
    nplanes = 1;
    s.thr = 1;
    while (maxc - s.thr >= s.thr) {
      s.thr *= 2;
      nplanes++;
    }
    for (uint8_t p = 0; p < nplanes; p++) {
      s.plane = 63 - __builtin_clzll(s.thr);
      speck2_sorting_pass(&s);
      if (s.bits.pos >= budget)
        break;
      for (size_t i = 0; i < s.n; i++) {
        if ((i & 63) == 0 && s.lsp[i >> 6] == 0) {
          i += 63;
          continue;
        }
        if (!mask_get(s.lsp, i))
          continue;
        const int b = s.coef[i] >= s.thr;
        if (b)
          s.coef[i] -= s.thr;
        fifo_put(&s.bits, b);
      }
      for (size_t k = 0; k < s.nfresh; k++) {
        s.coef[s.fresh[k]] -= s.thr;
        mask_set(s.lsp, s.fresh[k]);
      }
      s.nfresh = 0;
      if (s.bits.pos >= budget)
        break;
      s.thr /= 2;
      speck2_clean_lists(&s);
    }
    total_bits = s.bits.pos;
  }
  const size_t keep = total_bits < budget ? (size_t)total_bits : budget;
  const size_t nbytes = (keep + 7) / 8;
  uint8_t* out = (uint8_t*)calloc(9 + nbytes + 8, 1);
  out[0] = nplanes;
  memcpy(out + 1, &total_bits, 8);
  if (nbytes)
    memcpy(out + 9, s.bits.w, nbytes);
  if (keep % 8)
    out[9 + nbytes - 1] &= (uint8_t)((1u << (keep % 8)) - 1);
  *stream = out;
  *stream_len = 9 + nbytes;
  free(s.coef);
  speck2_free(&s);
  return 0;
}

/* src/SPECK_INT.cpp:79-108,165-228,359-469 with the 2D passes */
int orc_speck2d_decode(const uint8_t* stream, size_t len, size_t dx, size_t dy, uint64_t* coeffs,
                       uint64_t* signs)
{
  if (len < 9)
    return 1;
  speck2 s;
  speck2_alloc(&s, dx, dy, 0);
  const uint8_t nplanes = stream[0];
  uint64_t total_bits;
  memcpy(&total_bits, stream + 1, 8);
  uint64_t avail = (uint64_t)(len - 9) * 8;
  if (avail > total_bits)
    avail = total_bits;
  s.bits.cap_words = (size_t)(total_bits / 64 + 2);
  s.bits.w = (uint64_t*)calloc(s.bits.cap_words, 8);
  memcpy(s.bits.w, stream + 9, (size_t)((avail + 7) / 8));
  s.coef = coeffs;
  s.sign = signs;
  memset(coeffs, 0, s.n * sizeof(uint64_t));
  memset(signs, 0xff, ((s.n + 63) / 64) * 8);
  if (nplanes) {
    s.thr = (uint64_t)1 << (nplanes - 1);
    for (uint8_t p = 0; p < nplanes; p++) {
      speck2_sorting_pass(&s);
      if (s.bits.pos >= avail)
        break;
      const uint64_t half = s.thr / 2;
      int exhausted = 0;
      for (size_t i = 0; i < s.n && !exhausted; i++) {
        if ((i & 63) == 0 && s.lsp[i >> 6] == 0) {
          i += 63;
          continue;
        }
        if (!mask_get(s.lsp, i))
          continue;
        const int b = fifo_get(&s.bits);
        if (s.thr >= 2)
          s.coef[i] = b ? s.coef[i] + half : s.coef[i] - half;
        else if (b)
          s.coef[i]++;
        if (s.bits.pos == avail)
          exhausted = 1;
      }
      const uint64_t init = s.thr + s.thr - s.thr / 2 - 1;
      for (size_t k = 0; k < s.nfresh; k++) {
        s.coef[s.fresh[k]] = init;
        mask_set(s.lsp, s.fresh[k]);
      }
      s.nfresh = 0;
      if (s.bits.pos >= avail)
        break;
      s.thr /= 2;
      speck2_clean_lists(&s);
    }
    if (s.nfresh) {
      const uint64_t init = s.thr + s.thr - s.thr / 2 - 1;
      for (size_t k = 0; k < s.nfresh; k++)
        s.coef[s.fresh[k]] = init;
    }
  }
  speck2_free(&s);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* outlier coder                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* src/Outlier_Coder.cpp:71-129,179-197 : errors in units of the tolerance, rounded to nearest;
 * pos[] ascending.  Returns 1 on a bad list (an error within the tolerance). */
int orc_outlier_encode(const uint64_t* pos, const double* err, size_t count, size_t n, double tol,
                       uint8_t** stream, size_t* stream_len)
{
  if (n == 0 || tol <= 0.0 || count == 0)
    return 1;
  for (size_t k = 0; k < count; k++)
    if (pos[k] >= n || fabs(err[k]) <= tol)
      return 1;
  uint64_t* coef = (uint64_t*)calloc(n, sizeof(uint64_t));
  uint64_t* sign = (uint64_t*)malloc(((n + 63) / 64) * 8);
  memset(sign, 0xff, ((n + 63) / 64) * 8);
  const double inv = 1.0 / tol;
  for (size_t k = 0; k < count; k++) {
    const long long ll = llrint(err[k] * inv);
    if (ll >= 0)
      mask_set(sign, pos[k]);
    else
      mask_clr(sign, pos[k]);
    coef[pos[k]] = (uint64_t)(ll < 0 ? -ll : ll);
  }
  const int rtn = orc_speck1d_encode(coef, sign, n, stream, stream_len);
  free(coef);
  free(sign);
  return rtn;
}

/* src/Outlier_Coder.cpp:131-152,199-233 : correctors 1.1 tol for magnitude 1, (m - 0.25) tol
 * otherwise, added to vals[] */
int orc_outlier_decode_apply(const uint8_t* stream, size_t len, size_t n, double tol, double* vals)
{
  uint64_t* coef = (uint64_t*)malloc(n * sizeof(uint64_t));
  uint64_t* sign = (uint64_t*)malloc(((n + 63) / 64) * 8);
  const int rtn = orc_speck1d_decode(stream, len, n, coef, sign);
  if (!rtn)
    for (size_t i = 0; i < n; i++) {
      if (coef[i] == 0)
        continue;
      double e = coef[i] == 1 ? 1.1 : (double)coef[i] - 0.25;
      e *= tol * (mask_get(sign, i) ? 1.0 : -1.0);
      vals[i] += e;
    }
  free(coef);
  free(sign);
  return rtn;
}

/* ------------------------------------------------------------------------------------------ */
/* per-chunk float pipeline                                                                   */
/* ------------------------------------------------------------------------------------------ */
/* SPECK2D_FLT and SPECK3D_FLT differ only in the transform and the integer coder
 * (src/SPECK2D_FLT.cpp:48-58, src/SPECK3D_FLT.cpp); a slice is a chunk with dims[2] == 1, whose
 * 3D transform plan already is dwt2d (src/CDF97.cpp:102-112,327-385) */
static int int_encode(int two_d, const uint64_t* coef, const uint64_t* sign, const size_t dims[3],
                      size_t budget, uint8_t** stream, size_t* len)
{
  return two_d ? orc_speck2d_encode(coef, sign, dims[0], dims[1], budget, stream, len)
               : orc_speck3d_encode(coef, sign, dims, budget, stream, len);
}
static int int_decode(int two_d, const uint8_t* stream, size_t len, const size_t dims[3],
                      uint64_t* coef, uint64_t* sign)
{
  return two_d ? orc_speck2d_decode(stream, len, dims[0], dims[1], coef, sign)
               : orc_speck3d_decode(stream, len, dims, coef, sign);
}


/* src/SPECK_FLT.cpp:401-541, CompMode::Rate only */
static int chunk_compress_rate_impl(int two_d, double* vals, const size_t dims[3], double bpp, uint8_t** stream,
                            size_t* stream_len)
{
  const size_t n = dims[0] * dims[1] * dims[2];
  uint8_t condi[17];
  if (orc_condition(vals, n, condi)) {
    *stream = (uint8_t*)malloc(17);
    memcpy(*stream, condi, 17);
    *stream_len = 17;
    return 0;
  }
  orc_dwt3d(vals, dims);
  double maxabs = 0.0;
  for (size_t i = 0; i < n; i++)
    if (fabs(vals[i]) > maxabs)
      maxabs = fabs(vals[i]);

  uint64_t* coef = (uint64_t*)malloc(n * sizeof(uint64_t));
  uint64_t* sign = (uint64_t*)malloc(((n + 63) / 64) * sizeof(uint64_t));
  const size_t budget = (size_t)(bpp * (double)n);
  uint8_t* speck_stream = NULL;
  size_t speck_len = 0;
  int rtn = 0;
  for (int high_prec = 0; high_prec < 2; high_prec++) {
    /* SPECK_FLT.cpp:282-301 */
    const double q = high_prec ? maxabs / 0x1.fffffffffffffp52 : maxabs / 4294967295.0;
    memcpy(condi + 9, &q, 8);
    int width;
    rtn = orc_quantize(vals, n, q, coef, sign, &width);
    if (rtn)
      break;
    free(speck_stream);
    int_encode(two_d, coef, sign, dims, budget, &speck_stream, &speck_len);
    if (speck_len * 8 >= budget) /* SPECK_FLT.cpp:530-538 : enough bits, no retry */
      break;
  }
  if (!rtn) {
    *stream = (uint8_t*)malloc(17 + speck_len);
    memcpy(*stream, condi, 17);
    memcpy(*stream + 17, speck_stream, speck_len);
    *stream_len = 17 + speck_len;
  }
  free(speck_stream);
  free(coef);
  free(sign);
  return rtn;
}

/* src/SPECK_FLT.cpp:237-266 : strides of 4096 values are summed one after the other, each
 * sequentially; the canonical build contracts `init + diff * diff` into one fma */
double orc_estimate_mse_midtread(const double* vals, size_t n, double q)
{
  const size_t stride = 4096, nstrides = n / stride;
  const double rcp_q = 1.0 / q;
  double total = 0.0;
  for (size_t s = 0; s <= nstrides; s++) {
    const size_t beg = s * stride, end = s < nstrides ? beg + stride : n;
    double acc = 0.0;
    for (size_t i = beg; i < end; i++) {
      const double diff = fma(-q, rint(vals[i] * rcp_q), vals[i]);
      acc = fma(diff, diff, acc);
    }
    total += acc;
  }
  return total / (double)n;
}

/* src/SPECK_FLT.cpp:268-279 : q for a target PSNR, given the range of the conditioned data */
double orc_estimate_q_psnr(const double* coeffs, size_t n, double range, double psnr)
{
  const double t_mse = (range * range) * pow(10.0, -psnr / 10.0);
  double q = 2.0 * sqrt(t_mse * 3.0);
  while (orc_estimate_mse_midtread(coeffs, n, q) > t_mse)
    q /= exp2(0.25);
  return q;
}

/* src/SPECK_FLT.cpp:401-541, CompMode::PSNR: q from the target, every bit plane coded */
static int chunk_compress_psnr_impl(int two_d, double* vals, const size_t dims[3], double psnr, uint8_t** stream,
                            size_t* stream_len)
{
  const size_t n = dims[0] * dims[1] * dims[2];
  uint8_t condi[17];
  if (orc_condition(vals, n, condi)) {
    *stream = (uint8_t*)malloc(17);
    memcpy(*stream, condi, 17);
    *stream_len = 17;
    return 0;
  }
  double vmin = vals[0], vmax = vals[0]; /* SPECK_FLT.cpp:431-435 */
  for (size_t i = 1; i < n; i++) {
    if (vals[i] < vmin)
      vmin = vals[i];
    if (vals[i] > vmax)
      vmax = vals[i];
  }
  orc_dwt3d(vals, dims);
  const double q = orc_estimate_q_psnr(vals, n, vmax - vmin, psnr);
  memcpy(condi + 9, &q, 8);
  uint64_t* coef = (uint64_t*)malloc(n * sizeof(uint64_t));
  uint64_t* sign = (uint64_t*)malloc(((n + 63) / 64) * sizeof(uint64_t));
  int width;
  int rtn = orc_quantize(vals, n, q, coef, sign, &width);
  if (!rtn) {
    uint8_t* speck_stream = NULL;
    size_t speck_len = 0;
    int_encode(two_d, coef, sign, dims, 0, &speck_stream, &speck_len);
    *stream = (uint8_t*)malloc(17 + speck_len);
    memcpy(*stream, condi, 17);
    memcpy(*stream + 17, speck_stream, speck_len);
    *stream_len = 17 + speck_len;
    free(speck_stream);
  }
  free(coef);
  free(sign);
  return rtn;
}

/* src/SPECK_FLT.cpp:401-541, CompMode::PWE: q = 1.5 tol (:280-281), every bit plane coded, then
 * the values the decoder would see are rebuilt and every error above the tolerance goes to the
 * outlier coder (:461-486) */
static int chunk_compress_pwe_impl(int two_d, double* vals, const size_t dims[3], double tol, uint8_t** stream,
                           size_t* stream_len)
{
  const size_t n = dims[0] * dims[1] * dims[2];
  uint8_t condi[17];
  if (orc_condition(vals, n, condi)) {
    *stream = (uint8_t*)malloc(17);
    memcpy(*stream, condi, 17);
    *stream_len = 17;
    return 0;
  }
  double* orig = (double*)malloc(n * sizeof(double));
  memcpy(orig, vals, n * sizeof(double));
  orc_dwt3d(vals, dims);
  const double q = 1.5 * tol;
  memcpy(condi + 9, &q, 8);
  uint64_t* coef = (uint64_t*)malloc(n * sizeof(uint64_t));
  uint64_t* sign = (uint64_t*)malloc(((n + 63) / 64) * sizeof(uint64_t));
  int width;
  int rtn = orc_quantize(vals, n, q, coef, sign, &width);
  uint8_t* out_stream = NULL;
  size_t out_len = 0;
  if (!rtn) {
    orc_inv_quantize(coef, sign, n, q, vals);
    orc_idwt3d(vals, dims);
    size_t count = 0;
    for (size_t i = 0; i < n; i++)
      if (fabs(orig[i] - vals[i]) > tol)
        count++;
    if (count) {
      uint64_t* pos = (uint64_t*)malloc(count * sizeof(uint64_t));
      double* err = (double*)malloc(count * sizeof(double));
      size_t k = 0;
      for (size_t i = 0; i < n; i++) {
        const double diff = orig[i] - vals[i];
        if (fabs(diff) > tol) {
          pos[k] = i;
          err[k++] = diff;
        }
      }
      rtn = orc_outlier_encode(pos, err, count, n, tol, &out_stream, &out_len);
      free(pos);
      free(err);
    }
  }
  if (!rtn) {
    uint8_t* speck_stream = NULL;
    size_t speck_len = 0;
    int_encode(two_d, coef, sign, dims, 0, &speck_stream, &speck_len);
    *stream = (uint8_t*)malloc(17 + speck_len + out_len);
    memcpy(*stream, condi, 17);
    memcpy(*stream + 17, speck_stream, speck_len);
    if (out_len)
      memcpy(*stream + 17 + speck_len, out_stream, out_len);
    *stream_len = 17 + speck_len + out_len;
    free(speck_stream);
  }
  free(out_stream);
  free(orig);
  free(coef);
  free(sign);
  return rtn;
}

int orc_chunk_compress_rate(double* vals, const size_t dims[3], double bpp, uint8_t** stream,
                            size_t* stream_len)
{
  return chunk_compress_rate_impl(0, vals, dims, bpp, stream, stream_len);
}
int orc_chunk_compress_psnr(double* vals, const size_t dims[3], double psnr, uint8_t** stream,
                            size_t* stream_len)
{
  return chunk_compress_psnr_impl(0, vals, dims, psnr, stream, stream_len);
}
int orc_chunk_compress_pwe(double* vals, const size_t dims[3], double tol, uint8_t** stream,
                           size_t* stream_len)
{
  return chunk_compress_pwe_impl(0, vals, dims, tol, stream, stream_len);
}

static int chunk_decompress_impl(int two_d, const uint8_t* stream, size_t len,
                                 const size_t dims[3], double* out, double** hier);

/* src/SPECK_FLT.cpp:27-109,543-606 */
int orc_chunk_decompress(const uint8_t* stream, size_t len, const size_t dims[3], double* out)
{
  return chunk_decompress_impl(0, stream, len, dims, out, NULL);
}

/* src/SPECK_FLT.cpp:543-620 with multi_res: hier[h] receives the chunk at its h-th coarsened
 * resolution (orc_coarsened_resolutions), mean added back (:592-603).  A constant chunk fills
 * them with its value (the reference leaves its hierarchy untouched in that case). */
int orc_chunk_decompress_multi_res(const uint8_t* stream, size_t len, const size_t dims[3],
                                   double* out, double** hier)
{
  return chunk_decompress_impl(0, stream, len, dims, out, hier);
}

static int chunk_decompress_impl(int two_d, const uint8_t* stream, size_t len,
                                 const size_t dims[3], double* out, double** hier)
{
  const size_t n = dims[0] * dims[1] * dims[2];
  size_t res[16][3];
  const size_t nres = !hier ? 0 : two_d ? orc_coarsened_resolutions_2d(dims, res, 16)
                                         : orc_coarsened_resolutions(dims, res, 16);
  if (len < 17)
    return 1; /* WrongLength */
  if (stream[0] & 0x01) {
    if (len != 17)
      return 1;
    orc_inverse_condition(out, n, stream);
    for (size_t h = 0; h < nres; h++)
      orc_inverse_condition(hier[h], res[h][0] * res[h][1] * res[h][2], stream);
    return 0;
  }
  double q;
  memcpy(&q, stream + 9, 8);
  const uint8_t* sp = stream + 17;
  size_t remaining = len - 17;
  if (remaining < 9)
    return 1;
  uint64_t total_bits;
  memcpy(&total_bits, sp + 1, 8);
  size_t full = 9 + (size_t)((total_bits + 7) / 8);
  size_t speck_len = full < remaining ? full : remaining;
  uint64_t* coef = (uint64_t*)malloc(n * sizeof(uint64_t));
  uint64_t* sign = (uint64_t*)malloc(((n + 63) / 64) * sizeof(uint64_t));
  int_decode(two_d, sp, speck_len, dims, coef, sign);
  orc_inv_quantize(coef, sign, n, q, out);
  if (nres) {
    orc_idwt3d_multi_res(out, dims, hier);
    for (size_t h = 0; h < nres; h++)
      orc_inverse_condition(hier[h], res[h][0] * res[h][1] * res[h][2], stream);
  }
  else
    orc_idwt3d(out, dims);
  /* an outlier stream follows only when all of it is there (SPECK_FLT.cpp:88-103); its tolerance
   * is q / 1.5 (:578) */
  int rtn = 0;
  if (17 + speck_len < len) {
    const uint8_t* op = sp + speck_len;
    const size_t rem = len - 17 - speck_len;
    if (rem >= 9) {
      uint64_t obits;
      memcpy(&obits, op + 1, 8);
      if (rem == 9 + (size_t)((obits + 7) / 8))
        rtn = orc_outlier_decode_apply(op, rem, n, q / 1.5, out);
    }
  }
  orc_inverse_condition(out, n, stream);
  free(coef);
  free(sign);
  return rtn;
}

/* ------------------------------------------------------------------------------------------ */
/* container + C API mirrors                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* src/SPERR_C_API.cpp:7-97 : one slice through SPECK2D_FLT; the optional 10-byte header is
 * {version, flags (0x20: float input), u32 dimx, u32 dimy} */
int orc_comp_2d(const void* src, int is_float, size_t dimx, size_t dimy, int mode, double quality,
                int out_inc_header, void** dst, size_t* dst_len)
{
  if (*dst != NULL)
    return 1;
  if (quality <= 0.0)
    return 2;
  if (mode < 1 || mode > 3)
    return 2;
  const size_t n = dimx * dimy, dims[3] = {dimx, dimy, 1};
  double* buf = (double*)malloc(n * sizeof(double));
  for (size_t i = 0; i < n; i++)
    buf[i] = is_float ? (double)((const float*)src)[i] : ((const double*)src)[i];
  uint8_t* stream = NULL;
  size_t len = 0;
  const int rtn = mode == 1   ? chunk_compress_rate_impl(1, buf, dims, quality, &stream, &len)
                  : mode == 2 ? chunk_compress_psnr_impl(1, buf, dims, quality, &stream, &len)
                              : chunk_compress_pwe_impl(1, buf, dims, quality, &stream, &len);
  free(buf);
  if (rtn) {
    free(stream);
    return -1;
  }
  const size_t hlen = out_inc_header ? 10 : 0;
  uint8_t* out = (uint8_t*)malloc(hlen + len);
  if (out_inc_header) {
    out[0] = 0; /* SPERR_VERSION_MAJOR */
    out[1] = (uint8_t)(is_float ? 0x20 : 0);
    const uint32_t d2[2] = {(uint32_t)dimx, (uint32_t)dimy};
    memcpy(out + 2, d2, 8);
  }
  memcpy(out + hlen, stream, len);
  free(stream);
  *dst = out;
  *dst_len = hlen + len;
  return 0;
}

/* src/SPERR_C_API.cpp:99-134 : `src` is the stream without the optional header */
int orc_decomp_2d(const void* src, size_t src_len, int output_float, size_t dimx, size_t dimy,
                  void** dst)
{
  if (*dst != NULL)
    return 1;
  const size_t n = dimx * dimy, dims[3] = {dimx, dimy, 1};
  double* outd = (double*)malloc(n * sizeof(double));
  if (chunk_decompress_impl(1, (const uint8_t*)src, src_len, dims, outd, NULL)) {
    free(outd);
    return -1;
  }
  if (output_float) {
    float* f = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++)
      f[i] = (float)outd[i];
    free(outd);
    *dst = f;
  }
  else
    *dst = outd;
  return 0;
}

/* SPECK2D_FLT::decompress(multi_res = true), src/SPECK_FLT.cpp:543-606 with
 * src/SPECK2D_FLT.cpp:52-58: besides the slice (doubles), the slice at every coarsened resolution,
 * coarsest first, mean added back.  level_dims: x y per level; levels[h] is malloc'd. */
int orc_decomp_2d_multi_res(const void* src, size_t src_len, size_t dimx, size_t dimy, void** dst,
                            size_t* nlev, size_t* level_dims, double** levels)
{
  if (*dst != NULL)
    return 1;
  const size_t n = dimx * dimy, dims[3] = {dimx, dimy, 1};
  size_t res[16][3];
  const size_t nres = orc_coarsened_resolutions_2d(dims, res, 16);
  if (nres > 16)
    return -1;
  for (size_t h = 0; h < nres; h++) {
    level_dims[2 * h] = res[h][0];
    level_dims[2 * h + 1] = res[h][1];
    levels[h] = (double*)malloc(res[h][0] * res[h][1] * sizeof(double));
  }
  double* outd = (double*)malloc(n * sizeof(double));
  double* none = NULL;
  if (chunk_decompress_impl(1, (const uint8_t*)src, src_len, dims, outd, nres ? levels : &none)) {
    free(outd);
    for (size_t h = 0; h < nres; h++)
      free(levels[h]);
    return -1;
  }
  *nlev = nres;
  *dst = outd;
  return 0;
}

/* src/SPERR_C_API.cpp:156-216, src/SPERR3D_OMP_C.cpp:61-261 */
int orc_comp_3d(const void* src, int is_float, size_t dimx, size_t dimy, size_t dimz,
                size_t chunk_x, size_t chunk_y, size_t chunk_z, int mode, double quality,
                size_t nthreads, void** dst, size_t* dst_len)
{
  if (*dst != NULL)
    return 1;
  if (quality <= 0.0)
    return 2;
  if (mode < 1 || mode > 3)
    return 2;
  const size_t vol[3] = {dimx, dimy, dimz};
  size_t want[3] = {chunk_x, chunk_y, chunk_z}, cdim[3];
  for (int a = 0; a < 3; a++) { /* SPERR3D_OMP_C.cpp:23-30 */
    cdim[a] = want[a] < 1 ? 1 : want[a];
    if (cdim[a] > vol[a])
      cdim[a] = vol[a];
  }
  const size_t nchunks = orc_chunk_volume(vol, cdim, NULL, 0);
  size_t(*chunks)[6] = (size_t(*)[6])malloc(nchunks * sizeof(*chunks));
  orc_chunk_volume(vol, cdim, chunks, nchunks);
  uint8_t** streams = (uint8_t**)calloc(nchunks, sizeof(uint8_t*));
  size_t* lens = (size_t*)calloc(nchunks, sizeof(size_t));
  int* rtns = (int*)calloc(nchunks, sizeof(int));
  int nt = (int)nthreads;
  (void)nt;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(nt > 0 ? nt : 1) if (nt != 1)
#endif
  for (long long ci = 0; ci < (long long)nchunks; ci++) {
    const size_t* c = chunks[ci];
    const size_t cd[3] = {c[1], c[3], c[5]};
    const size_t cn = cd[0] * cd[1] * cd[2];
    double* buf = (double*)malloc(cn * sizeof(double));
    size_t k = 0; /* gather + widen (SPERR3D_OMP_C.cpp:236-261) */
    for (size_t z = c[4]; z < c[4] + c[5]; z++)
      for (size_t y = c[2]; y < c[2] + c[3]; y++) {
        const size_t row = (z * dimy + y) * dimx + c[0];
        if (is_float)
          for (size_t x = 0; x < cd[0]; x++)
            buf[k++] = (double)((const float*)src)[row + x];
        else
          for (size_t x = 0; x < cd[0]; x++)
            buf[k++] = ((const double*)src)[row + x];
      }
    rtns[ci] = mode == 1   ? orc_chunk_compress_rate(buf, cd, quality, &streams[ci], &lens[ci])
               : mode == 2 ? orc_chunk_compress_psnr(buf, cd, quality, &streams[ci], &lens[ci])
                           : orc_chunk_compress_pwe(buf, cd, quality, &streams[ci], &lens[ci]);
    free(buf);
  }
  int rtn = 0;
  for (size_t i = 0; i < nchunks; i++)
    if (rtns[i])
      rtn = -1;
  if (!rtn) { /* header (SPERR3D_OMP_C.cpp:163-234) */
    const size_t hlen = (nchunks > 1 ? 20 : 14) + 4 * nchunks;
    size_t total = hlen;
    for (size_t i = 0; i < nchunks; i++)
      total += lens[i];
    uint8_t* out = (uint8_t*)malloc(total);
    out[0] = 0; /* SPERR_VERSION_MAJOR (CMakeLists.txt:5) */
    out[1] = (uint8_t)(0x40 | (is_float ? 0x20 : 0) | (nchunks > 1 ? 0x10 : 0));
    size_t pos = 2;
    for (int a = 0; a < 3; a++) {
      uint32_t v = (uint32_t)vol[a];
      memcpy(out + pos, &v, 4);
      pos += 4;
    }
    if (nchunks > 1)
      for (int a = 0; a < 3; a++) {
        uint16_t v = (uint16_t)cdim[a];
        memcpy(out + pos, &v, 2);
        pos += 2;
      }
    for (size_t i = 0; i < nchunks; i++) {
      uint32_t v = (uint32_t)lens[i];
      memcpy(out + pos, &v, 4);
      pos += 4;
    }
    for (size_t i = 0; i < nchunks; i++) {
      memcpy(out + pos, streams[i], lens[i]);
      pos += lens[i];
    }
    *dst = out;
    *dst_len = total;
  }
  for (size_t i = 0; i < nchunks; i++)
    free(streams[i]);
  free(streams);
  free(lens);
  free(rtns);
  free(chunks);
  return rtn;
}

/* src/SPERR_C_API.cpp:218-258, src/SPERR3D_OMP_D.cpp:23-184, SPERR3D_Stream_Tools.cpp:46-105 */
static int decomp_3d_impl(const void* src, size_t src_len, int output_float, size_t nthreads,
                          size_t* dimx, size_t* dimy, size_t* dimz, void** dst, size_t* nlev,
                          size_t (*ldims)[3], double** levels);

int orc_decomp_3d(const void* src, size_t src_len, int output_float, size_t nthreads,
                  size_t* dimx, size_t* dimy, size_t* dimz, void** dst)
{
  return decomp_3d_impl(src, src_len, output_float, nthreads, dimx, dimy, dimz, dst, NULL, NULL,
                        NULL);
}

/* SPERR3D_OMP_D::decompress(p, multi_res = true), src/SPERR3D_OMP_D.cpp:50-150: besides the
 * volume (doubles), the volume at every coarsened resolution (src/sperr_helper.cpp:70-123),
 * coarsest first: *nlev levels (0 when the chunks are not dyadic or do not tile the volume),
 * ldims[h] their dims, levels[h] malloc'd doubles. */
int orc_decomp_3d_multi_res(const void* src, size_t src_len, size_t nthreads, size_t* dimx,
                            size_t* dimy, size_t* dimz, void** dst, size_t* nlev,
                            size_t (*ldims)[3], double** levels)
{
  return decomp_3d_impl(src, src_len, 0, nthreads, dimx, dimy, dimz, dst, nlev, ldims, levels);
}

static int decomp_3d_impl(const void* src, size_t src_len, int output_float, size_t nthreads,
                          size_t* dimx, size_t* dimy, size_t* dimz, void** dst, size_t* nlev,
                          size_t (*ldims)[3], double** levels)
{
  if (*dst != NULL)
    return 1;
  const uint8_t* p = (const uint8_t*)src;
  if (src_len < 18 || p[0] != 0 || !(p[1] & 0x40))
    return -1;
  const int multi = (p[1] & 0x10) != 0;
  uint32_t v3[3];
  memcpy(v3, p + 2, 12);
  size_t vol[3] = {v3[0], v3[1], v3[2]}, cdim[3] = {v3[0], v3[1], v3[2]};
  size_t pos = 14;
  if (multi) {
    uint16_t c3[3];
    memcpy(c3, p + 14, 6);
    for (int a = 0; a < 3; a++)
      cdim[a] = c3[a];
    pos = 20;
  }
  if (!vol[0] || !vol[1] || !vol[2] || !cdim[0] || !cdim[1] || !cdim[2])
    return -1;
  const size_t nchunks = orc_chunk_volume(vol, cdim, NULL, 0);
  size_t(*chunks)[6] = (size_t(*)[6])malloc(nchunks * sizeof(*chunks));
  orc_chunk_volume(vol, cdim, chunks, nchunks);
  size_t* offs = (size_t*)malloc((nchunks + 1) * sizeof(size_t));
  offs[0] = pos + 4 * nchunks;
  for (size_t i = 0; i < nchunks; i++) {
    uint32_t l;
    memcpy(&l, p + pos + 4 * i, 4);
    offs[i + 1] = offs[i] + l;
  }
  if (offs[nchunks] != src_len) {
    free(chunks);
    free(offs);
    return -1;
  }
  const size_t n = vol[0] * vol[1] * vol[2];
  double* outd = (double*)malloc(n * sizeof(double));
  int* rtns = (int*)calloc(nchunks, sizeof(int));
  /* multi-resolution: chunk resolutions scaled by the chunk grid (sperr_helper.cpp:99-123) */
  size_t cres[16][3], nres = 0, grid[3] = {1, 1, 1};
  if (nlev) {
    *nlev = 0;
    if (vol[0] % cdim[0] == 0 && vol[1] % cdim[1] == 0 && vol[2] % cdim[2] == 0) {
      nres = orc_coarsened_resolutions(cdim, cres, 16);
      for (int a = 0; a < 3; a++)
        grid[a] = vol[a] / cdim[a];
      for (size_t h = 0; h < nres; h++) {
        for (int a = 0; a < 3; a++)
          ldims[h][a] = cres[h][a] * grid[a];
        levels[h] = (double*)malloc(ldims[h][0] * ldims[h][1] * ldims[h][2] * sizeof(double));
      }
      *nlev = nres;
    }
  }
  int nt = (int)nthreads;
  (void)nt;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(nt > 0 ? nt : 1) if (nt != 1)
#endif
  for (long long ci = 0; ci < (long long)nchunks; ci++) {
    const size_t* c = chunks[ci];
    const size_t cd[3] = {c[1], c[3], c[5]};
    const size_t cn = cd[0] * cd[1] * cd[2];
    double* buf = (double*)malloc(cn * sizeof(double));
    if (nres) {
      double* hb[16];
      for (size_t h = 0; h < nres; h++)
        hb[h] = (double*)malloc(cres[h][0] * cres[h][1] * cres[h][2] * sizeof(double));
      rtns[ci] = orc_chunk_decompress_multi_res(p + offs[ci], offs[ci + 1] - offs[ci], cd, buf, hb);
      const size_t gi[3] = {c[0] / cdim[0], c[2] / cdim[1], c[4] / cdim[2]};
      for (size_t h = 0; h < nres; h++) { /* SPERR3D_OMP_D.cpp:121-130 */
        const size_t* r = cres[h];
        size_t kk = 0;
        for (size_t z = 0; z < r[2]; z++)
          for (size_t y = 0; y < r[1]; y++) {
            memcpy(levels[h] + ((gi[2] * r[2] + z) * ldims[h][1] + gi[1] * r[1] + y) * ldims[h][0] +
                       gi[0] * r[0],
                   hb[h] + kk, r[0] * sizeof(double));
            kk += r[0];
          }
        free(hb[h]);
      }
    }
    else
      rtns[ci] = orc_chunk_decompress(p + offs[ci], offs[ci + 1] - offs[ci], cd, buf);
    size_t k = 0; /* scatter (SPERR3D_OMP_D.cpp:167-184) */
    for (size_t z = c[4]; z < c[4] + c[5]; z++)
      for (size_t y = c[2]; y < c[2] + c[3]; y++) {
        memcpy(outd + (z * vol[1] + y) * vol[0] + c[0], buf + k, cd[0] * sizeof(double));
        k += cd[0];
      }
    free(buf);
  }
  int rtn = 0;
  for (size_t i = 0; i < nchunks; i++)
    if (rtns[i])
      rtn = -1;
  free(rtns);
  free(chunks);
  free(offs);
  if (rtn) {
    free(outd);
    for (size_t h = 0; h < nres; h++)
      free(levels[h]);
    if (nlev)
      *nlev = 0;
    return rtn;
  }
  *dimx = vol[0];
  *dimy = vol[1];
  *dimz = vol[2];
  if (output_float) {
    float* f = (float*)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; i++)
      f[i] = (float)outd[i];
    free(outd);
    *dst = f;
  }
  else
    *dst = outd;
  return 0;
}

/* src/SPERR_C_API.cpp:260-280, src/SPERR3D_Stream_Tools.cpp:134-226, src/sperr_helper.cpp:401-427:
 * keep `pct` percent of every chunk stream (never less than 64 bytes, never more than it has),
 * flag the container as a portion, rewrite the chunk lengths */
int orc_trunc_3d(const void* src, size_t src_len, unsigned pct, void** dst, size_t* dst_len)
{
  if (*dst != NULL)
    return 1;
  const uint8_t* h = (const uint8_t*)src;
  if (src_len < 20)
    return -1;
  const int multi = (h[1] & 0x10) != 0;
  uint32_t v3[3];
  memcpy(v3, h + 2, 12);
  size_t vol[3] = {v3[0], v3[1], v3[2]}, cd[3] = {v3[0], v3[1], v3[2]};
  size_t pos = 14;
  if (multi) {
    uint16_t c3[3];
    memcpy(c3, h + 14, 6);
    for (int a = 0; a < 3; a++)
      cd[a] = c3[a];
    pos = 20;
  }
  const size_t nchunks = orc_chunk_volume(vol, cd, NULL, 0);
  const size_t hlen = pos + 4 * nchunks;
  if (src_len < hlen)
    return -1;
  size_t* off = (size_t*)malloc(nchunks * sizeof(size_t));
  size_t* len = (size_t*)malloc(nchunks * sizeof(size_t));
  size_t at = hlen, total = hlen, far = 0;
  const int whole = pct == 0 || pct >= 100;
  for (size_t i = 0; i < nchunks; i++) {
    uint32_t l;
    memcpy(&l, h + pos + 4 * i, 4);
    off[i] = at;
    at += l;
    len[i] = l;
    if (!whole && l > 64) {
      size_t req = (size_t)((double)pct / 100.0 * (double)l);
      len[i] = req < 64 ? 64 : req;
    }
    total += len[i];
    if (off[i] + len[i] > far)
      far = off[i] + len[i];
  }
  int rtn = -1;
  if (src_len >= far) {
    uint8_t* out = (uint8_t*)malloc(total);
    memcpy(out, h, pos);
    if (!whole) {
      out[0] = 0;       /* SPERR_VERSION_MAJOR */
      out[1] |= 0x80;   /* "a portion of another complete bitstream" */
    }
    size_t w = hlen;
    for (size_t i = 0; i < nchunks; i++) {
      const uint32_t l = (uint32_t)len[i];
      memcpy(out + pos + 4 * i, &l, 4);
      memcpy(out + w, h + off[i], len[i]);
      w += len[i];
    }
    *dst = out;
    *dst_len = total;
    rtn = 0;
  }
  free(off);
  free(len);
  return rtn;
}
